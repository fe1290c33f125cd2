"""Inputs the fixture-based parity tests do not reach: degenerate point clouds for the grid searches, attention logits
far from unit scale, sizes that do not tile.  Runs tools/stress_*.py (also usable by hand) and checks their verdicts."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(name, **env):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", name)], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, **env))
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


def test_grid_searches_on_degenerate_clouds():
    text = _run("stress_geometry.py")
    assert "mismatches: 0" in text, text
    assert len(re.findall(r"chamfer True  knn True  graph True", text)) == 8, text


def test_attention_cores_far_from_unit_scale():
    text = _run("stress_scales.py")
    lines = [ln for ln in text.splitlines() if ln.startswith(("SA ", "SAev", "N2P"))]
    assert len(lines) == 11 and all("finite True" in ln for ln in lines), text
    for ln in lines:
        errs = [float(x) for x in re.findall(r"(\d\.\de-\d+)", ln)]
        assert errs and max(errs) < 5e-5, ln


def test_knn_k500_and_dist_loss_at_shipped_sizes():
    text = _run("stress_knn_dist.py")
    knn = [ln for ln in text.splitlines() if ln.startswith("knn ")]
    assert len(knn) == 5 and all(ln.endswith("equal: True") for ln in knn), text
    errs = [float(x) for x in re.findall(r"rel_err (\S+)", text)]
    assert len(errs) == 2 and max(errs) < 1e-5, text


def test_chamfer_backward_with_non_finite_coordinates_stays_in_bounds():
    """a diverged step (NaN / inf coordinates) leaves every distance comparison of the Chamfer search false; the
    arg-min index handed to the backward kernel must still be a valid row (the backward gathers and scatters through it
    unchecked).  The loss is allowed to be NaN; the indices and the gradient buffers are not allowed to be out of range."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
    from dvm import nn_ops, ops
    g = torch.Generator().manual_seed(3)
    for B, N, M in ((8, 512, 384), (2, 3000, 2200), (1, 9000, 8000)):     # LDS scan (few clouds), brute force, grid paths
        a = torch.rand(B, N, 3, generator=g).cuda()
        b = torch.rand(B, M, 3, generator=g).cuda()
        a[0, 5] = float("nan")
        b[B - 1] = float("nan")                                            # a whole target cloud non-finite
        a[B - 1, 7] = float("inf")
        d1, d2, i1, i2 = ops.chamfer(a, b, want_idx=True)
        assert int(i1.min()) >= 0 and int(i1.max()) < M and int(i2.min()) >= 0 and int(i2.max()) < N, (B, N, M)
        ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        e1, e2 = nn_ops.chamfer_nn(ar, br)
        (e1.mean() + e2.mean()).backward()
        torch.cuda.synchronize()
        assert ar.grad.shape == a.shape and br.grad.shape == b.shape
        if B > 2:                                                          # pairs without a non-finite coordinate are untouched
            assert bool(torch.isfinite(ar.grad[1:B - 1]).all()) and bool(torch.isfinite(br.grad[1:B - 1]).all())


def test_dynamic_lds_opt_in_grows_with_the_request():
    """the per-(device, kernel) MaxDynamicSharedMemorySize registry must follow the LARGEST request: the N2P backward's
    one-workgroup CSR build needs (2N + 1) * 4 bytes — a first call at N = 2048 (16 KB) followed by one at N = 10000 (80 KB,
    beyond the 64 KB that needs no opt-in) must launch, and give the same result as the three-kernel form."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
    from dvm import ops
    g = torch.Generator().manual_seed(11)

    def run(B, N, C=64, K=40):
        qkv = torch.randn(B, N, 3 * C, generator=g).cuda()
        idx = torch.randint(0, N, (B, N, K), generator=g, dtype=torch.int32).cuda()
        out, attn = ops.n2p_core_fwd(qkv, idx)
        gout = torch.randn(B, N, C, generator=g).cuda()
        return qkv, idx, attn, gout, ops.n2p_core_bwd(qkv, idx, attn, gout)

    run(4, 2048)                                   # freezes a small attribute in a registry that does not grow
    qkv, idx, attn, gout, d_big = run(4, 10000)    # LDS form: 80 KB of dynamic LDS
    torch.cuda.synchronize()
    ref = torch.cat([ops.n2p_core_bwd(qkv[b:b + 1], idx[b:b + 1], attn[b:b + 1], gout[b:b + 1]) for b in range(4)])   # B = 1: three-kernel form
    assert torch.isfinite(d_big).all()
    assert (d_big - ref).abs().max() <= 1e-4 * ref.abs().max()
