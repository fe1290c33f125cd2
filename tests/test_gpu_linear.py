"""-m gpu: dvm_linear_f32 / dvm_linear_prefix_f32 — the 1x1 convolutions of LG-Net as the reference's single-thread
fp32 fma chain on the matrix cores — bit for bit against the C oracle (oracle/dvm_oracle.c::dvo_linear, itself pinned
to torch's CPU Conv1d / matmul in tests/test_oracle_vs_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from dvm import ops as _ops
    assert torch.cuda.is_available()
    return _ops


def _case(g, M, K, Co, bias, res, bn):
    x = torch.randn(M, K, generator=g) * 2
    w = torch.randn(Co, K, generator=g) / K ** 0.5
    b = 0.1 * torch.randn(Co, generator=g) if bias else None
    r = torch.randn(M, Co, generator=g) if res else None
    ab = (1 + 0.2 * torch.randn(Co, generator=g), 0.3 * torch.randn(Co, generator=g)) if bn else None
    return x, w, b, r, ab


SHAPES = [  # (M, K, Co): the layer shapes of LG-Net + ragged sizes + K values that exercise every block rule
    (300, 1152, 384), (257, 384, 64), (513, 64, 256), (200, 256, 64), (333, 256, 512), (130, 768, 128), (4995, 256, 128),
    (190, 512, 128), (70, 64, 16), (100, 128, 384), (65, 128, 512), (77, 400, 96), (50, 772, 40), (31, 20, 7), (40, 1156, 33),
    (64, 6, 130), (129, 2304, 64)]


@pytest.mark.parametrize("M,K,Co", SHAPES)
def test_linear_point_major_bit_exact(ops, M, K, Co):
    g = torch.Generator().manual_seed(M * 7 + K)
    for bias, res, bn, slope in [(False, False, False, 1.0), (True, True, True, 0.2), (False, True, True, 0.0), (True, False, True, 0.2)]:
        x, w, b, r, ab = _case(g, M, K, Co, bias, res, bn)
        ref = O.linear(x.numpy(), w.numpy(), None if b is None else b.numpy(), None if r is None else r.numpy(),
                       None if ab is None else ab[0].numpy(), None if ab is None else ab[1].numpy(), slope)
        c = lambda t: None if t is None else t.cuda()  # noqa: E731
        out = ops.linear(x.cuda(), w.cuda(), bias=c(b), res=c(r), bn=None if ab is None else (ab[0].cuda(), ab[1].cuda()), slope=slope)
        assert np.array_equal(out.cpu().numpy(), ref), (M, K, Co, bias, res, bn, slope, np.abs(out.cpu().numpy() - ref).max())


@pytest.mark.parametrize("B,N,K,Co", [(2, 192, 1152, 384), (1, 4995, 64, 64), (2, 301, 64, 16), (2, 256, 256, 512), (1, 1023, 768, 128),
                                      (8, 2048, 64, 192), (3, 130, 512, 128), (2, 77, 128, 80), (1, 50, 400, 24)])
def test_linear_channel_major_bit_exact(ops, B, N, K, Co):
    """The reference's Conv1d layout (B,Cin,N) -> (B,Cout,N), incl. N % 4 != 0 (unaligned rows) and per-channel epilogue."""
    g = torch.Generator().manual_seed(B * 1000 + N + K)
    x = torch.randn(B, K, N, generator=g)
    w = torch.randn(Co, K, 1, generator=g) / K ** 0.5
    b = 0.1 * torch.randn(Co, generator=g)
    out = ops.linear(x.cuda(), w.cuda(), bias=b.cuda(), channel_major=True).cpu().numpy()
    for bb in range(B):
        ref = O.linear(x[bb].t().contiguous().numpy(), w[:, :, 0].numpy(), b.numpy())
        assert np.array_equal(out[bb], ref.T), (B, N, K, Co, bb)


def test_linear_equals_recorded_reference_conv(ops, golden):
    """Both layouts against nn.Conv1d(k=1) [+ eval BatchNorm + LeakyReLU] outputs recorded from the reference's torch
    (one thread, build container) — the live torch on this box's host CPU picks other oneDNN kernels and is no yardstick."""
    from oracle import oracle as O2
    g = golden("linear_chain")
    for i in range(int(g["n"])):
        x, w, b, bn = g["x%d" % i], g["w%d" % i], g.get("b%d" % i), g["bn%d" % i]
        a, be = O2.bn_eval_affine(bn[0], bn[1], bn[2], bn[3], 1e-5)
        xc, wc = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()
        bc = None if b is None else torch.from_numpy(b).cuda()
        y = ops.linear(xc, wc, bias=bc, channel_major=True)
        assert np.array_equal(y.cpu().numpy(), g["y%d" % i]), i
        z = ops.linear(xc, wc, bias=bc, bn=(torch.from_numpy(a).cuda(), torch.from_numpy(be).cuda()), slope=0.2, channel_major=True)
        assert np.array_equal(z.cpu().numpy(), g["z%d" % i]), i
        zp = ops.linear(xc.transpose(1, 2).contiguous(), wc, bias=bc, bn=(torch.from_numpy(a).cuda(), torch.from_numpy(be).cuda()), slope=0.2)
        assert np.array_equal(zp.transpose(1, 2).cpu().numpy(), g["z%d" % i]), i


def test_linear_prefix_equals_concatenation(ops):
    """conv3 / conv4 of Uni3FC: rows [g[b] | x[b,n]] without building the (B,N,768) concatenation."""
    g = torch.Generator().manual_seed(5)
    for (B, N, Cg, Cx, Co) in [(2, 300, 512, 256, 128), (1, 77, 8, 56, 16), (3, 64, 384, 128, 40)]:
        gl = torch.randn(B, 1, Cg, generator=g)
        x = torch.randn(B, N, Cx, generator=g)
        w = torch.randn(Co, Cg + Cx, generator=g) / 20
        al, be = 1 + 0.1 * torch.randn(Co, generator=g), 0.1 * torch.randn(Co, generator=g)
        out = ops.linear(x.cuda(), w.cuda(), bn=(al.cuda(), be.cuda()), slope=0.2, prefix=gl.cuda()).cpu().numpy()
        cat = torch.cat((gl.expand(-1, N, -1), x), -1).reshape(B * N, -1)
        ref = O.linear(cat.numpy(), w.numpy(), alpha=al.numpy(), beta=be.numpy(), slope=0.2).reshape(B, N, Co)
        assert np.array_equal(out, ref), (B, N, Cg, Cx, Co)


def test_linear_is_batch_and_tile_independent(ops):
    """The same row gives the same bits whatever batch it sits in (different tile configurations are picked for
    small and large problems) — the property that makes results independent of the batch size."""
    g = torch.Generator().manual_seed(6)
    x = torch.randn(8 * 2048, 256, generator=g).cuda()
    w = (torch.randn(128, 256, generator=g) / 16).cuda()
    big = ops.linear(x, w)
    small = ops.linear(x[:200].contiguous(), w)
    assert torch.equal(big[:200], small)
    xc = x.view(8, 2048, 256).transpose(1, 2).contiguous()
    cm = ops.linear(xc, w, channel_major=True)
    assert torch.equal(cm.transpose(1, 2).reshape(-1, 128), big)


def test_conv1x1_autograd(ops):
    """Forward on the HIP kernel, backward as the two GEMMs: gradients against torch's conv1d in float64."""
    from dvm import nn_ops
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 64, 150, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(96, 64, 1, generator=g) / 8).cuda().requires_grad_(True)
    b = torch.randn(96, generator=g).cuda().requires_grad_(True)
    go = torch.randn(2, 96, 150, generator=g).cuda()
    y = nn_ops.conv1x1(x, w, b)
    y.backward(go)
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    yd = torch.nn.functional.conv1d(xd, wd, bd)
    yd.backward(go.double())
    for a, r in ((y, yd), (x.grad, xd.grad), (w.grad, wd.grad), (b.grad, bd.grad)):
        assert (a.double() - r).abs().max() <= 2e-5 * (1 + r.abs().max()), (a.double() - r).abs().max()


def test_bn_affine_fold_matches_aten(ops, golden):
    """models.model._bn_affine (host-side fold) + the fused epilogue == conv -> nn.BatchNorm1d.eval() -> LeakyReLU as
    recorded from the reference's torch, bit for bit."""
    import models.model as mm
    g = golden("linear_chain")
    for i in range(int(g["n"])):
        w, bnp = g["w%d" % i], g["bn%d" % i]
        bn = torch.nn.BatchNorm1d(w.shape[0]).eval()
        with torch.no_grad():
            for t, v in zip((bn.weight, bn.bias, bn.running_mean, bn.running_var), bnp):
                t.copy_(torch.from_numpy(v))
        bn = bn.cuda()
        b = g.get("b%d" % i)
        out = ops.linear(torch.from_numpy(g["x%d" % i]).cuda(), torch.from_numpy(w).cuda(), bias=None if b is None else torch.from_numpy(b).cuda(),
                         bn=mm._bn_affine(bn), slope=0.2, channel_major=True)
        assert np.array_equal(out.cpu().numpy(), g["z%d" % i]), i


@pytest.mark.parametrize("cm", [False, True])
def test_linear_scaled_residual_epilogue(ops, cm):
    """r + s * (x W^T + b) in the GEMM's epilogue (dvm_linear_scaled_residual_f32, the JBU "fixup" form) == the same GEMM
    followed by torch's multiply and add, bit for bit (one multiply and one add, not an fma)."""
    g = torch.Generator().manual_seed(21)
    B, N, K, Co = 3, 777, 100, 52
    x = torch.randn(B, K, N, generator=g).cuda() if cm else torch.randn(B, N, K, generator=g).cuda()
    w, b = torch.randn(Co, K, generator=g).cuda(), torch.randn(Co, generator=g).cuda()
    plain = ops.linear(x, w, bias=b, channel_major=cm)
    r = torch.randn(plain.shape, generator=g).cuda()
    got = ops.linear(x, w, bias=b, channel_major=cm, post=(0.1, r))
    assert torch.equal(got, plain * 0.1 + r)
    with pytest.raises(Exception):
        ops.linear(x, w, bias=b, channel_major=cm, post=(0.1, r), slope=0.2)
