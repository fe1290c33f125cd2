"""-m gpu: LG-Net's native training forward / backward (dvm_uni3fc_train_{fwd,bwd}_f32, csrc/dvm_uni3fc_train.hip) against the
autograd path it replaces (Uni3FC._forward_train_pm, itself pinned to the reference by tests/test_gpu_network.py): same
launches in the same order, so the forward must agree BIT FOR BIT (features, second output, running statistics, batch
counters) and the parameter gradients to fp32 summation-order noise (the weight-gradient kernel combines row chunks with
atomics in both paths).  Reference: models/model.py:680-761 under autograd, train.py:93-112."""
import copy
import os
import subprocess
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from weights_init import reinit  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _nets(k, seed=0, gain=None):
    import models.model as mm
    torch.manual_seed(seed)
    a = mm.Uni3FC(k=k)
    if gain is not None:
        reinit(a, gain=gain)
    with torch.no_grad():       # non-trivial BatchNorm affines and biases
        g = torch.Generator().manual_seed(seed + 1)
        for m in a.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.copy_(0.5 + torch.rand(m.weight.shape, generator=g))
                m.bias.copy_(0.2 * (torch.rand(m.bias.shape, generator=g) - 0.5))
    a = a.cuda().train()
    return a, copy.deepcopy(a)


def _inputs(B, N, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(B, 3, N, generator=g) - 0.5).cuda()
    dino = torch.randn(B, N, 1152, generator=g).cuda()
    return x, dino


def _run(net, x, dino, native, with_tmp, gf, gt):
    net.native_train = bool(native)
    feat, tmp = net(x, dino, None)
    loss = (feat * gf).sum() + ((tmp * gt).sum() if with_tmp else 0.0)
    loss.backward()
    return feat.detach(), tmp.detach()


def _compare(a, b, tol):
    """Every gradient tensor within `tol` of the autograd path's, relative to that tensor's largest entry — but not finer than
    1e-2 of the largest gradient entry of its kind (weights / vectors) in the network: some gradients are sums that cancel
    exactly in real arithmetic and consist of rounding noise in BOTH paths (a bias in front of a BatchNorm — SA_Layer's
    trans_conv.bias — receives the column sums of a BatchNorm backward, which are zero; the bias of the BatchNorm in front of
    the max over the points collects per-shape column sums of one, which add up to zero over the batch)."""
    worst = 0.0
    floor = {}
    for _, q in b.named_parameters():
        if q.grad is not None:
            floor[q.dim() > 1] = max(floor.get(q.dim() > 1, 0.0), 1e-2 * float(q.grad.abs().max()))
    for (name, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        if q.grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name      # the 12 unused parameters
            continue
        assert p.grad is not None, name
        scale = max(float(q.grad.abs().max()), floor[q.dim() > 1]) + 1e-30
        err = float((p.grad - q.grad).abs().max()) / scale
        worst = max(worst, err)
        assert err <= tol, (name, err, scale)
    for (name, p), (_, q) in zip(a.named_buffers(), b.named_buffers()):
        assert torch.equal(p, q), name          # running statistics and batch counters: the same kernels in the same order
    return worst


@pytest.mark.parametrize("B,N,k,with_tmp", [(2, 300, 20, True), (1, 64, 40, False), (3, 515, 40, True)])
def test_native_training_step_equals_autograd_path(B, N, k, with_tmp):
    a, b = _nets(k, seed=N, gain=0.5)
    x, dino = _inputs(B, N, 7 + N)
    g = torch.Generator().manual_seed(3)
    gf, gt = torch.randn(B, N, 128, generator=g).cuda(), torch.randn(B, N, 64, generator=g).cuda()
    fa, ta = _run(a, x, dino, True, with_tmp, gf, gt)
    fb, tb = _run(b, x, dino, False, with_tmp, gf, gt)
    assert torch.equal(fa, fb) and torch.equal(ta, tb)
    worst = _compare(a, b, 2e-4)
    print("worst relative gradient difference %.2e" % worst)


def test_native_training_fused_accumulation_over_two_calls():
    """The driver's configuration: every p.grad a view of one flat bucket, the kernels ADD into it (no gradient tensors handed
    to autograd), two network calls per step as the criterion makes them."""
    from dvm import nn_ops
    from dvm.dist import FlatGradBucket
    a, b = _nets(40, seed=5, gain=0.5)
    x1, d1 = _inputs(2, 256, 11)
    x2, d2 = _inputs(2, 256, 12)
    g = torch.Generator().manual_seed(4)
    gf = torch.randn(2, 256, 128, generator=g).cuda()
    ba = FlatGradBucket(list(a.parameters()), attach=True)
    prev = nn_ops.fuse_grad_accumulation(True)
    try:
        a.native_train = True
        f1, _ = a(x1, d1, None)
        f2, _ = a(x2, d2, None)
        ((f1 * gf).sum() + (f2 * gf).sum()).backward()
        import models.model as mm
        mm.join_side_streams(torch.device("cuda", 0))
    finally:
        nn_ops.fuse_grad_accumulation(prev)
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(ba.params, ba.views))     # still the bucket's views
    b.native_train = False
    g1, _ = b(x1, d1, None)
    g2, _ = b(x2, d2, None)
    ((g1 * gf).sum() + (g2 * gf).sum()).backward()
    assert torch.equal(f1, g1) and torch.equal(f2, g2)
    _compare(a, b, 2e-4)


def test_native_training_full_size_and_frozen_parameters():
    """B = 8, N = 2048 (configs[2]'s shape): forward bits equal the autograd path's, gradients finite; a frozen parameter
    (requires_grad = False) gets no gradient and does not stop the others."""
    a, b = _nets(40, seed=9, gain=0.5)
    a.conv0[0].weight.requires_grad_(False)
    b.conv0[0].weight.requires_grad_(False)
    x, dino = _inputs(8, 2048, 99)
    g = torch.Generator().manual_seed(8)
    gf, gt = torch.randn(8, 2048, 128, generator=g).cuda(), torch.randn(8, 2048, 64, generator=g).cuda()
    fa, ta = _run(a, x, dino, True, True, gf, gt)
    fb, tb = _run(b, x, dino, False, True, gf, gt)
    assert torch.equal(fa, fb) and torch.equal(ta, tb)
    assert a.conv0[0].weight.grad is None
    assert all(bool(torch.isfinite(p.grad).all()) for p in a.parameters() if p.grad is not None)
    _compare(a, b, 5e-4)


def test_native_training_is_the_default_and_falls_back():
    """train mode + data inputs -> the native node; inputs that require a gradient -> the autograd path (same result)."""
    a, b = _nets(20, seed=2, gain=0.5)
    x, dino = _inputs(1, 128, 1)
    feat, _ = a(x, dino, None)
    assert type(feat.grad_fn).__name__ == "_Uni3FCTrainBackward"
    dino2 = dino.clone().requires_grad_(True)
    feat2, _ = b(x, dino2, None)
    assert type(feat2.grad_fn).__name__ != "_Uni3FCTrainBackward"
    assert torch.equal(feat.detach(), feat2.detach())
    feat2.sum().backward()
    assert dino2.grad is not None and bool(torch.isfinite(dino2.grad).all())


@pytest.mark.parametrize("native", [True, False], ids=["native", "autograd"])
def test_deterministic_switch_gives_bit_reproducible_gradients(native):
    """with dvm_set_deterministic(1) the three places of LG-Net's backward that combine partial sums of different
    workgroups with fp32 atomics (row chunks of the weight gradient, the split of the SA backward, the in-edge order of the N2P
    gather) sum in a fixed order: two runs of the same step give the SAME BITS in every parameter gradient, on the native node
    and on the autograd path; against the default (atomic) mode the gradients agree to summation-order noise."""
    from dvm import ops
    x, dino = _inputs(4, 1024, 5)
    g = torch.Generator().manual_seed(6)
    gf, gt = torch.randn(4, 1024, 128, generator=g).cuda(), torch.randn(4, 1024, 64, generator=g).cuda()

    def grads():
        a, _ = _nets(40, seed=3, gain=0.5)
        _run(a, x, dino, native, True, gf, gt)
        return a, [p.grad.clone() for p in a.parameters() if p.grad is not None]

    prev = ops.set_deterministic(True)
    try:
        _, g1 = grads()
        _, g2 = grads()
        _, g3 = grads()
    finally:
        ops.set_deterministic(prev)
    assert all(torch.equal(u, v) for u, v in zip(g1, g2)) and all(torch.equal(u, v) for u, v in zip(g1, g3))
    ref, g0 = grads()                                   # default mode
    worst = max(float((u - v).abs().max()) / (float(v.abs().max()) + 1e-30) for u, v in zip(g1, g0) if float(v.abs().max()) > 1e-3)
    assert worst < 1e-3, worst


def test_forward_pair_equals_two_calls():
    """Uni3FC.forward_pair — the step's two network calls as ONE native call with two groups (BatchNorm statistics,
    position-encoding range and running-statistics updates per call, in call order) — against two sequential forward() calls on
    an identical copy (merge_pair_calls = False): features, second outputs, running statistics and batch counters bit-identical;
    parameter gradients to summation-order noise."""
    from dvm import nn_ops
    from dvm.dist import FlatGradBucket
    import models.model as mm
    a, b = _nets(40, seed=7, gain=0.5)
    b.merge_pair_calls = False
    x1, d1 = _inputs(4, 1024, 31)
    x2, d2 = _inputs(4, 1024, 32)
    g = torch.Generator().manual_seed(9)
    gf1, gf2 = torch.randn(4, 1024, 128, generator=g).cuda(), torch.randn(4, 1024, 128, generator=g).cuda()
    ba, bb = FlatGradBucket(list(a.parameters()), attach=True), FlatGradBucket(list(b.parameters()), attach=True)
    prev = nn_ops.fuse_grad_accumulation(True)
    try:
        for _ in range(2):   # twice: the second round reuses the helper streams and the caching allocator's blocks
            ba.zero(), bb.zero()
            (f1, t1), (f2, t2) = a.forward_pair(x1, d1, x2, d2)
            node = f1.grad_fn.next_functions[0][0]      # (the outputs are row slices of ONE node's)
            assert type(node).__name__ == "_Uni3FCTrainBackward"
            ((f1 * gf1).sum() + (f2 * gf2).sum() + t1.sum()).backward()
            mm.join_side_streams(torch.device("cuda", 0))
            (g1, u1), (g2, u2) = b.forward_pair(x1, d1, x2, d2)
            assert g1.grad_fn is not f1.grad_fn and type(g1.grad_fn).__name__ == "_Uni3FCTrainBackward"   # two nodes, one per call
            ((g1 * gf1).sum() + (g2 * gf2).sum() + u1.sum()).backward()
            mm.join_side_streams(torch.device("cuda", 0))
            torch.cuda.synchronize()
            assert torch.equal(f1, g1) and torch.equal(f2, g2) and torch.equal(t1, u1) and torch.equal(t2, u2)
            _compare(a, b, 2e-4)
    finally:
        nn_ops.fuse_grad_accumulation(prev)
