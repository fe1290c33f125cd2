"""Point-major training operators (dvm_bn_act_train_{fwd,bwd}_pm_f32, dvm_linear_wgrad_f32, the _LinearPM / _BNActPM autograd
functions) against torch's own BatchNorm / matmul autograd on the same device."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from weights_init import reinit  # noqa: E402

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("R,C,slope,with_res", [(4096, 64, 0.2, False), (16384, 128, 1.0, True), (1000, 384, 0.2, False),
                                                (2 * 4995, 512, 0.2, False), (37, 64, 0.0, True)])
def test_bn_pm_matches_torch(R, C, slope, with_res):
    from dvm import nn_ops
    g = torch.Generator().manual_seed(R + C)
    x = (torch.randn(1, R, C, generator=g) * 1.7 + 0.3).to(_dev()).requires_grad_(True)
    res = torch.randn(1, R, C, generator=g).to(_dev()).requires_grad_(True) if with_res else None
    bn = torch.nn.BatchNorm1d(C).to(_dev())
    with torch.no_grad():   # (seeded: an element that lands within rounding of the activation's kink takes the other branch)
        bn.weight.copy_(0.5 + torch.rand(C, generator=g))
        bn.bias.copy_(torch.rand(C, generator=g) - 0.5)
    ref = torch.nn.BatchNorm1d(C).to(_dev()).double()
    ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    gy = torch.randn(1, R, C, generator=g).to(_dev())

    y = nn_ops.bn_act_pm(bn, x, res, slope)
    y.backward(gy)
    xd = x.detach().double().requires_grad_(True)
    rd = res.detach().double().requires_grad_(True) if with_res else None
    z = xd if rd is None else xd + rd
    yr = ref(z.transpose(1, 2)).transpose(1, 2)
    yr = yr if slope == 1.0 else torch.nn.functional.leaky_relu(yr, slope)
    yr.backward(gy.double())

    def close(a, b, tol):
        a, b = a.double(), b.double()
        assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max())), (float((a - b).abs().max()), float(b.abs().max()))

    # the activation's kink: a value within rounding of 0 may take the other branch; exclude nothing, the tolerance covers it
    close(y, yr, 2e-5)
    # gradients through the (Leaky)ReLU: elements whose pre-activation is within 1e-5 of the kink may take either slope
    zd = z.detach()
    pre = (zd - zd.mean(dim=(0, 1))) / torch.sqrt(zd.var(dim=(0, 1), unbiased=False) + ref.eps) * ref.weight.detach() + ref.bias.detach()
    safe = (pre.abs() > 1e-5).to(x.grad.dtype) if slope != 1.0 else torch.ones_like(x.grad)
    assert float(safe.mean()) > 0.999
    close(x.grad * safe, xd.grad * safe, 5e-5)
    if with_res:
        close(res.grad * safe, rd.grad * safe, 5e-5)
    close(bn.weight.grad, ref.weight.grad, 5e-5)
    close(bn.bias.grad, ref.bias.grad, 5e-5)
    close(bn.running_mean, ref.running_mean, 1e-5)
    close(bn.running_var, ref.running_var, 1e-5)
    assert int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("R,Co,K", [(16384, 64, 64), (4096, 384, 1152), (9990, 128, 768), (2048, 192, 64), (100, 64, 256), (16384, 512, 256)])
def test_linear_wgrad(R, Co, K):
    from dvm import ops
    g = torch.Generator().manual_seed(R + Co + K)
    gy = torch.randn(R, Co, generator=g).to(_dev())
    x = torch.randn(R, K, generator=g).to(_dev())
    dW = ops.linear_wgrad(gy, x)
    ref = gy.double().t() @ x.double()
    err = float((dW.double() - ref).abs().max())
    assert err <= 2e-5 * float(ref.abs().max()) + 1e-4 * np.sqrt(R) * 1e-2, err


@pytest.mark.parametrize("slope,bias", [(1.0, False), (0.2, True), (1.0, True)])
def test_linear_pm_autograd(slope, bias):
    from dvm import nn_ops
    g = torch.Generator().manual_seed(11)
    B, N, K, Co = 2, 1500, 256, 128
    x = torch.randn(B, N, K, generator=g).to(_dev()).requires_grad_(True)
    w = (torch.randn(Co, K, 1, generator=g) / 16).to(_dev()).requires_grad_(True)
    b = torch.randn(Co, generator=g).to(_dev()).requires_grad_(True) if bias else None
    gy = torch.randn(B, N, Co, generator=g).to(_dev())
    y = nn_ops.linear_pm(x, w, b, slope)
    y.backward(gy)
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    bd = b.detach().double().requires_grad_(True) if bias else None
    yr = torch.nn.functional.linear(xd, wd[..., 0], bd)
    yr = yr if slope == 1.0 else torch.nn.functional.leaky_relu(yr, slope)
    yr.backward(gy.double())
    for a, r in [(y, yr), (x.grad, xd.grad), (w.grad, wd.grad)] + ([(b.grad, bd.grad)] if bias else []):
        assert float((a.double() - r).abs().max()) <= 3e-5 * max(1.0, float(r.abs().max()))


def test_train_layouts_agree(monkeypatch):
    """One training forward/backward of Uni3FC in the point-major layout against the channel-major fallback: same GEMM
    chains, BatchNorm statistics summed in another order -> agreement to fp32 rounding when the kNN sets coincide."""
    import copy
    from models.model import Uni3FC
    torch.manual_seed(0)
    net = Uni3FC(k=40).to(_dev())
    reinit(net, 3, gain=0.5)
    net2 = copy.deepcopy(net)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 1024, generator=g).to(_dev())
    d = torch.randn(2, 1024, 1152, generator=g).to(_dev())
    outs = []
    for m, layout in ((net, "pm"), (net2, "cm")):
        m.point_major_train = layout == "pm"
        m.train()
        feat, cf = m(x, d)
        (feat.square().mean() + cf.square().mean()).backward()
        outs.append((feat.detach(), cf.detach(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    (f1, c1, g1), (f2, c2, g2) = outs
    assert float((c1 - c2).abs().max()) < 1e-4
    frac = float(((f1 - f2).abs().amax(-1) > 2e-3).float().mean())
    assert frac < 0.05, frac          # a kNN near-tie flipping moves a few points; everything else agrees
    assert set(g1) == set(g2)
    for k in g1:
        assert torch.isfinite(g1[k]).all()
    bn_run = dict(net.named_buffers())
    for k, v in net2.named_buffers():
        if "running" in k:
            assert torch.allclose(bn_run[k], v, atol=2e-3, rtol=2e-2), k


def test_grad_accumulation_fusion():
    """Two backward passes through one network (the criterion's two calls per step): parameter gradients added in place by
    the kernels == autograd's own accumulation, and a pre-existing .grad is added to, not overwritten."""
    import copy
    from dvm import nn_ops
    from models.model import Uni3FC
    torch.manual_seed(0)
    net = Uni3FC(k=40).to(_dev())
    reinit(net, 4, gain=0.5)
    net2 = copy.deepcopy(net)
    g = torch.Generator().manual_seed(6)
    xs = [torch.randn(2, 3, 640, generator=g).to(_dev()) for _ in range(2)]
    ds = [torch.randn(2, 640, 1152, generator=g).to(_dev()) for _ in range(2)]
    grads = []
    for m, fuse in ((net, False), (net2, True)):
        prev = nn_ops.fuse_grad_accumulation(fuse)
        try:
            m.train()
            for p in m.parameters():
                p.grad = torch.full_like(p, 0.25)
            loss = sum(m(x, d)[0].square().mean() for x, d in zip(xs, ds))
            loss.backward()
            torch.cuda.synchronize()
        finally:
            nn_ops.fuse_grad_accumulation(prev)
        grads.append({k: p.grad.clone() for k, p in m.named_parameters()})
    touched = 0
    for k in grads[0]:
        a, b = grads[0][k].double(), grads[1][k].double()
        scale = float((a - 0.25).abs().max())          # the gradient proper; 0.25 + g is rounded to 3e-8 either way
        touched += scale > 0
        assert float((a - b).abs().max()) <= 1.2e-7 + 2e-3 * scale, (k, float((a - b).abs().max()), scale)
    assert touched > 100
