import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
from weights_init import reinit
from dvm import nn_ops
from models.model import Uni3FC
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = Uni3FC(k=40).to(dev); reinit(net, 4, gain=0.5)
nets = [net, copy.deepcopy(net), copy.deepcopy(net)]
g = torch.Generator().manual_seed(6)
xs = [torch.randn(2, 3, 640, generator=g).to(dev) for _ in range(2)]
ds = [torch.randn(2, 640, 1152, generator=g).to(dev) for _ in range(2)]
grads = []; feats = []
for m, fuse in zip(nets, (False, True, False)):
    prev = nn_ops.fuse_grad_accumulation(fuse)
    m.train()
    for p in m.parameters():
        p.grad = torch.full_like(p, 0.25)
    outs = [m(x, d)[0] for x, d in zip(xs, ds)]
    loss = sum(o.square().mean() for o in outs)
    loss.backward(); torch.cuda.synchronize()
    nn_ops.fuse_grad_accumulation(prev)
    grads.append({k: p.grad.clone() for k, p in m.named_parameters()}); feats.append([o.detach() for o in outs])
print("fwd diff", [float((a - b).abs().max()) for a, b in zip(feats[0], feats[1])])
for tag, j in (("fused vs plain", 1), ("plain vs plain", 2)):
    rows = []
    for k in grads[0]:
        a, b = grads[0][k].double(), grads[j][k].double()
        rows.append((float((a - b).abs().max()) / max(1e-9, float((a - 0.25).abs().max())), k))
    rows.sort(reverse=True)
    print(tag, rows[:8])
