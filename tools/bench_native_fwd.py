import os, sys, time
ROOT = "/root/repo" if os.path.isdir("/root/repo/dv-matcher_amd") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
import torch
from models.model import Uni3FC
torch.manual_seed(0)
for (B, N) in ((2, 700), (8, 2048), (1, 4995)):
    net = Uni3FC(k=40).cuda().eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.3); m.running_var.uniform_(0.5, 1.5); m.weight.uniform_(0.7, 1.3); m.bias.normal_(0, 0.2)
    x = torch.rand(B, 3, N).cuda(); dino = torch.randn(B, N, 1152).cuda()
    os.environ["DVM_NATIVE_FWD"] = "0"
    with torch.no_grad(): ref, rtmp = net(x, dino, None)
    os.environ["DVM_NATIVE_FWD"] = "1"
    with torch.no_grad(): out, otmp = net(x, dino, None)
    torch.cuda.synchronize()
    d = (out - ref).abs()
    print("B=%d N=%d: feat equal %s  max |diff| %.3g  points > 1e-5: %.5f   tmp equal %s max %.3g" % (B, N, bool(torch.equal(out, ref)), float(d.max()), float((d.amax(-1) > 1e-5).float().mean()), bool(torch.equal(otmp, rtmp)), float((otmp - rtmp).abs().max())))
    for mode in ("0", "1"):
        os.environ["DVM_NATIVE_FWD"] = mode
        with torch.no_grad():
            for _ in range(3): net(x, dino, None)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(20): net(x, dino, None)
            th = time.perf_counter() - t
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
        print("   DVM_NATIVE_FWD=%s: %.2f ms per forward (host enqueue %.2f ms)" % (mode, dt * 1e3, th / 20 * 1e3))
