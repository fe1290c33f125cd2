"""One-off check of the fused preparation of the pair path (rownorm_split_kernel): the fp16 planes it leaves in the workspace and
the row norms, against a host computation of the same definition (planes: h = rn16(x s), m = rn16(x s - h), s the power of two
of dvm_softcorr_f16.h::scale_exp; norms: the oracle's ATen-order sum of squares).  The planes are found in the workspace by
their first row's bytes.  Usage: python tools/check_planes.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from dvm import ops

wl = ops.deformer_weight_list(dict(np.load(os.path.join(ROOT, "tests", "golden", "deformer_scape_r_weights.npz"))), "cuda")
g = torch.Generator().manual_seed(5)
bad = 0
for (B, N, M, spike) in ((3, 2048, 2048, False), (2, 1021, 777, False), (3, 2048, 2048, True)):
    f1, f2 = torch.randn(B, N, 128, generator=g), torch.randn(B, M, 128, generator=g)
    f1[0, 1, :8] = torch.tensor([1e-6, -3e-5, 7e-4, 0.0, -0.0, 2.5e-7, 1e-3, -1e-2])   # small values: subnormal m plane
    if spike:
        f2[1, 33, 7] = 300.0   # a row the 1/64 sample does not see: the planes are re-made
    v1, v2 = torch.rand(B, N, 3, generator=g), torch.rand(B, M, 3, generator=g)
    s0 = torch.zeros(B, dtype=torch.int32)
    ops.pair_forward(wl, f1.cuda(), f2.cuda(), v1.cuda(), v2.cuda(), 100.0, s0.cuda(), s0.cuda())
    torch.cuda.synchronize()
    ws = [v for k, v in ops._ws_cache.items() if k[2] == "pair2"][0].cpu().numpy()
    amax = max(float(f1.abs().max()), float(f2.abs().max()))
    e = int(np.floor(np.log2(amax)))
    sc = np.float32(2.0 ** (11 - e))
    for name, f in (("f1", f1), ("f2", f2)):
        xs = f.numpy().reshape(-1, 128).astype(np.float32) * sc
        h = xs.astype(np.float16)
        m = (xs - h.astype(np.float32)).astype(np.float16)
        planes = np.concatenate([h, m], axis=1).view(np.uint8).reshape(-1)   # row: 256 B of h, 256 B of m
        key = planes[:512].tobytes()
        pos = ws.tobytes().find(key)
        ok = pos >= 0 and np.array_equal(ws[pos:pos + planes.size], planes)
        if not ok and pos >= 0:
            d = np.nonzero(ws[pos:pos + planes.size] != planes)[0]
            print("   first mismatch at byte", d[0], "row", d[0] // 512, "of", d.size)
        print("B=%d N=%d M=%d spike=%s %s: planes %s" % (B, N, M, spike, name, "equal" if ok else "DIFFER (found=%s)" % (pos >= 0)))
        bad += 0 if ok else 1
print("mismatches:", bad)
