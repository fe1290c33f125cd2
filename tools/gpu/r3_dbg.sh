cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tiny_M" 2>&1 | grep -E "Mismatch|ACTUAL|DESIRED|Max|x:|y:|rror" | head -20
DVM_K1_SWEEP=0 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tiny_M" 2>&1 | tail -2
DVM_K1_SWEEP=1 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tiny_M" 2>&1 | tail -2
python - <<PY
import sys, numpy as np, torch, zlib
sys.path.insert(0, "dv-matcher_amd"); sys.path.insert(0, ".")
from dvm import ops
from oracle import oracle as O
g = np.random.default_rng(zlib.crc32(b"tiny_M"))
f1 = g.standard_normal((150, 128)).astype(np.float32); f2 = g.standard_normal((210, 128)).astype(np.float32)[:5]
val, idx, smax, ssum = ops.softcorr(torch.from_numpy(f1)[None].cuda(), torch.from_numpy(f2)[None].cuda(), 40.0, topk=10, variant=3)
oval, oidx, osmax, osum = O.softcorr(f1, f2, 40.0, topk=10)
bad = np.nonzero(smax[0].cpu().numpy() != osmax)[0]
print("rows with wrong smax:", bad[:20], len(bad))
print("idx eq", np.array_equal(idx[0].cpu().numpy(), oidx))
for r in bad[:4]:
    print(r, idx[0, r].cpu().numpy(), oidx[r], val[0, r].cpu().numpy()[:5], oval[r][:5], float(smax[0, r]), osmax[r])
PY
