"""Float64 twins of the training-step fixtures (bb_trainstep_scape256 / bb_trainstep_scape1024): the REFERENCE's own step
evaluated in float64 with the neighbour sets of its canonical (1-thread fp32) run forced, i.e. the quantity both the
reference's fp32 step and this repo's HIP step approximate.  Stored next to each fixture as <name>_f64.npz:

  g64_<param>, gd64_<param>   float64 gradients (the `keep` tensors of make_fixtures_backbone.py), cast to float32
  losses64                    the 5-tuple
  err32_<param>               rel. L2 distance of the canonical fp32 gradient (g_<param>) from the float64 one: the
                              reference's own rounding noise per tensor, which bounds what any fp32 implementation can match

tests/test_gpu_network.py prices this repo's gradients against the float64 ones and against err32.
Run from the repo root:  python tests/golden/make_fixtures_trainstep_f64.py [name ...]
"""
import contextlib
import copy
import os
import random
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from weights_init import dino_from_seed, reinit  # noqa: E402
import ref_import  # noqa: E402

STABLE_GAIN = 0.5


@contextlib.contextmanager
def force_knn(rm, sets):
    """The reference's knn_new replaced by the recorded neighbour sets, in call order."""
    orig, it = rm.knn_new, iter(sets)
    rm.knn_new = lambda a, b, k: next(it).to(torch.int64)
    try:
        yield
    finally:
        rm.knn_new = orig


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


def main(names):
    rm, rl, rdg = ref_import.import_reference()
    torch.set_num_threads(8)
    # the reference's graph construction indexes float32 work arrays with its input: the sampling runs on the float32
    # coordinates (the same node indices as in the canonical run), everything differentiable stays float64
    fps_orig = rdg.farthest_point_sample
    rdg.farthest_point_sample = lambda xyz, npoint: fps_orig(xyz.float(), npoint)
    for name in names:
        g = dict(np.load(os.path.join(HERE, name + ".npz")))
        v1, v2 = torch.from_numpy(g["verts1"]).double(), torch.from_numpy(g["verts2"]).double()
        B, N, _ = v1.shape
        d1, d2 = dino_from_seed(int(g["dino_seed1"]), B, N).double(), dino_from_seed(int(g["dino_seed2"]), B, N).double()
        net32 = reinit(rm.Uni3FC(k=40), salt=5, gain=float(g["gain"])).train()
        net = copy.deepcopy(net32).double()
        net.pos_encoding_sin_wave = lambda c: net32.pos_encoding_sin_wave(c.float()).double()   # (as f64_twin: part of the input)
        dfm = rm.Deformer(10)
        dfm.load_state_dict(torch.load(os.path.join(ref_import.REF, "ckpt/dvmatcher_scape_r/ep_deformer_val_best.pth"),
                                       weights_only=True, map_location="cpu"))
        dfm = dfm.double().train()
        crit = rl.GraphDeformLoss_Neural(k_deform=10, w_dist=0.02, w_map=0.005, k_dist=int(g["k_dist"]), N_dist=int(g["N_dist"]),
                                         partial=False, w_deform=0.5, w_img=0, w_rank=0, w_self_rec=0.5, w_cd=0.1, w_arap=0.01,
                                         save_name="fx")
        sets = [torch.from_numpy(s.astype(np.int64)) for s in g["knn_idx"]]
        cwd = os.getcwd()
        with tempfile.TemporaryDirectory() as td:
            os.chdir(td)
            try:
                random.seed(9001)
                torch.manual_seed(9002)
                with force_knn(rm, sets):
                    f1, _ = net(v1.permute(0, 2, 1), d1, None)
                    f2, _ = net(v2.permute(0, 2, 1), d2, None)
                out = crit(f1, f2, torch.cdist(v1, v1), torch.cdist(v2, v2), v1, v2, np.float64(g["alpha"]), dfm)
                out[0].backward()
            finally:
                os.chdir(cwd)
        arrs = dict(losses64=np.array([float(torch.as_tensor(o).detach()) for o in out]),
                    feat1_err32=np.float64(np.abs(f1.detach().numpy() - g["feat1"]).max()))
        named, nd = dict(net.named_parameters()), dict(dfm.named_parameters())
        for key in g:
            if key.startswith("g_") or key.startswith("gd_"):
                bb = key.startswith("g_")
                pname = key[2 if bb else 3:].replace("__", ".")
                g64 = (named[pname] if bb else nd[pname]).grad.numpy()
                arrs[("g64_" if bb else "gd64_") + key[2 if bb else 3:]] = g64.astype(np.float32)
                arrs["err32_" + key] = np.float64(rel(g[key], g64))
                k8 = ("g8_" if bb else "gd8_") + key[2 if bb else 3:]
                if k8 in g:
                    arrs["err32t8_" + key] = np.float64(rel(g[k8], g64))
        np.savez_compressed(os.path.join(HERE, name + "_f64.npz"), **arrs)
        print(name, "losses32", g["losses"], "losses64", arrs["losses64"])
        for key in sorted(k for k in arrs if k.startswith("err32_")):
            print("   %-52s fp32(1 thread) vs f64 %.2e   fp32(8 threads) vs f64 %s" % (key[6:], arrs[key],
                  "%.2e" % arrs["err32t8_" + key[6:]] if "err32t8_" + key[6:] in arrs else "-"))


if __name__ == "__main__":
    main(sys.argv[1:] or ["bb_trainstep_scape256", "bb_trainstep_scape1024"])
