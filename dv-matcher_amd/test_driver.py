#!/usr/bin/env python3
"""Inference driver reproducing the reference's test loop (test.py:95-133) on the MI355X path.

    feat1,_ = Uni3FC(verts1^T, dino1, upsampler); feat2,_ = Uni3FC(verts2^T, dino2, upsampler)     (eval mode)
    T12 = search_t(feat1, feat2) + 1;  T21 = search_t(feat2, feat1) + 1                           (test.py:19-28)
    result/<exp>_<dataset>/T/T_<a>_<b>.txt ('%i', 1-based), .../feature/usefeature_<a>.mat {'uphi': feat}

Pairs come from --data-root (a dataset directory in the reference's layout, read by models/dataset.py's
testDataset, with the visual features in feat/<shape>.mat), from --pairs (an .npz with verts1, verts2 (N,3),
dino1, dino2 (N,1152), name1, name2) or are synthetic (--synthetic P).  The DINOv2 feature pipeline that
produces the .mat files is outside this path.

  python dv-matcher_amd/test_driver.py --synthetic 2 --points 1024 --out result/demo [--ckpt ep_val_best.pth]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

from dvm import ops  # noqa: E402
from models.loss import search_t  # noqa: E402,F401
from models.model import Uni3FC  # noqa: E402


def load_pairs(args):
    """Yields (name1, name2, load1, load2); load() -> (verts (N,3), visual features (N,1152)) is only called for a shape
    whose backbone features are not cached yet."""
    if args.data_root:                                   # the reference's loop over testDataset (test.py:66-100)
        import scipy.io as sio
        from models.dataset import testDataset
        data = testDataset(args.data_root, name=args.data_name, train=False)

        def loader(i):
            path = os.path.join(args.data_root, "feat", data.used_shapes[i] + ".mat")
            return lambda: (data.verts_list[i].numpy(), np.asarray(sio.loadmat(path)["feat"], dtype=np.float32))
        for i1, i2 in data.combinations:
            yield data.used_shapes[i1], data.used_shapes[i2], loader(i1), loader(i2)
    elif args.pairs:
        for path in args.pairs:
            d = np.load(path, allow_pickle=False)
            yield (str(d["name1"]), str(d["name2"]), (lambda d=d: (d["verts1"], d["dino1"])), (lambda d=d: (d["verts2"], d["dino2"])))
    else:
        g = torch.Generator().manual_seed(args.seed)
        for p in range(args.synthetic):
            v1, v2 = torch.rand(args.points, 3, generator=g), torch.rand(args.points, 3, generator=g)
            d1, d2 = torch.randn(args.points, 1152, generator=g), torch.randn(args.points, 1152, generator=g)
            yield ("s%03da" % p, "s%03db" % p, (lambda v=v1, d=d1: (v.numpy(), d.numpy())), (lambda v=v2, d=d2: (v.numpy(), d.numpy())))


def write_results(save_path, name1, name2, T12, T21, feat1, feat2):
    """File names and formats of test.py:110-133."""
    import scipy.io
    tdir, fdir = os.path.join(save_path, "T"), os.path.join(save_path, "feature")
    os.makedirs(tdir, exist_ok=True)
    os.makedirs(fdir, exist_ok=True)
    np.savetxt(os.path.join(tdir, "T_%s_%s.txt" % (name1, name2)), T12, fmt='%i')
    np.savetxt(os.path.join(tdir, "T_%s_%s.txt" % (name2, name1)), T21, fmt='%i')
    scipy.io.savemat(os.path.join(fdir, "usefeature_%s.mat" % name1), {'uphi': feat1})
    scipy.io.savemat(os.path.join(fdir, "usefeature_%s.mat" % name2), {'uphi': feat2})


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", nargs="*", default=None, help=".npz files, one pair each")
    ap.add_argument("--data-root", default=None, help="dataset directory in the reference's layout (shapes_test/, feat/)")
    ap.add_argument("--data-name", default="scape_r")
    ap.add_argument("--synthetic", type=int, default=2)
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--ckpt", default=None, help="ep_val_best.pth (Uni3FC state_dict); default: random init")
    ap.add_argument("--out", default="result/dvmatcher_amd_synthetic")
    ap.add_argument("--no-feature-cache", action="store_true", help="recompute both shapes' features for every pair, like test.py")
    args = ap.parse_args(argv)
    assert torch.cuda.is_available(), "the inference path needs a HIP device"
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    net = Uni3FC(k=40).to(dev)
    if args.ckpt:
        net.load_state_dict(torch.load(args.ckpt, map_location=dev))
    net.eval()
    n, t0 = 0, time.perf_counter()
    feats = {}  # name -> (B=1) features: test.py re-runs the backbone for both shapes of every ordered pair; the eval-mode
    #             forward of one shape does not depend on its partner, so S forwards serve all S(S-1) pairs

    def features(name, load):
        if name not in feats:
            v, d = load()
            verts = torch.from_numpy(v).float().to(dev)[None]
            dino = torch.from_numpy(d).float().to(dev)[None]
            feats[name] = net(verts.permute(0, 2, 1), dino, None)[0]
        return feats[name]

    with torch.no_grad():
        for name1, name2, load1, load2 in load_pairs(args):
            if args.no_feature_cache:
                feats.clear()
            feat1, feat2 = features(name1, load1), features(name2, load2)
            T12, T21 = (t.long() + 1 for t in ops.argmin_pair(feat1, feat2))     # search_t both ways, 1-based like test.py:19-23
            write_results(args.out, name1, name2, T12.cpu().squeeze(0).numpy(), T21.cpu().squeeze(0).numpy(),
                          feat1.cpu().squeeze(0).numpy(), feat2.cpu().squeeze(0).numpy())
            n += 1
    torch.cuda.synchronize()
    print(json.dumps({"pairs": n, "seconds": time.perf_counter() - t0, "out": args.out}))


if __name__ == "__main__":
    main()
