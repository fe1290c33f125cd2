#!/usr/bin/env python3
"""Thin training driver reproducing the reference's step (train.py:93-112) on the MI355X path.

    feat1,_ = Uni3FC(verts1^T, dino1, upsampler); feat2,_ = Uni3FC(verts2^T, dino2, upsampler)
    loss,... = criterion(feat1, feat2, dist1, dist2, verts1, verts2, alpha_i, deformer)
    loss.backward(); optimizer.step(); optimizer.zero_grad()

with the reference's hyper-parameters read from its YAML (config/scape_r.yaml layout) and either
synthetic pairs or a dataset directory in the reference's layout (`--data-root`: models/dataset.py
reads `shapes_train/*.off` or its `.pt` cache, and `feat/<shape>.mat` visual features; producing those
features is outside this path).  One process
per GPU; with WORLD_SIZE > 1 the pair batch is sharded and the gradients are averaged with ONE
all-reduce over a flat fp32 bucket (RCCL over xGMI).  BatchNorm uses local batch statistics and the
positional encoding the local min/max (SURVEY §8e caveats).

  python dv-matcher_amd/train_driver.py --steps 5 --batch 2 --points 1024
  python -m torch.distributed.run --nproc-per-node 8 dv-matcher_amd/train_driver.py --batch 8 --points 2048
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

from dvm.dist import FlatGradBucket, shard_range  # noqa: E402
from models.loss import GraphDeformLoss_Neural  # noqa: E402
from models.model import Deformer, Uni3FC  # noqa: E402

DEFAULT_CFG = {  # the values of the reference's config/scape_r.yaml
    "expname": "dvmatcher_scape_r_std",
    "optimizer": {"lr": 2e-3, "b1": 0.9, "b2": 0.99, "decay_iter": 10, "decay_factor": 0.5},
    "training": {"batch_size": 2, "epochs": 20},
    "loss": {"k_deform": 10, "k_dist": 500, "N_dist": 1000, "partial": False, "min_alpha": 10, "max_alpha": 100,
             "w_dist": 0.02, "w_map": 0.005, "w_deform": 0.5, "w_self_rec": 0.5, "w_rank": 0, "w_img": 0,
             "deform": {"w_cd": 0.1, "w_arap": 0.01}},
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=None, help="a reference-style YAML (config/scape_r.yaml); default: its shipped values")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=None, help="global pair batch (default: training.batch_size)")
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--epoch", type=int, default=1, help="which epoch's alpha to use (1-based)")
    ap.add_argument("--sync-stats", action="store_true", help="DDP: BatchNorm statistics and the positional encoding's min/max "
                    "over the GLOBAL batch (SyncBatchNorm + two scalar all-reduces), i.e. the single-process semantics")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--data-root", default=None, help="dataset directory (shapes_train/, feat/, cache_*.pt); default: synthetic")
    ap.add_argument("--data-name", default="scape_r")
    ap.add_argument("--random-feat", action="store_true", help="with --data-root: random visual features instead of feat/*.mat")
    args = ap.parse_args()
    cfg = DEFAULT_CFG
    if args.config:
        import yaml
        cfg = yaml.safe_load(open(args.config))
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    assert torch.cuda.is_available(), "the training path needs a HIP device"
    local = local % torch.cuda.device_count()   # more ranks than devices (gloo tests on one GPU): share them round-robin
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, **({"device_id": dev} if args.backend == "nccl" else {}))
    Bg = args.batch or cfg["training"]["batch_size"]
    lo, hi = shard_range(Bg, rank, world)
    B, N = hi - lo, args.points
    L = cfg["loss"]
    torch.manual_seed(0)  # identical initial weights on every rank
    net, dfm = Uni3FC(k=40).to(dev), Deformer(k=L["k_deform"]).to(dev)
    if args.sync_stats and world > 1:
        net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)
        net.sync_minmax = True
    params = list(net.parameters()) + list(dfm.parameters())
    opt = torch.optim.Adam(params, lr=float(cfg["optimizer"]["lr"]), betas=(cfg["optimizer"]["b1"], cfg["optimizer"]["b2"]))
    crit = GraphDeformLoss_Neural(k_deform=L["k_deform"], w_dist=L["w_dist"], w_map=L["w_map"], k_dist=min(L["k_dist"], N // 2),
                                  N_dist=min(L["N_dist"], N // 2), partial=L["partial"], w_deform=L["w_deform"],
                                  w_img=L["w_img"], w_rank=L["w_rank"], w_self_rec=L["w_self_rec"], w_cd=L["deform"]["w_cd"],
                                  w_arap=L["deform"]["w_arap"], save_name=cfg["expname"])
    alpha = np.linspace(L["min_alpha"], L["max_alpha"] + 1, cfg["training"]["epochs"])[args.epoch - 1]
    # world > 1: every p.grad is a view into one flat buffer (one all-reduce, no pack/unpack); a single rank has nothing to
    # exchange and lets autograd hand Adam its gradient tensors directly (no accumulate-into-zeros adds either)
    bucket = FlatGradBucket(params, attach=world > 1)
    g = torch.Generator().manual_seed(100 + rank)
    random.seed(200 + rank)
    torch.manual_seed(300 + rank)
    if args.data_root:
        from models.dataset import Dataset
        data = Dataset(args.data_root, name=args.data_name, train=True, with_dino=not args.random_feat, feat_mat=True)
        order = torch.randperm(len(data), generator=torch.Generator().manual_seed(7)).tolist()      # same on every rank

        def batches():
            at = 0
            while True:                                      # this rank's slice [lo, hi) of every global batch
                items = [data[order[(at + j) % len(order)]] for j in range(lo, hi)]
                at += Bg
                cols = []
                for s in ("shape1", "shape2"):
                    n = min(it[s]["xyz"].shape[0] for it in items)
                    n = min(n, N)
                    xyz = torch.stack([it[s]["xyz"][:n] for it in items]).float().to(dev)
                    dd = torch.stack([it[s]["dist"][:n, :n] for it in items]).float().to(dev)
                    if args.random_feat:
                        ft = torch.randn(len(items), n, 1152, generator=g).to(dev)
                    else:
                        ft = torch.stack([it[s]["feat"][:n] for it in items]).float().to(dev)
                    cols.append((xyz, ft, dd))
                yield cols[0][0], cols[1][0], cols[0][1], cols[1][1], cols[0][2], cols[1][2]
    else:
        sv1, sv2 = torch.rand(B, N, 3, generator=g).to(dev), torch.rand(B, N, 3, generator=g).to(dev)
        sd1, sd2 = torch.randn(B, N, 1152, generator=g).to(dev), torch.randn(B, N, 1152, generator=g).to(dev)
        sdist1, sdist2 = torch.cdist(sv1, sv1), torch.cdist(sv2, sv2)

        def batches():
            while True:
                yield sv1, sv2, sd1, sd2, sdist1, sdist2
    feed = batches()
    net.train()
    dfm.train()
    losses = []
    sync_each = os.environ.get("DVM_SYNC_EACH_STEP", "0") == "1"   # train.py logs loss.item() every iteration

    def step():
        v1, v2, d1, d2, dist1, dist2 = next(feed)
        f1, _ = net(v1.permute(0, 2, 1), d1, None)
        f2, _ = net(v2.permute(0, 2, 1), d2, None)
        out = crit(f1, f2, dist1, dist2, v1, v2, alpha, dfm)
        out[0].backward()
        if world > 1:
            bucket.all_reduce_mean()
        opt.step()
        if world > 1:
            bucket.zero()
        else:
            opt.zero_grad(set_to_none=True)
        vals = torch.stack([torch.as_tensor(o, device=dev).detach().float().reshape(()) for o in out])
        return vals.tolist() if sync_each else vals      # the 5 loss terms; read back after the timed loop by default

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(step())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    losses = [l if isinstance(l, list) else l.tolist() for l in losses]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    if rank == 0:
        print(json.dumps({"metric": "training pairs/sec (fwd+loss+bwd+Adam)", "value": Bg * args.steps / dt, "unit": "pairs/s",
                          "n_gpus": world, "steps": args.steps, "ms_per_step": dt / args.steps * 1e3, "global_batch": Bg,
                          "points": N, "alpha": float(alpha), "grad_bucket_floats": bucket.numel,
                          "first_losses": losses[0], "last_losses": losses[-1]}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
