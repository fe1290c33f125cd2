#!/usr/bin/env python3
"""Thin training driver reproducing the reference's step (train.py:93-112) on the MI355X path.

    feat1,_ = Uni3FC(verts1^T, dino1, upsampler); feat2,_ = Uni3FC(verts2^T, dino2, upsampler)
    loss,... = criterion(feat1, feat2, dist1, dist2, verts1, verts2, alpha_i, deformer)
    loss.backward(); optimizer.step(); optimizer.zero_grad()

with the reference's hyper-parameters read from its YAML (config/scape_r.yaml layout) and
synthetic pairs (the dataset / DINO feature pipeline is outside this path: SURVEY §8f).  One process
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
    args = ap.parse_args()
    cfg = DEFAULT_CFG
    if args.config:
        import yaml
        cfg = yaml.safe_load(open(args.config))
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    assert torch.cuda.is_available(), "the training path needs a HIP device"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    Bg = args.batch or cfg["training"]["batch_size"]
    lo, hi = shard_range(Bg, rank, world)
    B, N = hi - lo, args.points
    L = cfg["loss"]
    torch.manual_seed(0)  # identical initial weights on every rank
    net, dfm = Uni3FC(k=40).to(dev), Deformer(k=L["k_deform"]).to(dev)
    params = list(net.parameters()) + list(dfm.parameters())
    opt = torch.optim.Adam(params, lr=float(cfg["optimizer"]["lr"]), betas=(cfg["optimizer"]["b1"], cfg["optimizer"]["b2"]))
    crit = GraphDeformLoss_Neural(k_deform=L["k_deform"], w_dist=L["w_dist"], w_map=L["w_map"], k_dist=min(L["k_dist"], N // 2),
                                  N_dist=min(L["N_dist"], N // 2), partial=L["partial"], w_deform=L["w_deform"],
                                  w_img=L["w_img"], w_rank=L["w_rank"], w_self_rec=L["w_self_rec"], w_cd=L["deform"]["w_cd"],
                                  w_arap=L["deform"]["w_arap"], save_name=cfg["expname"])
    alpha = np.linspace(L["min_alpha"], L["max_alpha"] + 1, cfg["training"]["epochs"])[args.epoch - 1]
    bucket = FlatGradBucket(params)
    g = torch.Generator().manual_seed(100 + rank)
    random.seed(200 + rank)
    torch.manual_seed(300 + rank)
    v1, v2 = torch.rand(B, N, 3, generator=g).to(dev), torch.rand(B, N, 3, generator=g).to(dev)
    d1, d2 = torch.randn(B, N, 1152, generator=g).to(dev), torch.randn(B, N, 1152, generator=g).to(dev)
    dist1, dist2 = torch.cdist(v1, v1), torch.cdist(v2, v2)
    net.train()
    dfm.train()
    losses = []

    def step():
        f1, _ = net(v1.permute(0, 2, 1), d1, None)
        f2, _ = net(v2.permute(0, 2, 1), d2, None)
        out = crit(f1, f2, dist1, dist2, v1, v2, alpha, dfm)
        out[0].backward()
        bucket.all_reduce_mean()
        opt.step()
        opt.zero_grad()
        return [float(torch.as_tensor(o).detach()) for o in out]

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
