#!/usr/bin/env python3
"""Training driver: the reference's training loop (train.py:75-169, train_partial.py:74-164) on the MI355X path.

Per epoch e = 1..epochs:  alpha = linspace(min_alpha, max_alpha + 1, epochs)[e-1];  lr *= decay_factor when
e % decay_iter == 0;  train pass (BatchNorm in train mode)

    feat1,_ = Uni3FC(verts1^T, dino1, upsampler); feat2,_ = Uni3FC(verts2^T, dino2, upsampler)
    loss,... = criterion(feat1, feat2, dist1, dist2, verts1, verts2, alpha, deformer)
    loss.backward(); optimizer.step(); optimizer.zero_grad()

then a no-grad validation pass in eval mode with the same alpha, and the reference's checkpoints under
ckpt/<expname>/: ep_{e}.pth + ep_deformer{e}.pth every misc.checkpoint_interval, ep_val_best.pth +
ep_deformer_val_best.pth whenever the validation loss does not get worse, ep_train_best.pth +
ep_deformer_train_best.pth every misc.log_interval iterations (state_dicts only, as the reference writes them;
test_driver.py / deform_driver.py load them).  `--partial` trains GraphDeformLoss_Neural_Partial on N != M pairs
(config/scape_partial.yaml).  Hyper-parameters come from a reference YAML (`--config`) or its shipped values.

Data: synthetic pairs (default), or a dataset directory in the reference's layout (`--data-root`, read by
models/dataset.py's Dataset / PartialDataset).

One process per GPU.  With WORLD_SIZE > 1 every global batch is sharded over the ranks and the gradients are exchanged
with ONE all-reduce (sum) over a flat fp32 bucket (RCCL over xGMI), started as soon as backward has produced them and
overlapped with the host-side bookkeeping; each rank back-propagates `criterion.data_parallel_loss(B_shard / B_global)`,
so the reduced gradient is that of the criterion on the whole batch (its dist / ARAP terms are sums over the pairs, the
rest are means).  `--sync-stats` makes BatchNorm statistics and the positional encoding's min / max batch-global too.

  python dv-matcher_amd/train_driver.py --epochs 2 --pairs-per-epoch 8 --batch 2 --points 1024 --ckpt-dir /tmp/ck
  python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048          # timing mode (one JSON line)
  python -m torch.distributed.run --nproc-per-node 8 dv-matcher_amd/train_driver.py --batch 8 --points 2048
"""
import argparse
import copy
import json
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)
from dvm import hostenv  # noqa: E402  (torch-free)
hostenv.apply_rank_host_limits()   # one of several ranks: pin to this rank's cores, size the thread pools — BEFORE torch loads

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from dvm.dist import FlatGradBucket, shard_range  # noqa: E402
from models.loss import GraphDeformLoss_Neural, GraphDeformLoss_Neural_Partial  # noqa: E402
from models.model import Deformer, Uni3FC, join_side_streams  # noqa: E402

FULL_CFG = {  # the values of the reference's config/scape_r.yaml
    "expname": "dvmatcher_scape_r_std", "with_dino": True, "feat_mat": True,
    "optimizer": {"lr": 2e-3, "b1": 0.9, "b2": 0.99, "decay_iter": 10, "decay_factor": 0.5},
    "training": {"batch_size": 2, "epochs": 20},
    "loss": {"k_deform": 10, "k_dist": 500, "N_dist": 1000, "partial": False, "min_alpha": 10, "max_alpha": 100,
             "w_dist": 0.02, "w_map": 0.005, "w_deform": 0.5, "w_self_rec": 0.5, "w_rank": 0, "w_img": 0,
             "deform": {"w_cd": 0.1, "w_arap": 0.01}},
    "misc": {"checkpoint_interval": 1, "log_interval": 5000},
}
PARTIAL_CFG = copy.deepcopy(FULL_CFG)  # ... and of config/scape_partial.yaml
PARTIAL_CFG.update(expname="dvmatcher_scape_partial", with_dino=False, feat_mat=False)
PARTIAL_CFG["training"]["batch_size"] = 5
PARTIAL_CFG["loss"].update(k_dist=300, N_dist=500, partial=True, w_deform=1000, w_self_rec=1000)


def rank_env():
    """(world, rank, local_rank) of this process under torch.distributed.run (1, 0, 0 when run alone)."""
    return tuple(int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))


def pick_device(local_rank, n_devices):
    """One process per GPU: LOCAL_RANK selects the device; more ranks than devices (the gloo tests on one GPU) share them
    round-robin."""
    if n_devices < 1:
        raise RuntimeError("the training path needs a HIP device")
    return local_rank % n_devices


def rank_seeds(rank):
    """Distinct, reproducible RNG streams per rank: (data generator, python `random`, torch global)."""
    return 100 + rank, 200 + rank, 300 + rank


def alpha_schedule(cfg):
    return np.linspace(cfg["loss"]["min_alpha"], cfg["loss"]["max_alpha"] + 1, cfg["training"]["epochs"])


def lr_at_epoch(cfg, epoch):
    """The learning rate in force DURING epoch `epoch` (1-based): multiplied by decay_factor at the start of every epoch
    divisible by decay_iter (train.py:78-82)."""
    return float(cfg["optimizer"]["lr"]) * float(cfg["optimizer"]["decay_factor"]) ** (epoch // int(cfg["optimizer"]["decay_iter"]))


def ckpt_paths(ckpt_dir, expname, tag):
    """The reference's file names (train.py:40-42, 124-126, 161-169): tag in {'val_best', 'train_best', <epoch>}."""
    d = os.path.join(ckpt_dir, str(expname))
    dtag = "deformer_%s" % tag if isinstance(tag, str) else "deformer%d" % tag
    return os.path.join(d, "ep_%s.pth" % tag), os.path.join(d, "ep_%s.pth" % dtag)


def save_ckpt(net, dfm, ckpt_dir, expname, tag):
    pb, pd = ckpt_paths(ckpt_dir, expname, tag)
    os.makedirs(os.path.dirname(pb), exist_ok=True)
    torch.save(net.state_dict(), pb)
    torch.save(dfm.state_dict(), pd)
    return pb, pd


class SyntheticPairs:
    """`pairs` fixed random pairs per split, N source / M target points, seeded per rank; dist = Euclidean cdist."""

    def __init__(self, pairs, N, M, seed, dev):
        g = torch.Generator().manual_seed(seed)
        self.v1, self.v2 = torch.rand(pairs, N, 3, generator=g).to(dev), torch.rand(pairs, M, 3, generator=g).to(dev)
        self.d1, self.d2 = torch.randn(pairs, N, 1152, generator=g).to(dev), torch.randn(pairs, M, 1152, generator=g).to(dev)
        self.dist1, self.dist2 = torch.cdist(self.v1, self.v1), torch.cdist(self.v2, self.v2)
        self.pairs = pairs

    def batch(self, idx):
        i = torch.as_tensor(idx, device=self.v1.device)
        self.last_names = (["syn-s%d" % int(j) for j in idx], ["syn-t%d" % int(j) for j in idx])   # identities of the shapes (--graph-cache)
        return self.v1[i], self.v2[i], self.d1[i], self.d2[i], self.dist1[i], self.dist2[i]


class DatasetPairs:
    """Pairs of a dataset directory (models/dataset.py), truncated to the first `N` / `M` points of each shape."""

    def __init__(self, data, N, M, dev, random_feat, seed):
        self.data, self.N, self.M, self.dev, self.random_feat = data, N, M, dev, random_feat
        self.g = torch.Generator().manual_seed(seed)
        self.pairs = len(data)

    def batch(self, idx):
        items = [self.data[i] for i in idx]
        # identities of the batch's shapes for --graph-cache: the dataset's shape names — or None where an item's coordinates are
        # drawn anew on every access (PartialDataset's random views): such shapes must not be cached
        self.last_names = None if getattr(self.data, "resamples_coordinates", False) else \
            ([str(it["shape1"]["name"]) for it in items], [str(it["shape2"]["name"]) for it in items])
        out = []
        for s, cap in (("shape1", self.N), ("shape2", self.M)):
            n = min(min(it[s]["xyz"].shape[0] for it in items), cap)
            xyz = torch.stack([it[s]["xyz"][:n] for it in items]).float().to(self.dev)
            dd = torch.stack([it[s]["dist"][:n, :n] for it in items]).float().to(self.dev)
            if self.random_feat:
                ft = torch.randn(len(items), n, 1152, generator=self.g).to(self.dev)
            elif "feat" not in items[0][s]:
                ft = None   # with_dino = False (train_partial.py:98-104): Uni3FC computes them through the upsampler
            else:
                ft = torch.stack([it[s]["feat"][:n] for it in items]).float().to(self.dev)
            out.append((xyz, ft, dd))
        return out[0][0], out[1][0], out[0][1], out[1][1], out[0][2], out[1][2]


def global_batches(n_pairs, Bg, shuffle_seed=None, keep_tail=False):
    """Index lists of the global batches of one pass.  The reference's DataLoader keeps the last, smaller batch
    (train.py:60-66, drop_last defaults to False): keep_tail=True does the same; a sharded run drops it, because a batch
    smaller than the world size cannot be split."""
    order = list(range(n_pairs))
    if shuffle_seed is not None:
        order = torch.randperm(n_pairs, generator=torch.Generator().manual_seed(shuffle_seed)).tolist()
    stop = n_pairs if keep_tail else n_pairs - Bg + 1
    return [order[i:i + Bg] for i in range(0, stop, Bg)]


def fit_criterion(crit, cfg, n_points):
    """Anchor / neighbour counts of the dist term for a batch of `n_points` points per shape: the configured values, clamped
    to what the batch holds (DatasetPairs truncates a batch to its smallest shape; random.sample(range(n), N_dist) would raise)."""
    crit.k_dist = min(int(cfg["loss"]["k_dist"]), max(1, n_points // 2))
    crit.N_dist = min(int(cfg["loss"]["N_dist"]), max(1, n_points // 2))


def step_flops(B, N, M):
    """Algorithmic flops of the matrix work of one training step (what the matrix cores are asked to do, not what the
    kernels issue): LG-Net's 1x1 convs forward + dX + dW (3 x 2 per multiply-add), its 7 feature-space kNN score matrices
    and the 4 SA energy / apply products forward (+ 2.5 x for their tile-recompute backward), the soft-correspondence
    distance matrix forward + backward, for both shapes of every pair.  models/model.py:480-761, models/loss.py:1339-1410."""
    conv_macs = 1152 * 384 + 384 * 64 + 4 * (64 * 192 + 2 * 64 * 256 + 80 * 64 + 64 * 64) + 2 * 256 * 512 + 2 * 768 * 128 + \
        256 * 128 + 3 * (128 * 384 + 2 * 128 * 512) + 512 * 128
    total = 0.0
    for n in (N, M):
        total += 3 * 2.0 * B * n * conv_macs                       # convs: forward, dX, dW
        total += 2.0 * B * n * n * (4 * 64 + 3 * 128)              # kNN scores (forward only: indices carry no gradient)
        total += 4 * 2.0 * B * n * n * (16 + 64) * 3.5             # SA: energy (16) + apply (64), forward + recompute backward
    total += 2.0 * B * N * M * 128 * (1 + 2.5)                     # soft correspondence: distances forward, recompute backward
    return total


def loss_values(out, dev):
    """The criterion's loss terms as one device vector.  Terms switched off by their weight are Python numbers: they become
    device scalars through a fill kernel — torch.as_tensor(0, device=...) is a copy from pageable memory, which blocks the
    host until the stream has drained, i.e. once per step behind the whole forward + backward."""
    return torch.stack([o.detach().float().reshape(()) if torch.is_tensor(o) and o.is_cuda else
                        torch.full((), float(o), dtype=torch.float32, device=dev) for o in out])


def build_criterion(cfg, partial, n_points):
    L = cfg["loss"]
    cls = GraphDeformLoss_Neural_Partial if partial else GraphDeformLoss_Neural
    # anchors / neighbours are drawn among the points that actually exist (clamp from the real count)
    k_dist, n_dist = min(int(L["k_dist"]), n_points // 2), min(int(L["N_dist"]), n_points // 2)
    return cls(k_deform=L["k_deform"], w_dist=L["w_dist"], w_map=L["w_map"], k_dist=k_dist, N_dist=n_dist, partial=L["partial"],
               w_deform=L["w_deform"], w_img=L["w_img"], w_rank=L["w_rank"], w_self_rec=L["w_self_rec"], w_cd=L["deform"]["w_cd"],
               w_arap=L["deform"]["w_arap"], save_name=cfg["expname"])


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=None, help="a reference-style YAML (config/scape_r.yaml / scape_partial.yaml); default: its shipped values")
    ap.add_argument("--partial", action="store_true", help="GraphDeformLoss_Neural_Partial on N != M pairs (train_partial.py)")
    ap.add_argument("--epochs", type=int, default=0, help="run the full loop for this many epochs (0: timing mode, see --steps)")
    ap.add_argument("--pairs-per-epoch", type=int, default=None, help="synthetic data: pairs per training pass (default 4 global batches)")
    ap.add_argument("--val-pairs", type=int, default=None, help="synthetic data: pairs of the validation pass (default 1 global batch)")
    ap.add_argument("--ckpt-dir", default="ckpt", help="checkpoints go to <ckpt-dir>/<expname>/ (reference names)")
    ap.add_argument("--steps", type=int, default=5, help="timing mode: timed steps")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=None, help="GLOBAL pair batch (default: training.batch_size)")
    ap.add_argument("--points", type=int, default=1024, help="source points N")
    ap.add_argument("--points-target", type=int, default=None, help="target points M (default: N; with --partial 0.44 N like 4995 / 2200)")
    ap.add_argument("--epoch", type=int, default=1, help="timing mode: which epoch's alpha / lr to use (1-based)")
    ap.add_argument("--graph", action="store_true", help="timing mode, one rank: capture the whole step (forward, criterion, backward, "
                    "Adam) into ONE HIP graph and replay it — ~2000 kernel launches per step become one graph launch")
    ap.add_argument("--sync-stats", action="store_true", help="DDP: BatchNorm statistics and the positional encoding's min/max "
                    "over the GLOBAL batch (SyncBatchNorm + two scalar all-reduces)")
    ap.add_argument("--graph-cache", action="store_true", help="opt-in per-shape deformation-graph cache (SURVEY 8f-2): every shape gets ONE "
                    "fixed FPS start index (from its id) instead of the reference's fresh random start per call (models/loss.py:1325-1337), and its "
                    "graph / xyz kNN are built once and reused whenever the shape comes up again")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--dist-always", action="store_true", help="initialise the process group, the flat-bucket all-reduce and the barriers "
                    "at world size 1 too (the RCCL branch then runs on a 1-GPU box: tests/test_gpu_ddp.py)")
    ap.add_argument("--sync-each-step", action="store_true", help="read the loss values back every step (train.py logs loss.item() every iteration)")
    ap.add_argument("--step-breakdown", action="store_true", help="timing mode: host seconds per phase of the step on stderr")
    ap.add_argument("--data-root", default=None, help="dataset directory (shapes_train/, shapes_test/, feat/, cache_*.pt); default: synthetic")
    ap.add_argument("--data-name", default=None)
    ap.add_argument("--random-feat", action="store_true", help="with --data-root: random visual features instead of feat/*.mat")
    ap.add_argument("--upsampler-weights", default=None, help="state_dict of the FeatUp / DINOv2 image backbone (hub layout) for "
                    "configurations with with_dino = False, whose visual features are rendered and back-projected on the fly "
                    "(train_partial.py:72, 98-104); default: the seeded random initialisation (parity unpinned)")
    args = ap.parse_args(argv)
    cfg = copy.deepcopy(PARTIAL_CFG if args.partial else FULL_CFG)
    if args.config:
        import yaml
        cfg = yaml.safe_load(open(args.config))
        cfg.setdefault("misc", {}).setdefault("checkpoint_interval", 1)
        cfg["misc"].setdefault("log_interval", 5000)
    partial = args.partial or bool(cfg["loss"].get("partial"))
    world, rank, local = rank_env()
    local = pick_device(local, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # --dist-always: the process group, the flat-bucket all-reduce and the barriers also at world size 1 — the RCCL branch
    # of this driver then runs on a 1-GPU box (tests/test_gpu_ddp.py) instead of first on the 8-GPU node
    dist_on = world > 1 or args.dist_always
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            # a port of its own only for the single-rank case (--dist-always): with more ranks every rank would pick a DIFFERENT
            # free port and the rendezvous would hang until the store timeout — the launcher has to export one
            if world > 1:
                raise SystemExit("train_driver: WORLD_SIZE = %d but MASTER_PORT is not set (start the ranks with "
                                 "`python -m torch.distributed.run --master-addr 127.0.0.1 --master-port P ...` or export it)" % world)
            import socket
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        dist.init_process_group(args.backend, rank=rank, world_size=world, **({"device_id": dev} if args.backend == "nccl" else {}))
    Bg = args.batch or int(cfg["training"]["batch_size"])
    if Bg < world:
        raise SystemExit("global batch %d < world size %d: every rank needs at least one pair" % (Bg, world))
    lo, hi = shard_range(Bg, rank, world)
    N = args.points
    M = args.points_target or (max(32, int(round(N * 2200 / 4995))) if partial else N)
    seed_data, seed_py, seed_torch = rank_seeds(rank)
    torch.manual_seed(0)  # identical initial weights on every rank
    net, dfm = Uni3FC(k=40).to(dev), Deformer(k=cfg["loss"]["k_deform"]).to(dev)
    if args.sync_stats and dist_on:
        # batch-global BatchNorm statistics and position-encoding range, as in the reference's single-process batch
        # (models/model.py:496-503, 548).  The native training node takes a collective for them (dvm_uni3fc_train_*_sync_f32: the
        # per-channel totals and the min / max are all-reduced between its launches); the autograd paths need converted modules.
        if net.native_train and net.point_major_train:
            from dvm.dist import TorchCollective
            net.sync_stats = TorchCollective()
        else:
            net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)
        net.sync_minmax = True      # (the eval / autograd paths' position encoding)
    params = list(net.parameters()) + list(dfm.parameters())
    use_graph = bool(args.graph) and world == 1 and args.epochs <= 0
    # one fused multi-tensor Adam launch per step (same update as the reference's torch.optim.Adam, train.py:44-45; the default
    # foreach form costs the host ~1.5 ms per step in ~10 launches over 151 tensors; a captured step needs the capturable form)
    fused_adam = not use_graph
    opt = torch.optim.Adam(params, lr=float(cfg["optimizer"]["lr"]), betas=(cfg["optimizer"]["b1"], cfg["optimizer"]["b2"]),
                           **({"fused": True} if fused_adam else {"capturable": use_graph}))
    # world > 1: every p.grad is a view into one flat buffer (one all-reduce, no pack/unpack); a single rank has nothing to
    # exchange and lets autograd hand Adam its gradient tensors directly — unless the step is captured into a graph, whose
    # gradient tensors must keep their addresses.
    attach = dist_on or use_graph
    bucket = FlatGradBucket(params, attach=attach)
    random.seed(seed_py)
    torch.manual_seed(seed_torch)
    timing = args.epochs <= 0
    if args.data_root:
        from models import dataset as ds
        name = args.data_name or ("scape_partial" if partial else "scape_r")
        cls = ds.PartialDataset if partial else ds.Dataset
        kw = dict(with_dino=bool(cfg.get("with_dino", True)) and not args.random_feat, feat_mat=bool(cfg.get("feat_mat", True)))
        train_set = DatasetPairs(cls(args.data_root, name=name, train=True, **kw), N, M, dev, args.random_feat, seed_data)
        try:
            val_set = DatasetPairs(cls(args.data_root, name=name, train=False, **kw), N, M, dev, args.random_feat, seed_data + 1)
        except Exception:  # noqa: BLE001  (a dataset directory without a test split: validate on the training pairs)
            val_set = train_set
    else:
        n_train = args.pairs_per_epoch or (Bg if timing else 4 * Bg)
        train_set = SyntheticPairs(n_train, N, M, seed_data if timing else 1000, dev)   # full loop: every rank holds the same pairs
        val_set = train_set if timing else SyntheticPairs(args.val_pairs or Bg, N, M, 2000, dev)
    crit = build_criterion(cfg, partial, min(N, M))
    if args.graph_cache:
        import collections
        crit.graph_cache = collections.OrderedDict()   # least-recently-used shapes are dropped beyond crit.graph_cache_max
    frac = (hi - lo) / Bg
    sync_each = args.sync_each_step   # train.py logs loss.item() every iteration

    from dvm import nn_ops
    nn_ops.fuse_grad_accumulation(True)   # .backward() only below, never autograd.grad()

    # with_dino = False on a dataset without stored features: the reference passes dino_feat = None and Uni3FC renders the
    # shapes, runs the image backbone and back-projects (train_partial.py:98-104, models/model.py:683-710).  Built only then.
    upsampler = None
    if args.data_root and not args.random_feat and not bool(cfg.get("with_dino", True)):
        from models.image_backbone import load_upsampler
        upsampler = load_upsampler(use_norm=True, weights=args.upsampler_weights, device=dev)
        if rank == 0 and not args.upsampler_weights:
            print("train_driver: with_dino = False and no --upsampler-weights: the image backbone is RANDOMLY initialised "
                  "(pass the FeatUp / DINOv2 state_dict, or --random-feat for explicit noise features)", file=sys.stderr)

    def forward_pair(v1, d1, v2, d2):
        # two calls, as in the reference (train-mode BatchNorm statistics are per call); inside each, LG-Net's local and
        # global chains already run on two streams (models/model.py:_two_branches)
        if (d1 is None or d2 is None) and upsampler is None:
            raise RuntimeError("the batch carries no visual features and no image backbone was built (use --random-feat for noise)")
        if d1 is not None and d2 is not None and hasattr(net, "forward_pair"):   # (SyncBatchNorm-converted nets keep the method: it falls back itself)
            (f1, _), (f2, _) = net.forward_pair(v1.permute(0, 2, 1), d1, v2.permute(0, 2, 1), d2, upsampler)
            return f1, f2
        return net(v1.permute(0, 2, 1), d1, upsampler)[0], net(v2.permute(0, 2, 1), d2, upsampler)[0]

    host_marks = [0.0, 0.0, 0.0, 0.0, 0] if args.step_breakdown else None   # host seconds per phase

    def mark(i, t0):
        if host_marks is not None:
            host_marks[i] += time.perf_counter() - t0
        return time.perf_counter()

    # the criterion's geometry (graphs: FPS is a 0.65 ms chain of dependent steps; xyz kNN) needs the coordinates only: it is
    # enqueued on its own stream BEFORE the network forward and joined in front of the criterion
    geo_stream = torch.cuda.Stream()

    def cached_keys(names, n1, n2):
        """--graph-cache: (fixed FPS starts, shape ids) of a batch from the NAMES of its shapes (`names` = (sources, targets), as the
        pair set recorded them for the batch; None = these coordinates are not cacheable).  A shape's start index is a hash of its
        name, so the same shape gets the same graph in every pair and every epoch; the cache key is (name, start, point count)."""
        if names is None:
            return None, None
        import zlib
        s1 = torch.tensor([zlib.crc32(("s:" + nm).encode()) % n1 for nm in names[0]], dtype=torch.long)
        s2 = torch.tensor([zlib.crc32(("s:" + nm).encode()) % n2 for nm in names[1]], dtype=torch.long)
        return (s1, s2), (list(names[0]), list(names[1]))

    def prefetch_geometry(v1, v2, starts=None, shape_ids=None):
        if geo_stream is None:
            return None
        geo_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(geo_stream), torch.no_grad():
            return crit.geometry(v1, v2, starts, shape_ids)

    def join_geometry(geo):
        if geo is None:
            return None
        cur = torch.cuda.current_stream()
        cur.wait_stream(geo_stream)
        for part in geo:
            for t in (part.values() if isinstance(part, dict) else (part,)):
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(cur)
        return geo

    def train_step(batch, alpha, pair_ids=None):
        v1, v2, d1, d2, dist1, dist2 = batch
        fit_criterion(crit, cfg, min(v1.shape[1], v2.shape[1]))
        t = time.perf_counter()
        starts, shape_ids = cached_keys(pair_ids, v1.shape[1], v2.shape[1]) if (args.graph_cache and pair_ids is not None) else (None, None)   # pair_ids: the batch's shape names
        geo = prefetch_geometry(v1, v2, starts, shape_ids)
        f1, f2 = forward_pair(v1, d1, v2, d2)
        t = mark(0, t)
        out = crit(f1, f2, dist1, dist2, v1, v2, alpha, dfm, fps_starts=starts, geometry=join_geometry(geo), shape_ids=shape_ids)
        t = mark(1, t)
        if dist_on:
            crit.data_parallel_loss(frac).backward()
            join_side_streams(dev)                           # gradients written from the helper-stream chain are complete
            work = bucket.all_reduce_sum(async_op=True, always=True)   # one 8.5 MB collective on RCCL's stream ...
        else:
            out[0].backward()
            join_side_streams(dev)
            work = None
        t = mark(2, t)
        # ... overlapped with the host-side bookkeeping of the step (5 loss terms; local values, as the reference logs them)
        vals = loss_values(out, dev)
        if work is not None:
            work.wait()
        opt.step()
        t = mark(3, t)
        if host_marks is not None:
            host_marks[4] += 1
        if attach:
            bucket.zero()
        else:
            opt.zero_grad(set_to_none=True)
        return vals.tolist() if sync_each else vals

    def shard(idx):
        return idx[lo:hi]

    # ------------------------------------------------------------------ timing mode: K steps on one resident batch
    if timing:
        alpha = alpha_schedule(cfg)[args.epoch - 1]
        for grp in opt.param_groups:
            grp["lr"] = lr_at_epoch(cfg, args.epoch)
        net.train()
        dfm.train()
        if args.data_root:
            feed_ids = [shard(b) for b in global_batches(train_set.pairs, Bg)[:max(1, args.steps)]]
        else:   # every rank owns its own Bg/world pairs (weak-scaling shape of the forward bench)
            feed_ids = [list(range(lo, hi)) if train_set.pairs >= Bg else list(range(hi - lo))]
        feed, feed_names = [], []
        for ix in feed_ids:
            feed.append(train_set.batch(ix))
            feed_names.append(train_set.last_names)
        losses = []
        if use_graph:
            # every draw the criterion makes on the host (dist-loss anchors, FPS starts) becomes a device-resident input of
            # the captured step; a training loop would refresh them with copy_() between replays
            batch = feed[0]
            n1, n2 = batch[0].shape[1], batch[1].shape[1]
            anchors = tuple(torch.as_tensor(random.sample(range(n), crit.N_dist), device=dev) for n in (n1, n2))
            starts = tuple(torch.randint(0, n, (hi - lo,), device=dev) for n in (n1, n2))

            def graph_step():
                v1, v2, d1, d2, dist1, dist2 = batch
                geo = prefetch_geometry(v1, v2, starts)
                f1, f2 = forward_pair(v1, d1, v2, d2)
                out = crit(f1, f2, dist1, dist2, v1, v2, alpha, dfm, fps_starts=starts, anchors=anchors, geometry=join_geometry(geo))
                out[0].backward()
                join_side_streams(dev)
                vals = loss_values(out, dev)
                opt.step()
                bucket.zero()
                return vals

            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(3, args.warmup)):
                    graph_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_vals = graph_step()

            def train_step(_batch, _alpha, _ids=None):   # noqa: F811  (the replay stands in for the eager step)
                graph.replay()
                return static_vals.clone()
        for i in range(args.warmup):
            train_step(feed[i % len(feed)], alpha, feed_names[i % len(feed)])
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        if host_marks is not None:
            host_marks[:] = [0.0, 0.0, 0.0, 0.0, 0]
        t0 = time.perf_counter()
        for i in range(args.steps):
            losses.append(train_step(feed[i % len(feed)], alpha, feed_names[i % len(feed)]))
        t_host = time.perf_counter() - t0          # all steps enqueued; the rest of dt is the GPU catching up
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        dt = time.perf_counter() - t0
        losses = [l if isinstance(l, list) else l.tolist() for l in losses]
        if dist_on:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        if rank == 0 and host_marks is not None and host_marks[4]:
            print("host ms per step (enqueue only): network forward x2 %.2f, criterion %.2f, backward %.2f, bookkeeping + Adam %.2f"
                  % tuple(1e3 * v / host_marks[4] for v in host_marks[:4]), file=sys.stderr)
        if rank == 0:
            print(json.dumps({"metric": "training pairs/sec (fwd+loss+bwd+Adam)", "value": Bg * args.steps / dt, "unit": "pairs/s",
                              "n_gpus": world, "steps": args.steps, "ms_per_step": dt / args.steps * 1e3,
                              "host_enqueue_ms_per_step": t_host / args.steps * 1e3, "global_batch": Bg,
                              "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": 157.3, "peak_name": "fp32 matrix",
                                           "achieved": step_flops(Bg, N, M) / (dt / args.steps) / 1e12 / world,
                                           "frac": step_flops(Bg, N, M) / (dt / args.steps) / 1e12 / world / 157.3,
                                           "flops_per_step": step_flops(Bg, N, M),
                                           "note": "algorithmic matrix flops of the whole step per GPU over the step time; the step is "
                                                   "~900 small launches, bound by the latency of its kernel chain (network forward -> criterion -> backward), not by the matrix pipe"},
                              "points": N, "points_target": M, "criterion": type(crit).__name__, "alpha": float(alpha),
                              "hip_graph": use_graph, "graph_cache": bool(args.graph_cache),
                              "sync_stats": (None if not (args.sync_stats and dist_on) else
                                             {"native_node": getattr(net, "sync_stats", None) is not None,
                                              "native_calls": net.__dict__.get("native_train_calls", 0),
                                              "collectives": getattr(getattr(net, "sync_stats", None), "calls", None)}), "process_group": (dist.get_backend() if dist_on else None),
                              "grad_bucket_floats": bucket.numel, "first_losses": losses[0], "last_losses": losses[-1]}))
        if dist_on:
            dist.destroy_process_group()
        return 0

    # ------------------------------------------------------------------ the reference's loop
    alphas = alpha_schedule(cfg)
    epochs = min(args.epochs, len(alphas))
    best_val = float("inf")
    history = []
    for epoch in range(1, epochs + 1):
        lr = lr_at_epoch(cfg, epoch)
        for grp in opt.param_groups:
            grp["lr"] = lr
        alpha = alphas[epoch - 1]
        net.train()
        dfm.train()
        sums, iters = torch.zeros(5, device=dev), 0
        for b in global_batches(train_set.pairs, Bg, shuffle_seed=1000 * epoch, keep_tail=(world == 1)):   # the same order on every rank
            ix = shard(b) if len(b) == Bg else b
            batch = train_set.batch(ix)
            sums += torch.as_tensor(train_step(batch, alpha, train_set.last_names), device=dev)
            iters += 1
            if rank == 0 and iters % int(cfg["misc"]["log_interval"]) == 0:    # per-epoch count, (i + 1) % log_interval (train.py:120-126)
                save_ckpt(net, dfm, args.ckpt_dir, cfg["expname"], "train_best")
        # validation: eval mode, no gradients, this epoch's alpha (train.py:135-156)
        net.eval()
        dfm.eval()
        vsum, viters = torch.zeros((), device=dev), 0
        with torch.no_grad():
            for b in global_batches(val_set.pairs, Bg, keep_tail=(world == 1)):   # the reference's loader keeps the tail batch
                v1, v2, d1, d2, dist1, dist2 = val_set.batch(shard(b) if len(b) == Bg else b)
                fit_criterion(crit, cfg, min(v1.shape[1], v2.shape[1]))
                f1, f2 = forward_pair(v1, d1, v2, d2)
                vsum += crit(f1, f2, dist1, dist2, v1, v2, alpha, dfm)[0].detach().float()
                viters += 1
        if dist_on:    # the validation loss that decides 'val_best' is the mean over all shards
            dist.all_reduce(vsum, op=dist.ReduceOp.SUM)
            vsum /= world
        val = float(vsum) / max(viters, 1)
        mean = (sums / max(iters, 1)).tolist()
        history.append(dict(epoch=epoch, alpha=float(alpha), lr=lr, train=mean, val=val))
        if rank == 0:
            print("epoch:%d, loss:%.6g, dist_loss:%.6g, deform_loss:%.6g, map_loss:%.6g, self_rec_loss:%.6g, val_loss:%.6g (alpha %.3f, lr %.3g)"
                  % ((epoch,) + tuple(mean) + (val, alpha, lr)), flush=True)
            if (epoch + 1) % int(cfg["misc"]["checkpoint_interval"]) == 0:
                save_ckpt(net, dfm, args.ckpt_dir, cfg["expname"], epoch)
            if val <= best_val:
                best_val = val
                save_ckpt(net, dfm, args.ckpt_dir, cfg["expname"], "val_best")
    if rank == 0:
        print(json.dumps({"epochs": epochs, "history": history, "best_val": best_val, "criterion": type(crit).__name__,
                          "ckpt": list(ckpt_paths(args.ckpt_dir, cfg["expname"], "val_best"))}))
    if dist_on:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
