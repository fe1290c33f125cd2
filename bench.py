#!/usr/bin/env python3
"""bench.py — point-cloud pairs/sec of DV-Matcher's correspondence hot path on MI355X.

Workload (BASELINE.json configs[1]): synthetic random pairs, N = M = 2048 points, d = 128,
"correspondence + deform forward only": for every pair and both directions
  graph(verts) -> soft correspondence (top-10) -> Pi@verts -> xyz kNN -> Deformer -> ED warp +
  ARAP -> 2x Chamfer (+ map term),
i.e. GraphDeformLoss_Neural.deform() x2 without the dumps (reference models/loss.py:1401-1411).
A "step" is one pass over a resident batch of `--pairs` pairs per GPU; inputs are in HBM before
the timed region.  One process per GPU; pairs shard across ranks with no data-path collective
(weak scaling: per-GPU work is fixed).

  python bench.py [--gpus N --steps K --warmup W --pairs P]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the soft-correspondence
sweep: exact fp16x2-split distances on the 16-bit matrix cores, pass A of K1), its launch time
measured with HIP events on the launch stream inside the timed region; `frac` = flops performed on
the f16 matrix pipe / its dense peak, `algorithmic` = SURVEY §8d's flops against the fp32 matrix peak; `cpu_baseline` is the C oracle ("port") timed on the host cores over a
bounded sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "dv-matcher_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

N_PTS, M_PTS, DIM, ALPHA = 2048, 2048, 128, 100.0
PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense f16 / bf16 matrix peak (same guide)


def rank_env():
    """(world, rank, local_rank) as torch.distributed.run exports them; (1, 0, 0) when run alone."""
    return tuple(int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))


def make_batch(P, seed, device):
    import torch
    g = torch.Generator().manual_seed(seed)
    f1 = torch.randn(P, N_PTS, DIM, generator=g)
    f2 = torch.randn(P, M_PTS, DIM, generator=g)
    v1 = torch.rand(P, N_PTS, 3, generator=g)
    v2 = torch.rand(P, M_PTS, 3, generator=g)
    s1 = torch.zeros(P, dtype=torch.int32)  # FPS start index 0 (explicit; the reference draws it at random)
    s2 = torch.zeros(P, dtype=torch.int32)
    return [t.to(device) for t in (f1, f2, v1, v2, s1, s2)]


def load_weights():
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "deformer_scape_r_weights.npz")
    return dict(np.load(path))  # the reference's shipped Deformer checkpoint (ckpt/dvmatcher_scape_r), as data


def cpu_baseline(sample_pairs):
    """The oracle (a C port of the reference's algorithm, OpenMP) on `sample_pairs` pairs, both directions."""
    import numpy as np
    import torch
    from oracle import oracle as O
    O.lib()
    w = load_weights()
    f1, f2, v1, v2, s1, s2 = [t.numpy() for t in make_batch(sample_pairs, 4242, "cpu")]
    O.pair_direction(w, f1[0][:256], f2[0][:256], v1[0][:256], v2[0][:256], ALPHA, 0)  # warm the thread pool
    t0 = time.perf_counter()
    for p in range(sample_pairs):
        O.pair_direction(w, f1[p], f2[p], v1[p], v2[p], ALPHA, int(s1[p]))
        O.pair_direction(w, f2[p], f1[p], v2[p], v1[p], ALPHA, int(s2[p]))
    dt = time.perf_counter() - t0
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    omp = int(os.environ.get("OMP_NUM_THREADS", cores))
    return {"value": sample_pairs / dt, "unit": "pairs/s", "cores": min(cores, omp), "kind": "port",
            "sample": "%d pairs, N=M=%d, d=%d, both directions, C oracle with OpenMP (%.1f s)" % (sample_pairs, N_PTS, DIM, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=512, help="pairs per GPU per step (resident batch)")
    ap.add_argument("--cpu-sample", type=int, default=48, help="pairs timed for cpu_baseline (0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo for the "
                    "two-ranks-on-one-GPU test of this script)")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="per-launch HBM bytes of the dominant kernel from the PMC passes (default: profiles/k1_traffic.json)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from dvm import _lib, ops

    world, rank, local = rank_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    local = local % torch.cuda.device_count()  # (more ranks than devices only happens in the gloo test)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    lib = _lib.load()

    P = args.pairs
    wl = ops.deformer_weight_list(load_weights(), dev)
    f1, f2, v1, v2, s1, s2 = make_batch(P, 1000 + rank, dev)  # every rank has its own shard of pairs
    out12 = out21 = None
    outs = None

    def step():
        nonlocal out12, out21, outs
        outs = ops.pair_forward(wl, f1, f2, v1, v2, ALPHA, s1, s2, with_map=True, out=outs)
        out12, out21 = outs

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ops.check(lib.dvm_profile_enable(args.steps + 4), "dvm_profile_enable")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms, nl = ctypes.c_double(), ctypes.c_int()
    ops.check(lib.dvm_profile_read(ctypes.byref(ms), ctypes.byref(nl)), "dvm_profile_read")
    lib.dvm_profile_disable()
    # the sweep kernel alone: the same launches with the helper-stream overlap switched off (outside the timed region)
    lib.dvm_pair_set_overlap(0)
    step()
    torch.cuda.synchronize()
    ops.check(lib.dvm_profile_enable(8), "dvm_profile_enable")
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ms1, nl1 = ctypes.c_double(), ctypes.c_int()
    ops.check(lib.dvm_profile_read(ctypes.byref(ms1), ctypes.byref(nl1)), "dvm_profile_read")
    lib.dvm_profile_disable()
    lib.dvm_pair_set_overlap(1)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # sanity: finite outputs (a bench number over garbage is worthless)
    assert torch.isfinite(out12["losses"]).all() and torch.isfinite(out21["warped"]).all()

    if rank == 0:
        pairs_total = P * args.steps * world
        value = pairs_total / dt
        # roofline of the dominant kernel: one launch covers BOTH directions of P pairs; its algorithmic
        # flops are P x 2*N*M*d (SURVEY §8d counts the distance tile once per pair; the kernel evaluates it
        # once per direction, i.e. performs twice that)
        traffic = args.traffic_bytes
        tpath = os.path.join(ROOT, "profiles", "k1_traffic.json")
        if traffic is None and os.path.exists(tpath):  # measured for one exact launch shape
            tj = json.load(open(tpath))
            if tj.get("pairs") == P:
                traffic = tj.get("bytes_per_launch")
        k1_ms = ms.value / max(nl.value, 1)
        flops_launch = P * (2.0 * N_PTS * M_PTS * DIM)
        achieved = flops_launch / (k1_ms * 1e-3) / 1e12 if k1_ms > 0 else 0.0
        res = {
            "metric": "point-cloud pairs/sec (N=2048, d=128)", "value": value, "unit": "pairs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: synthetic random pairs N=M=2048 d=128, correspondence+deform "
                                   "forward, both directions", "pairs_per_gpu_per_step": P, "alpha": ALPHA,
                       "deformer_weights": "reference ckpt/dvmatcher_scape_r (fixture)", "fps_start": 0,
                       "parallelism": "pairs sharded over %d GPU(s), no collective" % world},
            # The dominant kernel runs the N x M contraction on the 16-bit matrix pipe: 3 exact fp16 partial products
            # (2-way split of the scaled features, fp32 accumulate) per direction.  `achieved` / `peak` / `frac` price the
            # flops it PERFORMS (6 x the algorithmic count: 3 products x 2 directions of one distance tile) against the pipe
            # it runs on (dense f16 peak) — the utilisation figure.  `algorithmic` is SURVEY §8d's accounting: 2*N*M*d per
            # pair, the distance tile counted once, over the same launch time, against the fp32 matrix peak §8d prescribes
            # (a formulation-independent number: it can exceed what an fp32-MFMA kernel could ever reach).
            "roofline": {"bound": "mfma", "kernel": "softcorr_sweep_f16_kernel (K1 pass A, fp16x2-split sweep)",
                         "achieved": 6.0 * achieved, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": 6.0 * achieved / PEAK_F16_MFMA_TFLOPS, "pipe": "f16 matrix (v_mfma_f32_32x32x16_f16)",
                         "traffic": traffic, "launch_ms": k1_ms, "launches_timed": nl.value,
                         "flops_per_launch": 6.0 * flops_launch,
                         "algorithmic": {"flops_per_launch": flops_launch, "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS,
                                         "frac": achieved / PEAK_F32_MFMA_TFLOPS, "peak_name": "fp32 matrix"},
                         "share_of_step": (ms.value * 1e-3) / dt if dt > 0 else None,
                         # in the timed region the sweep shares the CUs with the geometry chain on the helper stream
                         # (FPS / graph / kNN), which stretches its launch; alone (overlap off, 3 launches after the
                         # timed region) it takes `launch_ms` below
                         "standalone": {"launch_ms": ms1.value / max(nl1.value, 1),
                                        "achieved": 6.0 * flops_launch / (ms1.value / max(nl1.value, 1) * 1e-3) / 1e12,
                                        "frac": 6.0 * flops_launch / (ms1.value / max(nl1.value, 1) * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS,
                                        "algorithmic_frac": flops_launch / (ms1.value / max(nl1.value, 1) * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS}},
        }
        if world == 1 and args.cpu_sample > 0:
            res["cpu_baseline"] = cpu_baseline(args.cpu_sample)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
