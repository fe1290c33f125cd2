#!/usr/bin/env python3
"""bench.py — point-cloud pairs/sec of DV-Matcher's correspondence hot path on MI355X.

Default workload (BASELINE.json configs[1]): synthetic random pairs, N = M = 2048 points, d = 128,
"correspondence + deform forward only": for every pair and both directions
  graph(verts) -> soft correspondence (top-10) -> Pi@verts -> xyz kNN -> Deformer -> ED warp +
  ARAP -> 2x Chamfer (+ map term),
i.e. GraphDeformLoss_Neural.deform() x2 without the dumps (reference models/loss.py:1401-1411).
A "step" is one pass over a resident batch of pairs; inputs are in HBM before the timed region.  The timed form is a
two-stage pipeline over two alternating resident batches (ops.PairPipeline: the graphs of batch t + 1 are built while
the feature half of batch t runs; nothing is cached); `single_call` reports the one-call form beside it.

  python bench.py [--gpus N --steps K --warmup W --pairs P | --pairs-total T] [--workload pair|train|partial]

One process per GPU.  `--gpus N` with N > 1 STARTS the N ranks itself: the parent process — which never
imports torch and never touches the GPU — spawns N children of this script with RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, relays rank 0's JSON line and exits non-zero if any child
did.  Under an external launcher (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`,
WORLD_SIZE already set) the process is a rank and spawns nothing.  Pairs are independent units: they are
sharded over the ranks with no data-path collective (barrier + MAX-over-ranks timing only).
  --pairs P        P pairs per GPU per step: per-GPU work fixed, "scaling": "weak" (default, P = 512)
  --pairs-total T  T pairs per step over ALL GPUs (T/N per rank): total work fixed, "scaling": "strong"
--workload train / partial time the training step of BASELINE configs[2] / configs[3] (train_driver.py's
timing mode: forward + criterion + backward + all-reduce + Adam), one JSON line in the same format.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the soft-correspondence sweep, pass A
of K1), its launch time measured with HIP events on the launch stream inside the timed region; `frac` =
flops performed on the f16 matrix pipe / its dense peak, `algorithmic_frac_f16` = SURVEY §8d's flops against the
same peak; `roofline.kernels` = the other kernels of the step, event-timed the same way over 3 extra steps
after the timed region.  `checked_pairs`: after the timed region 4 seeded-random pairs of the very batch that was
timed are recomputed by the CPU oracle and compared (arg-max maps bit-exact, coordinates <= 1e-4, losses rtol
1e-3); `cpu_baseline` is the same oracle ("port": a C/OpenMP restatement of the reference's algorithm)
timed on a bounded sample of pairs of that same batch, with a second figure `cpu_baseline.aten`: the same pairs in the
reference's dense ATen / MKL formulation on all host threads.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "dv-matcher_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

N_PTS, M_PTS, DIM, ALPHA = 2048, 2048, 128, 100.0
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense f16 / bf16 matrix peak (same guide)
PEAK_HBM_GBS = 8000.0          # HBM3E peak (same guide; 6.3 TB/s achievable)
CHECK_PAIRS = 4
TRAFFIC_FILE = "r6_k1_traffic.json"   # PMC passes of this round's dominant kernel (tools/k1_traffic.py)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=512, help="pairs per GPU per step (resident batch; weak scaling)")
    ap.add_argument("--pairs-total", type=int, default=None, help="pairs per step over all GPUs (strong scaling: T/N per rank)")
    ap.add_argument("--workload", default="pair", choices=["pair", "train", "partial"],
                    help="pair: configs[1] (default); train: configs[2] training step B=8/GPU N=2048; partial: configs[3] "
                         "training step B=2/GPU 4995 x 2200")
    ap.add_argument("--cpu-sample", type=int, default=48, help="pairs timed for cpu_baseline (0 = skip; rank 0 at N = 1 only)")
    ap.add_argument("--no-check", action="store_true", help="skip the oracle check of 4 pairs of the timed batches")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="time the one-call form (dvm_pair_fwd_f32 on ONE resident batch, rounds 1-5) instead of the two-stage "
                         "pipeline over two alternating batches")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo for the "
                    "two-ranks-on-one-GPU test of this script)")
    ap.add_argument("--dist-always", action="store_true", help="initialise the process group at world size 1 too (runs the RCCL "
                    "branch on a 1-GPU box; tests/test_gpu_ddp.py)")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="per-launch HBM bytes of the dominant kernel from the PMC passes (default: profiles/r6_k1_traffic.json, "
                         "used only if it was taken on this kernel source, for the kernel the timed launches ran, at this --pairs)")
    return ap.parse_args(argv)


def rank_env():
    """(world, rank, local_rank) as torch.distributed.run exports them; (1, 0, 0) when run alone."""
    return tuple(int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))


def shard_pairs(total, rank, world):
    """Contiguous, balanced share of `total` pairs for `rank` (strong scaling)."""
    base, rem = divmod(total, world)
    return base + (1 if rank < rem else 0)


# ---------------------------------------------------------------------------------------------------------------------
# Parent side of `--gpus N`: spawn N ranks.  Nothing here may import torch or touch HIP (a process that has initialised
# the GPU must not be replaced or forked into ranks on this pool); tests/test_dist_gloo.py asserts that.
def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_environment(rank, world, port, base=None, cpus=None):
    """Environment of rank `rank` of `world` on this node: rendezvous variables, and the rank's share of the host — a
    contiguous `cores // world` slice of the launcher's cores (DVM_RANK_CPUS, applied by the child before it imports torch)
    with OMP / MKL pools of that size, so that 8 ranks do not start 8 x all-cores threads (dvm/hostenv.py)."""
    from dvm import hostenv   # torch-free
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this pool
    env.update(hostenv.rank_host_env(rank, world, cpus=cpus, base=env))
    return env


def launch_ranks(argv, world):
    """Start `world` children of this script, one per GPU; relay rank 0's stdout; -> exit code (0 iff all children 0).
    If one rank dies the others would wait in a barrier forever: the survivors — exactly the processes started here — are
    ended and the failure is reported."""
    import threading
    port = free_port()
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=rank_environment(r, world, port),
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None, text=True))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs):
                break
            time.sleep(0.1)
    finally:
        for r, p in enumerate(procs):
            if p.poll() is None:
                p.kill()
                p.wait()
                print("bench.py: rank %d ended by the launcher because another rank failed" % r, file=sys.stderr)
            rc = rc or (p.returncode if p.returncode is not None else 1)
    reader.join(timeout=10)
    if out0 and out0[0]:
        sys.stdout.write(out0[0])
        sys.stdout.flush()
    return 0 if rc == 0 else (rc if rc > 0 else 1)


# ---------------------------------------------------------------------------------------------------------------------
def make_batch(P, seed, device):
    """P synthetic pairs, generated on the device from a per-rank seed: features N(0,1), coordinates U(0,1)^3, FPS start 0
    (explicit; the reference draws it at random)."""
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    f1 = torch.randn(P, N_PTS, DIM, generator=g, device=device)
    f2 = torch.randn(P, M_PTS, DIM, generator=g, device=device)
    v1 = torch.rand(P, N_PTS, 3, generator=g, device=device)
    v2 = torch.rand(P, M_PTS, 3, generator=g, device=device)
    s1 = torch.zeros(P, dtype=torch.int32, device=device)
    s2 = torch.zeros(P, dtype=torch.int32, device=device)
    return [f1, f2, v1, v2, s1, s2]


def load_weights():
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "deformer_scape_r_weights.npz")
    return dict(np.load(path))  # the reference's shipped Deformer checkpoint (ckpt/dvmatcher_scape_r), as data


def checked_pair_ids(P, n, seed=20260):
    """`n` pairs of a batch of P chosen by a seeded draw ("checked on 4 random pairs per run")."""
    import random
    return sorted(random.Random(seed + P).sample(range(P), min(n, P)))


def oracle_legs(timed, sample_pairs, check_pairs):
    """The CPU oracle (oracle/dvm_oracle.c, test infrastructure) on pairs of the batches that were timed.  `timed` = [(batch, out12,
    out21), ...] (two entries for the pipelined form): `check_pairs` seeded-random pairs, dealt round-robin over the batches, are
    compared with the GPU outputs of the timed region; the cpu_baseline is timed on those plus the first pairs of the first
    batch up to `sample_pairs` in all."""
    import numpy as np
    from oracle import oracle as O
    O.lib()
    w = load_weights()
    P = timed[0][0][0].shape[0]
    ids = [(n % len(timed), p) for n, p in enumerate(checked_pair_ids(P, check_pairs))] if check_pairs > 0 else []
    ids += [(0, p) for p in range(P) if (0, p) not in ids][:max(0, sample_pairs - len(ids))]
    host = {bp: [t[bp[1]].cpu().numpy() for t in timed[bp[0]][0]] for bp in ids}
    f1, f2, v1, v2, _, _ = host[ids[0]]
    O.pair_direction(w, f1[:256], f2[:256], v1[:256], v2[:256], ALPHA, 0)  # warm the thread pool
    check = {"T_exact": True, "max_abs_warped": 0.0, "max_abs_verts12": 0.0, "max_rel_losses": 0.0,
             "pairs": [{"batch": b, "pair": p} for b, p in ids[:check_pairs]]}
    kept = []
    dt = None
    t0 = time.perf_counter()
    for n, bp in enumerate(ids):
        f1, f2, v1, v2, s1, s2 = host[bp]
        o12 = O.pair_direction(w, f1, f2, v1, v2, ALPHA, int(s1))
        o21 = O.pair_direction(w, f2, f1, v2, v1, ALPHA, int(s2))
        if n == sample_pairs - 1:
            dt = time.perf_counter() - t0
        if n < check_pairs:
            kept.append((bp, o12, o21))
    for (b, p), o12, o21 in kept:
        for o, g in ((o12, timed[b][1]), (o21, timed[b][2])):
            check["T_exact"] = check["T_exact"] and bool(np.array_equal(g["T12"][p].cpu().numpy(), o["T12"]))
            check["max_abs_warped"] = max(check["max_abs_warped"], float(np.abs(g["warped"][p].cpu().numpy() - o["warped"]).max()))
            check["max_abs_verts12"] = max(check["max_abs_verts12"], float(np.abs(g["verts12"][p].cpu().numpy() - o["verts12"]).max()))
            gl, ol = g["losses"][p].cpu().numpy().astype(np.float64), o["losses"].astype(np.float64)
            check["max_rel_losses"] = max(check["max_rel_losses"], float((np.abs(gl - ol) / np.maximum(np.abs(ol), 1e-12)).max()))
    ok = check["T_exact"] and check["max_abs_warped"] <= 1e-4 and check["max_abs_verts12"] <= 1e-4 and check["max_rel_losses"] <= 1e-3
    check["ok"] = bool(ok) if check_pairs > 0 else None
    cpu = None
    if sample_pairs > 0 and dt is not None:
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        omp = int(os.environ.get("OMP_NUM_THREADS", cores))
        cpu = {"value": sample_pairs / dt, "unit": "pairs/s", "cores": min(cores, omp), "kind": "port",
               "sample": "%d pairs of the timed batches (the %d checked ones + the first %d; N=M=%d, d=%d, both directions), C oracle "
                         "with OpenMP: a scalar k-ordered fmaf chain per distance, the arithmetic the parity tests pin (%.1f s)"
                         % (sample_pairs, min(check_pairs, sample_pairs), max(0, sample_pairs - check_pairs), N_PTS, DIM, dt)}
    if cpu is not None:
        # second figure: the same pairs in the reference's dense ATen / MKL formulation on every host thread (cdist, softmax, topk,
        # matmul: what the reference's CPU run executes; oracle/torch_ref.py::pair_direction_aten, pinned to the C oracle by
        # tests/test_modules_cpu.py) — bounded to a few pairs
        import torch
        from oracle import torch_ref as TR
        n_aten = min(8, sample_pairs)
        f1, f2, v1, v2, s1, s2 = host[ids[0]]
        TR.pair_direction_aten(w, f1[:256], f2[:256], v1[:256], v2[:256], ALPHA, 0)   # warm up
        ta = time.perf_counter()
        agree = True
        for n, bp in enumerate(ids[:n_aten]):
            f1, f2, v1, v2, s1, s2 = host[bp]
            a12 = TR.pair_direction_aten(w, f1, f2, v1, v2, ALPHA, int(s1))
            TR.pair_direction_aten(w, f2, f1, v2, v1, ALPHA, int(s2))
            if n < len(kept):
                agree = agree and float(np.abs(a12["losses"] - kept[n][1]["losses"]).max() / np.abs(kept[n][1]["losses"]).max()) < 1e-3
        dta = time.perf_counter() - ta
        cpu["aten"] = {"value": n_aten / dta, "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
                       "losses_agree_with_c_oracle": bool(agree),
                       "sample": "%d of those pairs, both directions, dense N x M tensors on ATen / MKL as the reference's CPU run computes "
                                 "them (graphs from the C oracle: the reference's Python FPS loop + KDTree is not charged) (%.1f s)" % (n_aten, dta)}
    return check, cpu


def k1_sources_sha16():
    """Identity of the dominant kernel's source: sha256 over the files that define pass A of K1.  The PMC traffic file
    records it when it is measured (tools/k1_traffic.py); bench.py reports `traffic` only from a file taken on THIS source
    and for the kernel the timed launches actually ran."""
    import hashlib
    h = hashlib.sha256()
    for name in ("dvm_softcorr_coarse.hip", "dvm_softcorr_f16.hip", "dvm_softcorr_f16.h"):
        with open(os.path.join(ROOT, "dv-matcher_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def load_traffic(P, kernel_name):
    """(bytes per launch or None, provenance dict) from profiles/r6_k1_traffic.json."""
    tpath = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
    if not os.path.exists(tpath):
        return None, {"file": None, "why_null": "no traffic file for this round"}
    tj = json.load(open(tpath))
    prov = {"file": "profiles/" + TRAFFIC_FILE, "kernel": tj.get("kernel_slot_name"), "source_sha16": tj.get("source_sha16"),
            "pairs": tj.get("pairs")}
    if tj.get("pairs") != P:
        prov["why_null"] = "measured at --pairs %s, this run is --pairs %d" % (tj.get("pairs"), P)
    elif tj.get("kernel_slot_name") != kernel_name:
        prov["why_null"] = "measured for %r, the timed launches ran %r" % (tj.get("kernel_slot_name"), kernel_name)
    elif tj.get("source_sha16") != k1_sources_sha16():
        prov["why_null"] = "measured on another version of the kernel source (%s, now %s)" % (tj.get("source_sha16"), k1_sources_sha16())
    else:
        return tj.get("bytes_per_launch"), prov
    return None, prov


def load_pmc(P, kernel_name):
    """The SQ / TCC figures of the dominant kernel from profiles/r6_k1_traffic.json (tools/k1_traffic.py), under the same
    provenance rule as `traffic`: {} unless measured on this kernel source, for this kernel, at this --pairs."""
    tpath = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
    if not os.path.exists(tpath):
        return {}
    tj = json.load(open(tpath))
    if tj.get("pairs") != P or tj.get("kernel_slot_name") != kernel_name or tj.get("source_sha16") != k1_sources_sha16():
        return {}
    return {k: tj.get(k) for k in ("mfma_busy", "valu_per_mfma", "l2_hit_rate", "algorithmic_bytes_per_launch")}


def kernel_models(P):
    """Algorithmic work of the other kernels of one step (P pairs per GPU, both directions), per LAUNCH, and what bounds each.
    bytes = what has to cross HBM at least once (inputs read once, outputs written once); flops = performed on the pipe named."""
    R = 2 * P * N_PTS           # query rows of both directions = all feature rows of both clouds
    Rn = R // 2                 # deformation-graph nodes (N / 2 per cloud)
    row = DIM * 4               # one feature row, bytes
    return {
        1: dict(bound="hbm", work=2 * R * row + R * 12 * 8 + R * 10 * 8,
                note="pass B: every feature row once as a query and at least once as a candidate; the ~11 candidate rows of 512 B "
                     "gathered per query (of a list of 16 from the coarse screen, 12 from the first form) are served by L2"),
        2: dict(bound="mfma", work=3 * 2.0 * 299136 * Rn, peak=PEAK_F16_MFMA_TFLOPS,
                note="Deformer MLP 262-512-256-128-9 over all nodes on the f16 pipe, 3 partial products of the exact 2-way split"),
        3: dict(bound="hbm", work=8 * P * N_PTS * (16 + 16 + 4),
                note="8 nearest-neighbour problems per pair in one launch; a grid walk of ~110 candidates per query: latency / VALU bound"),
        4: dict(bound="hbm", work=2 * P * N_PTS * row,
                note="Conv2d(k->1) pooling of one cloud set: 10 neighbour rows of 512 B gathered per point (L2), one row written"),
        5: dict(bound="hbm", work=2 * P * N_PTS * (16 + 40), note="xyz kNN (k = 10) of all 2P clouds on the uniform grid: latency / VALU bound"),
        6: dict(bound="hbm", work=2 * P * N_PTS * 12 + 2 * P * (N_PTS // 2) * 4,
                note="farthest-point sampling: N/2 dependent steps per cloud, one workgroup per cloud: latency bound"),
        7: dict(bound="hbm", work=P * ((N_PTS // 2) * 264 * 4 + 2 * N_PTS * row),
                note="Deformer input rows of one direction: 264 floats written per node; the pooled rows of both clouds read at least "
                     "once (a node gathers its own pooled row and the 10 of its correspondences, mostly L2 hits)"),
    }


def run_pair(args):
    import torch
    import torch.distributed as dist
    from dvm import _lib, ops

    world, rank, local = rank_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    local = local % torch.cuda.device_count()  # (more ranks than devices only happens in the gloo test)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # --dist-always: initialise the process group (and run the barriers / the MAX all-reduce) even at world size 1, so
    # that the RCCL branch of this script executes on a 1-GPU box (tests/test_gpu_ddp.py) and not first on the 8-GPU node
    dist_on = world > 1 or args.dist_always
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            # (a port of its own only for the single-rank case: with more ranks each would pick a different one and hang)
            if world > 1:
                raise SystemExit("bench.py: WORLD_SIZE = %d but MASTER_PORT is not set (use `bench.py --gpus N`, which starts its own "
                                 "ranks, or torch.distributed.run --master-port P)" % world)
            os.environ["MASTER_PORT"] = str(free_port())
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    lib = _lib.load()

    strong = args.pairs_total is not None
    P = shard_pairs(args.pairs_total, rank, world) if strong else args.pairs
    if P < 1:
        raise SystemExit("--pairs-total %d gives rank %d of %d no pair" % (args.pairs_total, rank, world))
    pairs_per_step = args.pairs_total if strong else P * world
    wl = ops.deformer_weight_list(load_weights(), dev)
    batch = make_batch(P, 1000 + rank, dev)  # every rank has its own shard of pairs
    f1, f2, v1, v2, s1, s2 = batch
    out12 = out21 = None
    outs = None

    def step():
        nonlocal out12, out21, outs
        outs = ops.pair_forward(wl, f1, f2, v1, v2, ALPHA, s1, s2, with_map=True, out=outs)
        out12, out21 = outs

    # The timed form: a two-stage pipeline over TWO alternating resident batches (ops.PairPipeline).  Step t enqueues the
    # coordinate-only geometry of batch t + 1 (dvm_pair_geometry_f32: FPS -> node grid -> ring -> influence, vertex grid, xyz kNN)
    # on its own stream and runs the feature-dependent half of batch t (dvm_pair_fwd_cached_f32, reuse_geometry = 1) on the main
    # stream.  Every step builds the graphs of one batch and consumes the graphs of one batch — nothing is cached (the two batches
    # have different coordinates, and each workspace is rewritten every other step); K timed steps contain K geometry builds and
    # K feature halves.  `--no-pipeline` times the one-call form of rounds 1-5 on one batch instead; it is also reported below.
    pipelined = not args.no_pipeline
    batches = [batch]
    pouts = [None, None]
    if pipelined:
        batches.append(make_batch(P, 5000 + rank, dev))
        pipe = ops.PairPipeline(wl, P, N_PTS, M_PTS, with_map=True)
        resident = torch.cuda.Event()
        resident.record()                       # both batches are in HBM behind this point
        ticket = pipe.prefetch(*batches[0][2:], ready=resident)
        tstep = 0

    def pstep():
        nonlocal ticket, tstep
        cur, nxt = batches[tstep % 2], batches[(tstep + 1) % 2]
        pouts[tstep % 2], ticket = pipe.step(ticket, cur[0], cur[1], ALPHA, next_coords=nxt[2:], ready=resident, out=pouts[tstep % 2])
        tstep += 1

    timed_step = pstep if pipelined else step

    def read_slot(slot):
        ms, nl = ctypes.c_double(), ctypes.c_int()
        ops.check(lib.dvm_profile_read_kernel(slot, ctypes.byref(ms), ctypes.byref(nl)), "dvm_profile_read_kernel")
        return ms.value, nl.value

    for _ in range(args.warmup):
        timed_step()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    ops.check(lib.dvm_profile_enable(args.steps + 4), "dvm_profile_enable")   # slot 0 (the sweep) only
    # per-step durations for the median: one event between steps on the caller's stream (pair_forward joins its helper
    # streams back into it before returning, so consecutive events bracket whole steps)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        timed_step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # which pass-A kernel the last timed launch's (direction, pair) entries went to (the probe and the gate decide on the device)
    routes = (ctypes.c_int * 5)()
    ops.check(lib.dvm_k1_last_routes(routes), "dvm_k1_last_routes")
    routes = dict(zip(("full", "lean", "coarse", "reswept_lean", "entries"), list(routes)))
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    k1_total_ms, k1_launches = read_slot(0)
    k1_name = lib.dvm_profile_kernel_name(0).decode()     # what the timed launches actually ran (the probe / alpha routes pass A)
    lib.dvm_profile_disable()
    # the other kernels of the step, bracketed the same way, over 3 extra steps (outside the timed region: 16 more event
    # records per step)
    ops.check(lib.dvm_profile_select((1 << 8) - 1), "dvm_profile_select")
    ops.check(lib.dvm_profile_enable(3 * 16), "dvm_profile_enable")
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    slots = {k: read_slot(k) for k in range(1, 8)}
    lib.dvm_profile_disable()
    # the sweep kernel alone: the same launches with the helper-stream overlap switched off
    lib.dvm_pair_set_overlap(0)
    step()
    torch.cuda.synchronize()
    ops.check(lib.dvm_profile_enable(8), "dvm_profile_enable")
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    alone_ms, alone_n = read_slot(0)
    lib.dvm_profile_disable()
    lib.dvm_pair_set_overlap(1)
    # the one-call form of rounds 1-5 (dvm_pair_fwd_f32 on ONE resident batch: geometry and features of the same batch inside one
    # call, the geometry chain on the call's helper streams), timed over the same number of steps
    single = None
    if pipelined:
        for _ in range(max(1, args.warmup)):
            step()
        torch.cuda.synchronize()
        ts0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        single = time.perf_counter() - ts0
        if dist_on:
            t = torch.tensor([single], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            single = float(t.item())
        pipe_same = all(torch.equal(pouts[0][sd][k], outs[sd][k]) for sd in (0, 1) for k in outs[sd])   # batch 0: both forms, same bits
    # "graph build reported with and without caching".  The timed region above rebuilds both clouds' deformation graphs
    # every step, as the reference does (models/loss.py:1325-1337); here the same steps reuse the graphs / grids / xyz kNN of a
    # first call through the opt-in per-shape cache (dvm_pair_fwd_cached_f32: bit-identical outputs, checked below)
    gcache = ops.GeometryCache()
    ref12 = {k: v.clone() for k, v in out12.items()}
    couts = ops.pair_forward(wl, f1, f2, v1, v2, ALPHA, s1, s2, with_map=True, cache=gcache, key="bench-batch")   # builds
    for _ in range(max(1, args.warmup)):
        couts = ops.pair_forward(wl, f1, f2, v1, v2, ALPHA, s1, s2, with_map=True, out=couts, cache=gcache, key="bench-batch")
    torch.cuda.synchronize()
    tc0 = time.perf_counter()
    for _ in range(args.steps):
        couts = ops.pair_forward(wl, f1, f2, v1, v2, ALPHA, s1, s2, with_map=True, out=couts, cache=gcache, key="bench-batch")
    torch.cuda.synchronize()
    dtc = time.perf_counter() - tc0
    cached_same = all(torch.equal(couts[0][k], ref12[k]) for k in ref12)
    if dist_on:
        t = torch.tensor([dtc], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dtc = float(t.item())
    del gcache
    local_dt = dt
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # sanity: finite outputs (a bench number over garbage is worthless)
    assert torch.isfinite(out12["losses"]).all() and torch.isfinite(out21["warped"]).all()

    if rank == 0:
        value = pairs_per_step * args.steps / dt
        traffic, traffic_src = args.traffic_bytes, {"file": None, "why": "--traffic-bytes"}
        if traffic is None:
            traffic, traffic_src = load_traffic(P, k1_name)
        k1_ms = k1_total_ms / max(k1_launches, 1)
        k1_alone = alone_ms / max(alone_n, 1)
        flops_launch = P * (2.0 * N_PTS * M_PTS * DIM)   # the distance tile counted once per pair
        tf = lambda ms_: flops_launch / (ms_ * 1e-3) / 1e12 if ms_ > 0 else 0.0  # noqa: E731
        # fp16 products of the N x M contraction pass A PERFORMED per direction, from the routes the library reports for the last
        # timed launch: coarse screen 1 (hh), first forms 3 (hh + hm + mh); None when the launch was mixed or partly swept twice
        nprod = (1 if routes["coarse"] == routes["entries"] and routes["reswept_lean"] == 0
                 else 3 if routes["coarse"] == 0 else None)
        perf = (lambda ms_: 2.0 * nprod * tf(ms_)) if nprod else (lambda ms_: None)  # noqa: E731
        frac16 = lambda x: None if x is None else x / PEAK_F16_MFMA_TFLOPS  # noqa: E731
        kernels = []
        for k, (ms_tot, n) in sorted(slots.items()):
            if n == 0:
                continue
            m = kernel_models(P)[k]
            per_launch = ms_tot / n
            per_step = ms_tot / 3.0
            work = m["work"]
            if m["bound"] == "mfma":
                ach, peak, unit = work / (per_launch * 1e-3) / 1e12, m["peak"], "TFLOP/s"
            else:
                ach, peak, unit = work / (per_launch * 1e-3) / 1e9, PEAK_HBM_GBS, "GB/s"
            kernels.append({"kernel": lib.dvm_profile_kernel_name(k).decode(), "bound": m["bound"], "achieved": ach, "peak": peak,
                            "unit": unit, "frac": ach / peak, "launch_ms": per_launch, "launches_per_step": n / 3.0,
                            "ms_per_step": per_step, "algorithmic_per_launch": work, "note": m["note"]})
        pmc = load_pmc(P, k1_name)
        res = {
            "metric": "point-cloud pairs/sec (N=2048, d=128)", "value": value, "unit": "pairs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "median_ms_per_step": median_ms,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: synthetic random pairs N=M=2048 d=128, correspondence+deform "
                                   "forward, both directions", "pairs_per_gpu_per_step": P, "pairs_per_step": pairs_per_step,
                       "alpha": ALPHA, "deformer_weights": "reference ckpt/dvmatcher_scape_r (fixture)", "fps_start": 0,
                       "parallelism": "pairs sharded over %d GPU(s), no collective" % world,
                       "schedule": ("two-stage pipeline over 2 alternating resident batches: step t builds the graphs / grids / xyz kNN "
                                    "of batch t+1 (dvm_pair_geometry_f32, own stream) while it runs the feature-dependent half of batch t "
                                    "(dvm_pair_fwd_cached_f32); every step builds one batch's graphs and consumes one batch's graphs, "
                                    "nothing is cached" if pipelined else
                                    "one call per step on one resident batch (dvm_pair_fwd_f32)")},
            "per_gpu": {"pairs_per_s": value / world, "rank0_ms_per_step": local_dt / args.steps * 1e3},
            # the one-call form (geometry and features of the SAME batch inside one call; rounds 1-5 timed this)
            "single_call": (None if single is None else
                            {"value": pairs_per_step * args.steps / single, "unit": "pairs/s", "ms_per_step": single / args.steps * 1e3,
                             "bit_identical_to_pipelined": bool(pipe_same)}),
            # the same steps with the per-shape graph cache on (opt-in; NOT `value`: the reference rebuilds the graphs per call)
            "graph_cached": {"value": pairs_per_step * args.steps / dtc, "unit": "pairs/s", "ms_per_step": dtc / args.steps * 1e3,
                             "hits": args.steps, "bit_identical_to_uncached": bool(cached_same),
                             "what": "graphs (FPS nodes, rings, skinning), uniform grids and xyz kNN of both clouds reused from a first "
                                     "call (dvm_pair_fwd_cached_f32); everything feature-dependent recomputed"},
            "process_group": (dist.get_backend() if dist_on else None),
            # The dominant kernel runs the N x M contraction on the 16-bit matrix pipe.  ONE denominator: the dense f16 matrix peak.
            #   achieved / frac      = flops the kernel PERFORMS (`products_per_direction` fp16 products x 2 directions x B*2NMd; the norm
            #                          instruction not counted) / launch time            -> the pipe's utilisation
            #   algorithmic_frac_f16 = SURVEY §8d's formulation-independent count (B*2NMd: the distance tile once per pair) / launch time,
            #                          against the same peak
            #   mfma_busy, valu_per_mfma, l2_hit_rate, traffic = PMC passes of this kernel (profiles/r6_k1_traffic.json; null unless
            #                          taken on this kernel source, for this kernel, at this --pairs): why the pipe is not busier
            "roofline": {"bound": "mfma", "kernel": k1_name + " (K1 pass A: the N x M sweep on the f16 matrix pipe; named by the library "
                                                             "from what the timed launches ran)",
                         "routes_last_timed_launch": routes, "products_per_direction": nprod,
                         "achieved": perf(k1_ms), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": frac16(perf(k1_ms)), "pipe": "f16 matrix (v_mfma_f32_32x32x16_f16)",
                         "algorithmic_flops_per_launch": flops_launch, "algorithmic_frac_f16": frac16(tf(k1_ms)),
                         "traffic": traffic, "traffic_source": traffic_src, "launch_ms": k1_ms, "launches_timed": k1_launches,
                         "flops_per_launch": (2.0 * nprod * flops_launch) if nprod else None,
                         "mfma_busy": pmc.get("mfma_busy"), "valu_per_mfma": pmc.get("valu_per_mfma"), "l2_hit_rate": pmc.get("l2_hit_rate"),
                         "algorithmic_bytes_per_launch": pmc.get("algorithmic_bytes_per_launch"),
                         "share_of_step": (k1_total_ms * 1e-3) / local_dt if local_dt > 0 else None,
                         # in the timed region the sweep shares the CUs with the geometry chain of the NEXT batch (FPS / graph / kNN)
                         # and this batch's pooling on the helper streams, which stretches its launch; alone (one-call form, helper
                         # streams off, 3 launches after the timed region) it takes `launch_ms` below
                         "standalone": {"launch_ms": k1_alone, "achieved": perf(k1_alone), "frac": frac16(perf(k1_alone)),
                                        "algorithmic_frac_f16": frac16(tf(k1_alone))},
                         "kernels": kernels},
        }
        if args.no_check and not (world == 1 and args.cpu_sample > 0):
            res["checked_pairs"], res["check"], res["cpu_baseline"] = 0, None, None
        else:
            ncheck = 0 if args.no_check else min(CHECK_PAIRS, P)
            timed = [(batches[i], pouts[i][0], pouts[i][1]) for i in range(2)] if pipelined else [(batch, out12, out21)]
            check, cpu = oracle_legs(timed, args.cpu_sample if world == 1 else 0, ncheck)
            res["checked_pairs"], res["check"], res["cpu_baseline"] = ncheck, (check if ncheck else None), cpu
        print(json.dumps(res))
        if res["check"] is not None and not res["check"]["ok"]:
            print("bench.py: the GPU outputs of the timed batch differ from the oracle: %s" % res["check"], file=sys.stderr)
            if dist_on:
                dist.destroy_process_group()
            return 3
    if dist_on:
        dist.destroy_process_group()
    return 0


def run_train(args):
    """configs[2] / configs[3]: train_driver.py's timing mode (fwd + criterion + bwd + gradient all-reduce + Adam) per rank;
    rank 0 re-emits its line in this script's format."""
    import contextlib
    import io
    import train_driver
    world, rank, _ = rank_env()
    partial = args.workload == "partial"
    per_gpu = 2 if partial else 8
    batch = args.pairs_total if args.pairs_total is not None else per_gpu * world
    argv = ["--steps", str(args.steps), "--warmup", str(args.warmup), "--batch", str(batch), "--backend", args.backend]
    argv += ["--dist-always"] if args.dist_always else []
    argv += ["--partial", "--points", "4995", "--points-target", "2200"] if partial else ["--points", "2048"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = train_driver.main(argv)
    if rank == 0:
        line = [ln for ln in buf.getvalue().splitlines() if ln.startswith("{")][-1]
        tr = json.loads(line)
        res = {"metric": "training pairs/sec (fwd+loss+bwd+Adam)", "value": tr["value"], "unit": "pairs/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": tr["ms_per_step"], "higher_is_better": True,
               "scaling": "strong" if args.pairs_total is not None else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": ("BASELINE configs[3]: SCAPE-partial-shaped pairs 4995 x 2200, GraphDeformLoss_Neural_Partial"
                                       if partial else "BASELINE configs[2]: training loop step, N=2048, GraphDeformLoss_Neural") +
                                      ", synthetic pairs, random-init LG-Net", "global_batch": batch,
                          "parallelism": "data parallel over %d GPU(s), one flat 8.5 MB gradient all-reduce" % world,
                          "alpha": tr["alpha"], "criterion": tr["criterion"]},
               "host_enqueue_ms_per_step": tr["host_enqueue_ms_per_step"], "roofline": tr["roofline"], "cpu_baseline": None,
               "process_group": tr.get("process_group"),
               "first_losses": tr["first_losses"], "last_losses": tr["last_losses"]}
        print(json.dumps(res))
    return rc


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(argv, args.gpus)      # parent: spawns the ranks, imports no torch
    from dvm import hostenv
    hostenv.apply_rank_host_limits()              # a rank pins itself and sizes its thread pools BEFORE torch is imported
    return run_pair(args) if args.workload == "pair" else run_train(args)


if __name__ == "__main__":
    sys.exit(main())
