"""Host-side resources of one rank on a multi-GPU node: which CPU cores it may use and how many library threads it starts.

The training step is enqueued by one Python thread per rank (~2 ms of host work per step on the native path, ~22 ms on the
autograd path) and the CPU oracle / data preparation use OpenMP / ATen intra-op pools.  With 8 ranks on one node and the
default pool size (= all cores) every rank would start as many threads as the node has cores: 8x oversubscription of exactly
the resource that bounds a host-bound step.  Each rank therefore gets a contiguous share of the cores the launcher itself may
use (`cores // world`), pinned with sched_setaffinity, and OMP_NUM_THREADS / MKL_NUM_THREADS of that size.

Nothing here imports torch or touches HIP: the launcher (bench.py's parent) calls `rank_host_env` to build the children's
environment, and a rank calls `apply_rank_host_limits` BEFORE it imports torch (the OpenMP runtime sizes its pool at load).
The reference has no counterpart (train.py:30-33 runs one process on one device).
"""
import os

ENV_CPUS = "DVM_RANK_CPUS"   # comma-separated core ids this rank pins itself to (set by the launcher, or derived below)


def available_cpus():
    if hasattr(os, "sched_getaffinity"):
        return sorted(os.sched_getaffinity(0))
    return list(range(os.cpu_count() or 1))


def rank_cpu_share(local_rank, local_world, cpus=None):
    """Contiguous share of `cpus` for one of `local_world` ranks (at least one core; ranks beyond the core count wrap)."""
    cpus = available_cpus() if cpus is None else list(cpus)
    per = max(1, len(cpus) // max(1, local_world))
    lo = (local_rank * per) % len(cpus)
    return cpus[lo:lo + per] or cpus[:1]


def rank_host_env(local_rank, local_world, cpus=None, base=None):
    """Environment entries a launcher adds for one rank: its core list and matching thread-pool sizes (an OMP_NUM_THREADS /
    MKL_NUM_THREADS the user exported is respected)."""
    share = rank_cpu_share(local_rank, local_world, cpus)
    env = {ENV_CPUS: ",".join(str(c) for c in share)}
    base = os.environ if base is None else base
    for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        if k not in base:
            env[k] = str(len(share))
    return env


def apply_rank_host_limits(environ=None):
    """Called by a rank before `import torch`: pin to the launcher's core list (or, under an external launcher that set only
    LOCAL_RANK / LOCAL_WORLD_SIZE, to this rank's share) and size the thread pools.  Returns the core list, or None when the
    process is not one of several ranks."""
    env = os.environ if environ is None else environ
    local_world = int(env.get("LOCAL_WORLD_SIZE", env.get("WORLD_SIZE", "1")))
    if ENV_CPUS in env:
        share = [int(c) for c in env[ENV_CPUS].split(",") if c != ""]
    elif local_world > 1:
        share = rank_cpu_share(int(env.get("LOCAL_RANK", "0")), local_world)
    else:
        return None
    if not share:
        return None
    if hasattr(os, "sched_setaffinity"):
        try:
            os.sched_setaffinity(0, share)
        except OSError:   # a core of the list is not ours (cgroup changed between launcher and rank): keep the inherited mask
            share = available_cpus()
    for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        env.setdefault(k, str(len(share)))
    return share
