"""profiles/r6_k1_traffic.json from the PMC passes of tools/gpu/profiles_r6.sh: per-launch HBM bytes (and the SQ rows) of pass A of
K1 as bench.py launches it, with the provenance bench.py checks before it reports `roofline.traffic`: the kernel name the library
reports for the timed launches (dvm_profile_kernel_name(0), taken from the bench line measured in the same script) and the sha256
of the kernel's source files.  FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (128-byte
requests tallied as 64 B); units are KB.  usage: k1_traffic.py <pmc-dir> <bench-line.json> <out.json>"""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (torch-free at import)

d, line, out = sys.argv[1], json.load(open(sys.argv[2])), sys.argv[3]
SUBS = ("softcorr_coarse_kernel", "softcorr_sweep_f16_kernel")   # everything the slot-0 bracket encloses
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for sub in SUBS:
            if sub in r["Kernel_Name"]:
                acc[r["Counter_Name"]][r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
# per launch of the bracket = sum over the enclosed kernels of their per-launch means
per = {c: sum(sum(v) / len(v) for v in ks.values()) for c, ks in acc.items()}
P = line["config"]["pairs_per_gpu_per_step"]
fetch, write = per.get("FETCH_SIZE", 0.0) * 1024 * 2, per.get("WRITE_SIZE", 0.0) * 1024
res = {
    "kernel_slot_name": line["roofline"]["kernel"].split(" (K1 pass A")[0],
    "source_sha16": bench.k1_sources_sha16(),
    "pairs": P,
    "config": "bench.py --pairs %d (N=M=2048, d=128)" % P,
    "FETCH_SIZE_KB": per.get("FETCH_SIZE"), "WRITE_SIZE_KB": per.get("WRITE_SIZE"),
    "fetch_bytes_corrected_x2": fetch, "write_bytes": write, "bytes_per_launch": fetch + write,
    # per pair: the h plane of both clouds read once (2 x 2048 x 256 B; the first form reads both planes: 512 B) + per row and
    # direction 16 candidate (column, distance) pairs, the two partial softmax sums and the norm fragments (2 x 2048 x 168 B)
    "algorithmic_bytes_per_launch": P * (2 * 2048 * (256 if "softcorr_coarse_kernel" in line["roofline"]["kernel"] else 512) + 2 * 2048 * 168),
    "l2_hit_rate": (per["TCC_HIT_sum"] / (per["TCC_HIT_sum"] + per["TCC_MISS_sum"])) if "TCC_HIT_sum" in per and "TCC_MISS_sum" in per else None,
    # share of the chip's SIMD time the matrix pipe is busy: SQ_VALU_MFMA_BUSY_CYCLES (summed over the SIMDs) / (the launches' GPU
    # cycles x 256 CUs x 4 SIMDs), the cycles of the launch AS PROFILED (counter passes run the kernels one at a time: the launch
    # alone); vector instructions issued per matrix instruction
    # (rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs: cycles = GRBM / 8, SIMD-cycles = cycles x 1024 = GRBM x 128)
    "mfma_busy": (per["SQ_VALU_MFMA_BUSY_CYCLES"] / (per["GRBM_GUI_ACTIVE"] * 128.0)) if "SQ_VALU_MFMA_BUSY_CYCLES" in per and per.get("GRBM_GUI_ACTIVE") else None,
    "valu_per_mfma": (per["SQ_INSTS_VALU"] / per["SQ_INSTS_MFMA"]) if per.get("SQ_INSTS_MFMA") else None,
    "gpu_cycles_per_launch": (per["GRBM_GUI_ACTIVE"] / 8.0) if per.get("GRBM_GUI_ACTIVE") else None,
    "scratch_bytes_per_launch_estimate": "148 B per lane (36 spilled VGPRs) x 64 lanes x 4 waves x 8192 workgroups = 0.31 GB written once and re-read: the "
                                         "larger part of WRITE_SIZE - list bytes (0.27 GB)",
    "aggregation": "every counter = sum over the kernels of the slot-0 bracket of (mean over that kernel's launches of the value rocprofv3 "
                   "reports per dispatch, itself the sum over all XCDs / SEs); one launch = BOTH directions of `pairs` pairs (4x the rows of "
                   "a 256-pair one-direction run such as tools/gpu/r5_pmc_coarse.sh)",
    "SQ": {k: v for k, v in sorted(per.items()) if k.startswith("SQ_")},
    "kernels_in_bracket": sorted({k for ks in acc.values() for k in ks}),
    "how": "rocprofv3 --kernel-trace --pmc <one group per pass> -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-check "
           "(tools/gpu/profiles_r6.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md; Infinity-Cache hits are counted in FETCH_SIZE",
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("kernel_slot_name", "source_sha16", "pairs", "bytes_per_launch", "algorithmic_bytes_per_launch", "l2_hit_rate")}))
