"""Read the gfx950 ISA of every kernel of the library for two patterns that cost far more than they look in the source:

  * SERIALISED LOADS — a global load followed directly by `s_waitcnt vmcnt(0)`.  A predicated load (`x = j < M ? f(s[j]) : pad`)
    compiles to a branch around the load with a full wait in front of it, so a row of such loads arrives one dependent round trip
    at a time; the cure is to clamp the address, load unconditionally and select (round 3 found this in the wave top-k, the MLP's
    input staging, the grid build, `Pi @ V`, the map terms, the dist-loss kernels: DESIGN §0c);
  * SCRATCH — private-segment bytes / spilled registers per kernel (a spill inside a hot loop is a memory round trip per use).

Usage: python tools/isa_scan.py [min_serialised_loads=6]      (cross-compiles every csrc/*.hip with hipcc -S: no GPU needed)
"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dv-matcher_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -x hip --cuda-device-only -S".split()
EXTRA = {"dvm_softcorr_sweep2.hip": ["-fno-honor-nans"]}
thresh = int(sys.argv[1]) if len(sys.argv) > 1 else 6
tmp = tempfile.mkdtemp(prefix="isa_")
procs = []
for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
    out = os.path.join(tmp, os.path.basename(src)[:-4] + ".s")
    procs.append((src, out, subprocess.Popen(["/opt/rocm/bin/hipcc"] + FLAGS + EXTRA.get(os.path.basename(src), []) + [src, "-o", out],
                                             stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=CSRC)))
    if len(procs) % 8 == 0:
        for _, _, p in procs[-8:]: p.wait()
for _, _, p in procs: p.wait()
print("%-26s %-9s %-22s %s" % ("file", "scratch B", "loads waited at once", "kernel"))
for src, out, _ in procs:
    if not os.path.exists(out):
        print("%-26s (did not compile)" % os.path.basename(src)); continue
    txt = open(out).read()
    scratch = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", txt))
    name, body = None, {}
    for ln in txt.split("\n"):
        m = re.match(r"^(_Z\w+):", ln)
        if m: name = m.group(1); body[name] = []; continue
        if name and ln.startswith("\t") and not ln.startswith("\t."): body[name].append(ln.strip())
        if "s_endpgm" in ln: name = None
    for k, ins in body.items():
        n = tot = 0
        for i, x in enumerate(ins):
            if x.startswith("global_load") and "lds" not in x:
                tot += 1
                if any(y.startswith("s_waitcnt vmcnt(0)") for y in ins[i + 1:i + 3]): n += 1
        sc = scratch.get(k, 0)
        if n >= thresh or sc > 0:
            print("%-26s %-9d %3d of %-15d %s" % (os.path.basename(src), sc, n, tot, k[:110]))
