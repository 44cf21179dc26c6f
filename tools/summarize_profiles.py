#!/usr/bin/env python3
"""gpurun_out/profiles_raw -> profiles/rNN_*.txt (small, tracked)."""
import collections, csv, glob, os, re, shutil, subprocess, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raw = os.path.join(ROOT, "gpurun_out", "profiles_raw")
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)
shutil.copy(os.path.join(raw, "bench.json"), os.path.join(out, f"{tag}_bench.json"))
db = glob.glob(os.path.join(raw, "trace", "**", "*_results.db"), recursive=True)
if db:
    txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_stats.py"), db[0], "40"],
                         capture_output=True, text=True).stdout
    tl = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_timeline.py"), db[0]],
                        capture_output=True, text=True).stdout
    with open(os.path.join(out, f"{tag}_kernel_trace_stats.txt"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline\n"
                "# NB: kernel tracing serialises the two HIP queues; per-kernel durations are valid, overlap is not.\n")
        f.write(txt + "\n# last step:\n" + tl)
names = {258048: "in-proj 7072x2304x768 (256x256 tiles)", 86016: "N=768 GEMMs (out_proj K=768 / c_proj K=3072; 128x128 tiles)",
         344064: "c_fc 7072x3072x768 + QuickGELU (128x128 tiles)"}
lines = ["# PMC counters per launch (mean over launches) for the forward GEMMs of the B=32 step, from separate\n"
         "# `rocprofv3 --kernel-trace --pmc <set>` passes over tools/bench_gemm.py.  Grid_Size identifies the shape.\n"]
for d in sorted(glob.glob(os.path.join(raw, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "gemm_nt_kernel" in r["Kernel_Name"]:
            agg[(int(r["Grid_Size"]), int(r["Workgroup_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines.append(f"\n## {os.path.basename(d)[4:]}\n")
    for (grid, wg), v in sorted(agg.items(), reverse=True):
        lines.append(f"grid {grid:7d} wg {wg:4d} {names.get(grid, ''):60s} " +
                     "  ".join(f"{k}={sum(x) / len(x):.4g}" for k, x in sorted(v.items())) + "\n")
open(os.path.join(out, f"{tag}_gemm_pmc.txt"), "w").writelines(lines)
for fn in ("gemm_timeline.txt", "graph_phases.txt"):
    src = os.path.join(raw, fn)
    if os.path.exists(src):
        txt = "".join(l for l in open(src) if "amdgpu.ids" not in l)
        open(os.path.join(out, f"{tag}_{fn}"), "w").write(txt)
print(os.listdir(out))
