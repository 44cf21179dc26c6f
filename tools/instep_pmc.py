#!/usr/bin/env python3
"""Per-kernel PMC means INSIDE training steps (rocprofv3 --kernel-trace --pmc ... -- python bench.py ...): reads the
counter_collection.csv of one pass and prints mean counter values per kernel name for the big GEMM / attention kernels,
next to the kernel's mean duration.  usage: instep_pmc.py <dir with *_counter_collection.csv> [more dirs]"""
import collections, csv, glob, os, sys
KEYS = ("gemm_w4k_kernel", "gemm_w4g_kernel", "gemm_w4_kernel", "attn_fwd_kernel", "attn_bwd_kernel")
for d in sys.argv[1:]:
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = next((k for k in KEYS if k in r["Kernel_Name"]), None)
        if k is None:
            continue
        # split the 224x96 kernel's two shapes by grid/K is not in the csv: use LDS/duration-free key = name + workgroup count
        agg[k + " grid=" + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", d)
    for k in sorted(agg):
        vals = {c: sum(v) / len(v) for c, v in agg[k].items()}
        n = len(next(iter(agg[k].values())))
        extra = ""
        if "TCC_HIT_sum" in vals and "TCC_MISS_sum" in vals:
            extra = f"  L2 hit rate {vals['TCC_HIT_sum'] / (vals['TCC_HIT_sum'] + vals['TCC_MISS_sum']):.3f}"
        print(f"{k:44s} n={n:4d} " + " ".join(f"{c}={v:.4g}" for c, v in sorted(vals.items())) + extra)
