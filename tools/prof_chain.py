#!/usr/bin/env python3
"""Kernel-by-kernel timeline after the LAST sgd_kernel marker in a rocprofv3 rocpd db (see tools/probe_bwd_chain.py)."""
import re, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "sgd_kernel" in r[0]]
seg = rows[idx[-1] + 1:]
t0 = seg[0][1]
print(f"{len(seg)} kernels, span {(seg[-1][2] - t0) / 1e3:.1f} us, busy {sum(r[2] - r[1] for r in seg) / 1e3:.1f} us")
prev = None
agg = {}
for r in seg:
    nm = re.sub(r"\(anonymous namespace\)::", "", r[0]); nm = re.sub(r"\(.*$", "", nm); nm = re.sub(r"^void ", "", nm)[:64]
    gap = 0.0 if prev is None else (r[1] - prev) / 1e3
    prev = r[2]
    a = agg.setdefault(nm, [0, 0.0, 0.0]); a[0] += 1; a[1] += (r[2] - r[1]) / 1e3; a[2] += gap
    if len(sys.argv) > 2:
        print(f" +{(r[1] - t0) / 1e3:8.1f}  dur {(r[2] - r[1]) / 1e3:6.1f}  gap {gap:5.1f}  {nm}")
print(f"{'kernel':66s} calls  dur_sum  dur_avg  gap_before_avg")
for nm, (n, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{nm:66s} {n:5d} {d:8.1f} {d / n:8.2f} {g / n:8.2f}")
