#!/usr/bin/env python3
"""Timeline of the LAST complete step in a rocprofv3 rocpd db: per-stream busy time, span, gaps."""
import re, sqlite3, sys
db = sys.argv[1]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, start, end, stream_id, queue_id from kernels order by start").fetchall()
# steps are delimited by the sgd kernel
idx = [i for i, r in enumerate(rows) if "sgd_kernel" in r[0]]
a, b = idx[-2] + 1, idx[-1] + 1
step = rows[a:b]
t0, t1 = step[0][1], max(r[2] for r in step)
print(f"step span {(t1 - t0) / 1e3:.1f} us, {len(step)} kernels")
byq = {}
for r in step:
    byq.setdefault(r[4], []).append(r)
for q, rs in byq.items():
    busy = sum(r[2] - r[1] for r in rs)
    print(f" queue {q}: {len(rs)} kernels, busy {busy / 1e3:.1f} us, first +{(rs[0][1] - t0) / 1e3:.1f} last end +{(max(r[2] for r in rs) - t0) / 1e3:.1f}")
# union busy
ev = sorted((r[1], r[2]) for r in step)
u, cs, ce = 0, ev[0][0], ev[0][1]
for s, e in ev[1:]:
    if s > ce:
        u += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
u += ce - cs
print(f" union busy {u / 1e3:.1f} us -> idle {(t1 - t0 - u) / 1e3:.1f} us")
if len(sys.argv) > 2:
    for r in step[: int(sys.argv[2])]:
        nm = re.sub(r"\(anonymous namespace\)::", "", r[0]); nm = re.sub(r"\(.*$", "", nm)[:60]
        print(f"  q{r[4]} +{(r[1] - t0) / 1e3:8.1f} {(r[2] - r[1]) / 1e3:7.1f} us  {nm}")
