#!/usr/bin/env python3
"""Per-kernel summary (count / total / avg / min / max, % of GPU kernel time) from a rocprofv3
rocpd database (`rocprofv3 --kernel-trace -d DIR -o NAME` writes NAME_results.db).
Usage: python tools/prof_stats.py path/to/x_results.db [top_n]"""
import re
import sqlite3
import sys


def main():
    db = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                       "max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# {db}: {sum(r[1] for r in rows)} dispatches, {tot / 1e3:.3f} ms of kernel time")
    print(f"{'kernel':92s} {'calls':>6s} {'total_us':>11s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
    for r in rows[:top]:
        nm = re.sub(r"\(anonymous namespace\)::", "", r[0])
        nm = re.sub(r"\(.*$", "", nm)[:92]
        print(f"{nm:92s} {r[1]:6d} {r[2]:11.1f} {r[3]:9.1f} {r[4]:9.1f} {r[5]:9.1f} {100 * r[2] / tot:6.1f}")


if __name__ == "__main__":
    main()
