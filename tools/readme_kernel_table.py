#!/usr/bin/env python3
"""The per-kernel table of profiles/README.md, generated from the committed bench lines so that the prose cannot
disagree with the files next to it (round-3 review, item 8).

    python tools/readme_kernel_table.py r04 [r03 ...]      -> markdown on stdout
    python tools/readme_kernel_table.py --write r04 r03    -> rewrites the block between the markers in profiles/README.md

Every number comes from profiles/<tag>_bench.json (`roofline.kernels`: HIP events around each launch INSIDE real steps)
and profiles/<tag>_bench_*.json (the other configs); nothing is typed by hand."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
BEGIN, END = "<!-- kernel-table:begin (tools/readme_kernel_table.py) -->", "<!-- kernel-table:end -->"


def _line(path):
    try:
        with open(path) as f:
            rows = [json.loads(l) for l in f if l.strip().startswith("{")]
        return rows
    except (OSError, ValueError):
        return []


def table(tags):
    out = ["| refresh | images/s | ms/step | frac (executed) | in-proj | attention | out-proj | c_fc | c_proj | f16 | ViT-L/14 | K = 4 / 8 / 16 / 24 / 48 | B = 4 / 8 / 16 / 64 / 128 | f32 | eval (B = 100) |",
           "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for tag in tags:
        b = _line(os.path.join(PROF, f"{tag}_bench.json"))
        if not b:
            continue
        b = b[-1]
        r = b["roofline"]
        k = r.get("kernels", {})
        us = lambda n: f"{k[n]['avg_us']:.1f} us ({k[n]['frac_of_peak']:.3f})" if n in k else "-"
        one = lambda fn: (lambda rows: f"{rows[-1]['value']:.0f}" if rows else "-")(_line(os.path.join(PROF, f"{tag}_{fn}.json")))
        many = lambda fn: (lambda rows: " / ".join(f"{x['value']:.0f}" for x in rows) if rows else "-")(_line(os.path.join(PROF, f"{tag}_{fn}.json")))
        fe = r.get("frac_executed", r.get("frac"))
        out.append(f"| `{tag}` | {b['value']:.0f} | {b['ms_per_step']:.3f} | {r['frac']:.3f} ({fe:.3f}) | {us('in_proj')} | {us('attn_fwd')} | "
                   f"{us('out_proj')} | {us('c_fc')} | {us('c_proj')} | {one('bench_f16')} | {one('bench_vitl14')} | {many('bench_ksweep')} | "
                   f"{many('bench_batchsweep')} | {one('bench_f32')} | {one('bench_eval')} |")
    return "\n".join(out)


def main():
    args = sys.argv[1:]
    write = "--write" in args
    tags = [a for a in args if not a.startswith("--")] or ["r04"]
    md = table(tags)
    if not write:
        print(md)
        return
    path = os.path.join(PROF, "README.md")
    txt = open(path).read()
    block = f"{BEGIN}\n{md}\n{END}"
    if BEGIN in txt and END in txt:
        txt = txt[:txt.index(BEGIN)] + block + txt[txt.index(END) + len(END):]
    else:
        raise SystemExit(f"markers not found in {path}")
    open(path, "w").write(txt)
    print("rewrote the kernel table of", path)


if __name__ == "__main__":
    main()
