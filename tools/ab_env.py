#!/usr/bin/env python3
"""Same-box A/B of environment switches: alternates short bench.py runs (own process each, so that switches read at
engine construction take effect) and prints ms/step per arm.  Usage: python tools/ab_env.py [--rounds 3] [--steps 40]
[--extra "<bench args>"] NAME=VALUE[,NAME=VALUE] ...   (the baseline arm, no switches, is always included)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
rounds, steps, extra = 3, 40, []
while args and args[0].startswith("--"):
    k = args.pop(0)
    if k == "--rounds": rounds = int(args.pop(0))
    elif k == "--steps": steps = int(args.pop(0))
    elif k == "--extra": extra = args.pop(0).split()
arms = [("baseline", {})] + [(a, dict(kv.split("=", 1) for kv in a.split(","))) for a in args]
res = {n: [] for n, _ in arms}
for r in range(rounds):
    for name, env in arms:
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "10",
                              "--no-cpu-baseline", "--no-precision", *extra], env=e, capture_output=True, text=True)
        try:
            line = json.loads(out.stdout.strip().splitlines()[-1])
            res[name].append(line["ms_per_step"])
        except Exception:
            print(name, "FAILED", out.stderr[-800:])
for name, _ in arms:
    v = res[name]
    if v: print(f"{name:40s} ms/step min {min(v):.4f}  all {[round(x, 4) for x in v]}")
