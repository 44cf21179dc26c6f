#!/usr/bin/env python3
"""In-kernel phase timeline of rpo_attn_readonly_fwd (debug build with -DRPO_TIMELINE)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rpo_amd import _lib, ops
dbg = os.path.join(ROOT, "rpo_amd", "build", "librpo_hip_dbg.so")
if not os.path.exists(dbg) or os.environ.get("RPO_REBUILD_DBG"):
    # one recipe for the -DRPO_TIMELINE library (tools/build_debug.sh: EVERY translation unit the ABI needs); compiler output
    # goes to a log next to the library, never into the timeline this script prints
    os.makedirs(os.path.dirname(dbg), exist_ok=True)
    with open(dbg + ".log", "w") as log:
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "build_debug.sh")], stdout=log, stderr=log)
lib = _lib.load(dbg); _lib._lib = lib
lib.rpo_debug_set_timeline.argtypes = [C.c_void_p]
dev = torch.device("cuda:0")
buf = torch.zeros(8 * 64, dtype=torch.int64, device=dev)
assert lib.rpo_debug_set_timeline(buf.data_ptr()) == 0
H, N, Kp, d = 12, 197, 24, 768
# round 6 (attn_fwd16_kernel, the 16-bit modes' default): two-phase staging -- stamps 1 / 2 = V^T fragments written / the
# first 128 K rows committed, 3 = barrier A passed, 4 = key tiles 0 .. 3 done, 5 = rest of K committed + barrier B passed,
# 6 = remaining key tiles + partial tile + stores issued.  (The one-barrier kernel -- f32 mode, -DRPO_ATTN_ONE_BARRIER --
# has no stamps 4 / 5: absent stamps are skipped.)
names2 = ["start", "q + V landed, V^T built", "K rows [0,128) committed", "barrier A passed", "key tiles 0-3 (phase A)",
          "rest of K committed, barrier B passed", "remaining key tiles + stores issued", "stores drained"]
names1 = ["start", "K staged (issued+written)", "V^T built", "barrier passed", "-", "-",
          "key loop (S^T, softmax, P.V) + stores issued", "stores drained"]
for B in [int(a) for a in sys.argv[1:]] or [16, 32]:
    qkv = torch.randn(B * (N + Kp), 3 * d, device=dev).to(torch.bfloat16)
    out = torch.empty(B * (N + Kp), d, dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        buf.zero_()
        ops.attn_readonly_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], out, B, H, N, Kp)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.attn_readonly_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], out, B, H, N, Kp)
    e.record(); e.synchronize()
    print(f"== B={B}: {B * H} workgroups, warm loop {1e3 * s.elapsed_time(e) / 20:.1f} us per launch (debug build, stamps on)")
    t = buf.view(8, 64).cpu()
    for b in range(4):
        r = t[b]
        if r[0] == 0: continue
        have = [i for i in range(8) if r[i] != 0]
        names = names2 if (r[4] != 0 and r[5] != 0) else names1       # which kernel the launcher took for this shape
        assert have[0] == 0 and have[-1] == 7 and all(r[a] <= r[b] for a, b in zip(have, have[1:])), r
        print(f"wg {b*97}: " + " | ".join(f"{names[i]} +{int(r[i]-r[j])}" for j, i in zip(have, have[1:])) + f" | total {int(r[7]-r[0])}")
