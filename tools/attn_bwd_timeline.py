#!/usr/bin/env python3
"""In-kernel phase timeline of rpo_attn_readonly_bwd_proj / rpo_attn_readonly_bwd (debug build, -DRPO_TIMELINE)."""
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
B, H, N, Kp, d = 32, 12, 197, 24, 768
Rf = B * N
names = {11: "Wo/dx operands + 24 MFMA", 12: "partials through LDS", 13: "K/V landed in LDS", 14: "phase 1 (row max / sum)",
         15: "phase 2 (dP, U, W)", 16: "cross-wave sum + store"}
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for cold in (False, True):
    for fused in (True, False):
        qkv = torch.randn(B * (N + Kp), 3 * d, device=dev).to(torch.bfloat16)
        dx = torch.randn(B * Kp, d, device=dev).to(torch.bfloat16)
        w = (torch.randn(d, d, device=dev) * d ** -0.5).to(torch.bfloat16)
        dq = torch.empty(B * Kp, d, dtype=torch.bfloat16, device=dev)
        for _ in range(3):
            if cold: flush.fill_(1)                       # push the operands out of L2 / MALL
            buf.zero_()
            if fused: ops.attn_readonly_bwd_proj(qkv[Rf:, :d], qkv[:Rf, d:2 * d], qkv[:Rf, 2 * d:], dx, w, dq, B, H, N, Kp)
            else: ops.attn_readonly_bwd(qkv[Rf:, :d], qkv[:Rf, d:2 * d], qkv[:Rf, 2 * d:], dx, dq, B, H, N, Kp)
        torch.cuda.synchronize()
        t = buf.view(8, 64).cpu()
        print(f"== {'fused d out-proj' if fused else 'plain'}, operands {'cold (512 MB written in between)' if cold else 'warm'}")
        for b in range(4):
            r = t[b]
            if r[10] == 0: continue
            stamps = [i for i in range(10, 17) if r[i] != 0]
            print(f" wg {b * 97}: " + " | ".join(f"{names.get(i, i)} +{int(r[i] - r[stamps[stamps.index(i) - 1]])}" for i in stamps[1:])
                  + f" | total {int(r[stamps[-1]] - r[10])}")
