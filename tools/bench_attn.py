#!/usr/bin/env python3
"""rpo_attn_readonly_fwd alone: warm-loop time per launch (HIP events on the launch stream) and the error against an fp64
softmax of the same 16-bit operands.  RPO_HIP_LIB=<variant .so> selects a build (tools/build_variant.sh, SRC=attn_image).
Usage: python tools/bench_attn.py [--dtype bf16|f16] [--iters 200]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rpo_amd import ops

args = sys.argv[1:]
dt = torch.float16 if "--dtype" in args and args[args.index("--dtype") + 1] == "f16" else torch.bfloat16
iters = int(args[args.index("--iters") + 1]) if "--iters" in args else 200
dev = torch.device("cuda:0")
torch.manual_seed(0)
for name, B, H, N, Kp, gain in (("ViT-B/16 B=32", 32, 12, 197, 24, 1.0), ("ViT-B/16 B=32 x4 logits", 32, 12, 197, 24, 2.0),
                                ("ViT-B/16 B=32 K=48", 32, 12, 197, 48, 1.0), ("ViT-L/14 B=16", 16, 16, 257, 24, 1.0),
                                ("ViT-B/16 B=4", 4, 12, 197, 24, 1.0)):
    d = 64 * H
    S = N + Kp
    qkv = (torch.randn(B * S, 3 * d, device=dev) * gain).to(dt)
    out = torch.empty(B * S, d, dtype=dt, device=dev)
    run = lambda: ops.attn_readonly_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], out, B, H, N, Kp)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            run()
        e.record(); e.synchronize()
        best = min(best, 1e3 * s.elapsed_time(e) / iters)
    # reference: rows are [B*N frozen | B*Kp prompt]; every query reads the N frozen keys of its image
    q = torch.cat([qkv[:B * N, :d].view(B, N, H, 64), qkv[B * N:, :d].view(B, Kp, H, 64)], 1).double()
    k = qkv[:B * N, d:2 * d].view(B, N, H, 64).double()
    v = qkv[:B * N, 2 * d:].view(B, N, H, 64).double()
    p = torch.softmax(torch.einsum("bshd,bnhd->bhsn", q, k) * 0.125, -1)
    ref = torch.einsum("bhsn,bnhd->bshd", p, v).reshape(B, S, d)
    got = torch.cat([out[:B * N].view(B, N, d), out[B * N:].view(B, Kp, d)], 1).double()
    err = (got - ref).abs().max().item()
    flops = 4.0 * B * H * S * N * 64
    import hashlib
    sha = hashlib.sha1(out.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:10]
    print(f"{name:28s} {str(dt)[6:]:9s} {best:7.2f} us  {flops / best * 1e-6:7.1f} TFLOP/s  max abs err {err:.2e} (|out| <= {ref.abs().max().item():.2f}) sha1 {sha}")
