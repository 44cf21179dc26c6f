#!/usr/bin/env python3
"""The text tower's prompt-row attention alone (rpo_text_attn_fwd / rpo_text_attn_bwd at the bench's shapes: 19 classes,
K = 24 prompt rows, 8 heads, prompt lengths of the Oxford-Pets base classes): 200 launches captured in ONE HIP graph (a
ctypes call costs the host ~12 us, more than the kernel), HIP events around the replay.  With rpo_amd/build/ab/librpo_valu.so present (SRC=attn_text tools/build_variant.sh valu
-DRPO_TEXT_ATTN_VALU) the same loop runs on the VALU kernel of the same tree in a second process."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from rpo_amd import ops, synth
    dev = torch.device("cuda:0")
    toks = synth.oxford_pets_base_tokens()
    lens = torch.tensor(toks.argmax(1) + 1, dtype=torch.int32, device=dev)
    for name, n_rep in (("Oxford-Pets base prompts", 1), ("same prompts x 20 (380 classes)", 20)):
        ln = lens.repeat(n_rep)
        n, K, H, d = ln.numel(), 24, 8, 512
        Lmax = int(ln.max())
        for dt in (torch.bfloat16, torch.float16):
            kv = torch.randn(n * Lmax, 2 * d, device=dev).to(dt)
            q, da = torch.randn(n * K, d, device=dev).to(dt), torch.randn(n * K, d, device=dev).to(dt)
            out, dq = torch.empty_like(q), torch.empty_like(q)
            res = []
            for fn in (lambda: ops.text_attn_fwd(q, kv[:, :d], kv[:, d:], out, ln, n, K, Lmax, H, causal=False),
                       lambda: ops.text_attn_bwd(q, kv[:, :d], kv[:, d:], da, dq, ln, n, K, Lmax, H)):
                for _ in range(10): fn()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for _ in range(200): fn()
                g.replay(); torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); g.replay(); e.record(); e.synchronize()
                res.append(1e3 * s.elapsed_time(e) / 200)
            lib = os.path.basename(os.environ.get("RPO_HIP_LIB", "default library (one wave per class and head)"))
            print(f"{lib:48s} {name:34s} {str(dt):15s} n_cls {n:4d} keys <= {Lmax:2d}: "
                  f"fwd {res[0]:6.2f} us  bwd {res[1]:6.2f} us per launch")


if __name__ == "__main__":
    main()
    alt = os.path.join(ROOT, "rpo_amd", "build", "ab", "librpo_valu.so")
    if "RPO_HIP_LIB" not in os.environ and os.path.exists(alt):
        subprocess.check_call([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, RPO_HIP_LIB=alt))
