#!/usr/bin/env python3
"""rpo_chain_bwd (one persistent launch) against the launch-per-stage chain on the same inputs: the image tower's
prompt-row backward of a real step, both ways, then timing of both and the stage timeline of workgroup 0.
Usage: python tools/check_chain.py [batch] [K] [dtype]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rpo_amd import synth
from rpo_amd.config import vit_b16
from rpo_amd.custom_clip import CustomCLIP
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K = int(sys.argv[2]) if len(sys.argv) > 2 else 24
act = {"bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[3] if len(sys.argv) > 3 else "bf16"]
cfg = vit_b16(K=K)
toks = synth.default_tokens(cfg)
sd = synth.clip_state_dict(cfg, seed=0, token_rows=np.unique(toks).tolist() + [49407])
m = CustomCLIP(cfg, sd, toks, "cuda:0", act, max_batch=B, prompts=synth.prompts(cfg, sd, seed=7))
eng = m.engine
img = torch.from_numpy(synth.images(cfg, B)).cuda(); lab = torch.from_numpy(synth.labels(cfg, B)).cuda()
eng.forward_backward(img, lab)
torch.cuda.synchronize()
def run(chain: bool):
    os.environ["RPO_CHAIN"] = "1" if chain else "0"       # (read per call by Engine.chain_ok)
    eng._image_backward(B)
    torch.cuda.synchronize()
    return eng.g_img.clone(), eng.dxa_v[:B * K].clone()
g0, d0 = run(False)
g1, d1 = run(True)
st = eng.chain_state_v.cpu().numpy()
print(f"B={B} K={K} {act}: give-ups {st[0]}, groups on the safe protocol {st[1]}, counters {st[16:16 + 16 * 8:16]}")
print("xcc of group 0:", st[256:256 + 32] - 1)
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
print(f"dL/d(block-0 input) rows: max rel diff {rel(d1, d0):.3e};  g_img: {rel(g1, g0):.3e}  (|g| max {float(g0.abs().max()):.3e})")
os.environ["RPO_CHAIN_SAFE"] = "1"
g2, d2 = run(True)
print(f"safe protocol == fast protocol bitwise: {bool(torch.equal(d2, d1))}; groups on the safe protocol {int(eng.chain_state_v[1])}")
del os.environ["RPO_CHAIN_SAFE"]
def timed(chain, n=30):
    os.environ["RPO_CHAIN"] = "1" if chain else "0"
    for _ in range(3): eng._image_backward(B)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): eng._image_backward(B)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
print(f"eager image backward: launches {timed(False):.1f} us, chain {timed(True):.1f} us")
# graph-captured, as in the step
def graphed(chain):
    os.environ["RPO_CHAIN"] = "1" if chain else "0"
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): eng._image_backward(B)
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 50 * 1e6
print(f"graph replay:         launches {graphed(False):.1f} us, chain {graphed(True):.1f} us")
# stage timeline of workgroup 0 (100 MHz ticks)
L = cfg.layers_v
eng.chain_timeline = torch.zeros(1 + 7 * L, dtype=torch.int64, device="cuda")
os.environ["RPO_CHAIN"] = "1"
for _ in range(3): eng._image_backward(B)
torch.cuda.synchronize()
t = eng.chain_timeline.cpu().numpy()[:1 + 7 * L].astype(np.int64)
dt = np.diff(t) / 100.0
per = dt.reshape(L, 7)
names = ["A d c_proj", "B d c_fc", "C ln_2", "D0 d out", "D1 attn", "E d q", "F ln_1"]
print("stage times of workgroup 0 incl. the hand-off in front of each stage, us (mean over layers | first processed layer):")
for i, n in enumerate(names):
    print(f"  {n:10s} {per[:, i].mean():6.2f} | {per[0, i]:6.2f}")
print(f"  layer total {per.sum(1).mean():.2f} us; chain {dt.sum():.1f} us")
