#!/usr/bin/env python3
"""Does the image tower run faster as SEVERAL independent part-batches on separate streams than as one batch?

One-round kernels put every CU in its prologue, k-loop and epilogue at the same time.  Images are
independent through the whole image tower, so the batch can be cut into P parts whose kernel chains run on P streams:
each part's kernels fill 256 / P CUs, the parts drift out of phase, and one part's store burst / launch ramp overlaps
another part's k-loop.  This probe measures the image forward (+ backward) graph of P engines of B / P images
replayed concurrently against the single B-image graph, same box, same library.

usage: probe_half_batch.py [B] [parts ...]   e.g.  probe_half_batch.py 32 1 2 4
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rpo_amd import synth
from rpo_amd.config import vit_b16
from rpo_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
parts_list = [int(a) for a in sys.argv[2:]] or [1, 2, 4]
cfg = vit_b16()
toks = synth.default_tokens(cfg)
sd = synth.clip_state_dict(cfg, seed=0, token_rows=np.unique(toks).tolist() + [49407])
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
tp, ip = synth.prompts(cfg, sd, seed=7)
img_all = torch.from_numpy(synth.images(cfg, B)).to(dev)
lab_all = torch.from_numpy(synth.labels(cfg, B)).to(dev)


def capture(fn):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        fn()
    return g


def bench(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return 1e6 * ts[len(ts) // 2]


for P in parts_list:
    b = B // P
    engs, imgs, labs, streams, gf, gb = [], [], [], [], [], []
    for i in range(P):
        e = Engine(cfg, sd, toks, dev, torch.bfloat16, max_batch=b)
        e.params[:cfg.K * cfg.d_t].copy_(torch.from_numpy(tp).reshape(-1))
        e.params[cfg.K * cfg.d_t:].copy_(torch.from_numpy(ip).reshape(-1))
        im, lb = img_all[i * b:(i + 1) * b].contiguous(), lab_all[i * b:(i + 1) * b].contiguous()
        e.forward_backward(im, lb)                      # warm-up, text cache, kernel attributes
        torch.cuda.synchronize()
        engs.append(e); imgs.append(im); labs.append(lb); streams.append(torch.cuda.Stream())
    for i, e in enumerate(engs):
        gf.append(capture(lambda e=e, i=i: e._image_forward(imgs[i], train=True)))
        gb.append(capture(lambda e=e: e._image_backward(b)))
    main = torch.cuda.current_stream()

    def run(graphs, delay_us=0.0):
        # part i starts i * delay_us later (torch.cuda._sleep spins the given number of cycles on its stream): parts
        # launched at the same instant run the same kernel sequence in lock-step, i.e. stay IN phase
        for i, (s, g) in enumerate(zip(streams, graphs)):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                if i and delay_us:
                    torch.cuda._sleep(int(i * delay_us * 2000))          # ~2 GHz
                g.replay()
        for s in streams:
            main.wait_stream(s)

    t_f = bench(lambda: run(gf))
    t_b = bench(lambda: run(gb))
    t_f1 = bench(lambda: run(gf[:1]))
    print(f"B={B} parts={P} ({b} images each): image fwd {t_f:8.1f} us (one part alone {t_f1:8.1f})   image bwd {t_b:8.1f} us",
          flush=True)
    if P > 1:
        for d in (5, 10, 15, 20, 30, 40, 60):
            print(f"    part i delayed by i x {d:3d} us: image fwd {bench(lambda: run(gf, d)):8.1f} us (includes the delay)", flush=True)
    # staggered start: part i begins i/P of a kernel later -- a crude phase offset via a short spin kernel is not
    # needed: different parts see different L2 states and drift apart on their own; report the spread of part end times
    del engs, gf, gb
    torch.cuda.empty_cache()
