import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
from rpo_amd import synth
from rpo_amd.config import vit_b16
from rpo_amd.trainer import RPO, OptimConfig
cfg = vit_b16(); toks = synth.default_tokens(cfg)
sd = synth.clip_state_dict(cfg, seed=0, token_rows=np.unique(toks).tolist() + [49407])
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
tr = RPO(cfg, sd, toks, OptimConfig(), dev, torch.bfloat16, batch_size=32, num_batches=10**9, prompts=synth.prompts(cfg, sd, seed=7))
img = torch.from_numpy(synth.images(cfg, 32)).to(dev); lab = torch.from_numpy(synth.labels(cfg, 32)).to(dev)
eng = tr.engine
for _ in range(3): eng.forward_backward(img, lab)
torch.cuda.synchronize()
p = bench.KernelProbe(); eng.probe = p
for _ in range(3): eng.forward_backward(img, lab)
eng.probe = None; torch.cuda.synchronize()
for k, v in p.ev.items():
    ts = [1e3 * s.elapsed_time(e) for s, e in v]
    n = len(ts) // 3
    print(k, " ".join(f"{t:5.1f}" for t in ts[-n:]))
# same without the text tower running beside
eng._text_forward_saved = eng._text_forward
eng._text_forward = lambda train: None
eng._text_backward_saved = eng._text_backward
p = bench.KernelProbe(); eng.probe = p
for _ in range(3): eng._image_forward(img, False)
eng.probe = None; torch.cuda.synchronize()
print("image forward alone, eval (no pre-activation store):")
for k, v in p.ev.items():
    ts = [1e3 * s.elapsed_time(e) for s, e in v]
    n = len(ts) // 3
    print(k, " ".join(f"{t:5.1f}" for t in ts[-n:]))
