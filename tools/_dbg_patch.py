import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import workload
from rpo_amd import synth
from rpo_amd.trainer import RPO
cfg, sd, toks, tp, ip, image, label = workload("d2_k8_b3")
B = image.shape[0]
imgs = [torch.from_numpy(synth.images(cfg, B, seed=50 + i)).cuda() for i in range(6)]
labs = [torch.from_numpy(synth.labels(cfg, B, seed=60 + i)).cuda() for i in range(6)]
for name, use_graph, promise in (("eager", False, False), ("eager2", False, False), ("graph", True, False), ("graph+patch", True, True), ("graph+patch2", True, True)):
    tr = RPO(cfg, sd, toks, None, "cuda:0", torch.bfloat16, batch_size=B, num_batches=10 ** 9, use_graph=use_graph, prompts=(tp, ip))
    rec = []
    for s in range(5):
        loss = tr.step_async(imgs[s], labs[s], imgs[s + 1] if promise else None)
        rec.append(float(loss.item()))
    torch.cuda.synchronize()
    print(f"{name:14s}", " ".join(f"{v:.7f}" for v in rec), float(tr.engine.params.double().sum()))
