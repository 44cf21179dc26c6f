"""Worker of tests/test_gpu_model.py::test_two_ranks_equal_one_rank_global_batch: one rank of a data-parallel run of
the HIP trainer (launched by torch.distributed.run).  Every rank takes its shard of the same global batches, runs
`steps` optimisation steps and rank 0 saves the prompts."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from rpo_amd import synth  # noqa: E402
from rpo_amd.config import vit_b16  # noqa: E402
from rpo_amd.dist import GradSync  # noqa: E402
from rpo_amd.trainer import RPO, OptimConfig  # noqa: E402


def run(out_path: str, global_batch: int, steps: int, act: torch.dtype, sync: GradSync):
    cfg = vit_b16(layers_v=2, layers_t=2, K=8)
    toks = synth.oxford_pets_base_tokens()
    sd = synth.clip_state_dict(cfg, seed=0, token_rows=np.unique(toks).tolist() + [49407])
    tp, ip = synth.prompts(cfg, sd, seed=7)
    first, count = sync.shard(global_batch)
    dev = torch.device(f"cuda:{sync.local_rank}")
    torch.cuda.set_device(dev)
    if sync.rank != 0:                      # the broadcast in build_model must make every rank start from rank 0's prompts
        tp = tp + 1.0
    oc = OptimConfig(lr=0.01, warmup_epoch=0, lr_scheduler="constant")
    tr = RPO(cfg, sd, toks, oc, dev, act, batch_size=count, num_batches=10 ** 9, sync=sync, prompts=(tp, ip))
    losses = []
    for s in range(steps):
        img = synth.images(cfg, global_batch, seed=500 + s)[first:first + count]
        lab = synth.labels(cfg, global_batch, seed=600 + s)[first:first + count]
        losses.append(tr.forward_backward({"img": torch.from_numpy(img), "label": torch.from_numpy(lab)})["loss"])
    if sync.rank == 0:
        np.savez(out_path, params=tr.engine.params.cpu().numpy(), losses=np.asarray(losses), world=sync.world_size)
    sync.barrier()
    sync.close()


if __name__ == "__main__":
    out, gb, steps, act = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    run(out, gb, steps, torch.float32 if act == "f32" else torch.bfloat16, GradSync())
