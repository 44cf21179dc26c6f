"""Shared builders for tests: synthetic workload -> oracle objects."""
import functools
import os

import numpy as np
import torch

from oracle.rpo_oracle import OracleRPO
from rpo_amd import synth
from rpo_amd.config import vit_b16

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# tag -> (depth, K, B, logit_scale)   (must match tools/make_golden.py)
CASES = {
    "d1_k4_b2": (1, 4, 2, np.log(100.0)),
    "d2_k8_b3": (2, 8, 3, np.log(100.0)),
    "d2_k24_b2_init": (2, 24, 2, np.log(1 / 0.07)),
    "d2_k16_b2": (2, 16, 2, np.log(100.0)),
    "d2_k48_b2": (2, 48, 2, np.log(100.0)),
    "d12_k24_b4": (12, 24, 4, np.log(100.0)),
}


def load_golden(tag):
    return dict(np.load(os.path.join(GOLDEN, f"ref_{tag}.npz")))


@functools.lru_cache(maxsize=2)
def workload(tag):
    depth, K, B, ls = CASES[tag]
    cfg = vit_b16(layers_v=depth, layers_t=depth, K=K)
    toks = synth.oxford_pets_base_tokens()
    sd = synth.clip_state_dict(cfg, seed=0, logit_scale=float(ls))
    tp, ip = synth.prompts(cfg, sd, seed=7)
    image = synth.images(cfg, B)
    label = synth.labels(cfg, B)
    return cfg, sd, toks, tp, ip, image, label


def oracle_for(tag):
    cfg, sd, toks, tp, ip, image, label = workload(tag)
    m = OracleRPO(sd, toks, cfg.K, cfg.patch)
    m.set_prompts(tp, ip)
    return m, image, label
