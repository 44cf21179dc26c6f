"""Static description of one RPO train-step workload.

The reference hard-codes the ViT-B/16 numbers inside ``trainers/rpo.py``
(``d_v = 768`` at :52, ``14 * 14`` patches at :154, 8 text heads at :142,
embed dim 512 at :185, context length 77 at :141).  Here they are fields so the
same kernels serve ViT-L/14 (BASELINE.json configs[3]) as well.
"""
from __future__ import annotations

from dataclasses import dataclass, replace

# Oxford-Pets base split: first ceil(37/2)=19 class names in label order
# (datasets/oxford_pets.py:55-72 lower-cases "<breed>_<n>.jpg" -> breed;
#  :161-167 keeps the first half).  RPO keeps the underscores
# (trainers/rpo.py:133 replaces only the template's "_").
OXFORD_PETS_BASE_CLASSES = (
    "abyssinian", "american_bulldog", "american_pit_bull_terrier", "basset_hound",
    "beagle", "bengal", "birman", "bombay", "boxer", "british_shorthair",
    "chihuahua", "egyptian_mau", "english_cocker_spaniel", "english_setter",
    "german_shorthaired", "great_pyrenees", "havanese", "japanese_chin", "keeshond",
)
PROMPT_TEMPLATE = "a photo of a _."

SOT_TOKEN = 49406
EOT_TOKEN = 49407
HEAD_DIM = 64


@dataclass(frozen=True)
class RPOConfig:
    """Dimensions of the two towers plus the few-shot step parameters."""

    name: str = "ViT-B/16"
    # image tower
    image_size: int = 224
    patch: int = 16
    d_v: int = 768
    layers_v: int = 12
    # text tower
    d_t: int = 512
    layers_t: int = 12
    context: int = 77
    vocab: int = 49408
    # joint space
    embed: int = 512
    # RPO
    K: int = 24
    n_cls: int = 19

    @property
    def heads_v(self) -> int:
        return self.d_v // HEAD_DIM

    @property
    def heads_t(self) -> int:
        return self.d_t // HEAD_DIM

    @property
    def grid(self) -> int:
        return self.image_size // self.patch

    @property
    def n_patches(self) -> int:
        return self.grid * self.grid

    @property
    def n_frozen(self) -> int:
        """CLS + patches: the tokens every query may read (trainers/rpo.py:154)."""
        return 1 + self.n_patches

    @property
    def seq_v(self) -> int:
        return self.n_frozen + self.K

    @property
    def patch_dim(self) -> int:
        return 3 * self.patch * self.patch

    def with_(self, **kw) -> "RPOConfig":
        return replace(self, **kw)


def vit_b16(**kw) -> RPOConfig:
    return RPOConfig(**kw)


def vit_l14(**kw) -> RPOConfig:
    """ViT-L/14: d=1024, 24 layers, 16 heads, patch 14 (257 frozen tokens);
    text width 768 / 12 heads, embed 768.  The reference cannot run it
    (SURVEY.md finding 7); only the oracle restatement pins it."""
    base = dict(name="ViT-L/14", patch=14, d_v=1024, layers_v=24, d_t=768,
                layers_t=12, embed=768)
    base.update(kw)
    return RPOConfig(**base)


# ---------------------------------------------------------------------------
# Algorithmic FLOPs: mask-aware minimal work, 2 FLOP per
# MAC.  Used by bench.py for roofline.achieved; cannot be inflated by wasted
# compute because it counts only what the read-only structure requires.
# ---------------------------------------------------------------------------

def flops_image(cfg: RPOConfig) -> tuple[float, float]:
    """(forward, backward) FLOPs per image for the image tower."""
    L, d, N, K, S, e = cfg.layers_v, cfg.d_v, cfg.n_frozen, cfg.K, cfg.seq_v, cfg.embed
    fwd = L * (20 * S * d * d + 4 * N * d * d + 4 * S * N * d) \
        + 2 * (N - 1) * d * cfg.patch_dim + 2 * K * d * e
    bwd = L * (20 * K * d * d + 4 * K * N * d) + 2 * K * d * e
    return float(fwd), float(bwd)


def flops_text(cfg: RPOConfig, len_prompts) -> float:
    """FLOPs per step for the K prompt rows of every class (fwd + bwd);
    the frozen rows' K/V are cached once per run and excluded."""
    L, d, K, e = cfg.layers_t, cfg.d_t, cfg.K, cfg.embed
    tot = 0.0
    for lc in len_prompts:
        tot += L * (20 * K * d * d + 4 * K * int(lc) * d)
    tot += len(len_prompts) * 2 * K * d * e
    return 2.0 * tot


def flops_step(cfg: RPOConfig, batch: int, len_prompts) -> float:
    f, b = flops_image(cfg)
    return batch * (f + b) + flops_text(cfg, len_prompts) + batch * 2.0 * cfg.K * cfg.embed * cfg.n_cls


def flops_coop_step(cfg: RPOConfig, batch: int, lens, replicas: int = 1) -> float:
    """Algorithmic FLOPs of one CoOp (replicas = 1) / CoCoOp (replicas = batch) train step, trainers/coop.py:258-281 /
    trainers/cocoop.py:255-275: the plain image tower forward for `batch` images (all N tokens through all blocks; no
    backward -- nothing trainable sits in front of it), and the DENSE text tower forward + backward-to-input for the
    tokens [0, len_c) of every class (causal attention counted as the lower triangle; backward: the dX GEMMs of the four
    linears -- weights frozen -- and 2x the attention's forward), once per replica; projections and head on top."""
    Lv, dv, N, e = cfg.layers_v, cfg.d_v, cfg.n_frozen, cfg.embed
    Lt, dt = cfg.layers_t, cfg.d_t
    img = Lv * (24.0 * N * dv * dv + 4.0 * N * N * dv) + 2.0 * (N - 1) * dv * cfg.patch_dim + 2.0 * dv * e
    T = float(sum(int(l) for l in lens))
    tri = float(sum(int(l) * (int(l) + 1) // 2 for l in lens))
    text = Lt * (48.0 * T * dt * dt + 3.0 * 4.0 * tri * dt) + 2.0 * 2.0 * len(lens) * dt * e
    return batch * img + replicas * text + 4.0 * batch * len(lens) * e


def flops_last_block_dead(cfg: RPOConfig) -> float:
    """FLOPs per image that flops_image (the SURVEY 8d contract figure) counts but the engine does not execute: in the
    LAST image block only the K prompt rows are consumed (ln_post, trainers/rpo.py:210), so the N frozen rows skip
    their q-projection, attention, out-proj and MLP (they still provide K / V)."""
    d, N = cfg.d_v, cfg.n_frozen
    return float(20 * N * d * d + 4 * N * N * d)


def act_dtype_for_prec(prec: str):
    """`TRAINER.RPO.PREC` (configs/trainers/RPO/main_K24.yaml:35, trainers/rpo.py:247-249,278,298-304) -> the
    activation / weight storage dtype of the HIP engine.

    "fp32": exact-f32 MFMA path (the parity mode).  "fp16" (the reference's GPU default: fp16 weights and
    activations, plain SGD, no loss scaling) -> native IEEE half storage: f16 weights / activations on
    v_mfma_f32_32x32x16_f16 with fp32 accumulation.  Unlike the reference's fp16 mode the residual stream, LayerNorm,
    softmax, logits, the prompts, the prompt GRADIENTS and the optimiser state stay fp32, and there is no loss
    scaling.  What IS stored in fp16 in the backward are the A operands of the dX GEMMs (the back-propagated row
    gradients dx / du / dq and the two feature gradients the head writes): entries below 6e-5 go subnormal and below
    6e-8 flush, exactly as in the reference's own fp16 run (plain SGD, no GradScaler: main_K24.yaml:35,
    trainers/rpo.py:278 builds the scaler for "amp" only).  With the synthetic weights the row gradients are
    1e-4 .. 1e-1 and the goldens bound the effect (gradients within 0.6 % of their largest entry); a user whose
    gradients are much smaller should pick torch.bfloat16 (fp32 exponent range) -- no loss scale is applied for them.
    "amp" (fp32 master weights + autocast + GradScaler) maps onto the same storage mode: autocast's fp16 GEMMs with
    fp32 accumulate are what the f16 mode computes and its fp32 master copy of the only trainable state is what the
    engine keeps anyway; of GradScaler only the skipping of a step whose gradient holds Inf / NaN is reproduced --
    `RPO(..., amp=True)` (rpo_sgd_step_guarded) -- its scaling of small gradients is NOT.  bf16 remains available as
    `torch.bfloat16` (same MFMA rate, fp32 exponent range, 8 x the rounding error)."""
    import torch
    table = {"fp32": torch.float32, "fp16": torch.float16, "amp": torch.float16}
    if prec not in table:
        raise ValueError(f"TRAINER.RPO.PREC must be one of {sorted(table)} (trainers/rpo.py:247), got {prec!r}")
    return table[prec]
