"""Deterministic synthetic workload: CLIP weights, images, labels, prompts, tokens.

There is no network, so neither CLIP checkpoints nor datasets exist on the build
or GPU boxes.  Everything is generated from counter-based RNG streams
(``numpy.random.Philox`` keyed by (seed, crc32(name))) so that

* the build container (where the reference is imported to make golden vectors),
* the GPU box (where only this repo exists), and
* every rank of a data-parallel job

regenerate the *same bits* independently and in any order.

Weight statistics follow CLIP's own initialiser (reference
``clip/model.py:303-330`` for the text tower, ``:217-225`` for the visual
embeddings) with two deliberate differences that make parity tests stronger:
biases and LayerNorm affine parameters are non-trivial (a trained CLIP has
non-trivial values; zeros/ones would hide bias/affine bugs), and
``logit_scale = ln(100)`` (trained-CLIP regime).

State-dict key names and shapes are those of the reference ``CLIP`` module
 so ``tools/make_golden.py`` can ``load_state_dict`` them
into the real reference.
"""
from __future__ import annotations

import json
import math
import os
import zlib
from typing import Optional, Dict, Iterable, List, Sequence

import numpy as np

from .config import EOT_TOKEN, SOT_TOKEN, RPOConfig

_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def _rng(seed: int, name: str) -> np.random.Generator:
    key = (int(seed) & 0xFFFFFFFF) | (zlib.crc32(name.encode()) << 32)
    return np.random.Generator(np.random.Philox(key=key))


def normal(seed: int, name: str, shape: Sequence[int], std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    g = _rng(seed, name)
    out = g.standard_normal(size=tuple(shape), dtype=np.float32)
    if std != 1.0:
        out *= np.float32(std)
    if mean != 0.0:
        out += np.float32(mean)
    return out


# ---------------------------------------------------------------------------
# weights
# ---------------------------------------------------------------------------

def _block(seed: int, prefix: str, d: int, n_layers: int, out: Dict[str, np.ndarray]) -> None:
    attn_std = d ** -0.5
    proj_std = (d ** -0.5) * ((2 * n_layers) ** -0.5)
    fc_std = (2 * d) ** -0.5
    out[prefix + "attn.in_proj_weight"] = normal(seed, prefix + "attn.in_proj_weight", (3 * d, d), attn_std)
    out[prefix + "attn.in_proj_bias"] = normal(seed, prefix + "attn.in_proj_bias", (3 * d,), 0.02)
    out[prefix + "attn.out_proj.weight"] = normal(seed, prefix + "attn.out_proj.weight", (d, d), proj_std)
    out[prefix + "attn.out_proj.bias"] = normal(seed, prefix + "attn.out_proj.bias", (d,), 0.02)
    out[prefix + "ln_1.weight"] = normal(seed, prefix + "ln_1.weight", (d,), 0.1, 1.0)
    out[prefix + "ln_1.bias"] = normal(seed, prefix + "ln_1.bias", (d,), 0.05)
    out[prefix + "mlp.c_fc.weight"] = normal(seed, prefix + "mlp.c_fc.weight", (4 * d, d), fc_std)
    out[prefix + "mlp.c_fc.bias"] = normal(seed, prefix + "mlp.c_fc.bias", (4 * d,), 0.02)
    out[prefix + "mlp.c_proj.weight"] = normal(seed, prefix + "mlp.c_proj.weight", (d, 4 * d), proj_std)
    out[prefix + "mlp.c_proj.bias"] = normal(seed, prefix + "mlp.c_proj.bias", (d,), 0.02)
    out[prefix + "ln_2.weight"] = normal(seed, prefix + "ln_2.weight", (d,), 0.1, 1.0)
    out[prefix + "ln_2.bias"] = normal(seed, prefix + "ln_2.bias", (d,), 0.05)


def clip_state_dict(cfg: RPOConfig, seed: int = 0, token_rows: Iterable[int] | None = None,
                    logit_scale: float = math.log(100.0)) -> Dict[str, np.ndarray]:
    """fp32 numpy state dict with the reference ``CLIP`` key names.

    ``token_rows``: if given, only these rows of ``token_embedding.weight`` are
    materialised (others stay zero) -- the hot path touches the table only at
    ``make_prompts`` time (trainers/rpo.py:135-136) and for the EOT row
    (:63), so the GPU box need not generate all 25 M entries.  Rows are
    generated per-row so both variants agree bit-for-bit on the rows they share.
    """
    sd: Dict[str, np.ndarray] = {}
    dt, dv, e = cfg.d_t, cfg.d_v, cfg.embed
    sd["positional_embedding"] = normal(seed, "positional_embedding", (cfg.context, dt), 0.01)
    sd["text_projection"] = normal(seed, "text_projection", (dt, e), dt ** -0.5)
    sd["logit_scale"] = np.array(logit_scale, dtype=np.float32)
    tok = np.zeros((cfg.vocab, dt), dtype=np.float32)
    rows = range(cfg.vocab) if token_rows is None else sorted(set(int(r) for r in token_rows))
    if token_rows is None:
        # one stream per 1024-row chunk keeps the full table fast to generate and
        # still lets the sparse variant reproduce any row (chunk stream, then slice)
        for c0 in range(0, cfg.vocab, 1024):
            n = min(1024, cfg.vocab - c0)
            tok[c0:c0 + n] = normal(seed, f"token_embedding.chunk{c0 // 1024}", (1024, dt), 0.02)[:n]
    else:
        chunks: Dict[int, np.ndarray] = {}
        for r in rows:
            c = r // 1024
            if c not in chunks:
                chunks[c] = normal(seed, f"token_embedding.chunk{c}", (1024, dt), 0.02)
            tok[r] = chunks[c][r - c * 1024]
    sd["token_embedding.weight"] = tok
    sd["ln_final.weight"] = normal(seed, "ln_final.weight", (dt,), 0.1, 1.0)
    sd["ln_final.bias"] = normal(seed, "ln_final.bias", (dt,), 0.05)
    for l in range(cfg.layers_t):
        _block(seed, f"transformer.resblocks.{l}.", dt, cfg.layers_t, sd)
    scale = dv ** -0.5
    sd["visual.class_embedding"] = normal(seed, "visual.class_embedding", (dv,), scale)
    sd["visual.positional_embedding"] = normal(seed, "visual.positional_embedding", (cfg.n_frozen, dv), scale)
    sd["visual.proj"] = normal(seed, "visual.proj", (dv, e), scale)
    sd["visual.conv1.weight"] = normal(seed, "visual.conv1.weight", (dv, 3, cfg.patch, cfg.patch),
                                       cfg.patch_dim ** -0.5)
    for nm in ("ln_pre", "ln_post"):
        sd[f"visual.{nm}.weight"] = normal(seed, f"visual.{nm}.weight", (dv,), 0.1, 1.0)
        sd[f"visual.{nm}.bias"] = normal(seed, f"visual.{nm}.bias", (dv,), 0.05)
    for l in range(cfg.layers_v):
        _block(seed, f"visual.transformer.resblocks.{l}.", dv, cfg.layers_v, sd)
    return sd


def clip_state_dict_shared(cfg: RPOConfig, seed: int, token_rows, path: str, writer: bool, barrier) -> Dict[str, np.ndarray]:
    """One generation per NODE instead of one per rank: the local writer rank generates the state dict and leaves it in
    `path` as one flat fp32 file + a JSON index (written to a temporary name and renamed, so a reader never sees a
    partial file), `barrier()` is called by every rank, and the other ranks memory-map it read-only (page cache: one
    physical copy for the node).  Bit-identical to clip_state_dict on every rank by construction.  The writer removes
    nothing: the caller owns `path` (bench.py deletes it after the engines are built)."""
    idx_path = path + ".json"
    if writer:
        sd = clip_state_dict(cfg, seed=seed, token_rows=token_rows)
        index, off = {}, 0
        for k in sorted(sd):
            a = np.asarray(sd[k])                       # (ascontiguousarray would turn the 0-d logit_scale into [1])
            index[k] = [off, list(a.shape)]
            off += a.size
        flat = np.lib.format.open_memmap(path + ".tmp", mode="w+", dtype=np.float32, shape=(off,))
        for k, (o, shape) in index.items():
            flat[o:o + int(np.prod(shape, dtype=np.int64))] = np.asarray(sd[k], dtype=np.float32).reshape(-1)
        flat.flush()
        del flat
        with open(idx_path + ".tmp", "w") as f:
            json.dump(index, f)
        os.replace(path + ".tmp", path)
        os.replace(idx_path + ".tmp", idx_path)
        barrier()
        return sd
    barrier()
    with open(idx_path) as f:
        index = json.load(f)
    flat = np.load(path, mmap_mode="r")
    return {k: (flat[o:o + int(np.prod(shape, dtype=np.int64))].reshape(tuple(shape)) if shape else np.array(flat[o], dtype=np.float32))
            for k, (o, shape) in index.items()}


def state_dict_checksum(sd: Dict[str, np.ndarray]) -> str:
    """Order-independent fingerprint used by fixtures to detect generator drift."""
    h = 0
    for k in sorted(sd):
        a = np.ascontiguousarray(sd[k])
        h = zlib.crc32(k.encode(), h)
        h = zlib.crc32(a.view(np.uint8).reshape(-1).data, h)
    return f"{h:08x}"


# ---------------------------------------------------------------------------
# batches
# ---------------------------------------------------------------------------

def images(cfg: RPOConfig, batch: int, seed: int = 1234, rank: int = 0) -> np.ndarray:
    """[B,3,H,W] fp32 i.i.d. N(0,1): what a mean/std-normalised image looks like to
    the patch embedding (configs/trainers/RPO/main_K24.yaml:11-12)."""
    return normal(seed + rank, "images", (batch, 3, cfg.image_size, cfg.image_size))


def labels(cfg: RPOConfig, batch: int, seed: int = 4321, rank: int = 0) -> np.ndarray:
    g = _rng(seed + rank, "labels")
    return g.integers(0, cfg.n_cls, size=(batch,), dtype=np.int64)


def prompts(cfg: RPOConfig, sd: Dict[str, np.ndarray], seed: int = 7) -> tuple[np.ndarray, np.ndarray]:
    """(text_prompt[K,d_t], img_prompt[K,d_v]) following the reference formula
    (trainers/rpo.py:63-67, :77-81): base token repeated K times plus 0.1 x a
    unit-norm gaussian direction.  Values are injected into the reference when
    golden vectors are made, so torch's global RNG order (:65, :79) is irrelevant."""
    tn = normal(seed, "text_prompt_noise", (cfg.K, cfg.d_t))
    tn /= np.linalg.norm(tn, axis=-1, keepdims=True)
    vn = normal(seed, "img_prompt_noise", (cfg.K, cfg.d_v))
    vn /= np.linalg.norm(vn, axis=-1, keepdims=True)
    text = sd["token_embedding.weight"][EOT_TOKEN][None, :] + np.float32(0.1) * tn
    img = sd["visual.class_embedding"][None, :] + np.float32(0.1) * vn
    return text.astype(np.float32), img.astype(np.float32)


# ---------------------------------------------------------------------------
# tokens
# ---------------------------------------------------------------------------

def oxford_pets_base_tokens() -> np.ndarray:
    """[19,77] int64 token ids of "a photo of a <class>." for the Oxford-Pets base
    split, captured from the reference tokenizer (clip/clip.py:185-221) by
    tools/make_golden.py -- the BPE tokenizer itself is out of scope (SURVEY.md
    section 2 row 7)."""
    with open(os.path.join(_DATA_DIR, "tokens_oxford_pets_base.json")) as f:
        obj = json.load(f)
    return np.asarray(obj["tokens"], dtype=np.int64)


def synthetic_tokens(cfg: RPOConfig, lengths: Sequence[int], seed: int = 99) -> np.ndarray:
    """Token ids for classes whose tokenised prompt has the given lengths
    (SOT ... EOT counted): used for ragged / maximum-length edge cases."""
    g = _rng(seed, "synthetic_tokens")
    out = np.zeros((len(lengths), cfg.context), dtype=np.int64)
    for c, n in enumerate(lengths):
        n = int(n)
        assert 3 <= n and n + cfg.K <= cfg.context, "need len_c + K <= context (SURVEY appendix B.3)"
        out[c, 0] = SOT_TOKEN
        out[c, 1:n - 1] = g.integers(1000, 40000, size=(n - 2,))
        out[c, n - 1] = EOT_TOKEN
    return out


def coop_tokens(base_tokens: np.ndarray, n_ctx: int, placeholder: Optional[int] = None) -> np.ndarray:
    """Token ids of CoOp's "X X .. X name." prompts (trainers/coop.py:95-98: prompt_prefix + " " + name + ".") built from a
    table of "SOS words EOT" rows: SOS, n_ctx placeholder ids, then the row's own words and its EOT.  The tokenizer is out
    of scope; benchmarks and tests only need rows of that SHAPE whose ids have embedding rows."""
    base = np.asarray(base_tokens, dtype=np.int64)
    out = np.zeros_like(base)
    ph = int(base[0, 1]) if placeholder is None else int(placeholder)
    for c in range(base.shape[0]):
        n = int(base[c].argmax()) + 1                     # SOS .. EOT
        assert n + n_ctx <= base.shape[1]
        out[c, 0] = base[c, 0]
        out[c, 1:1 + n_ctx] = ph
        out[c, 1 + n_ctx:n_ctx + n] = base[c, 1:n]
    return out


def len_prompts(tokens: np.ndarray) -> np.ndarray:
    """argmax(ids)+1 (trainers/rpo.py:137): EOT has the largest id."""
    return tokens.argmax(axis=-1).astype(np.int64) + 1


def default_tokens(cfg: RPOConfig) -> np.ndarray:
    if cfg.n_cls == 19:
        return oxford_pets_base_tokens()
    base = [10, 10, 14, 11, 8, 8, 9, 8, 8, 11, 8, 10, 13, 10, 11, 10, 10, 10, 10]
    return synthetic_tokens(cfg, [base[i % 19] for i in range(cfg.n_cls)])
