"""On-device input transforms with the reference's configuration surface.

The reference names its transforms in `configs/trainers/RPO/main_K24.yaml:8-13`:

    INPUT.SIZE (224, 224) · INTERPOLATION "bicubic" · PIXEL_MEAN / PIXEL_STD (CLIP)
    TRANSFORMS ["random_resized_crop", "random_flip", "normalize"]

and Dassl (un-vendored) turns them into torchvision's `RandomResizedCrop`, `RandomHorizontalFlip`,
`ToTensor`, `Normalize` for training and `Resize(max(SIZE))`, `CenterCrop(SIZE)`, `ToTensor`, `Normalize` for
testing, each applied per sample on CPU workers (`DATALOADER.NUM_WORKERS: 16`, `:6`).  At MI355X step rates
(> 8 k images/s per GPU) that CPU path cannot keep up, so here the random decisions are drawn on the host and the
pixel work -- Pillow's 8-bit bicubic resample, flip, /255, normalise -- runs in `rpo_preprocess_batch`
(`rpo_amd/csrc/preprocess.hip`) on whole batches of decoded uint8 images, bit-identical to the CPU path.

    tf = build_transform(InputConfig(), is_train=True, device="cuda:0", max_batch=32)
    x = tf(list_of_uint8_HWC_arrays)            # -> float32 [B, 3, 224, 224] on the device

The HIP library is required; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes
import math
import random
from collections import defaultdict
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .ops import check

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # main_K24.yaml:11
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)      # main_K24.yaml:12


@dataclass
class InputConfig:
    """The `INPUT` node of the reference's yacs config (main_K24.yaml:8-13; RRCROP_SCALE is Dassl's default)."""
    SIZE: Tuple[int, int] = (224, 224)
    INTERPOLATION: str = "bicubic"
    PIXEL_MEAN: Tuple[float, float, float] = CLIP_MEAN
    PIXEL_STD: Tuple[float, float, float] = CLIP_STD
    TRANSFORMS: Tuple[str, ...] = ("random_resized_crop", "random_flip", "normalize")
    RRCROP_SCALE: Tuple[float, float] = (0.08, 1.0)


class TorchRng:
    """Draws from torch's global generator the way torchvision's transforms do (`torch.empty(1).uniform_`,
    `torch.randint`, `torch.rand(1)`), so seeding with `torch.manual_seed` gives torchvision's crop sequence."""

    def uniform(self, a: float, b: float) -> float:
        return torch.empty(1).uniform_(a, b).item()

    def randint(self, lo: int, hi: int) -> int:
        return int(torch.randint(lo, hi, size=(1,)).item())

    def rand(self) -> float:
        return torch.rand(1).item()


def random_resized_crop_params(height: int, width: int, rng, scale=(0.08, 1.0),
                               ratio=(3.0 / 4.0, 4.0 / 3.0)) -> Tuple[int, int, int, int]:
    """torchvision `RandomResizedCrop.get_params` -> (top, left, h, w): ten attempts at an area fraction in
    `scale` with a log-uniform aspect ratio, then the central-crop fallback."""
    area = height * width
    lo, hi = math.log(ratio[0]), math.log(ratio[1])
    for _ in range(10):
        target = area * rng.uniform(scale[0], scale[1])
        aspect = math.exp(rng.uniform(lo, hi))
        w = int(round(math.sqrt(target * aspect)))
        h = int(round(math.sqrt(target / aspect)))
        if 0 < w <= width and 0 < h <= height:
            i = rng.randint(0, height - h + 1)
            j = rng.randint(0, width - w + 1)
            return i, j, h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def center_crop_window(height: int, width: int, size: int) -> Tuple[int, int, int, int]:
    """torchvision `Resize(size)` (shorter side -> size, `int()` truncation of the longer) followed by
    `CenterCrop(size)` -> (resize_w, resize_h, left, top)."""
    if width <= height:
        rw, rh = size, int(size * height / width)
    else:
        rh, rw = size, int(size * width / height)
    return rw, rh, int(round((rw - size) / 2.0)), int(round((rh - size) / 2.0))


@dataclass
class SamplePlan:
    """Everything random about one sample, decided on the host."""
    crop: Tuple[int, int, int, int]          # top, left, h, w
    resize: Tuple[int, int]                  # w, h
    window: Tuple[int, int]                  # left, top
    flip: bool


class DeviceTransform:
    """Batch transform: host plans + one packed H2D copy + `rpo_preprocess_batch`.

    Two pinned staging slots alternate so the copy of batch t+1 can be filled while batch t is in flight; the
    device buffers of a slot are reused only after the event recorded behind its kernels has completed."""

    def __init__(self, cfg: InputConfig, is_train: bool, device, max_batch: int = 32,
                 max_image_bytes: int = 3 * 1024 * 1024, rng=None):
        if cfg.INTERPOLATION != "bicubic":
            raise NotImplementedError("only INPUT.INTERPOLATION == 'bicubic' (main_K24.yaml:10) is implemented")
        unknown = set(cfg.TRANSFORMS) - {"random_resized_crop", "random_flip", "normalize"}
        if unknown:
            raise NotImplementedError(f"transforms not used by the RPO configs: {sorted(unknown)}")
        if cfg.SIZE[0] != cfg.SIZE[1]:
            raise NotImplementedError("square INPUT.SIZE only")
        if "normalize" not in cfg.TRANSFORMS:
            raise NotImplementedError("the RPO configs always normalise")
        self.cfg, self.is_train = cfg, is_train
        self.size = int(cfg.SIZE[0])
        self.dev = torch.device(device)
        self.max_batch = max_batch
        self.rng = rng if rng is not None else TorchRng()
        self.lib = _lib.load()
        self.mean = (ctypes.c_float * 3)(*cfg.PIXEL_MEAN)
        self.std = (ctypes.c_float * 3)(*cfg.PIXEL_STD)
        self.desc_bytes = (ctypes.sizeof(_lib.ImageDesc) * max_batch + 15) // 16 * 16
        self.cap = self.desc_bytes + max_batch * max_image_bytes
        self.slots = []
        for _ in range(2):
            self.slots.append({
                "host": torch.empty(self.cap, dtype=torch.uint8).pin_memory() if self.dev.type == "cuda"
                else torch.empty(self.cap, dtype=torch.uint8),
                "dev": torch.empty(self.cap, dtype=torch.uint8, device=self.dev),
                "ws": None, "done": None})
        self.turn = 0

    # ---- host-side decisions -----------------------------------------------------------------------------
    def plan(self, height: int, width: int) -> SamplePlan:
        S = self.size
        if self.is_train and "random_resized_crop" in self.cfg.TRANSFORMS:
            crop = random_resized_crop_params(height, width, self.rng, self.cfg.RRCROP_SCALE)
            resize, window = (S, S), (0, 0)
        else:                                   # test: Resize(max(SIZE)) + CenterCrop(SIZE)
            rw, rh, left, top = center_crop_window(height, width, S)
            crop, resize, window = (0, 0, height, width), (rw, rh), (left, top)
        flip = bool(self.is_train and "random_flip" in self.cfg.TRANSFORMS and self.rng.rand() < 0.5)
        return SamplePlan(crop, resize, window, flip)

    # ---- device work -------------------------------------------------------------------------------------
    def __call__(self, images: Sequence[np.ndarray], plans: Optional[Sequence[SamplePlan]] = None,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
        B, S = len(images), self.size
        if not 0 < B <= self.max_batch:
            raise ValueError(f"batch of {B} images, transform built for 1..{self.max_batch}")
        if plans is None:
            plans = [self.plan(im.shape[0], im.shape[1]) for im in images]
        slot = self.slots[self.turn]
        self.turn ^= 1
        if slot["done"] is not None:
            slot["done"].synchronize()          # the device side of this slot is free again
        host = slot["host"]
        descs = (_lib.ImageDesc * B)()
        off = self.desc_bytes
        max_rows, kmax = 1, 1
        hv = host.numpy()
        for b, (im, pl) in enumerate(zip(images, plans)):
            if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
                raise ValueError("images must be uint8 arrays of shape [H, W, 3] (decoded RGB)")
            H, W = im.shape[:2]
            n = H * W * 3
            if off + n > self.cap:
                raise ValueError("staging buffer too small: raise max_image_bytes")
            hv[off:off + n] = np.ascontiguousarray(im).reshape(-1)
            top, left, ch, cw = pl.crop
            d = descs[b]
            d.src_offset, d.width, d.height = off - self.desc_bytes, W, H
            d.crop_x, d.crop_y, d.crop_w, d.crop_h = left, top, cw, ch
            d.resize_w, d.resize_h = pl.resize
            d.win_x, d.win_y = pl.window
            d.flip = int(pl.flip)
            max_rows = max(max_rows, ch)
            kmax = max(kmax, self.lib.rpo_preprocess_ksize(cw, pl.resize[0]),
                       self.lib.rpo_preprocess_ksize(ch, pl.resize[1]))
            off += (n + 15) // 16 * 16
        ctypes.memmove(host.data_ptr(), descs, ctypes.sizeof(descs))
        need = self.lib.rpo_preprocess_workspace_bytes(B, S, max_rows, kmax)
        if slot["ws"] is None or slot["ws"].numel() < need:
            slot["ws"] = torch.empty(int(need * 1.25) + 1024, dtype=torch.uint8, device=self.dev)
        if out is None:
            out = torch.empty(B, 3, S, S, dtype=torch.float32, device=self.dev)
        assert out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == (B, 3, S, S)
        dev = slot["dev"]
        dev[:off].copy_(host[:off], non_blocking=True)
        stream = torch.cuda.current_stream(self.dev)
        check(self.lib.rpo_preprocess_batch(dev.data_ptr() + self.desc_bytes, off - self.desc_bytes,
                                            ctypes.addressof(descs), dev.data_ptr(), B, S, max_rows, kmax,
                                            ctypes.addressof(self.mean), ctypes.addressof(self.std),
                                            out.data_ptr(), slot["ws"].data_ptr(), slot["ws"].numel(),
                                            stream.cuda_stream), "rpo_preprocess_batch")
        slot["done"] = torch.cuda.Event()
        slot["done"].record(stream)
        return out


def build_transform(cfg: InputConfig, is_train: bool, device, max_batch: int = 32, **kw) -> DeviceTransform:
    """Counterpart of Dassl's `build_transform(cfg, is_train)` for the choices the RPO configs make."""
    return DeviceTransform(cfg, is_train, device, max_batch, **kw)


# ---- few-shot sampling and the base / new class split ---------------------------------------------------------

@dataclass
class Datum:
    """Dassl's `Datum` as the reference uses it (datasets/oxford_pets.py:65-72, :176-180)."""
    impath: str
    label: int
    classname: str = ""


def subsample_classes(*datasets: Sequence[Datum], subsample: str = "all") -> List[List[Datum]]:
    """`OxfordPets.subsample_classes` (datasets/oxford_pets.py:140-186): sorted label set split at ceil(n/2);
    "base" keeps the first half, "new" the second, labels renumbered from 0."""
    if subsample not in ("all", "base", "new"):
        raise ValueError(subsample)
    if subsample == "all":
        return [list(d) for d in datasets]
    labels = sorted({item.label for item in datasets[0]})
    m = math.ceil(len(labels) / 2)
    selected = labels[:m] if subsample == "base" else labels[m:]
    relabel = {y: i for i, y in enumerate(selected)}
    return [[Datum(it.impath, relabel[it.label], it.classname) for it in d if it.label in relabel]
            for d in datasets]


def generate_fewshot_dataset(data: Sequence[Datum], num_shots: int, repeat: bool = False,
                             rng: Optional[random.Random] = None) -> List[Datum]:
    """Dassl `DatasetBase.generate_fewshot_dataset` as called at datasets/oxford_pets.py:44-45: group by label in
    first-seen order, `random.sample` num_shots per class (all of them, or sampling with replacement when
    `repeat`, if a class has fewer).  Dassl is un-vendored: restated from its published source, unpinned."""
    if num_shots < 1:
        return list(data)
    rng = rng if rng is not None else random
    tracker: Dict[int, List[Datum]] = defaultdict(list)
    for it in data:
        tracker[it.label].append(it)
    out: List[Datum] = []
    for _, items in tracker.items():
        if len(items) >= num_shots:
            out.extend(rng.sample(items, num_shots))
        elif repeat:
            out.extend(rng.choices(items, k=num_shots))
        else:
            out.extend(items)
    return out
