#!/usr/bin/env python3
"""Generates tests/golden/resample_golden.npz with Pillow + torch -- the libraries the reference's transforms end in
(Dassl -> torchvision -> `PIL.Image.crop/resize(BICUBIC)/transpose`, `ToTensor`, `Normalize`;
configs/trainers/RPO/main_K24.yaml:8-13).  Run in the build container (Pillow 12.2.0, torch 2.10):

    python tools/make_resample_golden.py

Each case: a random uint8 image, a crop box, the size the crop is resized to, the output window, a flip flag and
the fp32 [3, S, S] tensor the PIL/torch path produces.  Small output sizes keep the fixture small; the code
paths (shrink with antialias support, enlarge, skipped pass, windowed center crop) do not depend on S.
"""
import os
import sys

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MEAN = torch.tensor([0.48145466, 0.4578275, 0.40821073])
STD = torch.tensor([0.26862954, 0.26130258, 0.27577711])


def pil_path(img, crop, resize, window, flip, S):
    top, left, h, w = crop
    im = Image.fromarray(img).crop((left, top, left + w, top + h))
    im = im.resize(resize, Image.BICUBIC)
    im = im.crop((window[0], window[1], window[0] + S, window[1] + S))
    if flip:
        im = im.transpose(Image.FLIP_LEFT_RIGHT)
    t = torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    t.sub_(MEAN[:, None, None]).div_(STD[:, None, None])
    return t.numpy()


def main():
    rng = np.random.default_rng(20260929)
    cases = [  # H, W, crop(top,left,h,w), resize(w,h), window(left,top), flip, S
        (47, 61, (0, 0, 47, 61), (32, 32), (0, 0), False, 32),          # shrink both axes
        (47, 61, (5, 7, 30, 41), (32, 32), (0, 0), True, 32),           # crop + flip
        (20, 17, (0, 0, 20, 17), (32, 32), (0, 0), False, 32),          # enlarge
        (32, 90, (0, 10, 32, 70), (32, 32), (0, 0), False, 32),         # vertical pass skipped
        (90, 32, (3, 0, 80, 32), (32, 32), (0, 0), True, 32),           # horizontal pass skipped
        (32, 32, (0, 0, 32, 32), (32, 32), (0, 0), True, 32),           # both skipped (copy)
        (120, 80, (0, 0, 120, 80), (32, 48), (0, 8), False, 32),        # test-time: resize + center crop
        (75, 100, (0, 0, 75, 100), (42, 32), (5, 0), False, 32),        # test-time, landscape
        (1, 1, (0, 0, 1, 1), (32, 32), (0, 0), False, 32),              # single pixel
        (401, 7, (100, 2, 290, 3), (48, 48), (0, 0), True, 48),         # extreme aspect, long filter
    ]
    out = {"n": np.int64(len(cases))}
    for i, (H, W, crop, resize, window, flip, S) in enumerate(cases):
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        if i == 3:
            img[:] = np.where(rng.random((H, W, 1)) < 0.5, 0, 255)      # saturating input: exercises clip8
        out[f"img{i}"] = img
        out[f"meta{i}"] = np.array([*crop, *resize, *window, int(flip), S], np.int64)
        out[f"ref{i}"] = pil_path(img, crop, resize, window, flip, S)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "resample_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", "Pillow", Image.__version__, "torch", torch.__version__)


if __name__ == "__main__":
    main()
