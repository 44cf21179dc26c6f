"""Pins oracle/resample_oracle.py (the checker of the on-device input transforms): bit-exact against the committed
Pillow/torch fixtures everywhere, and against Pillow itself where it is importable (this container and the GPU
image both ship it; the fixtures cover a box where it is not)."""
import os

import numpy as np
import pytest

from oracle import resample_oracle as R

GOLD = os.path.join(os.path.dirname(__file__), "golden", "resample_golden.npz")


def oracle_case(img, meta):
    top, left, h, w, rw, rh, wx, wy, flip, S = [int(v) for v in meta]
    crop = np.ascontiguousarray(img[top:top + h, left:left + w])
    out = R.resize_bicubic(crop, rw, rh, (wx, wy, S, S))
    if flip:
        out = out[:, ::-1]
    return R.to_tensor_normalize(out)


def test_oracle_matches_pillow_fixtures_bit_exact():
    g = np.load(GOLD)
    for i in range(int(g["n"])):
        got = oracle_case(g[f"img{i}"], g[f"meta{i}"])
        ref = g[f"ref{i}"]
        assert got.dtype == np.float32 and got.shape == ref.shape
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"case {i}: {np.abs(got - ref).max()}"


def test_oracle_matches_pillow_directly():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    for (W, H, ow, oh) in [(37, 53, 224, 224), (500, 375, 224, 224), (224, 300, 224, 224), (1000, 17, 224, 224),
                           (3, 3, 224, 224), (640, 480, 224, 298), (225, 224, 224, 224), (1500, 1125, 224, 224)]:
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BICUBIC))
        assert np.array_equal(R.resize_bicubic(img, ow, oh), ref), (W, H, ow, oh)
    for (W, H) in [(500, 375), (375, 500), (224, 224), (1023, 517), (230, 225)]:
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        ow, oh, left, top = R.eval_window(H, W)
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BICUBIC).crop((left, top, left + 224, top + 224)))
        assert np.array_equal(R.resize_bicubic(img, ow, oh, (left, top, 224, 224)), ref), (W, H)


def test_coefficients_are_normalised_fixed_point():
    for (n_in, n_out) in [(500, 224), (100, 224), (224, 224 * 3), (4000, 224)]:
        ksize, bounds, kk = R.precompute_coeffs(n_in, 0.0, float(n_in), n_out)
        assert kk.shape == (n_out, ksize)
        s = kk.sum(axis=1)
        assert np.all(np.abs(s - (1 << R.PRECISION_BITS)) <= ksize), "rows sum to 1.0 within rounding"
        assert np.all(bounds[:, 0] >= 0) and np.all(bounds[:, 0] + bounds[:, 1] <= n_in)


def test_crop_sampler_and_eval_window():
    class Seq:
        def __init__(self, u, r): self.u, self.r = list(u), list(r)
        def uniform(self, a, b): return a + (b - a) * self.u.pop(0)
        def randint(self, lo, hi): return lo + int(self.r.pop(0) * (hi - lo))
    # area fraction 0.5, aspect exp(0) = 1 -> square crop of side round(sqrt(0.5 * 375 * 500)) = 306
    i, j, h, w = R.random_resized_crop_params(375, 500, Seq([(0.5 - 0.08) / 0.92, 0.5], [0.0, 0.999]))
    assert (h, w) == (306, 306) and i == 0 and j == 500 - 306
    # ten failures -> central fallback clipped to the 3:4 .. 4:3 range
    always_big = Seq([1.0, 1.0] * 10, [])
    i, j, h, w = R.random_resized_crop_params(100, 1000, always_big)
    assert (h, w) == (100, 133) and i == 0 and j == (1000 - 133) // 2
    assert R.eval_window(375, 500) == (298, 224, 37, 0)
    assert R.eval_window(500, 375) == (224, 298, 0, 37)
    assert R.eval_window(224, 224) == (224, 224, 0, 0)
