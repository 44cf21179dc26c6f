"""On-device input transforms (rpo_preprocess_batch through the C ABI) against the Pillow-pinned oracle: bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import resample_oracle as R

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "resample_golden.npz")


def make(size, is_train=True, max_batch=32, **kw):
    from rpo_amd.input_pipeline import InputConfig, build_transform
    return build_transform(InputConfig(SIZE=(size, size)), is_train, "cuda:0", max_batch, **kw)


def oracle_plan(img, pl, S):
    top, left, h, w = pl.crop
    out = R.resize_bicubic(np.ascontiguousarray(img[top:top + h, left:left + w]), pl.resize[0], pl.resize[1],
                           (pl.window[0], pl.window[1], S, S))
    if pl.flip:
        out = out[:, ::-1]
    return R.to_tensor_normalize(out)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_golden_fixtures_bit_exact():
    from rpo_amd.input_pipeline import SamplePlan
    g = np.load(GOLD)
    tfs = {}
    for i in range(int(g["n"])):
        top, left, h, w, rw, rh, wx, wy, flip, S = [int(v) for v in g[f"meta{i}"]]
        tf = tfs.setdefault(S, make(S))
        got = tf([g[f"img{i}"]], [SamplePlan((top, left, h, w), (rw, rh), (wx, wy), bool(flip))]).cpu().numpy()[0]
        assert np.array_equal(bits(got), bits(g[f"ref{i}"])), f"case {i}: max diff {np.abs(got - g[f'ref{i}']).max()}"


def test_train_batch_mixed_sizes_bit_exact():
    """A ragged batch (every image its own size, crop and flip) in one call, random_resized_crop plans drawn by
    the product's sampler from torch's generator."""
    torch.manual_seed(11)
    rng = np.random.default_rng(3)
    tf = make(224)
    sizes = [(375, 500), (500, 333), (224, 224), (97, 1024), (1024, 97), (60, 40), (2, 2), (768, 1024)]
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    plans = [tf.plan(h, w) for h, w in sizes]
    assert any(p.flip for p in plans) and not all(p.flip for p in plans)
    got = tf(imgs, plans).cpu().numpy()
    for b, (im, pl) in enumerate(zip(imgs, plans)):
        assert np.array_equal(bits(got[b]), bits(oracle_plan(im, pl, 224))), (b, sizes[b], pl)


def test_eval_transform_bit_exact():
    rng = np.random.default_rng(4)
    tf = make(224, is_train=False)
    sizes = [(375, 500), (500, 375), (224, 224), (225, 230), (517, 1023), (1200, 900)]
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    got = tf(imgs).cpu().numpy()
    for b, im in enumerate(imgs):
        assert np.array_equal(bits(got[b]), bits(R.eval_transform(im))), sizes[b]


def test_identity_plan_is_normalise_only_and_slots_alternate():
    """224x224 input, whole-image crop: both passes are skipped (Pillow copies), so the result is ToTensor +
    Normalize of the input; repeated calls cycle the two staging slots without corrupting results."""
    from rpo_amd.input_pipeline import SamplePlan
    rng = np.random.default_rng(6)
    tf = make(224)
    for it in range(5):
        imgs = [rng.integers(0, 256, (224, 224, 3), dtype=np.uint8) for _ in range(3)]
        plans = [SamplePlan((0, 0, 224, 224), (224, 224), (0, 0), False)] * 3
        got = tf(imgs, plans).cpu().numpy()
        for b in range(3):
            assert np.array_equal(bits(got[b]), bits(R.to_tensor_normalize(imgs[b])))


def test_full_batch_properties():
    """B = 32 photographs-sized inputs (the bench workload): flip of a plan == mirrored output, output of a constant
    image is the normalised constant, and saturated inputs stay inside the normalised [0, 1] range."""
    from rpo_amd.input_pipeline import SamplePlan
    rng = np.random.default_rng(8)
    tf = make(224)
    imgs = [rng.integers(0, 256, (375, 500, 3), dtype=np.uint8) for _ in range(30)]
    imgs.append(np.full((375, 500, 3), 200, np.uint8))
    imgs.append(np.where(rng.random((375, 500, 1)) < 0.5, 0, 255).astype(np.uint8).repeat(3, axis=2))
    torch.manual_seed(2)
    plans = [tf.plan(375, 500) for _ in imgs]
    a = tf(imgs, plans).clone()
    flipped = [SamplePlan(p.crop, p.resize, p.window, not p.flip) for p in plans]
    b = tf(imgs, flipped)
    assert torch.equal(a, b.flip(-1))
    const = R.to_tensor_normalize(np.full((1, 1, 3), 200, np.uint8))[:, 0, 0]
    assert np.array_equal(a[30].cpu().numpy(), np.broadcast_to(const[:, None, None], (3, 224, 224)))
    lo = R.to_tensor_normalize(np.zeros((1, 1, 3), np.uint8))[:, 0, 0]
    hi = R.to_tensor_normalize(np.full((1, 1, 3), 255, np.uint8))[:, 0, 0]
    s = a[31].cpu().numpy()
    assert np.all(s >= lo[:, None, None]) and np.all(s <= hi[:, None, None])


def test_bad_descriptors_fail_loudly():
    from rpo_amd._lib import RPOLibraryError
    from rpo_amd.input_pipeline import SamplePlan
    tf = make(224)
    img = np.zeros((100, 100, 3), np.uint8)
    with pytest.raises(RPOLibraryError):
        tf([img], [SamplePlan((50, 50, 80, 80), (224, 224), (0, 0), False)])      # crop leaves the image
    with pytest.raises(RPOLibraryError):
        tf([img], [SamplePlan((0, 0, 100, 100), (224, 224), (8, 0), False)])      # window leaves the resize
    with pytest.raises(ValueError):
        tf([img.astype(np.float32)])
    out = tf([img], [SamplePlan((0, 0, 100, 100), (224, 224), (0, 0), False)])    # still usable afterwards
    assert torch.isfinite(out).all()


def test_trainer_consumes_decoded_images():
    """RPO.forward_backward / model_inference fed with decoded uint8 images (device transforms) == the same trainer
    fed with the float tensors the Pillow-pinned oracle produces for the same plans."""
    from helpers import workload
    from rpo_amd.trainer import RPO
    cfg, sd, toks, tp, ip, _, _ = workload("d1_k4_b2")
    prompts = (tp, ip)
    rng = np.random.default_rng(12)
    B = 4
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in [(300, 400), (256, 256), (500, 333), (224, 224)]]
    labels = np.array([1, 0, 3, 2])
    losses = []
    for mode in ("uint8", "float"):
        tr = RPO(cfg, sd, toks, device="cuda:0", act_dtype=torch.float32, batch_size=B, prompts=prompts)
        torch.manual_seed(77)
        if mode == "uint8":
            out = tr.forward_backward({"img": imgs, "label": labels})
        else:
            tf = tr.transform(True)
            plans = [tf.plan(im.shape[0], im.shape[1]) for im in imgs]
            x = np.stack([oracle_plan(im, pl, 224) for im, pl in zip(imgs, plans)])
            out = tr.forward_backward({"img": torch.from_numpy(x), "label": torch.from_numpy(labels)})
        losses.append((out["loss"], tr.engine.params.clone()))
        if mode == "uint8":
            ev = tr.model_inference(imgs).cpu().numpy()
        else:
            xe = torch.from_numpy(np.stack([R.eval_transform(im) for im in imgs]))
            assert np.array_equal(ev, tr.model_inference(xe.cuda()).cpu().numpy())
    assert losses[0][0] == losses[1][0]
    assert torch.equal(losses[0][1], losses[1][1])
