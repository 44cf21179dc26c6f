"""Pin the oracle: the dense CPU restatement must reproduce what the REAL
reference (trainers/rpo.py CustomCLIP, imported by tools/make_golden.py in the
build container) produced on the same generated weights / inputs."""
import os

import numpy as np
import pytest
import torch

from oracle import rpo_oracle
from rpo_amd import synth
from rpo_amd.config import flops_image, flops_text, vit_b16, vit_l14

from helpers import CASES, GOLDEN, load_golden, oracle_for, workload

FAST = ["d1_k4_b2", "d2_k8_b3", "d2_k24_b2_init", "d2_k16_b2", "d2_k48_b2"]


def test_tokens_fixture_matches_survey():
    toks = synth.oxford_pets_base_tokens()
    assert toks.shape == (19, 77)
    assert synth.len_prompts(toks).tolist() == [10, 10, 14, 11, 8, 8, 9, 8, 8, 11, 8, 10, 13, 10, 11, 10, 10, 10, 10]
    assert (toks[:, 0] == 49406).all()
    for c, n in enumerate(synth.len_prompts(toks)):
        assert toks[c, n - 1] == 49407 and (toks[c, n:] == 0).all()


def test_generator_is_stable():
    """Fixtures store a checksum of the generated weights: generator drift (numpy
    upgrade, edited stds) would silently unpin every golden otherwise."""
    for tag in ("d1_k4_b2",):
        cfg, sd, *_ = workload(tag)
        assert synth.state_dict_checksum(sd) == load_golden(tag)["weights_crc"].item().decode()


def test_sparse_token_table_matches_full():
    cfg = vit_b16(layers_v=0, layers_t=0)
    rows = [0, 5, 1023, 1024, 49406, 49407]
    full = synth.clip_state_dict(cfg)["token_embedding.weight"]
    sparse = synth.clip_state_dict(cfg, token_rows=rows)["token_embedding.weight"]
    assert np.array_equal(full[rows], sparse[rows])


@pytest.mark.parametrize("tag", FAST + ["d12_k24_b4"])
def test_dense_oracle_matches_reference(tag):
    g = load_golden(tag)
    m, image, label = oracle_for(tag)
    with torch.no_grad():
        logits = m.forward(image).logits
    # fp32 noise floor of the reference itself is ~4e-6 on logits
    np.testing.assert_allclose(logits.numpy(), g["logits"], atol=3e-5, rtol=0)
    out, gt, gi = m.loss_and_grads(image, label)
    assert abs(float(out.loss) - float(g["loss"])) < 2e-5
    for mine, ref in ((gt.numpy(), g["g_text"]), (gi.numpy(), g["g_img"])):
        assert np.abs(mine - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def test_dense_oracle_matches_reference_at_bench_size():
    """The oracle against the reference's own output at the bench's shape (12 layers, K = 24, B = 32:
    tools/make_golden_fullsize.py) -- it is also bench.py's cpu_baseline at exactly this size."""
    import os
    from helpers import GOLDEN
    g = dict(np.load(os.path.join(GOLDEN, "ref_full_k24_b32.npz")))
    cfg = vit_b16(K=24)
    toks = synth.oxford_pets_base_tokens()
    sd = synth.clip_state_dict(cfg, seed=0, token_rows=np.unique(toks).tolist() + [49407])
    tp, ip = synth.prompts(cfg, sd, seed=7)
    m = rpo_oracle.OracleRPO(sd, toks, cfg.K, cfg.patch)
    m.set_prompts(tp, ip)
    image, label = synth.images(cfg, 32), synth.labels(cfg, 32)
    assert np.array_equal(label, g["label"])
    out, gt, gi = m.loss_and_grads(image, label)
    np.testing.assert_allclose(out.logits.detach().numpy(), g["logits"], atol=3e-5, rtol=0)
    assert abs(float(out.loss) - float(g["loss"])) < 2e-5
    for mine, ref in ((gt.numpy(), g["g_text"]), (gi.numpy(), g["g_img"])):
        assert np.abs(mine - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def test_dense_oracle_matches_reference_at_vitl14_widths():
    """ViT-L/14 (BASELINE configs[3]): the oracle against the reference's OWN CustomCLIP run at ViT-L/14 widths
    (tools/make_golden_vitl14_ref.py: trainers/rpo.py's forward and autograd, its four ViT-B/16 dimension literals
    supplied from outside the module) -- 2 + 2 layers, batch 2: eval logits, loss, both gradients and the prompt rows
    after every block of both towers (width 1024 / 16 heads / 281 tokens, width 768 / 12 heads)."""
    import os
    from helpers import GOLDEN
    g = dict(np.load(os.path.join(GOLDEN, "ref_vitl14_d2_k24_b2.npz")))
    cfg = vit_l14(layers_v=2, layers_t=2, K=24)
    toks = synth.oxford_pets_base_tokens()
    sd = synth.clip_state_dict(cfg, seed=0, token_rows=np.unique(toks).tolist() + [49407])
    assert synth.state_dict_checksum(sd) == g["weights_crc"].item().decode()
    tp, ip = synth.prompts(cfg, sd, seed=7)
    m = rpo_oracle.OracleRPO(sd, toks, cfg.K, cfg.patch)
    m.set_prompts(tp, ip)
    image, label = synth.images(cfg, 2), synth.labels(cfg, 2)
    assert np.array_equal(label, g["label"])
    with torch.no_grad():
        _, trows = m.text_tower(m.text_prompt, return_rows=True)
        _, irows = m.image_tower(torch.from_numpy(image), m.img_prompt, return_rows=True)
    np.testing.assert_allclose(torch.stack(trows).numpy()[:, :4], g["text_rows"], atol=3e-5)
    np.testing.assert_allclose(torch.stack(irows).numpy(), g["img_rows"], atol=3e-5)
    out, gt, gi = m.loss_and_grads(image, label)
    np.testing.assert_allclose(out.logits.detach().numpy(), g["logits"], atol=3e-5, rtol=0)
    assert abs(float(out.loss) - float(g["loss"])) < 2e-5
    for mine, ref in ((gt.numpy(), g["g_text"]), (gi.numpy(), g["g_img"])):
        assert np.abs(mine - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def test_prompt_rows_per_block():
    g = load_golden("d2_k8_b3")
    m, image, label = oracle_for("d2_k8_b3")
    with torch.no_grad():
        _, trows = m.text_tower(m.text_prompt, return_rows=True)
        _, irows = m.image_tower(torch.from_numpy(image), m.img_prompt, return_rows=True)
    np.testing.assert_allclose(torch.stack(trows).numpy(), g["text_rows"], atol=2e-5)
    np.testing.assert_allclose(torch.stack(irows).numpy(), g["img_rows"], atol=2e-5)


@pytest.mark.parametrize("tag", ["d2_k8_b3", "d12_k24_b4"])
def test_sgd_steps_match_reference(tag):
    g = load_golden(tag)
    cfg, sd, toks, tp, ip, _, _ = workload(tag)
    depth, K, B, _ = CASES[tag]
    m, _, _ = oracle_for(tag)
    lr, mom, wd = g["sgd_hparams"]
    opt = rpo_oracle.OracleSGD(lr, mom, wd)
    losses = []
    for step in range(4):
        im = synth.images(cfg, B, seed=1234 + 10 * step)
        lb = synth.labels(cfg, B, seed=4321 + 10 * step)
        losses += rpo_oracle.train_steps(m, opt, [(im, lb)])
        if step in (0, 3):
            np.testing.assert_allclose(m.text_prompt.detach().numpy(), g[f"text_prompt_step{step + 1}"], atol=2e-6)
            np.testing.assert_allclose(m.img_prompt.detach().numpy(), g[f"img_prompt_step{step + 1}"], atol=2e-6)
    np.testing.assert_allclose(losses, g["sgd_losses"], atol=3e-5)


def test_oracle_follows_reference_trajectory_through_the_warmup_boundary():
    """The 60-step run of the REAL reference (tools/make_golden_trajectory.py: BASELINE.json configs[0]): both
    restatements of the schedule give the reference's learning rate for all 15 epochs, and the oracle -- stepped with
    them -- reproduces the first two epochs (constant warm-up, then the first epoch at the full rate): the prompts
    after epoch 1 and the 8 losses, the last four of which already depend on updates made at lr 0.01."""
    from rpo_amd.trainer import OptimConfig, lr_at_epoch
    g = dict(np.load(os.path.join(GOLDEN, "ref_traj_d12_k24_b4_e15.npz")))
    lr, mom, wd, max_epoch, iters, B, warm, cons = g["hparams"]
    max_epoch, iters, B, warm = int(max_epoch), int(iters), int(B), int(warm)
    oc = OptimConfig(lr=lr, max_epoch=max_epoch, warmup_epoch=warm, warmup_cons_lr=cons, momentum=mom, weight_decay=wd)
    for e in range(max_epoch):
        assert abs(lr_at_epoch(oc, e) - g["lrs"][e]) < 1e-15
        assert abs(rpo_oracle.cosine_lr_with_constant_warmup(lr, e, max_epoch, warm, cons) - g["lrs"][e]) < 1e-15
    cfg, sd, toks, tp, ip, _, _ = workload("d12_k24_b4")
    assert g["weights_crc"].item().decode() == synth.state_dict_checksum(sd)
    m, _, _ = oracle_for("d12_k24_b4")
    opt = rpo_oracle.OracleSGD(lr, mom, wd)
    batches = [(synth.images(cfg, B, seed=1234 + 10 * i), synth.labels(cfg, B, seed=4321 + 10 * i)) for i in range(iters)]
    losses = []
    for e in range(2):
        opt.lr = rpo_oracle.cosine_lr_with_constant_warmup(lr, e, max_epoch, warm, cons)
        losses += rpo_oracle.train_steps(m, opt, batches)
        if e == 0:
            np.testing.assert_allclose(m.text_prompt.detach().numpy(), g["text_prompt_e1"], atol=2e-6)
            np.testing.assert_allclose(m.img_prompt.detach().numpy(), g["img_prompt_e1"], atol=2e-6)
    np.testing.assert_allclose(losses, g["losses"][:2 * iters], atol=5e-5)


def test_algorithmic_flops_match_survey():
    """SURVEY.md section 8d quotes these to 2 decimals (GFLOP)."""
    lens = [10, 10, 14, 11, 8, 8, 9, 8, 8, 11, 8, 10, 13, 10, 11, 10, 10, 10, 10]
    f, b = flops_image(vit_b16())
    assert round(f / 1e9, 2) == 38.72 and round(b / 1e9, 2) == 3.59
    assert round(flops_text(vit_b16(), lens) / 1e9, 2) == 58.08
    for K, fi, ft in ((4, 36.32, 9.68), (8, 37.52, 19.36), (16, 39.91, 38.72), (48, 49.49, 116.16)):
        c = vit_b16(K=K)
        assert round(sum(flops_image(c)) / 1e9, 2) == fi
        assert round(flops_text(c, lens) / 1e9, 2) == ft
    f, b = flops_image(vit_l14())
    assert round(f / 1e9, 2) == 174.75 and round(b / 1e9, 2) == 12.72
    assert round(flops_text(vit_l14(), lens) / 1e9, 2) == 130.51


@pytest.mark.parametrize("tag,depth,B", [("d2_b3", 2, 3), ("d12_b2", 12, 2)])
def test_plain_clip_oracle_matches_reference_clip_forward(tag, depth, B):
    """The unmasked towers (clip/model.py:344-372; what trainers/zsclip.py and the sibling trainers run): the oracle's
    restatement against logits / features produced by the reference's own CLIP.forward (tools/make_golden_plainclip.py)."""
    import os
    from rpo_amd import synth
    from rpo_amd.config import vit_b16
    from oracle.rpo_oracle import plain_clip_forward
    gold = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", f"ref_plainclip_{tag}.npz")))
    cfg = vit_b16(layers_v=depth, layers_t=depth, K=1)
    sd = synth.clip_state_dict(cfg, seed=0, logit_scale=float(np.log(100.0)))
    assert gold["weights_crc"].item().decode() == synth.state_dict_checksum(sd), "fixture made from other weights"
    logits, img_f, txt_f = plain_clip_forward(sd, synth.images(cfg, B), synth.oxford_pets_base_tokens(), cfg.patch)
    assert np.abs(logits.numpy() - gold["logits"]).max() <= 3e-5
    assert np.abs(img_f.numpy() - gold["image_features"]).max() <= 2e-5 * max(1.0, np.abs(gold["image_features"]).max())
    assert np.abs(txt_f.numpy() - gold["text_features"]).max() <= 2e-5 * max(1.0, np.abs(gold["text_features"]).max())


@pytest.mark.parametrize("tag,depth,B,n_ctx", [("d2_b3_ctx4", 2, 3, 4), ("d2_b2_ctx16", 2, 2, 16)])
def test_coop_oracle_matches_reference_trainer(tag, depth, B, n_ctx):
    """CoOp (trainers/coop.py:117-134,196-208,266-270): logits, loss and the gradient of the context vectors of the
    oracle against the reference's own CustomCLIP + cross_entropy + backward (tools/make_golden_plainclip.py)."""
    import os
    from oracle.rpo_oracle import coop_loss_and_grad
    gold = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", f"ref_coop_{tag}.npz")))
    cfg = vit_b16(layers_v=depth, layers_t=depth, K=1)
    sd = synth.clip_state_dict(cfg, seed=0, logit_scale=float(np.log(100.0)))
    assert gold["weights_crc"].item().decode() == synth.state_dict_checksum(sd)
    assert gold["ctx"].shape == (n_ctx, cfg.d_t)
    logits, loss, g = coop_loss_and_grad(sd, synth.images(cfg, B), gold["tokenized_prompts"], gold["ctx"], gold["label"], cfg.patch)
    assert np.abs(logits.numpy() - gold["logits"]).max() <= 3e-5
    assert abs(float(loss) - float(gold["loss"])) <= 1e-5
    assert np.abs(g.numpy() - gold["ctx_grad"]).max() <= 2e-5 * np.abs(gold["ctx_grad"]).max()


@pytest.mark.parametrize("tag,depth,B,n_ctx", [("d2_b3_ctx4_csc", 2, 3, 4), ("d2_b2_ctx4_middle", 2, 2, 4),
                                               ("d2_b2_ctx5_middle_csc", 2, 2, 5), ("d2_b2_ctx4_front", 2, 2, 4)])
def test_coop_oracle_options_match_reference_trainer(tag, depth, B, n_ctx):
    """The reference's non-default CoOp options -- class-specific contexts (trainers/coop.py:84-86), class token in the
    "middle" / at the "front" (:136-183) -- of the oracle against the reference's own CustomCLIP + cross_entropy +
    backward with those options set (fixtures carry `csc`, `class_token_position`, and the tokenizer's `name_lens`,
    which the oracle must recover from the token ids alone)."""
    import os
    import torch
    from oracle.rpo_oracle import coop_loss_and_grad
    gold = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", f"ref_coop_{tag}.npz")))
    cfg = vit_b16(layers_v=depth, layers_t=depth, K=1)
    sd = synth.clip_state_dict(cfg, seed=0, logit_scale=float(np.log(100.0)))
    assert gold["weights_crc"].item().decode() == synth.state_dict_checksum(sd)
    csc, pos = bool(gold["csc"]), str(gold["class_token_position"])
    assert gold["ctx"].shape == ((cfg.n_cls, n_ctx, cfg.d_t) if csc else (n_ctx, cfg.d_t))
    toks = gold["tokenized_prompts"]
    assert np.array_equal(toks.argmax(-1) - n_ctx - 2, gold["name_lens"])      # name_len from the ids = the tokenizer's
    logits, loss, g = coop_loss_and_grad(sd, synth.images(cfg, B), toks, gold["ctx"], gold["label"], cfg.patch,
                                         class_token_position=pos)
    assert np.abs(logits.numpy() - gold["logits"]).max() <= 3e-5
    assert abs(float(loss) - float(gold["loss"])) <= 1e-5
    assert np.abs(g.numpy() - gold["ctx_grad"]).max() <= 2e-5 * np.abs(gold["ctx_grad"]).max()
    if pos != "end":                                                          # the position matters: "end" gives other logits
        other, _, _ = coop_loss_and_grad(sd, synth.images(cfg, B), toks, gold["ctx"], gold["label"], cfg.patch)
        assert np.abs(other.numpy() - gold["logits"]).max() > 1e-3


@pytest.mark.parametrize("tag,depth,B", [("d2_b1_ctx4", 2, 1), ("d2_b3_ctx4", 2, 3)])
def test_cocoop_oracle_matches_reference_trainer(tag, depth, B):
    """oracle.rpo_oracle.cocoop_loss_and_grads against the reference's own cocoop.CustomCLIP + backward
    (tests/golden/ref_cocoop_*.npz, tools/make_golden_cocoop.py): logits, loss, and the gradient of every trained tensor
    (ctx, meta-net weights and biases)."""
    import os
    from helpers import GOLDEN
    g = dict(np.load(os.path.join(GOLDEN, f"ref_cocoop_{tag}.npz")))
    cfg = vit_b16(layers_v=depth, layers_t=depth, K=1)
    sd = synth.clip_state_dict(cfg, seed=0, logit_scale=float(np.log(100.0)))
    image = synth.images(cfg, B)
    meta = {k: g[k] for k in ("w1", "b1", "w2", "b2")}
    logits, loss, grads = rpo_oracle.cocoop_loss_and_grads(sd, image, g["tokenized_prompts"], g["ctx"], meta, g["label"], cfg.patch)
    np.testing.assert_allclose(logits.numpy(), g["logits"], atol=3e-5, rtol=0)
    assert abs(float(loss) - float(g["loss"])) < 2e-5
    for k in ("ctx", "w1", "b1", "w2", "b2"):
        ref = g["g_" + k]
        assert np.abs(grads[k].numpy() - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), k
