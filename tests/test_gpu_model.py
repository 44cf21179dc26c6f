"""End-to-end parity on the MI355X: CustomCLIP / RPO (HIP path, through the C ABI)
against the golden vectors captured from the REAL reference (tests/golden/) and
against the CPU oracle.

Tolerance (BASELINE.json north_star): logits and learned prompts within 1e-3 of
the reference in fp32.  The f32 mode is asserted at 1e-3 absolute on logits / loss /
updated prompts and 1e-3 relative-to-max on gradients (measured ~1e-5).
The bf16 throughput mode cannot meet 1e-3 at logit scale 100;
it is asserted at a documented looser bound and its measured error is printed.
"""
import functools
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# Tests of the measured-slower experiments (include/rpo_amd_experimental.h): they need the
# -DRPO_EXPERIMENTAL library and run only when the whole pytest process is started with RPO_EXPERIMENTAL=1
experimental = pytest.mark.skipif(os.environ.get("RPO_EXPERIMENTAL") != "1",
                                  reason="experiment: run with RPO_EXPERIMENTAL=1 (loads the -DRPO_EXPERIMENTAL library)")

from helpers import CASES, load_golden, workload  # noqa: E402
from rpo_amd import synth  # noqa: E402

TOL_F32 = 1e-3
BF16_LOGIT_ATOL = 0.12         # logits are O(1..8) at scale 100; measured <= 0.06 (printed by the test)
BF16_GRAD_REL = 0.05           # relative to max |grad|; measured 2.0-2.3 %
F16_LOGIT_ATOL = 1e-2          # native IEEE-half storage mode (TRAINER.RPO.PREC = fp16 / amp): 8x finer than bf16
F16_GRAD_REL = 6e-3


def _model(tag, act, max_batch=None):
    from rpo_amd.custom_clip import CustomCLIP
    cfg, sd, toks, tp, ip, image, label = workload(tag)
    m = CustomCLIP(cfg, sd, toks, "cuda:0", act, max_batch=max_batch or image.shape[0], prompts=(tp, ip))
    return m, torch.from_numpy(image).cuda(), torch.from_numpy(label).cuda()


def _relmax(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("tag", list(CASES))
def test_f32_matches_reference_golden(tag):
    g = load_golden(tag)
    m, image, label = _model(tag, torch.float32)
    m.prompt_learner.eval()
    logits = m(image).cpu().numpy()
    assert np.abs(logits - g["logits"]).max() <= TOL_F32
    m.prompt_learner.train()
    loss = m(image, label)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) <= TOL_F32
    gt = m.prompt_learner.text_prompt.grad.cpu().numpy()
    gi = m.prompt_learner.img_prompt.grad.cpu().numpy()
    rt, ri = _relmax(gt, g["g_text"]), _relmax(gi, g["g_img"])
    print(f"[f32 {tag}] logits err {np.abs(logits - g['logits']).max():.2e} loss err "
          f"{abs(loss.item() - float(g['loss'])):.2e} g_text rel {rt:.2e} g_img rel {ri:.2e}")
    assert rt <= TOL_F32 and ri <= TOL_F32


def test_f32_prompt_rows_per_block():
    """Bisecting aid: prompt rows after every block vs the reference's forward hooks."""
    tag = "d2_k8_b3"
    g = load_golden(tag)
    m, image, label = _model(tag, torch.float32)
    eng, cfg = m.engine, m.cfg
    eng.forward_backward(image, label)
    torch.cuda.synchronize()
    B, N, K = image.shape[0], cfg.n_frozen, cfg.K
    for l in range(cfg.layers_v):
        rows = eng.x[l + 1][B * N:B * (N + K)].view(B, K, cfg.d_v).cpu().numpy()
        assert np.abs(rows - g["img_rows"][l]).max() <= TOL_F32, f"image block {l}"
    for l in range(cfg.layers_t):
        rows = eng.xt[l + 1].view(cfg.n_cls, K, cfg.d_t).cpu().numpy()
        assert np.abs(rows - g["text_rows"][l]).max() <= TOL_F32, f"text block {l}"


def test_f32_vitl14_blocks_match_reference_modules():
    """ViT-L/14 widths (width 1024 / 16 heads / 257 + 24 tokens; text width 768 / 12 heads) against the reference's own
    CustomCLIP run at those widths (tools/make_golden_vitl14_ref.py: `ref_vitl14_d2_k24_b2.npz`, 2 + 2 layers): the
    prompt rows after every block of both towers, eval logits, loss and both gradients, fp32 mode within 1e-3."""
    from rpo_amd.config import vit_l14
    from rpo_amd.custom_clip import CustomCLIP
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vitl14_d2_k24_b2.npz")))
    cfg = vit_l14(layers_v=2, layers_t=2, K=24)
    toks = synth.oxford_pets_base_tokens()
    sd = synth.clip_state_dict(cfg, seed=0, token_rows=np.unique(toks).tolist() + [49407])
    tp, ip = synth.prompts(cfg, sd, seed=7)
    image, label = torch.from_numpy(synth.images(cfg, 2)).cuda(), torch.from_numpy(synth.labels(cfg, 2)).cuda()
    m = CustomCLIP(cfg, sd, toks, "cuda:0", torch.float32, max_batch=2, prompts=(tp, ip))
    m.prompt_learner.eval()
    logits = m(image).cpu().numpy()
    assert np.abs(logits - g["logits"]).max() <= TOL_F32
    eng = m.engine
    eng.forward_backward(image, label)
    torch.cuda.synchronize()
    B, N, K = 2, cfg.n_frozen, cfg.K
    for l in range(cfg.layers_v):
        rows = eng.x[l + 1][B * N:B * (N + K)].view(B, K, cfg.d_v).cpu().numpy()
        assert np.abs(rows - g["img_rows"][l]).max() <= TOL_F32, f"image block {l}"
    for l in range(cfg.layers_t):
        rows = eng.xt[l + 1].view(cfg.n_cls, K, cfg.d_t)[:4].cpu().numpy()
        assert np.abs(rows - g["text_rows"][l]).max() <= TOL_F32, f"text block {l}"
    assert abs(float(eng.loss.item()) - float(g["loss"])) <= TOL_F32
    assert _relmax(eng.g_text.cpu().numpy(), g["g_text"]) <= TOL_F32 and _relmax(eng.g_img.cpu().numpy(), g["g_img"]) <= TOL_F32


@pytest.mark.parametrize("tag", ["d2_k8_b3", "d12_k24_b4"])
def test_f32_sgd_steps_match_reference(tag):
    from rpo_amd.trainer import RPO, OptimConfig
    g = load_golden(tag)
    cfg, sd, toks, tp, ip, _, _ = workload(tag)
    B = CASES[tag][2]
    lr, mom, wd = (float(v) for v in g["sgd_hparams"])
    oc = OptimConfig(lr=lr, momentum=mom, weight_decay=wd, warmup_epoch=0, lr_scheduler="constant")
    tr = RPO(cfg, sd, toks, oc, "cuda:0", torch.float32, batch_size=B, num_batches=10 ** 9, prompts=(tp, ip))
    losses = []
    for step in range(4):
        batch = {"img": torch.from_numpy(synth.images(cfg, B, seed=1234 + 10 * step)),
                 "label": torch.from_numpy(synth.labels(cfg, B, seed=4321 + 10 * step))}
        losses.append(tr.forward_backward(batch)["loss"])
        if step in (0, 3):
            t = tr.model.prompt_learner.text_prompt.detach().cpu().numpy()
            i = tr.model.prompt_learner.img_prompt.detach().cpu().numpy()
            assert np.abs(t - g[f"text_prompt_step{step + 1}"]).max() <= TOL_F32
            assert np.abs(i - g[f"img_prompt_step{step + 1}"]).max() <= TOL_F32
    assert np.abs(np.asarray(losses) - g["sgd_losses"]).max() <= TOL_F32


@pytest.mark.parametrize("tag", ["d2_k8_b3", "d12_k24_b4"])
def test_bf16_error_is_bounded_and_reported(tag):
    g = load_golden(tag)
    m, image, label = _model(tag, torch.bfloat16)
    m.prompt_learner.eval()
    logits = m(image).cpu().numpy()
    m.prompt_learner.train()
    loss = m(image, label)
    loss.backward()
    gt = m.prompt_learner.text_prompt.grad.cpu().numpy()
    gi = m.prompt_learner.img_prompt.grad.cpu().numpy()
    le = np.abs(logits - g["logits"]).max()
    rt, ri = _relmax(gt, g["g_text"]), _relmax(gi, g["g_img"])
    print(f"[bf16 {tag}] logits err {le:.3e} loss err {abs(loss.item() - float(g['loss'])):.3e} "
          f"g_text rel {rt:.3e} g_img rel {ri:.3e}")
    assert np.isfinite(logits).all() and le <= BF16_LOGIT_ATOL
    assert rt <= BF16_GRAD_REL and ri <= BF16_GRAD_REL
    assert (logits.argmax(-1) == g["logits"].argmax(-1)).mean() >= 0.5


@pytest.mark.parametrize("tag", ["d2_k8_b3", "d2_k16_b2", "d12_k24_b4"])
def test_f16_mode_matches_reference_golden(tag):
    """Row f4: the native fp16 storage mode (f16 weights / activations on v_mfma_f32_32x32x16_f16, fp32 accumulate,
    fp32 residual stream / LayerNorm / softmax / head / optimiser) against the reference goldens at a stated bound:
    logits within 1e-2 absolute (they are O(1..8)), prompt gradients within 0.6 % of their largest entry."""
    g = load_golden(tag)
    m, image, label = _model(tag, torch.float16)
    m.prompt_learner.eval()
    logits = m(image).cpu().numpy()
    m.prompt_learner.train()
    loss = m(image, label)
    loss.backward()
    gt = m.prompt_learner.text_prompt.grad.cpu().numpy()
    gi = m.prompt_learner.img_prompt.grad.cpu().numpy()
    le = np.abs(logits - g["logits"]).max()
    rt, ri = _relmax(gt, g["g_text"]), _relmax(gi, g["g_img"])
    print(f"[f16 {tag}] logits err {le:.3e} loss err {abs(loss.item() - float(g['loss'])):.3e} "
          f"g_text rel {rt:.3e} g_img rel {ri:.3e}")
    assert np.isfinite(logits).all() and le <= F16_LOGIT_ATOL
    assert abs(loss.item() - float(g["loss"])) <= F16_LOGIT_ATOL
    assert rt <= F16_GRAD_REL and ri <= F16_GRAD_REL
    assert (logits.argmax(-1) == g["logits"].argmax(-1)).all()


@pytest.mark.parametrize("early", ["0", "1"])
def test_early_text_forward_changes_nothing(early, monkeypatch, tmp_path):
    """RPO_EARLY_TEXT (on by default at small batches): the next step's text forward launched behind this step's text
    backward + text SGD.  Same bits as eager launches over steps that include an epoch boundary (new learning rate), an
    eval call between two steps and a checkpoint reload (prompts changed from outside: the early features are stale)."""
    from rpo_amd.trainer import RPO
    monkeypatch.setenv("RPO_EARLY_TEXT", early)
    tag = "d2_k8_b3"
    cfg, sd, toks, tp, ip, image, label = workload(tag)
    B = image.shape[0]
    outs = []
    for use_graph in (False, True):
        tr = RPO(cfg, sd, toks, None, "cuda:0", torch.bfloat16, batch_size=B, num_batches=2, use_graph=use_graph, prompts=(tp, ip))
        rec = []
        for step in range(7):
            batch = {"img": torch.from_numpy(synth.images(cfg, B, seed=50 + step)),
                     "label": torch.from_numpy(synth.labels(cfg, B, seed=60 + step))}
            rec.append(tr.forward_backward(batch)["loss"])
            if step == 2:
                rec.append(tr.model_inference(batch["img"].cuda()).float().cpu().numpy().tobytes())
            if step == 3:
                tr.save_model(str(tmp_path / f"g{int(use_graph)}"), epoch=1)
            if step == 5:
                tr.load_model(str(tmp_path / f"g{int(use_graph)}"), epoch=1)
        if use_graph:
            assert tr._early_text == (early == "1")
        outs.append((rec, tr.engine.params.clone()))
    assert outs[0][0] == outs[1][0], "early text forward changed a loss / the eval logits"
    assert torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("early_text", ["0", "1"])
def test_early_patch_embed_changes_nothing(early_text, monkeypatch):
    """step_async(next_image=...): the next batch's im2col + patch GEMM under this step's backward (side stream, behind the
    text chain).  Same bits as eager launches, over a sequence that keeps the promise (next call's image IS next_image),
    breaks it (another tensor; the named tensor modified in place afterwards), has an eval call between two steps (which
    overwrites the image tower's input buffers) and more distinct buffers than the trainer keeps patch graphs for."""
    from rpo_amd.trainer import RPO
    monkeypatch.setenv("RPO_EARLY_TEXT", early_text)
    cfg, sd, toks, tp, ip, image, label = workload("d2_k8_b3")
    B = image.shape[0]
    nb = 12
    imgs = [torch.from_numpy(synth.images(cfg, B, seed=50 + i)).cuda() for i in range(nb)]
    labs = [torch.from_numpy(synth.labels(cfg, B, seed=60 + i)).cuda() for i in range(nb)]
    outs = []
    for use_graph in (False, True):
        tr = RPO(cfg, sd, toks, None, "cuda:0", torch.bfloat16, batch_size=B, num_batches=10 ** 9, use_graph=use_graph, prompts=(tp, ip))
        rec, used = [], 0
        cur = [t.clone() for t in imgs]
        for step in range(nb - 1):
            promise = cur[step + 1]
            if step == 4:
                promise = cur[0]                                  # a promise that is not kept: the next call passes cur[5]
            loss = tr.step_async(cur[step], labs[step], promise if use_graph else None)
            if step == 6:
                cur[7].mul_(0.5)                                  # the named buffer changes after it was named
            rec.append(float(loss.item()))
            if step == 2:
                rec.append(tr.model_inference(cur[9]).float().cpu().numpy().tobytes())
            if use_graph and step >= 1:
                used += tr._patch_tag is not None
        if use_graph:
            assert used >= nb - 3 and tr._g_img_fwd_np is not None and len(tr._g_patch) <= 9
        torch.cuda.synchronize()
        outs.append((rec, tr.engine.params.clone()))
    assert outs[0][0] == outs[1][0], "the early patch embed changed a loss / the eval logits"
    assert torch.equal(outs[0][1], outs[1][1])


def test_one_graph_per_step_equals_eager(monkeypatch):
    """RPO_ONE_GRAPH=1: the whole step -- both streams, fork and join inside the capture -- as one HIP graph."""
    from rpo_amd.trainer import RPO
    monkeypatch.setenv("RPO_ONE_GRAPH", "1")
    monkeypatch.setenv("RPO_EARLY_TEXT", "0")
    cfg, sd, toks, tp, ip, image, label = workload("d2_k8_b3")
    B = image.shape[0]
    outs = []
    for use_graph in (False, True):
        tr = RPO(cfg, sd, toks, None, "cuda:0", torch.bfloat16, batch_size=B, num_batches=2, use_graph=use_graph, prompts=(tp, ip))
        ls = []
        for step in range(5):
            batch = {"img": torch.from_numpy(synth.images(cfg, B, seed=50 + step)),
                     "label": torch.from_numpy(synth.labels(cfg, B, seed=60 + step))}
            ls.append(tr.forward_backward(batch)["loss"])
        if use_graph:
            assert tr._g_step is not None, "the one-graph capture fell back"
        outs.append((ls, tr.engine.params.clone()))
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("act", [torch.float32, torch.bfloat16, torch.float16])
def test_graph_replay_equals_eager(act):
    from rpo_amd.trainer import RPO
    tag = "d2_k8_b3"
    cfg, sd, toks, tp, ip, image, label = workload(tag)
    B = image.shape[0]
    outs = []
    for use_graph in (False, True):
        tr = RPO(cfg, sd, toks, None, "cuda:0", act, batch_size=B, use_graph=use_graph, prompts=(tp, ip))
        ls = []
        for step in range(3):
            batch = {"img": torch.from_numpy(synth.images(cfg, B, seed=50 + step)),
                     "label": torch.from_numpy(synth.labels(cfg, B, seed=60 + step))}
            ls.append(tr.forward_backward(batch)["loss"])
        outs.append((ls, tr.engine.params.clone()))
    assert outs[0][0] == outs[1][0], "graph replay must be bit-identical to eager launches"
    assert torch.equal(outs[0][1], outs[1][1])


def test_eval_graph_equals_eager_and_tracks_prompt_updates():
    """The eval branch (trainers/rpo.py:229-232) replays a captured graph per batch size: same bits as eager
    launches, per-image results independent of the batch they came in, and fresh text features after the prompts
    change."""
    m, image, label = _model("d2_k8_b3", torch.bfloat16)
    m.prompt_learner.eval()
    eng = m.engine
    a = eng.forward_eval(image, use_graph=False).clone()
    b = eng.forward_eval(image).clone()
    c = eng.forward_eval(image).clone()
    assert torch.equal(a, b) and torch.equal(b, c)
    b2 = eng.forward_eval(image[:2].contiguous()).clone()
    assert torch.equal(b2, a[:2]) and set(eng._eval_graphs) == {3, 2}
    with torch.no_grad():
        m.prompt_learner.text_prompt.add_(0.01)
    d = m(image)
    e = eng.forward_eval(image, use_graph=False).clone()
    assert torch.equal(d, e) and not torch.equal(d, a)


def test_ragged_and_max_length_classes_f32():
    """len_c from 3 up to the maximum 77-K, 5 classes; vs the dense CPU oracle."""
    from oracle.rpo_oracle import OracleRPO
    from rpo_amd.config import vit_b16
    from rpo_amd.custom_clip import CustomCLIP
    cfg = vit_b16(layers_v=1, layers_t=2, K=6, n_cls=5)
    toks = synth.synthetic_tokens(cfg, [3, 71, 20, 8, 71])
    sd = synth.clip_state_dict(cfg, seed=3, token_rows=np.unique(toks).tolist() + [49407])
    tp, ip = synth.prompts(cfg, sd, seed=11)
    image, label = synth.images(cfg, 2), synth.labels(cfg, 2)
    o = OracleRPO(sd, toks, cfg.K, cfg.patch)
    o.set_prompts(tp, ip)
    out, gt, gi = o.loss_and_grads(image, label)
    m = CustomCLIP(cfg, sd, toks, "cuda:0", torch.float32, max_batch=2, prompts=(tp, ip))
    loss = m(torch.from_numpy(image).cuda(), torch.from_numpy(label).cuda())
    loss.backward()
    assert abs(loss.item() - out.loss.item()) <= TOL_F32
    assert _relmax(m.prompt_learner.text_prompt.grad.cpu().numpy(), gt.numpy()) <= TOL_F32
    assert _relmax(m.prompt_learner.img_prompt.grad.cpu().numpy(), gi.numpy()) <= TOL_F32


def test_smaller_batch_than_max_and_linearity():
    """Size-independent property at full width: the loss gradient is linear in the per-image
    losses, so g(batch of 4) == mean of g(each image alone) (mean-CE)."""
    tag = "d2_k8_b3"
    cfg, sd, toks, tp, ip, _, _ = workload(tag)
    from rpo_amd.custom_clip import CustomCLIP
    m = CustomCLIP(cfg, sd, toks, "cuda:0", torch.float32, max_batch=4, prompts=(tp, ip))
    img = torch.from_numpy(synth.images(cfg, 4, seed=5)).cuda()
    lab = torch.from_numpy(synth.labels(cfg, 4, seed=6)).cuda()
    m.engine.forward_backward(img, lab)
    g_all = m.engine.grads.clone()
    acc = torch.zeros_like(g_all)
    for b in range(4):
        m.engine.forward_backward(img[b:b + 1].contiguous(), lab[b:b + 1].contiguous())
        acc += m.engine.grads
    torch.cuda.synchronize()
    assert (acc / 4 - g_all).abs().max().item() <= 1e-5 * max(1.0, g_all.abs().max().item())


def test_checkpoint_roundtrip(tmp_path):
    from rpo_amd.trainer import RPO
    tag = "d1_k4_b2"
    cfg, sd, toks, tp, ip, image, label = workload(tag)
    tr = RPO(cfg, sd, toks, None, "cuda:0", torch.float32, batch_size=2, prompts=(tp, ip))
    tr.forward_backward({"img": torch.from_numpy(image), "label": torch.from_numpy(label)})
    fn = tr.save_model(str(tmp_path), epoch=3)
    ck = torch.load(fn, map_location="cpu", weights_only=True)
    assert set(ck["state_dict"]) == {"text_prompt", "img_prompt"} and ck["epoch"] == 3
    tr2 = RPO(cfg, sd, toks, None, "cuda:0", torch.float32, batch_size=2, prompts=(tp, ip))
    tr2.load_model(str(tmp_path), epoch=3)
    assert torch.equal(tr2.engine.params, tr.engine.params)
    a = tr.model_inference(torch.from_numpy(image).cuda())
    b = tr2.model_inference(torch.from_numpy(image).cuda())
    assert torch.equal(a, b)


@pytest.mark.parametrize("act,tol", [(torch.float32, TOL_F32), (torch.bfloat16, None)])
def test_vit_l14_dims_against_oracle(act, tol):
    """BASELINE.json configs[3]: ViT-L/14 (d=1024, 16 heads, patch 14 -> 257 frozen tokens, text width 768,
    embed 768).  The reference cannot run it (SURVEY.md finding 7); the dimension-generic oracle pins it.
    Depth is reduced to keep the CPU oracle fast; every width / head count / tile path is the real one."""
    from oracle.rpo_oracle import OracleRPO
    from rpo_amd.config import vit_l14
    from rpo_amd.custom_clip import CustomCLIP
    cfg = vit_l14(layers_v=2, layers_t=2, K=24)
    toks = synth.oxford_pets_base_tokens()
    sd = synth.clip_state_dict(cfg, seed=5, token_rows=np.unique(toks).tolist() + [49407])
    tp, ip = synth.prompts(cfg, sd, seed=9)
    image, label = synth.images(cfg, 2), synth.labels(cfg, 2)
    o = OracleRPO(sd, toks, cfg.K, cfg.patch)
    o.set_prompts(tp, ip)
    out, gt, gi = o.loss_and_grads(image, label)
    m = CustomCLIP(cfg, sd, toks, "cuda:0", act, max_batch=2, prompts=(tp, ip))
    loss = m(torch.from_numpy(image).cuda(), torch.from_numpy(label).cuda())
    loss.backward()
    el = abs(loss.item() - out.loss.item())
    rt = _relmax(m.prompt_learner.text_prompt.grad.cpu().numpy(), gt.numpy())
    ri = _relmax(m.prompt_learner.img_prompt.grad.cpu().numpy(), gi.numpy())
    print(f"[ViT-L/14 {act}] loss err {el:.2e} g_text rel {rt:.2e} g_img rel {ri:.2e}")
    if tol is not None:
        assert el <= tol and rt <= tol and ri <= tol
    else:
        assert el <= 0.1 and rt <= BF16_GRAD_REL and ri <= BF16_GRAD_REL


def test_dry_scale_schedules_agree_under_real_rccl():
    """`bench.py --dry-scale` (round 6): one GPU, a real RCCL communicator of one rank.  The three ways a rank may issue the
    step's collectives -- captured in the step's graphs (the N > 1 default: asserted captured), eager between the graphs (the
    per-process-group fallback), ONE all-reduce after the join (RPO_ONE_COLLECTIVE=1, what gloo uses) -- give bit-identical
    losses and prompts over five steps across an epoch boundary, and the line carries what the communicator reports."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "RPO_FORCE_DIST")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry-scale", "--batch", "4"], capture_output=True,
                       text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    ds = d["dry_scale"]
    assert ds["ok"] and ds["bit_identical"]
    assert ds["runs"]["graph"]["collectives_in_graph"] and ds["runs"]["graph"]["text_allreduce_in_graph"]
    assert ds["runs"]["graph"]["tail_graphs"] >= 2 and not ds["runs"]["eager"]["collectives_in_graph"]
    assert not ds["runs"]["one_collective"]["split_collective"] and ds["runs"]["graph"]["split_collective"]
    assert d["rccl_ranks"] == 1 and d["backend"] == "nccl" and len(d["rank_devices"]) == 1 and "rank 0" in d["rank_devices"][0]


def test_bench_runs_under_torchrun_with_rccl(tmp_path):
    """The launch line the driver uses for N > 1, with one rank (this box has one GPU): RCCL process group,
    prompt broadcast, gradient all-reduce and barriers all execute (RPO_FORCE_DIST=1)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, RPO_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
           "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "4", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["parallelism"] == "dp1"
    assert np.isfinite(d["config"]["final_loss"]) and d["config"]["collective"].startswith("rccl")
    # round 5: the two all-reduces and the SGD launch are captured in the step's HIP graphs (RCCL under stream capture);
    # the same run with eager collectives (RPO_NO_GRAPH_COLLECTIVES=1) must end at the same loss
    assert d["config"]["collectives_in_graph"] is True and d["config"]["host_us_per_step"] > 0
    # round 6: what the communicator itself reports, and the collective's DEVICE time (events around graph-replayed all-reduces)
    assert d["config"]["rccl_ranks"] == 1 and len(d["config"]["rank_devices"]) == 1
    assert d["config"]["collective_us"] > 0 and "graph replays" in d["config"]["collective_us_how"]
    assert d["config"]["collective_host_clocked_us"] > 0
    r2 = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(env, RPO_NO_GRAPH_COLLECTIVES="1"), cwd=root)
    assert r2.returncode == 0, r2.stderr[-2000:]
    d2 = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][-1])
    assert d2["config"]["collectives_in_graph"] is False
    assert d2["config"]["final_loss"] == d["config"]["final_loss"]
    # batch 4 runs the early text forward (text half of the tail captured on the side stream); the one-tail order that
    # batch 32 per GPU uses -- the driver's scaling run -- must end at the same loss as well
    r3 = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(env, RPO_EARLY_TEXT="0"), cwd=root)
    assert r3.returncode == 0, r3.stderr[-2000:]
    d3 = json.loads([l for l in r3.stdout.splitlines() if l.startswith("{")][-1])
    assert d3["config"]["collectives_in_graph"] is True and d3["config"]["final_loss"] == d["config"]["final_loss"]


def test_bench_two_rank_flow_on_one_gpu(tmp_path):
    """The N = 2 control flow of bench.py end to end on this one-GPU box, launched the way the driver launches N = 1:
    plain `python bench.py --gpus 2 ...` (no launcher, no WORLD_SIZE) -- the script re-executes itself under
    torch.distributed.run.  Both ranks on cuda:0, gloo instead of RCCL (RCCL refuses two ranks on one device).
    Sharded synthetic batches, prompt broadcast, per-step gradient all-reduce, barriers, max-over-ranks timing, rank 0
    prints the one JSON line with the GLOBAL batch."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(RPO_DIST_BACKEND="gloo", RPO_ALL_RANKS_ON_GPU0="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--batch", "4", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0)"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["config"]["parallelism"] == "dp2"
    assert d["scaling"] == "weak" and d["value"] > 0 and np.isfinite(d["config"]["final_loss"])
    assert "2 ranks" in d["config"]["collective"]
    assert "cpu_baseline" not in d and "precision" not in d      # N = 1 only
    assert d["roofline"]["dominant_kernel"]["avg_us"] > 0


def test_bench_eight_rank_flow_on_one_gpu():
    """BASELINE.json configs[2]'s control flow (N = 8, global batch = 8 x per-GPU batch) end to end on this one-GPU box:
    `python bench.py --gpus 8` re-executes itself under torch.distributed.run; all ranks on cuda:0, gloo instead of
    RCCL (RCCL refuses several ranks on one device).  Exercises: ONE weight generation per node (local rank 0 writes,
    seven ranks memory-map), eight shards of distinct synthetic data, prompt broadcast, per-step all-reduce of the
    flat gradient buffer over 8 ranks, barriers, max-over-ranks timing, exactly one JSON line with the GLOBAL batch and
    the measured collective time.  No scaling curve: 8 ranks share one GPU here."""
    import glob
    import json
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(RPO_DIST_BACKEND="gloo", RPO_ALL_RANKS_ON_GPU0="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--batch", "4", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0)"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 32 and d["config"]["parallelism"] == "dp8"
    assert d["scaling"] == "weak" and d["value"] > 0 and np.isfinite(d["config"]["final_loss"])
    assert "8 ranks" in d["config"]["collective"]
    assert d["config"]["collective_us"] > 0 and d["config"]["collective_bytes"] == 4 * 24 * (512 + 768)
    assert "cpu_baseline" not in d and "precision" not in d      # N = 1 only
    # round 4: the text half of the gradient goes out behind the text chain, the image half after the image chain; the
    # line carries every rank's own step time and what the host pinning did
    # (the split schedule needs an asynchronous backend: under this test's gloo stand-in the trainer keeps ONE all-reduce
    #  after the join -- round-4 advisor; test_bench_runs_under_torchrun_with_rccl covers the RCCL schedule)
    assert d["config"]["collective_schedule"].startswith("one all-reduce") and 0 < d["config"]["collective_share_of_step"] < 1
    rk = d["config"]["rank_ms_per_step"]
    assert len(rk["all"]) == 8 and rk["min"] <= rk["max"] and abs(rk["max"] - d["ms_per_step"]) < 1e-3
    assert "pinned" in d["config"]["host"]
    assert not glob.glob(os.path.join(tempfile.gettempdir(), "rpo_amd_weights_*")), "shared weight directory left behind"


_EDGE_ORACLE = {}


def _edge_oracle(K, lens, B, layers_t, seed_sd, seed_p):
    """(cfg, toks, sd, prompts, image, label, oracle outputs) of an edge-shape case: the dense CPU oracle runs ONCE per
    case however many storage modes are compared against it (1000 classes x 77 tokens take it tens of seconds)."""
    from oracle.rpo_oracle import OracleRPO
    from rpo_amd.config import vit_b16
    key = (K, tuple(lens), B, layers_t, seed_sd, seed_p)
    if key not in _EDGE_ORACLE:
        cfg = vit_b16(layers_v=1, layers_t=layers_t, K=K, n_cls=len(lens))
        toks = synth.synthetic_tokens(cfg, lens)
        sd = synth.clip_state_dict(cfg, seed=seed_sd, token_rows=np.unique(toks).tolist() + [49407])
        tp, ip = synth.prompts(cfg, sd, seed=seed_p)
        image, label = synth.images(cfg, B), synth.labels(cfg, B)
        o = OracleRPO(sd, toks, cfg.K, cfg.patch)
        o.set_prompts(tp, ip)
        out, gt, gi = o.loss_and_grads(image, label)
        _EDGE_ORACLE[key] = (cfg, toks, sd, (tp, ip), image, label, float(out.loss.item()), out.logits.detach().numpy().copy(),
                             gt.numpy().copy(), gi.numpy().copy())
        del o
    return _EDGE_ORACLE[key]


# the reference's large-class-count workload (configs/trainers/RPO/imagenet_k24_ep15.yaml:1-35 + scripts/rpo/xd_train.sh:20-28:
# K = 24 on 1000 ImageNet classes; SUN397 has 397): 24 000 prompt rows in the text tower, 19 x 8-14 -> 1000 x 3-24 keys
MANY_CLASS_LENS = {n: [3 + (7 * c) % 22 for c in range(n)] for n in (100, 397, 1000)}


@pytest.mark.parametrize("n_cls", [100, 397, 1000])
@pytest.mark.parametrize("act", [torch.float32, torch.float16, torch.bfloat16], ids=["float32", "float16", "bfloat16"])
def test_many_classes_k24_against_oracle(n_cls, act):
    """K = 24 prompts over 100 / 397 / 1000 classes (the reference trains RPO on ImageNet's 1000: the text tower is then
    24 000 prompt rows, 2.3x the image tower's FLOPs -- the inverse of the Oxford-Pets bench): loss, both prompt gradients
    and the eval logits against the dense CPU oracle at depth 1, in every storage mode, through the kernels the engine
    selects at that size (rpo_gemm_nt's wide tiles from 2048 rows on, the head on the fp32 matrix pipe, 8000 attention waves)."""
    from rpo_amd.custom_clip import CustomCLIP
    cfg, toks, sd, prm, image, label, o_loss, o_logits, gt, gi = _edge_oracle(24, MANY_CLASS_LENS[n_cls], 2, 1, 5, 13)
    m = CustomCLIP(cfg, sd, toks, "cuda:0", act, max_batch=2, prompts=prm)
    loss = m(torch.from_numpy(image).cuda(), torch.from_numpy(label).cuda())
    loss.backward()
    rt = _relmax(m.prompt_learner.text_prompt.grad.cpu().numpy(), gt)
    ri = _relmax(m.prompt_learner.img_prompt.grad.cpu().numpy(), gi)
    m.prompt_learner.eval()
    le = float(np.abs(m(torch.from_numpy(image).cuda()).cpu().numpy() - o_logits).max())
    print(f"[{act} n_cls={n_cls} K=24] loss err {abs(loss.item() - o_loss):.3e} logits err {le:.3e} g_text rel {rt:.3e} g_img rel {ri:.3e}")
    if act == torch.float32:
        assert abs(loss.item() - o_loss) <= TOL_F32 and rt <= TOL_F32 and ri <= TOL_F32 and le <= TOL_F32
    else:
        la, gr = (F16_LOGIT_ATOL, F16_GRAD_REL) if act == torch.float16 else (BF16_LOGIT_ATOL, BF16_GRAD_REL)
        # (logits: the modes' bounds were set on |logits| <= 8.7 -- 19 classes; over 1000 classes the largest logit is 12.0 and
        #  the maximum runs over 2000 of them: the bound scales with the logits' magnitude, as in the 300-class edge case below.
        #  Measured at 1000 classes: f16 1.05e-2, i.e. 8.7e-4 of the largest logit)
        scale = max(1.0, float(np.abs(o_logits).max()) / 8.7)
        assert abs(loss.item() - o_loss) <= la and rt <= gr and ri <= gr and le <= la * scale


@pytest.mark.parametrize("K,n_cls,B", [(1, 19, 2), (5, 40, 3), (53, 3, 1), (4, 300, 2)])
def test_edge_shapes_against_oracle_f32(K, n_cls, B):
    """K = 1 (reference asserts K >= 1), a class count that is not 19, the largest K that still fits the
    context next to the longest prompt (len 24 -> K = 53), batch 1, and a class count beyond one 256-thread pass of
    the head kernels (the reference's ImageNet configs have 500 / 1000 classes)."""
    from oracle.rpo_oracle import OracleRPO
    from rpo_amd.config import vit_b16
    from rpo_amd.custom_clip import CustomCLIP
    cfg = vit_b16(layers_v=1, layers_t=1, K=K, n_cls=n_cls)
    lens = [3 + (7 * c) % 22 for c in range(n_cls)]
    lens[0] = 24 if K == 53 else lens[0]
    toks = synth.synthetic_tokens(cfg, lens)
    sd = synth.clip_state_dict(cfg, seed=2, token_rows=np.unique(toks).tolist() + [49407])
    tp, ip = synth.prompts(cfg, sd, seed=4)
    image, label = synth.images(cfg, B), synth.labels(cfg, B)
    o = OracleRPO(sd, toks, cfg.K, cfg.patch)
    o.set_prompts(tp, ip)
    out, gt, gi = o.loss_and_grads(image, label)
    m = CustomCLIP(cfg, sd, toks, "cuda:0", torch.float32, max_batch=B, prompts=(tp, ip))
    loss = m(torch.from_numpy(image).cuda(), torch.from_numpy(label).cuda())
    loss.backward()
    assert abs(loss.item() - out.loss.item()) <= TOL_F32
    assert _relmax(m.prompt_learner.text_prompt.grad.cpu().numpy(), gt.numpy()) <= TOL_F32
    assert _relmax(m.prompt_learner.img_prompt.grad.cpu().numpy(), gi.numpy()) <= TOL_F32
    m.prompt_learner.eval()
    np.testing.assert_allclose(m(torch.from_numpy(image).cuda()).cpu().numpy(), out.logits.detach().numpy(), atol=TOL_F32)


@pytest.mark.parametrize("act", [torch.float16, torch.bfloat16], ids=["float16", "bfloat16"])
@pytest.mark.parametrize("K,lens", [(6, [3, 71, 20, 8, 71]), (53, [24, 5, 17]), (4, [3 + (7 * c) % 22 for c in range(300)])],
                         ids=["71keys", "K53", "300classes"])
def test_edge_shapes_against_oracle_16bit(K, lens, act):
    """The 16-bit modes' text tower at the shapes the Oxford-Pets fixtures do not reach: prompts of 71 tokens (three key
    tiles of the one-wave attention kernel), K = 53 (two query tiles) and 300 classes -- against the dense CPU oracle,
    at the modes' stated bounds."""
    from oracle.rpo_oracle import OracleRPO
    from rpo_amd.config import vit_b16
    from rpo_amd.custom_clip import CustomCLIP
    cfg = vit_b16(layers_v=1, layers_t=2, K=K, n_cls=len(lens))
    toks = synth.synthetic_tokens(cfg, lens)
    sd = synth.clip_state_dict(cfg, seed=3, token_rows=np.unique(toks).tolist() + [49407])
    tp, ip = synth.prompts(cfg, sd, seed=11)
    image, label = synth.images(cfg, 2), synth.labels(cfg, 2)
    o = OracleRPO(sd, toks, cfg.K, cfg.patch)
    o.set_prompts(tp, ip)
    out, gt, gi = o.loss_and_grads(image, label)
    m = CustomCLIP(cfg, sd, toks, "cuda:0", act, max_batch=2, prompts=(tp, ip))
    loss = m(torch.from_numpy(image).cuda(), torch.from_numpy(label).cuda())
    loss.backward()
    la, gr = (F16_LOGIT_ATOL, F16_GRAD_REL) if act == torch.float16 else (BF16_LOGIT_ATOL, BF16_GRAD_REL)
    rt = _relmax(m.prompt_learner.text_prompt.grad.cpu().numpy(), gt.numpy())
    ri = _relmax(m.prompt_learner.img_prompt.grad.cpu().numpy(), gi.numpy())
    print(f"[{act} K={K} n_cls={len(lens)}] loss err {abs(loss.item() - out.loss.item()):.3e} g_text rel {rt:.3e} g_img rel {ri:.3e}")
    assert abs(loss.item() - out.loss.item()) <= la and rt <= gr and ri <= gr
    m.prompt_learner.eval()
    # (logits: 2 x the mode's bound -- with K = 4 a logit is 100 x the mean of only 4 cosines and the maximum runs over 600
    #  of them: measured 1.22e-2 in f16 here, 1.13e-2 with the fp32-VALU attention kernel of rounds 1-4)
    np.testing.assert_allclose(m(torch.from_numpy(image).cuda()).cpu().numpy(), out.logits.detach().numpy(), atol=2 * la)


FULL_SIZE = [("ViT-B/16", 24, 32, torch.float32), ("ViT-B/16", 24, 32, torch.bfloat16),
             ("ViT-B/16", 24, 32, torch.float16),
             ("ViT-B/16", 4, 32, torch.bfloat16), ("ViT-B/16", 8, 32, torch.bfloat16),
             ("ViT-B/16", 16, 32, torch.float32), ("ViT-B/16", 16, 32, torch.bfloat16),
             ("ViT-B/16", 48, 32, torch.bfloat16),
             ("ViT-L/14", 24, 16, torch.float32), ("ViT-L/14", 24, 16, torch.bfloat16),
             ("ViT-L/14", 24, 16, torch.float16)]


@pytest.mark.parametrize("model,K,B,act", FULL_SIZE, ids=lambda v: str(v).replace("torch.", ""))
def test_full_size_properties(model, K, B, act):
    """BASELINE.json configs[1], [3] and every point of the configs[4] K sweep at FULL size (all layers, the bench's
    batch), where the CPU oracle is too slow to be the checker: size-independent properties of the step, and the
    throughput mode against the exact-f32 mode of the same engine (which tests/ pin to the reference at small depth).
    These are the shapes bench.py runs, i.e. the kernels the heuristics select there -- the one-wave-per-SIMD 256x256
    in-proj, the one-round 224x384 c_fc and 224x96 (ViT-L/14: 288x256 / 288x64) out-proj / c_proj with the hi / lo
    residual stream, the generic tiles at K = 48 -- are compared at model level here (and against the reference's own
    outputs in test_full_size_matches_reference_golden)."""
    from rpo_amd.config import vit_b16, vit_l14
    from rpo_amd.trainer import RPO
    cfg = (vit_b16 if model == "ViT-B/16" else vit_l14)(K=K)
    toks = synth.oxford_pets_base_tokens()
    sd = synth.clip_state_dict(cfg, seed=0, token_rows=np.unique(toks).tolist() + [49407])
    tp, ip = synth.prompts(cfg, sd, seed=7)
    img = torch.from_numpy(synth.images(cfg, B)).cuda()
    lab = torch.from_numpy(synth.labels(cfg, B)).cuda()
    tr = RPO(cfg, sd, toks, None, "cuda:0", act, batch_size=B, prompts=(tp, ip))
    eng = tr.engine
    # (1) determinism: the same batch twice -> bit-identical loss and gradients
    eng.forward_backward(img, lab); torch.cuda.synchronize()
    g0, l0, lg0 = eng.grads.clone(), eng.loss.clone(), eng.logits.clone()
    eng.forward_backward(img, lab); torch.cuda.synchronize()
    assert torch.equal(eng.grads, g0) and torch.equal(eng.loss, l0)
    assert torch.isfinite(g0).all() and torch.isfinite(l0).all()
    # (2) the eval branch returns the logits the train step computed
    tr.model.prompt_learner.eval()
    lg_eval = tr.model(img)
    tr.model.prompt_learner.train()
    assert torch.equal(lg_eval, lg0)
    # (3) loss = mean CE of those logits (fp32 head): recompute on the host
    ce = torch.nn.functional.cross_entropy(lg0.double().cpu(), lab.cpu()).item()
    assert abs(ce - l0.item()) <= 1e-5 * max(1.0, abs(ce))
    # (4) linearity of mean-CE gradients in the batch: g(B) == (g(first half) + g(second half)) / 2
    h = B // 2
    eng.forward_backward(img[:h].contiguous(), lab[:h].contiguous()); torch.cuda.synchronize()
    ga = eng.grads.clone()
    eng.forward_backward(img[h:].contiguous(), lab[h:].contiguous()); torch.cuda.synchronize()
    gb = eng.grads.clone()
    rel = ((ga + gb) / 2 - g0).abs().max().item() / g0.abs().max().item()
    assert rel <= (2e-5 if act == torch.float32 else 2e-2), rel
    # (5) the graph path reproduces the eager path bit-for-bit, and SGD moves the prompts
    before = eng.params.clone()
    loss_graph = tr.forward_backward({"img": img, "label": lab})["loss"]
    assert loss_graph == l0.item()
    assert not torch.equal(eng.params, before)
    # (6) throughput mode vs the exact-f32 mode at this size (f32 run first in the parametrisation caches its result)
    key = (model, K, B)
    if act == torch.float32:
        _F32_FULL[key] = (lg0.cpu(), g0.cpu())
    elif key in _F32_FULL:
        lf, gf = _F32_FULL[key]
        le = (lg0.cpu() - lf).abs().max().item()
        nt = cfg.K * cfg.d_t
        rt = (g0.cpu()[:nt] - gf[:nt]).abs().max().item() / gf[:nt].abs().max().item()
        ri = (g0.cpu()[nt:] - gf[nt:]).abs().max().item() / gf[nt:].abs().max().item()
        print(f"[full size {model} K={K} B={B} {act}] vs f32 mode: logits err {le:.3e} g_text rel {rt:.3e} g_img rel {ri:.3e}")
        # (against the goldens the f16 bound is 1e-2, measured 4e-3 .. 9e-3; at B = 32 the largest of 608 logits
        #  against the f32 MODE reads 1.0e-2, so this cross-check gets twice that)
        la, gr = (BF16_LOGIT_ATOL, BF16_GRAD_REL) if act == torch.bfloat16 else (2 * F16_LOGIT_ATOL, F16_GRAD_REL)
        assert le <= la and rt <= gr and ri <= gr


_F32_FULL = {}


# Full-size fixtures (tools/make_golden_fullsize.py): the REAL reference run on the bench's own shapes (12 layers,
# B = 32, every K of the configs[4] sweep), so that the model-level comparison goes through the kernels the bench
# selects -- gemm_w4 / gemm_w4g / gemm_w4k at M = 32 x 221, the K = 48 path -- and not only through the generic tiles
# that the small-batch goldens reach.  ViT-L/14 (configs[3]): the reference's shipped trainer hard-codes four ViT-B/16
# dimensions (SURVEY.md finding 7); its fixture `ref_full_vitl14_k24_b16` comes from the reference's own CustomCLIP run at
# L/14 widths with those four values supplied from outside (tools/make_golden_vitl14_ref.py).
FULL_GOLDEN = [("ref_full_k24_b32", "ViT-B/16", 24, 32), ("ref_full_k4_b32", "ViT-B/16", 4, 32),
               ("ref_full_k8_b32", "ViT-B/16", 8, 32), ("ref_full_k16_b32", "ViT-B/16", 16, 32),
               ("ref_full_k48_b32", "ViT-B/16", 48, 32), ("ref_full_vitl14_k24_b16", "ViT-L/14", 24, 16)]
FULL_TOL = {torch.float32: (TOL_F32, TOL_F32), torch.float16: (F16_LOGIT_ATOL, F16_GRAD_REL),
            torch.bfloat16: (BF16_LOGIT_ATOL, BF16_GRAD_REL)}


@functools.lru_cache(maxsize=2)
def _full_workload(model, K, B):
    from rpo_amd.config import vit_b16, vit_l14
    cfg = (vit_b16 if model == "ViT-B/16" else vit_l14)(K=K)
    toks = synth.oxford_pets_base_tokens()
    sd = synth.clip_state_dict(cfg, seed=0, token_rows=np.unique(toks).tolist() + [49407])
    tp, ip = synth.prompts(cfg, sd, seed=7)
    return cfg, sd, toks, tp, ip, synth.images(cfg, B), synth.labels(cfg, B)


@pytest.mark.parametrize("act", [torch.float32, torch.float16, torch.bfloat16], ids=lambda v: str(v).replace("torch.", ""))
@pytest.mark.parametrize("fixture,model,K,B", FULL_GOLDEN, ids=[f[0] for f in FULL_GOLDEN])
def test_full_size_matches_reference_golden(fixture, model, K, B, act):
    """Eval logits, train loss and both prompt gradients of the full-depth model at the bench's batch against the
    reference's own outputs: f32 mode within the north star's 1e-3, f16 within 1e-2 (logits) / 0.6 % (gradients,
    relative to the largest entry), bf16 within 0.12 / 5 %."""
    from rpo_amd.custom_clip import CustomCLIP
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", fixture + ".npz")))
    cfg, sd, toks, tp, ip, image, label = _full_workload(model, K, B)
    assert np.array_equal(label, g["label"])
    m = CustomCLIP(cfg, sd, toks, "cuda:0", act, max_batch=B, prompts=(tp, ip))
    image, label = torch.from_numpy(image).cuda(), torch.from_numpy(label).cuda()
    m.prompt_learner.eval()
    logits = m(image).cpu().numpy()
    m.prompt_learner.train()
    loss = m(image, label)
    loss.backward()
    gt = m.prompt_learner.text_prompt.grad.cpu().numpy()
    gi = m.prompt_learner.img_prompt.grad.cpu().numpy()
    le, ll = np.abs(logits - g["logits"]).max(), abs(loss.item() - float(g["loss"]))
    rt, ri = _relmax(gt, g["g_text"]), _relmax(gi, g["g_img"])
    print(f"[full {fixture} {act}] logits err {le:.3e} loss err {ll:.3e} g_text rel {rt:.3e} g_img rel {ri:.3e}")
    la, gr = FULL_TOL[act]
    assert np.isfinite(logits).all() and le <= la and ll <= la
    assert rt <= gr and ri <= gr
    if act != torch.bfloat16:
        assert (logits.argmax(-1) == g["logits"].argmax(-1)).all()


@pytest.mark.parametrize("act,tol", [(torch.float32, 1e-6), (torch.float16, 1e-4), (torch.bfloat16, TOL_F32)],
                         ids=["float32", "float16", "bfloat16"])
def test_full_size_sgd_steps_match_reference(act, tol):
    """configs[1] through the trainer (graphs, fused SGD): the prompts after 1 and 2 optimiser steps at B = 32 against
    the reference's torch.optim.SGD run with the same explicit hyper-parameters.  The prompts move by lr x gradient
    ~ 1e-2 x 1e-1 per step, so even the bf16 mode's 2.5 % gradient error leaves the LEARNED PROMPTS within the north
    star's 1e-3 of the reference (measured: f32 1.5e-8, f16 5.8e-6, bf16 6.3e-5 after two steps)."""
    from rpo_amd.trainer import RPO, OptimConfig
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_full_k24_b32.npz")))
    cfg, sd, toks, tp, ip, _, _ = _full_workload("ViT-B/16", 24, 32)
    lr, mom, wd = (float(v) for v in g["sgd_hparams"])
    oc = OptimConfig(lr=lr, momentum=mom, weight_decay=wd, warmup_epoch=0, lr_scheduler="constant")
    tr = RPO(cfg, sd, toks, oc, "cuda:0", act, batch_size=32, num_batches=10 ** 9, prompts=(tp, ip))
    losses = []
    for step in range(2):
        batch = {"img": torch.from_numpy(synth.images(cfg, 32, seed=1234 + 10 * step)),
                 "label": torch.from_numpy(synth.labels(cfg, 32, seed=4321 + 10 * step))}
        losses.append(tr.forward_backward(batch)["loss"])
        t = tr.model.prompt_learner.text_prompt.detach().cpu().numpy()
        i = tr.model.prompt_learner.img_prompt.detach().cpu().numpy()
        et = np.abs(t - g[f"text_prompt_step{step + 1}"]).max()
        ei = np.abs(i - g[f"img_prompt_step{step + 1}"]).max()
        print(f"[full sgd {act}] step {step + 1}: text prompt err {et:.2e} img prompt err {ei:.2e}")
        assert et <= tol and ei <= tol
    lt = {torch.float32: TOL_F32, torch.float16: F16_LOGIT_ATOL, torch.bfloat16: BF16_LOGIT_ATOL}[act]
    assert np.abs(np.asarray(losses) - g["sgd_losses"]).max() <= lt


# bounds on the LEARNED PROMPTS after the whole run (max abs difference to the reference's, per tensor): the north star's
# 1e-3 for the fp32 mode; the 16-bit modes' are what was measured on the GPU (printed by the test) with head-room
TRAJ_TOL = {torch.float32: 1e-5, torch.float16: 1e-3, torch.bfloat16: 5e-3}   # measured: 2.3e-7 / 1.1e-4 / 1.4e-3


@pytest.mark.parametrize("act", [torch.float32, torch.float16, torch.bfloat16], ids=lambda v: str(v).replace("torch.", ""))
def test_sixty_step_trajectory_matches_reference_run(act):
    """BASELINE configs[0] end to end: 15 epochs x 4 iterations of ViT-B/16, K = 24, batch 4 through RPO.forward_backward
    -- captured graphs, fused SGD (momentum 0.9, weight decay 5e-4), one constant warm-up epoch at 1e-5, cosine decay,
    the rate updated after the last batch of every epoch (trainers/rpo.py:306-314, main_K24.yaml:15-22) -- against the
    REAL reference's run of the same schedule on the same inputs (tools/make_golden_trajectory.py): both prompt tensors
    after epochs 1, 5 and 15, the learning rate of every epoch, the 60 losses.  This is what "learned prompt embeddings
    match the reference" means over a run rather than over two steps."""
    from rpo_amd.trainer import RPO, OptimConfig
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_traj_d12_k24_b4_e15.npz")))
    lr, mom, wd, max_epoch, iters, B, warm, cons = (float(v) for v in g["hparams"])
    max_epoch, iters, B, warm = int(max_epoch), int(iters), int(B), int(warm)
    cfg, sd, toks, tp, ip, _, _ = _full_workload("ViT-B/16", 24, B)
    oc = OptimConfig(lr=lr, max_epoch=max_epoch, lr_scheduler="cosine", warmup_epoch=warm, warmup_type="constant",
                     warmup_cons_lr=cons, momentum=mom, weight_decay=wd)
    tr = RPO(cfg, sd, toks, oc, "cuda:0", act, batch_size=B, num_batches=iters, prompts=(tp, ip))
    batches = [{"img": torch.from_numpy(synth.images(cfg, B, seed=1234 + 10 * i)),
                "label": torch.from_numpy(synth.labels(cfg, B, seed=4321 + 10 * i))} for i in range(iters)]
    losses, worst = [], 0.0
    for epoch in range(max_epoch):
        assert abs(tr.lr - float(g["lrs"][epoch])) <= 1e-12 * max(1.0, lr), (epoch, tr.lr, float(g["lrs"][epoch]))
        for it in range(iters):
            losses.append(tr.forward_backward(batches[it])["loss"])
        if f"text_prompt_e{epoch + 1}" in g:
            t = tr.model.prompt_learner.text_prompt.detach().cpu().numpy()
            i = tr.model.prompt_learner.img_prompt.detach().cpu().numpy()
            et, ei = np.abs(t - g[f"text_prompt_e{epoch + 1}"]).max(), np.abs(i - g[f"img_prompt_e{epoch + 1}"]).max()
            moved = max(np.abs(g[f"text_prompt_e{epoch + 1}"] - tp).max(), np.abs(g[f"img_prompt_e{epoch + 1}"] - ip).max())
            print(f"[trajectory {act}] after epoch {epoch + 1:2d}: text prompt err {et:.2e} img prompt err {ei:.2e} "
                  f"(the prompts have moved by up to {moved:.2e})")
            worst = max(worst, et, ei)
    le = np.abs(np.asarray(losses) - g["losses"]).max()
    print(f"[trajectory {act}] 60 losses: max |diff| {le:.2e} (reference {g['losses'][0]:.4f} -> {g['losses'][-1]:.4f})")
    assert worst <= TRAJ_TOL[act], (worst, TRAJ_TOL[act])
    lt = {torch.float32: TOL_F32, torch.float16: F16_LOGIT_ATOL, torch.bfloat16: BF16_LOGIT_ATOL}[act]
    assert le <= lt, (le, lt)
    tr.model.prompt_learner.eval()
    logits = tr.model(batches[0]["img"].cuda()).cpu().numpy()
    assert np.abs(logits - g["final_logits"]).max() <= lt


def test_two_ranks_equal_one_rank_global_batch(tmp_path):
    """SURVEY.md section 8e's own correctness test on the HIP path: 2 ranks x B/2 images (both on this box's one GPU,
    gloo instead of RCCL, which refuses two ranks on one device) for two SGD steps must leave the prompts a 1-rank
    run on the global batch leaves, to summation-order noise.  Rank 1 starts from perturbed prompts, so the
    start-up broadcast is covered too."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    worker = os.path.join(root, "tests", "dp_equiv_worker.py")
    G, steps = 4, 2
    outs = {}
    for world in (1, 2):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, RPO_DIST_BACKEND="gloo", RPO_ALL_RANKS_ON_GPU0="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        out = str(tmp_path / f"w{world}.npz")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(port), worker, out, str(G), str(steps), "f32"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[world] = dict(np.load(out))
    assert int(outs[2]["world"]) == 2
    a, b = outs[1]["params"], outs[2]["params"]
    assert np.abs(a - b).max() <= 1e-6, np.abs(a - b).max()
    # each rank's loss is the mean over ITS shard: rank 0's differs from the global mean, the prompts do not
    assert np.isfinite(outs[2]["losses"]).all()


def test_reference_shaped_constructor_and_seeded_init():
    """`CustomCLIP(cfg, classnames, prompt, clipmodel)` exactly as trainers/rpo.py:255 calls it; the prompts it draws
    under torch.manual_seed are the reference's own (G7 fixture: the REAL PromptLearner.initialization_token under the
    same seed), and so are the logits they give."""
    import types
    from rpo_amd.config import OXFORD_PETS_BASE_CLASSES, PROMPT_TEMPLATE, vit_b16
    from rpo_amd.custom_clip import CustomCLIP
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_init_seed3_d1_k4.npz")))
    cfg = vit_b16(layers_v=1, layers_t=1, K=4)
    sd = synth.clip_state_dict(cfg, seed=0, logit_scale=float(np.log(100.0)))

    class FakeCLIP:                                   # stands in for clip.model.CLIP: only .state_dict() is used
        def state_dict(self):
            return {k: torch.from_numpy(v) for k, v in sd.items()}

    ns = types.SimpleNamespace
    rcfg = ns(TRAINER=ns(RPO=ns(K=4, PREC="fp32")), INPUT=ns(SIZE=(224, 224)))
    torch.manual_seed(int(g["seed"]))
    m = CustomCLIP(rcfg, list(OXFORD_PETS_BASE_CLASSES), PROMPT_TEMPLATE, FakeCLIP())
    assert m.engine.act == torch.float32 and m.cfg == cfg
    assert np.abs(m.prompt_learner.text_prompt.detach().cpu().numpy() - g["text_prompt"]).max() <= 1e-7
    assert np.abs(m.prompt_learner.img_prompt.detach().cpu().numpy() - g["img_prompt"]).max() <= 1e-7
    m.prompt_learner.eval()
    logits = m(torch.from_numpy(synth.images(cfg, 2)).cuda()).cpu().numpy()
    assert np.abs(logits - g["logits"]).max() <= TOL_F32
    # a tokenizer callable replaces the bundled table
    toks = synth.oxford_pets_base_tokens()
    table = {PROMPT_TEMPLATE.replace("_", c): toks[i:i + 1] for i, c in enumerate(OXFORD_PETS_BASE_CLASSES)}
    m2 = CustomCLIP(rcfg, list(OXFORD_PETS_BASE_CLASSES)[:5], PROMPT_TEMPLATE, FakeCLIP(), tokenize=lambda t: table[t])
    assert m2.cfg.n_cls == 5 and m2.len_prompts.tolist() == [10, 10, 14, 11, 8]
    with pytest.raises(ValueError):
        CustomCLIP(rcfg, ["cat", "dog"], PROMPT_TEMPLATE, FakeCLIP())
    with pytest.raises(IndexError):
        m2.prompt_learner.train()
        m2(torch.from_numpy(synth.images(cfg, 2)).cuda(), torch.tensor([0, 7]))


def test_reference_checkpoint_fixture_loads(tmp_path):
    """Row f2: the committed checkpoint directory was written by the reference side (tools/make_golden.py G8: the REAL
    prompt_learner.state_dict() after one torch.optim.SGD step, in the `prompt_learner/model-best.pth.tar` +
    `model.pth.tar-2` layout trainers/rpo.py:325-357 reads, with the token_prefix / token_suffix keys it drops).
    Loading it must give the reference's prompts bit-for-bit, its eval logits and its momentum buffers; what
    `save_model(is_best=True)` writes back must satisfy the same reader contract."""
    from rpo_amd.config import vit_b16
    from rpo_amd.trainer import RPO
    gold = os.path.join(os.path.dirname(__file__), "golden")
    g = dict(np.load(os.path.join(gold, "ref_ckpt_d1_k4.npz")))
    cfg = vit_b16(layers_v=1, layers_t=1, K=4)
    toks = synth.oxford_pets_base_tokens()
    sd = synth.clip_state_dict(cfg, seed=0, logit_scale=float(np.log(100.0)))
    image = torch.from_numpy(synth.images(cfg, 2)).cuda()
    for epoch in (None, 2):                           # default = model-best (:333)
        tr = RPO(cfg, sd, toks, None, "cuda:0", torch.float32, batch_size=2, prompts=synth.prompts(cfg, sd, seed=99))
        tr.load_model(os.path.join(gold, "ckpt_d1_k4"), epoch=epoch)
        assert np.array_equal(tr.model.prompt_learner.text_prompt.detach().cpu().numpy(), g["text_prompt"])
        assert np.array_equal(tr.model.prompt_learner.img_prompt.detach().cpu().numpy(), g["img_prompt"])
        assert tr.epoch == 2
        nt = cfg.K * cfg.d_t
        assert np.array_equal(tr.engine.mom[:nt].cpu().numpy().reshape(g["momentum_text"].shape), g["momentum_text"])
        assert np.array_equal(tr.engine.mom[nt:].cpu().numpy().reshape(g["momentum_img"].shape), g["momentum_img"])
        assert np.abs(tr.model_inference(image).cpu().numpy() - g["logits"]).max() <= TOL_F32
    with pytest.raises(FileNotFoundError):
        tr.load_model(os.path.join(gold, "ckpt_d1_k4"), epoch=7)
    # writer: model-best + model.pth.tar-N, readable without arbitrary unpickling, reader-contract keys
    fn = tr.save_model(str(tmp_path), epoch=5, is_best=True, val_result=71.0)
    best = os.path.join(str(tmp_path), "prompt_learner", "model-best.pth.tar")
    assert os.path.exists(best) and fn.endswith("model.pth.tar-5")
    ck = torch.load(best, map_location="cpu", weights_only=True)
    assert {"state_dict", "epoch", "optimizer", "scheduler", "val_result"} <= set(ck) and ck["epoch"] == 5
    assert set(ck["state_dict"]) == {"text_prompt", "img_prompt"}
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros_like(ck["state_dict"][k])) for k in ("text_prompt", "img_prompt")],
                          lr=0.1, momentum=0.9)
    opt.load_state_dict(ck["optimizer"])              # torch.optim.SGD's own layout: a reference run can resume from it
    assert torch.equal(opt.state_dict()["state"][0]["momentum_buffer"], torch.from_numpy(g["momentum_text"]))
    assert tr.after_epoch_eval(str(tmp_path), 80.0) and not tr.after_epoch_eval(str(tmp_path), 10.0)
    assert torch.load(best, map_location="cpu", weights_only=True)["val_result"] == 80.0


def test_anomaly_scan_and_bad_labels():
    """trainers/rpo.py:287-288 (nan detector) and F.cross_entropy's target check (:230)."""
    from rpo_amd.trainer import RPO
    tag = "d1_k4_b2"
    cfg, sd, toks, tp, ip, image, label = workload(tag)
    tr = RPO(cfg, sd, toks, None, "cuda:0", torch.float32, batch_size=2, prompts=(tp, ip))
    tr.detect_anomaly = True
    tr.forward_backward({"img": torch.from_numpy(image), "label": torch.from_numpy(label)})      # finite: passes
    with pytest.raises(IndexError):
        tr.forward_backward({"img": torch.from_numpy(image), "label": torch.tensor([0, cfg.n_cls])})
    bad = torch.from_numpy(image).clone()
    bad[0, 0, 0, 0] = float("nan")
    with pytest.raises(FloatingPointError):
        tr.forward_backward({"img": bad, "label": torch.from_numpy(label)})
    # a device-resident out-of-range label cannot raise from the kernel: the loss is NaN, not a plausible number
    tr.detect_anomaly = False
    tr.engine.forward_backward(torch.from_numpy(image).cuda(), torch.tensor([0, 400], device="cuda"))
    assert torch.isnan(tr.engine.loss).all()
    # ... and so are the prompt gradients (the softmax without its one-hot term would be a finite, wrong update): the
    # guarded optimiser step then skips it and leaves the prompts untouched
    assert torch.isnan(tr.engine.grads).any()
    from rpo_amd import ops
    tr.engine.params.fill_(0.25)                 # (the NaN image above left NaN prompts behind)
    tr.engine.mom.zero_()
    before = tr.engine.params.clone()
    found = torch.zeros(2, dtype=torch.int32, device="cuda")
    ops.sgd_step_guarded(tr.engine.params, tr.engine.grads, tr.engine.mom, 0.01, 0.9, 5e-4, 1.0, first_step=False,
                         found_inf=found)
    assert torch.equal(tr.engine.params, before) and int(found[0]) == 1


@pytest.mark.gpu
def test_end_to_end_few_shot_run_learns():
    """The whole path the reference's `train.py` drives, on a synthetic few-shot problem: decoded uint8 images ->
    device transforms -> RPO.forward_backward for a few epochs with the yaml's schedule (constant warm-up epoch, then
    cosine) -> model_inference on held-out images.  The loss must fall and held-out accuracy must beat chance clearly
    (the class signal is a colour cast; the towers are frozen RANDOM networks, so this is a plumbing check of the
    train / eval loop, not a statement about accuracy)."""
    from rpo_amd.config import vit_b16
    from rpo_amd.trainer import RPO, OptimConfig
    cfg = vit_b16(layers_v=2, layers_t=2, K=4)
    toks = synth.oxford_pets_base_tokens()
    sd = synth.clip_state_dict(cfg, seed=0, logit_scale=float(np.log(100.0)))
    rng = np.random.default_rng(0)
    n_cls, shots, B = 4, 8, 16
    tint = rng.integers(40, 216, (n_cls, 3))

    def sample(c):
        h, w = int(rng.integers(230, 400)), int(rng.integers(230, 400))
        noise = rng.integers(-40, 41, (h, w, 3))
        return np.clip(tint[c][None, None, :] + noise, 0, 255).astype(np.uint8)

    train = [(sample(c), c) for c in range(n_cls) for _ in range(shots)]
    test = [(sample(c), c) for c in range(n_cls) for _ in range(8)]
    epochs, nb = 6, len(train) // B
    tr = RPO(cfg, sd, toks, optim=OptimConfig(lr=0.02, max_epoch=epochs), device="cuda:0",
             act_dtype=torch.bfloat16, batch_size=B, num_batches=nb, prompts=synth.prompts(cfg, sd, seed=7))
    torch.manual_seed(0)
    first, last = [], []
    for ep in range(epochs):
        order = rng.permutation(len(train))
        for b in range(nb):
            idx = order[b * B:(b + 1) * B]
            out = tr.forward_backward({"img": [train[i][0] for i in idx], "label": np.array([train[i][1] for i in idx])})
            (first if ep == 1 else last if ep == epochs - 1 else []).append(out["loss"])
    assert tr.epoch == epochs
    assert np.mean(last) < 0.7 * np.mean(first), (first, last)
    logits = torch.cat([tr.model_inference([t[0] for t in test[i:i + B]]) for i in range(0, len(test), B)])
    pred = logits[:, :n_cls].argmax(1).cpu().numpy()
    acc = float((pred == np.array([t[1] for t in test])).mean())
    print('held-out accuracy', acc, 'loss', np.mean(first), '->', np.mean(last))
    assert acc >= 0.45, acc            # chance: 0.25 among the 4 classes used, 0.05 over the 19 class prompts


def test_amp_mode_skips_a_step_with_nonfinite_gradients():
    """PREC "amp" (trainers/rpo.py:298-304): GradScaler.step skips the optimiser step when the gradients hold Inf / NaN.
    Gradients are fp32 and unscaled here, so that skip is all that is left of the scaler: RPO(amp=True)."""
    from rpo_amd import synth
    from rpo_amd.config import vit_b16
    from rpo_amd.trainer import RPO
    cfg = vit_b16(layers_v=2, layers_t=2, K=4)
    toks = synth.oxford_pets_base_tokens()
    sd = synth.clip_state_dict(cfg, seed=0, token_rows=np.unique(toks).tolist() + [49407])
    pr = synth.prompts(cfg, sd, seed=7)
    img = torch.from_numpy(synth.images(cfg, 4, seed=1)).cuda()
    lab = torch.from_numpy(synth.labels(cfg, 4, seed=2)).cuda()
    plain = RPO(cfg, sd, toks, device="cuda:0", act_dtype=torch.float16, batch_size=4, prompts=pr)
    amp = RPO(cfg, sd, toks, device="cuda:0", act_dtype=torch.float16, batch_size=4, prompts=pr, amp=True)
    for _ in range(3):
        plain.step_async(img, lab); amp.step_async(img, lab)
    torch.cuda.synchronize()
    assert torch.equal(plain.engine.params, amp.engine.params), "finite gradients: amp must take the ordinary step"
    before = amp.engine.params.clone()
    bad = img.clone(); bad[1, 2, 100, 50] = float("nan")
    amp.step_async(bad, lab); torch.cuda.synchronize()
    assert torch.equal(amp.engine.params, before) and amp._found_inf.tolist() == [1, 1]
    amp.step_async(img, lab); plain.step_async(img, lab); torch.cuda.synchronize()
    assert amp._found_inf.tolist() == [0, 1] and torch.equal(plain.engine.params, amp.engine.params)


@pytest.mark.parametrize("tag,depth,B", [("d2_b3", 2, 3), ("d12_b2", 12, 2)])
@pytest.mark.parametrize("mode", ["f32", "f16", "bf16"])
def test_plain_clip_inference_matches_reference_clip_forward(tag, depth, B, mode):
    """The unmasked towers (trainers/zsclip.py:58-63, trainers/coop.py:196-208 -> clip/model.py:344-372): logits of
    rpo_amd.zeroshot.ZeroshotCLIP against the reference's own CLIP.forward (tests/golden/ref_plainclip_*.npz)."""
    from rpo_amd.config import vit_b16
    from rpo_amd.zeroshot import ZeroshotCLIP
    gold = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", f"ref_plainclip_{tag}.npz")))
    cfg = vit_b16(layers_v=depth, layers_t=depth, K=1)
    sd = synth.clip_state_dict(cfg, seed=0, logit_scale=float(np.log(100.0)))
    dt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[mode]
    m = ZeroshotCLIP(sd, device="cuda:0", act_dtype=dt, max_batch=4)
    image = torch.from_numpy(synth.images(cfg, B))
    logits = m.model_inference(image).cpu().numpy()
    tol = {"f32": 1e-3, "f16": 1e-2, "bf16": 0.12}[mode]
    err = np.abs(logits - gold["logits"]).max()
    assert err <= tol, f"{mode}: logits differ from the reference by {err:.3e} (bound {tol})"
    assert (logits.argmax(1) == gold["logits"].argmax(1)).all()
    again = m.model_inference(image).cpu().numpy()                  # cached text features, same bits
    assert np.array_equal(again, logits)
    img_f = m.engine.img_cls_f[:B].cpu().numpy()
    if mode == "f32":
        assert np.abs(img_f - gold["image_features"]).max() <= 1e-4 * max(1.0, np.abs(gold["image_features"]).max())
        assert np.abs(m.engine.plain_text_f.cpu().numpy() - gold["text_features"]).max() <= 1e-4 * max(1.0, np.abs(gold["text_features"]).max())


@pytest.mark.parametrize("tag,depth,B", [("d2_b3_ctx4", 2, 3), ("d2_b2_ctx16", 2, 2)])
@pytest.mark.parametrize("mode", ["f32", "f16", "bf16"])
def test_coop_context_inference_matches_reference_trainer(tag, depth, B, mode):
    """CoOp's forward (trainers/coop.py:117-134,196-208): learned context vectors in front of the class name on the
    unmasked towers -- logits against the reference's own coop.CustomCLIP (tests/golden/ref_coop_*.npz)."""
    from rpo_amd.config import vit_b16
    from rpo_amd.zeroshot import ZeroshotCLIP
    gold = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", f"ref_coop_{tag}.npz")))
    cfg = vit_b16(layers_v=depth, layers_t=depth, K=1)
    sd = synth.clip_state_dict(cfg, seed=0, logit_scale=float(np.log(100.0)))
    dt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[mode]
    m = ZeroshotCLIP(sd, gold["tokenized_prompts"], device="cuda:0", act_dtype=dt, max_batch=4)
    image = torch.from_numpy(synth.images(cfg, B))
    before = m.model_inference(image).cpu().numpy()               # "X X .." placeholders: not the answer
    m.set_context(gold["ctx"])
    logits = m.model_inference(image).cpu().numpy()
    tol = {"f32": 1e-3, "f16": 1e-2, "bf16": 0.12}[mode]
    err = np.abs(logits - gold["logits"]).max()
    assert err <= tol, f"{mode}: logits differ from the reference by {err:.3e} (bound {tol})"
    assert np.abs(before - gold["logits"]).max() > min(10 * tol, 0.5)


@pytest.mark.parametrize("tag,depth,B,n_ctx", [("d2_b3_ctx4", 2, 3, 4), ("d2_b2_ctx16", 2, 2, 16)])
@pytest.mark.parametrize("mode", ["f32", "f16", "bf16"])
def test_coop_context_training_matches_reference_trainer(tag, depth, B, n_ctx, mode):
    """Row f4, CoOp TRAINING (trainers/coop.py:258-281): logits, cross-entropy and d loss / d ctx of the HIP path -- the
    dense text-tower backward, causal attention with dK / dV -- against the reference's own coop.CustomCLIP +
    F.cross_entropy + backward (tests/golden/ref_coop_*.npz, written by tools/make_golden_plainclip.py)."""
    from rpo_amd.config import vit_b16
    from rpo_amd.coop import CoOpCustomCLIP
    gold = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", f"ref_coop_{tag}.npz")))
    cfg = vit_b16(layers_v=depth, layers_t=depth, K=1)
    sd = synth.clip_state_dict(cfg, seed=0, logit_scale=float(np.log(100.0)))
    dt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[mode]
    m = CoOpCustomCLIP(sd, gold["tokenized_prompts"], n_ctx, "cuda:0", dt, max_batch=4, ctx=gold["ctx"])
    image = torch.from_numpy(synth.images(cfg, B)).cuda()
    label = torch.from_numpy(gold["label"]).cuda()
    eng = m.engine
    logits = eng.coop_forward_backward(image, label).cpu().numpy()
    loss, g = eng.loss.item(), eng.coop_grad.cpu().numpy()
    le, ll, gr = np.abs(logits - gold["logits"]).max(), abs(loss - float(gold["loss"])), _relmax(g, gold["ctx_grad"])
    print(f"[coop {tag} {mode}] logits err {le:.3e} loss err {ll:.3e} ctx_grad rel {gr:.3e}")
    lt, gt = {"f32": (TOL_F32, TOL_F32), "f16": (F16_LOGIT_ATOL, F16_GRAD_REL), "bf16": (BF16_LOGIT_ATOL, BF16_GRAD_REL)}[mode]
    assert le <= lt and ll <= lt and gr <= gt
    assert np.array_equal(m(image).cpu().numpy(), logits)                       # eval branch: same logits, no backward
    # determinism: the same batch again -> the same bits
    eng.coop_forward_backward(image, label)
    assert np.array_equal(eng.coop_grad.cpu().numpy(), g)


@pytest.mark.parametrize("tag,depth,B,n_ctx", [("d2_b3_ctx4_csc", 2, 3, 4), ("d2_b2_ctx4_middle", 2, 2, 4),
                                               ("d2_b2_ctx5_middle_csc", 2, 2, 5), ("d2_b2_ctx4_front", 2, 2, 4)])
@pytest.mark.parametrize("mode", ["f32", "f16", "bf16"])
def test_coop_options_match_reference_trainer(tag, depth, B, n_ctx, mode):
    """CoOp's non-default options on the HIP path -- class-specific contexts (TRAINER.COOP.CSC, trainers/coop.py:84-86:
    ctx [n_cls, n_ctx, d], no sum over the classes in the backward) and CLASS_TOKEN_POSITION "middle" / "front"
    (:136-183: the class-name tokens re-ordered around the context before the positional embedding is added) -- against
    the reference's own coop.CustomCLIP + F.cross_entropy + backward with those options set."""
    from rpo_amd.config import vit_b16
    from rpo_amd.coop import CoOpCustomCLIP
    gold = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", f"ref_coop_{tag}.npz")))
    cfg = vit_b16(layers_v=depth, layers_t=depth, K=1)
    sd = synth.clip_state_dict(cfg, seed=0, logit_scale=float(np.log(100.0)))
    dt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[mode]
    csc, pos = bool(gold["csc"]), str(gold["class_token_position"])
    m = CoOpCustomCLIP(sd, gold["tokenized_prompts"], n_ctx, "cuda:0", dt, max_batch=4, ctx=gold["ctx"], csc=csc,
                       class_token_position=pos)
    image = torch.from_numpy(synth.images(cfg, B)).cuda()
    label = torch.from_numpy(gold["label"]).cuda()
    eng = m.engine
    logits = eng.coop_forward_backward(image, label).cpu().numpy()
    loss, g = eng.loss.item(), eng.coop_grad.cpu().numpy()
    assert g.shape == gold["ctx_grad"].shape
    le, ll, gr = np.abs(logits - gold["logits"]).max(), abs(loss - float(gold["loss"])), _relmax(g, gold["ctx_grad"])
    print(f"[coop {tag} {mode}] logits err {le:.3e} loss err {ll:.3e} ctx_grad rel {gr:.3e}")
    lt, gt = {"f32": (TOL_F32, TOL_F32), "f16": (F16_LOGIT_ATOL, F16_GRAD_REL), "bf16": (BF16_LOGIT_ATOL, BF16_GRAD_REL)}[mode]
    assert le <= lt and ll <= lt and gr <= gt
    assert np.array_equal(m(image).cpu().numpy(), logits)
    eng.coop_forward_backward(image, label)
    assert np.array_equal(eng.coop_grad.cpu().numpy(), g)


def test_coop_checkpoints_amp_and_cocoop_test_batches(tmp_path):
    """The rest of the sibling trainers' surface (advisor, round 3): Dassl-layout checkpoints written and read back
    (trainers/coop.py:283-325 drops token_prefix / token_suffix), a class-specific-context trainer taking SGD steps, the
    `amp` branch skipping a step with a non-finite gradient (:263-270), CTX_INIT word embeddings (:72-80), and CoCoOp
    inference on a test batch larger than its training batch (the reference tests at batch 100 with a batch-1 model)."""
    from rpo_amd.config import vit_b16
    from rpo_amd.coop import CoCoOp, CoOp
    from rpo_amd.trainer import OptimConfig
    gold = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_coop_d2_b3_ctx4_csc.npz")))
    cfg = vit_b16(layers_v=2, layers_t=2, K=1)
    sd = synth.clip_state_dict(cfg, seed=0, logit_scale=float(np.log(100.0)))
    toks = gold["tokenized_prompts"]
    oc = OptimConfig(lr=0.002, momentum=0.9, weight_decay=5e-4, warmup_epoch=0, lr_scheduler="constant")
    tr = CoOp(sd, toks, 4, oc, "cuda:0", torch.float32, batch_size=3, num_batches=2, ctx=gold["ctx"], csc=True, amp=True)
    batch = {"img": torch.from_numpy(synth.images(cfg, 3)), "label": torch.from_numpy(gold["label"])}
    out = tr.forward_backward(batch)
    assert abs(out["loss"] - float(gold["loss"])) <= TOL_F32 and tr.skipped_steps == 0
    ctx1 = tr.engine.coop_ctx.cpu().numpy().copy()
    assert ctx1.shape == gold["ctx"].shape and np.abs(ctx1 - gold["ctx"]).max() > 0
    # first SGD step: p -= lr * (g + wd * p)
    np.testing.assert_allclose(ctx1, gold["ctx"] - oc.lr * (gold["ctx_grad"] + oc.weight_decay * gold["ctx"]), atol=2e-7)
    fn = tr.save_model(str(tmp_path), is_best=True)
    ck = torch.load(fn, weights_only=True)
    assert set(ck["state_dict"]) == {"ctx", "token_prefix", "token_suffix"} and ck["state_dict"]["ctx"].shape == ctx1.shape
    assert ck["state_dict"]["token_suffix"].shape == (cfg.n_cls, 77 - 1 - 4, cfg.d_t)
    tr.forward_backward(batch)                                   # a second step, then back to the checkpoint
    tr2 = CoOp(sd, toks, 4, oc, "cuda:0", torch.float32, batch_size=3, num_batches=2, csc=True)
    tr2.load_model(str(tmp_path))                                # weights only, as trainers/coop.py:283-325
    assert np.array_equal(tr2.engine.coop_ctx.cpu().numpy(), ctx1)
    assert float(tr2.engine.coop_moms.abs().max()) == 0.0 and tr2.epoch == 0 and tr2._steps == 0
    tr2.resume_model(str(tmp_path))                              # ... resuming also restores the optimiser state
    assert torch.equal(tr2.engine.coop_moms.cpu(), ck["optimizer"]["state"][0]["momentum_buffer"].reshape(-1))
    tr2.forward_backward(batch)
    assert np.array_equal(tr2.engine.coop_ctx.cpu().numpy(), tr.engine.coop_ctx.cpu().numpy()), "resume != continue"
    # amp: a non-finite gradient skips the step and is counted
    before = tr.engine.coop_params.clone()
    bad = {"img": batch["img"].clone(), "label": batch["label"]}
    bad["img"][0, 0, 0, 0] = float("inf")
    tr.forward_backward(bad)
    assert tr.skipped_steps == 1 and torch.equal(tr.engine.coop_params, before)
    # CTX_INIT: the embeddings of the initialisation words, given as token ids
    ids = np.array([320, 1125, 539, 320])                        # "a photo of a"
    tr3 = CoOp(sd, toks, 4, oc, "cuda:0", torch.float32, batch_size=3, ctx_init=ids)
    assert np.array_equal(tr3.engine.coop_ctx.cpu().numpy(), sd["token_embedding.weight"][ids])
    # CoCoOp: test batch 5 on a model set up for training batch 2 -> chunks of 2, image by image the same logits
    gc = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_cocoop_d2_b3_ctx4.npz")))
    meta = {k: gc[k] for k in ("w1", "b1", "w2", "b2")}
    co = CoCoOp(sd, gc["tokenized_prompts"], 4, oc, "cuda:0", torch.float32, batch_size=2, ctx=gc["ctx"], meta=meta)
    imgs = torch.from_numpy(synth.images(cfg, 5, seed=91)).cuda()
    co.model.prompt_learner.eval()
    big = co.model_inference(imgs).cpu().numpy()
    assert big.shape == (5, cfg.n_cls)
    for b in range(5):
        one = co.model_inference(imgs[b:b + 1]).cpu().numpy()
        np.testing.assert_allclose(big[b:b + 1], one, atol=2e-5)
    fn = co.save_model(str(tmp_path / "cocoop"), epoch=3)
    ck = torch.load(fn, weights_only=True)
    assert {"ctx", "meta_net.linear1.weight", "meta_net.linear2.bias", "token_prefix"} <= set(ck["state_dict"])
    co2 = CoCoOp(sd, gc["tokenized_prompts"], 4, oc, "cuda:0", torch.float32, batch_size=2)
    co2.load_model(str(tmp_path / "cocoop"), epoch=3)
    assert torch.equal(co2.engine.coop_params, co.engine.coop_params) and co2.epoch == 0
    assert co2.resume_model(str(tmp_path / "cocoop"), epoch=3) == 3 and co2.epoch == 3
    with pytest.raises((ValueError, RuntimeError)):              # a checkpoint of another context shape (CSC) is refused,
        CoOp(sd, toks, 4, oc, "cuda:0", torch.float32, batch_size=3).resume_model(str(tmp_path))   # not half-applied


def test_coop_trainer_sgd_steps_match_oracle():
    """Three optimiser steps of the CoOp trainer (f32 mode) against the CPU oracle's autograd + torch-equivalent SGD on
    the same batches: ctx within 1e-3 (measured ~1e-7), losses within 1e-3; the oracle itself is pinned to the
    reference's coop trainer by tests/test_oracle_golden.py."""
    from oracle.rpo_oracle import OracleSGD, coop_loss_and_grad
    from rpo_amd.config import vit_b16
    from rpo_amd.coop import CoOp
    from rpo_amd.trainer import OptimConfig
    gold = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_coop_d2_b3_ctx4.npz")))
    cfg = vit_b16(layers_v=2, layers_t=2, K=1)
    sd = synth.clip_state_dict(cfg, seed=0, logit_scale=float(np.log(100.0)))
    toks, ctx0 = gold["tokenized_prompts"], gold["ctx"]
    oc = OptimConfig(lr=0.002, momentum=0.9, weight_decay=5e-4, warmup_epoch=0, lr_scheduler="constant")
    tr = CoOp(sd, toks, 4, oc, "cuda:0", torch.float32, batch_size=3, num_batches=10 ** 9, ctx=ctx0)
    ctx = torch.from_numpy(ctx0.copy())
    opt = OracleSGD(oc.lr, oc.momentum, oc.weight_decay)
    for step in range(3):
        im, lb = synth.images(cfg, 3, seed=70 + step), synth.labels(cfg, 3, seed=80 + step)
        _, loss, g = coop_loss_and_grad(sd, im, toks, ctx.numpy(), lb, cfg.patch)
        opt.step([ctx], [g])
        out = tr.forward_backward({"img": torch.from_numpy(im), "label": torch.from_numpy(lb)})
        assert abs(out["loss"] - float(loss)) <= TOL_F32 and 0.0 <= out["acc"] <= 100.0
        err = (tr.model.prompt_learner.ctx.cpu() - ctx).abs().max().item()
        print(f"[coop sgd] step {step + 1}: loss {out['loss']:.5f} ctx err {err:.2e}")
        assert err <= 1e-5
    sdict = tr.model.prompt_learner.state_dict()                              # the reference's checkpoint keys
    assert set(sdict) == {"ctx", "token_prefix", "token_suffix"} and sdict["token_suffix"].shape == (19, 77 - 1 - 4, 512)


@pytest.mark.parametrize("tag,depth,B,n_ctx", [("d2_b1_ctx4", 2, 1, 4), ("d2_b3_ctx4", 2, 3, 4)])
@pytest.mark.parametrize("mode", ["f32", "f16", "bf16"])
def test_cocoop_training_matches_reference_trainer(tag, depth, B, n_ctx, mode):
    """Row f4, CoCoOp (trainers/cocoop.py:137-153,166-192,255-275): meta-net conditioned context, per-image text
    features.  Logits, cross-entropy and the gradient of every trained tensor (ctx, meta_net.linear1 / linear2 weight and
    bias) of the HIP path against the reference's own cocoop.CustomCLIP + backward (tests/golden/ref_cocoop_*.npz)."""
    from rpo_amd.config import vit_b16
    from rpo_amd.coop import CoCoOpCustomCLIP
    gold = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", f"ref_cocoop_{tag}.npz")))
    cfg = vit_b16(layers_v=depth, layers_t=depth, K=1)
    sd = synth.clip_state_dict(cfg, seed=0, logit_scale=float(np.log(100.0)))
    dt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[mode]
    meta = {k: gold[k] for k in ("w1", "b1", "w2", "b2")}
    m = CoCoOpCustomCLIP(sd, gold["tokenized_prompts"], n_ctx, "cuda:0", dt, max_batch=B, ctx=gold["ctx"], meta=meta)
    image = torch.from_numpy(synth.images(cfg, B)).cuda()
    label = torch.from_numpy(gold["label"]).cuda()
    eng = m.engine
    m.prompt_learner.eval()
    logits = m(image).cpu().numpy()
    m.prompt_learner.train()
    loss = m(image, label).item()
    le, ll = np.abs(logits - gold["logits"]).max(), abs(loss - float(gold["loss"]))
    got = dict(ctx=eng.coop_grad, w1=eng.meta_grad[0], b1=eng.meta_grad[1], w2=eng.meta_grad[2], b2=eng.meta_grad[3])
    rel = {k: _relmax(v.cpu().numpy(), gold["g_" + k]) for k, v in got.items()}
    print(f"[cocoop {tag} {mode}] logits err {le:.3e} loss err {ll:.3e} grads rel " +
          " ".join(f"{k} {v:.2e}" for k, v in rel.items()))
    lt, gt = {"f32": (TOL_F32, TOL_F32), "f16": (F16_LOGIT_ATOL, F16_GRAD_REL), "bf16": (BF16_LOGIT_ATOL, BF16_GRAD_REL)}[mode]
    assert le <= lt and ll <= lt
    assert all(v <= gt for v in rel.values()), rel


def test_cocoop_trainer_sgd_steps_match_oracle():
    """Two optimiser steps of the CoCoOp trainer (f32 mode, batch 2) against the CPU oracle's autograd + SGD on the same
    batches: every trained tensor within 1e-5."""
    from oracle.rpo_oracle import OracleSGD, cocoop_loss_and_grads
    from rpo_amd.config import vit_b16
    from rpo_amd.coop import CoCoOp
    from rpo_amd.trainer import OptimConfig
    gold = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_cocoop_d2_b3_ctx4.npz")))
    cfg = vit_b16(layers_v=2, layers_t=2, K=1)
    sd = synth.clip_state_dict(cfg, seed=0, logit_scale=float(np.log(100.0)))
    toks = gold["tokenized_prompts"]
    keys = ("ctx", "w1", "b1", "w2", "b2")
    P = {k: torch.from_numpy(gold[k].copy()) for k in keys}
    oc = OptimConfig(lr=0.002, momentum=0.9, weight_decay=5e-4, warmup_epoch=0, lr_scheduler="constant")
    tr = CoCoOp(sd, toks, 4, oc, "cuda:0", torch.float32, batch_size=2, num_batches=10 ** 9, ctx=gold["ctx"],
                meta={k: gold[k] for k in keys[1:]})
    opt = OracleSGD(oc.lr, oc.momentum, oc.weight_decay)
    for step in range(2):
        im, lb = synth.images(cfg, 2, seed=70 + step), synth.labels(cfg, 2, seed=80 + step)
        _, loss, g = cocoop_loss_and_grads(sd, im, toks, P["ctx"].numpy(), {k: P[k].numpy() for k in keys[1:]}, lb, cfg.patch)
        opt.step([P[k] for k in keys], [g[k] for k in keys])
        out = tr.forward_backward({"img": torch.from_numpy(im), "label": torch.from_numpy(lb)})
        assert abs(out["loss"] - float(loss)) <= TOL_F32
        have = dict(zip(keys, [tr.engine.coop_ctx] + list(tr.engine.meta)))
        err = max((have[k].cpu() - P[k]).abs().max().item() for k in keys)
        print(f"[cocoop sgd] step {step + 1}: loss {out['loss']:.5f} max parameter err {err:.2e}")
        assert err <= 1e-5
    assert set(tr.model.prompt_learner.state_dict()) == {"ctx", "meta_net.linear1.weight", "meta_net.linear1.bias",
                                                         "meta_net.linear2.weight", "meta_net.linear2.bias",
                                                         "token_prefix", "token_suffix"}


def test_weight_prefetch_hints_do_not_change_the_step(monkeypatch):
    """The prefetch hints of the step (every GEMM names the weights its successor reads) are
    timing-only: two optimiser steps at the bench's shape with them and without them (RPO_NO_WPREFETCH=1) leave
    bit-identical losses and prompts."""
    from rpo_amd.config import vit_b16
    from rpo_amd.trainer import RPO
    cfg, sd, toks, tp, ip, image, label = _full_workload("ViT-B/16", 24, 32)
    res = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("RPO_NO_WPREFETCH", "1")
        else:
            monkeypatch.delenv("RPO_NO_WPREFETCH", raising=False)
        tr = RPO(cfg, sd, toks, None, "cuda:0", torch.bfloat16, batch_size=32, num_batches=10 ** 9, prompts=(tp, ip))
        assert tr.engine._pf_chains == (not off)
        losses = [tr.forward_backward({"img": torch.from_numpy(image), "label": torch.from_numpy(label)})["loss"] for _ in range(2)]
        res.append((losses, tr.engine.params.clone()))
        del tr
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])


@experimental
@pytest.mark.parametrize("B", [4, 32])
def test_joint_backward_equals_the_two_chains(monkeypatch, B):
    """Engine._joint_backward issues every stage of the image tower's and the text tower's backward chain as ONE launch
    (rpo_gemm_nt_pair, rpo_layernorm_bwd_pair, rpo_attn_bwd_proj_pair).  Every problem of a paired launch is bit-identical
    to its own launch, so two optimiser steps with the chains paired (RPO_JOINT_BWD=1) and as two chains on two streams
    (the default) must leave bit-identical losses, gradients and prompts -- at the reference's batch 4 and at the bench's
    32, where the text problem's workgroups walk several tiles."""
    from rpo_amd.trainer import RPO
    cfg, sd, toks, tp, ip, image, label = _full_workload("ViT-B/16", 24, 32)
    image, label = image[:B], label[:B]
    res = []
    # (the paired form runs the text tower's attention backward on the MFMA kernel with d out-proj folded in: the two-chain
    #  run is given the same kernel, RPO_TEXT_BWD_FOLD=1, so that the comparison is launch structure only)
    monkeypatch.setenv("RPO_TEXT_BWD_FOLD", "1")
    # (... and both runs keep the chains' GEMMs on rpo_gemm_nt, which is what the paired launch pairs: the default chains
    #  run them on rpo_gemm_ws since round 5, whose four-way k split gives other last bits)
    monkeypatch.setenv("RPO_NO_WS", "1")
    for joint in ("1", "0"):
        monkeypatch.setenv("RPO_JOINT_BWD", joint)
        tr = RPO(cfg, sd, toks, None, "cuda:0", torch.bfloat16, batch_size=B, num_batches=10 ** 9, prompts=(tp, ip))
        losses = [tr.forward_backward({"img": torch.from_numpy(image), "label": torch.from_numpy(label)})["loss"] for _ in range(2)]
        assert tr._joint_bwd == (joint == "1")
        res.append((losses, tr.engine.grads.clone(), tr.engine.params.clone()))
        del tr
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])


@experimental
@pytest.mark.parametrize("how", ["1", "safe"])
def test_mlp_fused_step_equals_the_default_schedule(monkeypatch, how):
    """RPO_MLP_FUSED=1 / =safe (opt-in, Engine._image_forward): c_fc -> c_proj of every whole-batch image block as one
    launch (rpo_mlp_fused).  The same tile arithmetic, so loss and both gradients of a B = 32 step must be the BITS of
    the default schedule's, and within the golden bounds of the reference."""
    from rpo_amd.custom_clip import CustomCLIP
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_full_k24_b32.npz")))
    cfg, sd, toks, tp, ip, image, label = _full_workload("ViT-B/16", 24, 32)
    out = {}
    for mode in ("0", how):
        monkeypatch.setenv("RPO_MLP_FUSED", mode)
        m = CustomCLIP(cfg, sd, toks, "cuda:0", torch.float16, max_batch=32, prompts=(tp, ip))
        loss = m(torch.from_numpy(image).cuda(), torch.from_numpy(label).cuda())
        loss.backward()
        torch.cuda.synchronize()
        out[mode] = (loss.item(), m.prompt_learner.img_prompt.grad.cpu().numpy().copy(),
                     m.prompt_learner.text_prompt.grad.cpu().numpy().copy(), int(m.engine.mlp_counters[:-1].sum().item()))
        assert int(m.engine.mlp_counters[-1].item()) == 0, "a hand-off poll gave up"
        del m
    assert out["0"][3] == 0 and out[how][3] == 32 * 8 * 11, "11 whole-batch blocks x 32 images x 8 arrivals"
    assert out[how][0] == out["0"][0] and np.array_equal(out[how][1], out["0"][1]) and np.array_equal(out[how][2], out["0"][2])
    assert abs(out[how][0] - float(g["loss"])) <= F16_LOGIT_ATOL
    assert _relmax(out[how][1], g["g_img"]) <= F16_GRAD_REL and _relmax(out[how][2], g["g_text"]) <= F16_GRAD_REL


@experimental
def test_split_row_launches_match_reference_golden(monkeypatch):
    """RPO_SPLIT=1 (opt-in, Engine._split_rows): at K = 48 an image's 245 rows do not fit the one-round 224-row tiles, its
    197 frozen rows do -- they keep the row-unit kernels (units of 197 + 0 rows) and the prompt rows run as their own
    launches in EVERY block.  Same bounds against the reference's own outputs as the default schedule."""
    from rpo_amd.custom_clip import CustomCLIP
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_full_k48_b32.npz")))
    cfg, sd, toks, tp, ip, image, label = _full_workload("ViT-B/16", 48, 32)
    monkeypatch.setenv("RPO_SPLIT", "force")     # ("1" no longer engages at K = 48: the 256x96 geometry takes the whole rows)
    m = CustomCLIP(cfg, sd, toks, "cuda:0", torch.float16, max_batch=32, prompts=(tp, ip))
    assert m.engine._split_rows(32)
    loss = m(torch.from_numpy(image).cuda(), torch.from_numpy(label).cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["loss"])) <= F16_LOGIT_ATOL
    assert _relmax(m.prompt_learner.img_prompt.grad.cpu().numpy(), g["g_img"]) <= F16_GRAD_REL
    assert _relmax(m.prompt_learner.text_prompt.grad.cpu().numpy(), g["g_text"]) <= F16_GRAD_REL


@experimental
@pytest.mark.parametrize("case", ["d2_k8_b3", "full_b32", "full_b16_f16"])
def test_persistent_backward_chain_matches_the_launch_chain(monkeypatch, case):
    """rpo_chain_bwd (csrc/chain.hip, opt-in RPO_CHAIN=1): the 7 x layers stages of a tower's prompt-row backward as ONE
    persistent launch, 8 XCD-local groups handing tiles over through one counter each.  Same inputs, both towers:
    gradients within the 16-bit modes' rounding of the launch-per-stage chain (different k-slicing of the fp32 sums,
    16-bit intermediates rounded at the same points); no bounded spin gave up; the placement-independent protocol
    (RPO_CHAIN_SAFE=1: agent-scope release per hand-off) gives the SAME BITS as the XCD-local one the kernel picks after
    checking HW_REG_XCC_ID; uneven unit counts (3 images, 19 classes over 8 groups) leave no row behind."""
    from rpo_amd.custom_clip import CustomCLIP
    if case == "d2_k8_b3":
        cfg, sd, toks, tp, ip, image, label = workload(case)
        act, tol = torch.bfloat16, 2e-2
    else:
        cfg, sd, toks, tp, ip, image, label = _full_workload("ViT-B/16", 24, 32)
        act, tol = (torch.float16, 2e-3) if case.endswith("f16") else (torch.bfloat16, 2e-2)
        if "b16" in case:
            image, label = image[:16], label[:16]
    B = image.shape[0]
    m = CustomCLIP(cfg, sd, toks, "cuda:0", act, max_batch=B, prompts=(tp, ip))
    eng = m.engine
    monkeypatch.delenv("RPO_CHAIN", raising=False)
    eng.forward_backward(torch.from_numpy(image).cuda(), torch.from_numpy(label).cuda())
    torch.cuda.synchronize()
    ref_i, ref_t = eng.g_img.clone(), eng.g_text.clone()
    monkeypatch.setenv("RPO_CHAIN", "1")
    monkeypatch.setenv("RPO_CHAIN_TEXT", "1")
    assert eng.chain_ok("v", B) and eng.chain_ok("t", cfg.n_cls)
    eng.g_img.zero_(); eng.g_text.zero_()
    eng._image_backward(B); eng._text_backward()
    torch.cuda.synchronize()
    got_i, got_t = eng.g_img.clone(), eng.g_text.clone()
    for st in (eng.chain_state_v, eng.chain_state_t):
        assert int(st[0]) == 0, "a bounded spin of the persistent chain gave up"
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    assert rel(got_i, ref_i) < tol and rel(got_t, ref_t) < tol, (rel(got_i, ref_i), rel(got_t, ref_t))
    fast_groups = 8 - int(eng.chain_state_v[1])
    monkeypatch.setenv("RPO_CHAIN_SAFE", "1")
    eng._image_backward(B); eng._text_backward()
    torch.cuda.synchronize()
    assert int(eng.chain_state_v[1]) == min(8, B) and int(eng.chain_state_v[0]) == 0
    assert torch.equal(eng.g_img, got_i) and torch.equal(eng.g_text, got_t), "results depend on the hand-off protocol"
    assert fast_groups >= 0


# The residual stream as the 16-bit hi half alone (RPO_RESID16=1, opt-in; in fp16 what the reference's own `PREC: fp16`
# run does, clip/model.py:379-400,153-159): its own tolerance rows against the reference's fp32 CPU outputs -- max abs
# error on logits and loss, max relative error (to the largest entry) on the prompt gradients.  (fp16: 7.6e-3 measured
# at K = 24, 1.2e-2 at K = 4 -- the default mode's 1e-2 does not hold for it.)
RESID16_TOL = {torch.float16: (2e-2, 6e-3), torch.bfloat16: (0.12, 5e-2)}


@pytest.mark.parametrize("act", [torch.float16, torch.bfloat16], ids=["float16", "bfloat16"])
def test_16bit_residual_stream_against_reference(monkeypatch, act):
    """The bench's shape (ViT-B/16, K = 24, B = 32) with the residual stream as the 16-bit hi half alone and as hi + lo
    halves, both against the reference's own outputs (`ref_full_k24_b32.npz`) on the same inputs."""
    from rpo_amd.custom_clip import CustomCLIP
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_full_k24_b32.npz")))
    cfg, sd, toks, tp, ip, image, label = _full_workload("ViT-B/16", 24, 32)
    image, label = torch.from_numpy(image).cuda(), torch.from_numpy(label).cuda()
    errs = {}
    for resid16 in ("0", "1"):
        monkeypatch.setenv("RPO_RESID16", resid16)
        m = CustomCLIP(cfg, sd, toks, "cuda:0", act, max_batch=32, prompts=(tp, ip))
        m.prompt_learner.eval()
        logits = m(image).cpu().numpy()
        m.prompt_learner.train()
        loss = m(image, label)
        loss.backward()
        gt = m.prompt_learner.text_prompt.grad.cpu().numpy()
        gi = m.prompt_learner.img_prompt.grad.cpu().numpy()
        errs[resid16] = (np.abs(logits - g["logits"]).max(), abs(loss.item() - float(g["loss"])),
                         _relmax(gt, g["g_text"]), _relmax(gi, g["g_img"]))
        print(f"[{act}, 16-bit residual stream = {resid16}] logits {errs[resid16][0]:.3e} loss {errs[resid16][1]:.3e} "
              f"g_text {errs[resid16][2]:.3e} g_img {errs[resid16][3]:.3e}")
        assert np.isfinite(logits).all()
        del m
    for key, (la, gr) in (("1", RESID16_TOL[act]), ("0", FULL_TOL[act])):
        le, ll, rt, ri = errs[key]
        assert le <= la and ll <= la and rt <= gr and ri <= gr
