"""The minimal (prompt-rows-only, hand-written backward) oracle must agree with
the dense autograd oracle: this is the structural claim the HIP path relies on
(SURVEY.md finding 4 / appendix A.5)."""
import numpy as np
import pytest
import torch

from oracle import rows_oracle as R
from oracle.rpo_oracle import OracleRPO
from rpo_amd import synth
from rpo_amd.config import vit_b16

from helpers import load_golden, oracle_for


@pytest.mark.parametrize("tag", ["d1_k4_b2", "d2_k8_b3", "d2_k16_b2", "d2_k48_b2"])
def test_rows_step_matches_dense_and_reference(tag):
    g = load_golden(tag)
    m, image, label = oracle_for(tag)
    rows = R.RowsRPO(m)
    out = rows.step(image, label, m.text_prompt.detach(), m.img_prompt.detach())
    np.testing.assert_allclose(out["logits"].numpy(), g["logits"], atol=3e-5)
    assert abs(float(out["loss"]) - float(g["loss"])) < 2e-5
    for mine, ref in ((out["g_text"].numpy(), g["g_text"]), (out["g_img"].numpy(), g["g_img"])):
        assert np.abs(mine - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def test_ragged_and_max_length_classes():
    """len_c from 3 (SOT x EOT) up to the maximum 77-K; dense vs rows."""
    cfg = vit_b16(layers_v=1, layers_t=2, K=6, n_cls=5)
    toks = synth.synthetic_tokens(cfg, [3, 71, 20, 8, 71])
    sd = synth.clip_state_dict(cfg, seed=3, token_rows=np.unique(toks).tolist() + [49407])
    tp, ip = synth.prompts(cfg, sd, seed=11)
    m = OracleRPO(sd, toks, cfg.K, cfg.patch)
    m.set_prompts(tp, ip)
    image, label = synth.images(cfg, 2), synth.labels(cfg, 2)
    out_d, gt, gi = m.loss_and_grads(image, label)
    out_r = R.RowsRPO(m).step(image, label, tp, ip)
    np.testing.assert_allclose(out_r["logits"].detach().numpy(), out_d.logits.detach().numpy(), atol=3e-5)
    np.testing.assert_allclose(out_r["g_text"].numpy(), gt.numpy(), atol=2e-6 * max(1, gt.abs().max()))
    np.testing.assert_allclose(out_r["g_img"].numpy(), gi.numpy(), atol=2e-6 * max(1, gi.abs().max()))


def test_op_level_against_torch_autograd():
    torch.manual_seed(0)
    x = torch.randn(5, 7, 192, dtype=torch.float64, requires_grad=True)
    w, b = torch.randn(192, dtype=torch.float64), torch.randn(192, dtype=torch.float64)
    dy = torch.randn_like(x)
    y = torch.nn.functional.layer_norm(x, (192,), w, b, 1e-5)
    (gx,) = torch.autograd.grad(y, x, dy)
    np.testing.assert_allclose(R.ln_fwd(x.detach(), w, b).numpy(), y.detach().numpy(), atol=1e-12)
    np.testing.assert_allclose(R.ln_bwd(dy, x.detach(), w).numpy(), gx.numpy(), atol=1e-12)

    u = torch.randn(11, 33, dtype=torch.float64, requires_grad=True)
    (gu,) = torch.autograd.grad((u * torch.sigmoid(1.702 * u)).sum(), u)
    np.testing.assert_allclose(R.qgelu_grad(u.detach()).numpy(), gu.numpy(), atol=1e-12)

    # read-only attention on odd sizes (221 queries / 197 keys etc.): rows oracle vs
    # torch's own MHA math with the additive -inf column mask the reference builds
    for (n_q, n_k, H) in ((24, 197, 3), (48, 257, 2), (4, 5, 1)):
        D = 64 * H
        q = torch.randn(n_q, D, dtype=torch.float64, requires_grad=True)
        k, v = torch.randn(n_k, D, dtype=torch.float64), torch.randn(n_k, D, dtype=torch.float64)
        da = torch.randn(n_q, D, dtype=torch.float64)
        qh = q.reshape(n_q, H, 64).transpose(0, 1)
        kh = k.reshape(n_k, H, 64).transpose(0, 1)
        vh = v.reshape(n_k, H, 64).transpose(0, 1)
        a = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh).transpose(0, 1).reshape(n_q, D)
        (gq,) = torch.autograd.grad(a, q, da)
        np.testing.assert_allclose(R.attn_rows_fwd(q.detach(), k, v, H).numpy(), a.detach().numpy(), atol=1e-12)
        np.testing.assert_allclose(R.attn_rows_bwd(q.detach(), k, v, da, H).numpy(), gq.numpy(), atol=1e-12)


def test_head_against_autograd():
    torch.manual_seed(1)
    B, C, K, e = 5, 7, 3, 32
    i_f = torch.randn(B, K, e, dtype=torch.float64, requires_grad=True)
    t_f = torch.randn(C, K, e, dtype=torch.float64, requires_grad=True)
    lab = torch.tensor([0, 6, 3, 3, 1])
    ih = i_f / i_f.norm(dim=-1, keepdim=True)
    th = t_f / t_f.norm(dim=-1, keepdim=True)
    logits = sum(100.0 * ih[:, i] @ th[:, i].t() for i in range(K)) / K
    loss = torch.nn.functional.cross_entropy(logits, lab)
    gi, gt = torch.autograd.grad(loss, (i_f, t_f))
    lg, ls, di, dt = R.head_fwd_bwd(i_f.detach(), t_f.detach(), lab, 100.0)
    np.testing.assert_allclose(lg.numpy(), logits.detach().numpy(), atol=1e-10)
    assert abs(float(ls) - float(loss)) < 1e-12
    np.testing.assert_allclose(di.numpy(), gi.numpy(), atol=1e-12)
    np.testing.assert_allclose(dt.numpy(), gt.numpy(), atol=1e-12)
