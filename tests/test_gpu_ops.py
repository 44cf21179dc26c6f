"""Op-level parity on the MI355X: every C-ABI kernel against its CPU oracle
(oracle/rows_oracle.py, float64) on seeded inputs, all three storage modes.

Tolerances (written here, asserted below):
  f32 mode : max|err| <= 2e-5 * max|ref|   (exact-f32 MFMA; only summation order differs)
  bf16 mode: max|err| <= 1.5e-2 * max|ref| (inputs are pre-rounded to bf16 so the error
             budget is the bf16 rounding of outputs / of P inside attention)
  f16 mode : max|err| <= 2e-3 * max|ref|   (IEEE half: 3 more mantissa bits than bf16)
"""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# Tests of the measured-slower experiments (include/rpo_amd_experimental.h): they need the
# -DRPO_EXPERIMENTAL library and run only when the whole pytest process is started with RPO_EXPERIMENTAL=1
experimental = pytest.mark.skipif(os.environ.get("RPO_EXPERIMENTAL") != "1",
                                  reason="experiment: run with RPO_EXPERIMENTAL=1 (loads the -DRPO_EXPERIMENTAL library)")

from oracle import rows_oracle as R  # noqa: E402  (checker only)

DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
TOL = {"f32": 2e-5, "bf16": 1.5e-2, "f16": 2e-3}


def dev():
    return torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float32) * scale


def q(t, mode):
    """value the kernel will actually see, as float64 on the CPU"""
    return t.to(DT[mode]).to(torch.float64)


def close(got, ref, mode, what, tol=None):
    got = got.detach().to(torch.float64).cpu()
    ref = ref.to(torch.float64)
    err = (got - ref).abs().max().item()
    den = max(ref.abs().max().item(), 1e-30)
    lim = (tol if tol is not None else TOL[mode])
    assert math.isfinite(err) and err <= lim * den, f"{what} [{mode}]: max err {err:.3e} vs max ref {den:.3e} (rel {err/den:.2e} > {lim})"


def ops():
    from rpo_amd import ops as o
    return o


# ------------------------------------------------------------------------------------------
def test_library_loads_and_probe_layouts():
    """The fragment layouts every MFMA kernel assumes, checked on the device with an
    ASYMMETRIC operand pair (a transposed C/D map cannot pass)."""
    o = ops()
    assert o.version() == 8
    for which, kdim in ((0, 16), (1, 2)):
        g = torch.Generator().manual_seed(which)
        a = torch.randint(-4, 5, (32, kdim), generator=g).float()
        b = torch.randint(-4, 5, (32, kdim), generator=g).float() + torch.arange(32).float()[:, None] % 3
        d = o.probe_mfma(which, a.to(dev()), b.to(dev())).cpu()
        assert torch.equal(d, a @ b.t()), f"MFMA layout probe {which} failed"


@pytest.mark.parametrize("mode", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("M,N,K", [(130, 260, 128), (333, 768, 768), (64, 128, 3072)])
def test_gemm_epilogues(mode, M, N, K):
    from rpo_amd import _lib as L
    o = ops()
    a, w = rnd((M, K), 1, 0.5), rnd((N, K), 2, K ** -0.5)
    bias, resid, u = rnd((N,), 3), rnd((M, N), 4), rnd((M, N), 5)
    a64, w64 = q(a, mode), q(w, mode)
    acc = a64 @ w64.t()
    ad, wd = a.to(dev(), DT[mode]), w.to(dev(), DT[mode])
    bd, rd, ud = bias.to(dev()), resid.to(dev()), u.to(dev())
    act = DT[mode]

    out = torch.empty(M, N, dtype=act, device=dev())
    close(o.gemm_nt(ad, wd, out, L.EPI_NONE), acc, mode, "gemm none")
    close(o.gemm_nt(ad, wd, out, L.EPI_BIAS, bias=bd), acc + bias.double(), mode, "gemm bias")
    of = torch.empty(M, N, dtype=torch.float32, device=dev())
    close(o.gemm_nt(ad, wd, of, L.EPI_NONE), acc, mode, "gemm none f32-out", tol=TOL["f32"] if mode == "f32" else 1e-4)
    close(o.gemm_nt(ad, wd, of, L.EPI_BIAS_RESID, bias=bd, resid=rd), acc + bias.double() + resid.double(), mode,
          "gemm resid", tol=TOL["f32"] if mode == "f32" else 1e-4)
    # QuickGELU with the pre-activation of the bottom rows saved
    row0 = M // 3
    aux = torch.full((M - row0, N), float("nan"), device=dev())
    pre = acc + bias.double()
    close(o.gemm_nt(ad, wd, out, L.EPI_BIAS_QGELU, bias=bd, aux=aux, aux_row0=row0), R.qgelu(pre), mode, "gemm qgelu")
    close(aux, pre[row0:], mode, "gemm qgelu saved u", tol=TOL["f32"] if mode == "f32" else 1e-4)
    close(o.gemm_nt(ad, wd, out, L.EPI_QGELU_BWD, aux=ud), acc * R.qgelu_grad(u.double()), mode, "gemm qgelu bwd")
    if mode != "f32":
        # 16-bit modes: aux may hold d quickgelu / du in the act dtype instead (rpo_gemm_args.aux_dtype)
        aux16 = torch.full((M - row0, N), float("nan"), dtype=act, device=dev())
        close(o.gemm_nt(ad, wd, out, L.EPI_BIAS_QGELU, bias=bd, aux=aux16, aux_row0=row0), R.qgelu(pre), mode, "gemm qgelu (aux16)")
        close(aux16, R.qgelu_grad(pre[row0:]), mode, "gemm qgelu saved derivative")
        d16 = R.qgelu_grad(u.double()).float().to(dev(), act)
        close(o.gemm_nt(ad, wd, out, L.EPI_QGELU_BWD, aux=d16), acc * d16.double().cpu(), mode, "gemm qgelu bwd (aux16)")
    # skipped rectangle must stay untouched, the rest must be computed
    out.fill_(7.0)
    sr, sc = 128, 128
    o.gemm_nt(ad, wd, out, L.EPI_BIAS, bias=bd, skip_row0=sr, skip_col0=sc)
    ref = acc + bias.double()
    got = out.double().cpu()
    if M > sr and N > sc:
        assert torch.all(got[sr:, sc:] == 7.0)
    close(got[:sr], ref[:sr], mode, "gemm skip (top)")
    close(got[:, :sc], ref[:, :sc], mode, "gemm skip (left)")


@pytest.mark.parametrize("mode", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("M,N,K", [(3000, 3072, 128), (6400, 1024, 128), (2000, 3072, 64), (7072, 3072, 768)])
def test_gemm_many_tiles_all_epilogues(mode, M, N, K):
    """Multi-round launches (more tiles than resident workgroups) of every tile shape the heuristic picks
    (128x128, 64x128, 64x64), every epilogue, M tail included, against float64; a forced alternative tile
    shape must give the same bits (the k-order per output element does not depend on the tiling)."""
    from rpo_amd import _lib as L
    o = ops()
    if K == 768 and mode == "f32":
        pytest.skip("covered by the bf16 run; keeps the CPU reference matmul short")
    a, w = rnd((M, K), 1, 0.5), rnd((N, K), 2, K ** -0.5)
    bias, resid, u = rnd((N,), 3), rnd((M, N), 4), rnd((M, N), 5)
    acc = q(a, mode) @ q(w, mode).t()
    ad, wd = a.to(dev(), DT[mode]), w.to(dev(), DT[mode])
    bd, rd, ud = bias.to(dev()), resid.to(dev()), u.to(dev())
    row0 = M - 700
    for epi, kw, ref, odt in (
            (L.EPI_BIAS, dict(bias=bd), acc + bias.double(), DT[mode]),
            (L.EPI_BIAS_QGELU, dict(bias=bd, aux_row0=row0), R.qgelu(acc + bias.double()), DT[mode]),
            (L.EPI_BIAS_RESID, dict(bias=bd, resid=rd), acc + bias.double() + resid.double(), torch.float32),
            (L.EPI_QGELU_BWD, dict(aux=ud), acc * R.qgelu_grad(u.double()), DT[mode]),
            (L.EPI_NONE, dict(), acc, torch.float32)):
        out = torch.full((M, N), float("nan"), dtype=odt, device=dev())
        if epi == L.EPI_BIAS_QGELU:
            kw = dict(kw, aux=torch.full((M - row0, N), float("nan"), device=dev()))
        o.gemm_nt(ad, wd, out, epi, **kw)
        tol = TOL[mode] if odt != torch.float32 or mode == "f32" else 1e-4
        close(out, ref, mode, f"gemm epi {epi}", tol=tol)
        if epi == L.EPI_BIAS_QGELU:
            close(kw["aux"], (acc + bias.double())[row0:], mode, "gemm saved u", tol=TOL["f32"] if mode == "f32" else 1e-4)
        out2 = torch.full((M, N), float("nan"), dtype=odt, device=dev())
        kw2 = dict(kw)
        if epi == L.EPI_BIAS_QGELU:
            kw2["aux"] = torch.empty_like(kw["aux"])
        o.gemm_nt(ad, wd, out2, epi, tile_config=2, **kw2)
        assert torch.equal(out, out2), "tile shape must not change the result bits"


@pytest.mark.parametrize("M,N,K", [(7072, 2304, 768), (4096, 4096, 64), (2048, 1536, 128), (7000, 2304, 192),
                                   (6500, 2560, 3072), (3000, 2312, 128), (257, 264, 64)])
def test_gemm_pingpong_256_bit_identical_and_race_screen(M, N, K):
    """The 256x256 ping-pong kernel (tile_config 7; what the heuristic picks for the image in-proj): against
    float64, bit-identical to the lock-step 256x256 and 128x128 kernels (same k-order per output), M tails, K from
    two 32-deep tiles up to 96, and a race screen -- 25 back-to-back launches must all give the same bits."""
    from rpo_amd import _lib as L
    o = ops()
    a, w = rnd((M, K), 1, 0.5), rnd((N, K), 2, K ** -0.5)
    bias = rnd((N,), 3)
    acc = q(a, "bf16") @ q(w, "bf16").t()
    ad, wd, bd = a.to(dev(), torch.bfloat16), w.to(dev(), torch.bfloat16), bias.to(dev())
    row0 = max(M - 300, 0)
    for epi, ref in ((L.EPI_BIAS, acc + bias.double()), (L.EPI_BIAS_QGELU, R.qgelu(acc + bias.double()))):
        outs = {}
        for cfg in (7, 3, 2, 0):
            out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev())
            kw = dict(bias=bd)
            if epi == L.EPI_BIAS_QGELU:
                kw.update(aux_row0=row0, aux=torch.full((M - row0, N), float("nan"), device=dev()))
            o.gemm_nt(ad, wd, out, epi, tile_config=cfg, **kw)
            outs[cfg] = (out, kw.get("aux"))
        close(outs[7][0], ref, "bf16", f"pingpong gemm epi {epi}")
        for cfg in (3, 2, 0):
            assert torch.equal(outs[7][0], outs[cfg][0]), f"tile_config {cfg} differs from the ping-pong kernel"
            if epi == L.EPI_BIAS_QGELU:
                assert torch.equal(outs[7][1], outs[cfg][1])
        if epi == L.EPI_BIAS:
            for _ in range(25):
                out = torch.empty((M, N), dtype=torch.bfloat16, device=dev())
                o.gemm_nt(ad, wd, out, epi, tile_config=7, bias=bd)
                assert torch.equal(out, outs[7][0]), "ping-pong kernel is not deterministic: LDS race"


@pytest.mark.parametrize("M,N,K", [(7072, 2304, 768), (7072, 3072, 768), (4096, 4096, 128), (2048, 1536, 256),
                                   (7000, 2304, 192), (6500, 2560, 3072), (3000, 2312, 192), (257, 264, 128)])
@pytest.mark.parametrize("mode", ["bf16", "f16"])
def test_gemm_one_wave_per_simd_256_bit_identical_and_race_screen(mode, M, N, K):
    """The one-wave-per-SIMD 256x256 kernel (tile_config 8, hand-scheduled asm k-loop): against float64,
    bit-identical to the lock-step 128x128 kernel (same k-order per output), M / N tails, K from 128 (four 32-deep
    tiles: one steady-state iteration + the three-tile tail) upwards, and a race screen -- 25 back-to-back launches
    must all give the same bits."""
    from rpo_amd import _lib as L
    o = ops()
    if mode == "f16" and K == 3072:
        pytest.skip("one long-K case per format is enough (CPU reference matmul time)")
    a, w = rnd((M, K), 1, 0.5), rnd((N, K), 2, K ** -0.5)
    bias = rnd((N,), 3)
    acc = q(a, mode) @ q(w, mode).t()
    ad, wd, bd = a.to(dev(), DT[mode]), w.to(dev(), DT[mode]), bias.to(dev())
    row0 = max(M - 300, 0)
    ref_cfg = 2
    for epi, ref in ((L.EPI_BIAS, acc + bias.double()), (L.EPI_BIAS_QGELU, R.qgelu(acc + bias.double()))):
        outs = {}
        for cfg in (8, ref_cfg):
            out = torch.full((M, N), float("nan"), dtype=DT[mode], device=dev())
            kw = dict(bias=bd)
            if epi == L.EPI_BIAS_QGELU:
                kw.update(aux_row0=row0, aux=torch.full((M - row0, N), float("nan"), device=dev()))
            o.gemm_nt(ad, wd, out, epi, tile_config=cfg, **kw)
            outs[cfg] = (out, kw.get("aux"))
        close(outs[8][0], ref, mode, f"w4 gemm epi {epi}")
        assert torch.equal(outs[8][0], outs[ref_cfg][0]), f"tile_config {ref_cfg} differs from the one-wave-per-SIMD kernel"
        if epi == L.EPI_BIAS_QGELU:
            assert torch.equal(outs[8][1], outs[ref_cfg][1])
        if epi == L.EPI_BIAS:
            for _ in range(25):
                out = torch.empty((M, N), dtype=DT[mode], device=dev())
                o.gemm_nt(ad, wd, out, epi, tile_config=8, bias=bd)
                assert torch.equal(out, outs[8][0]), "one-wave-per-SIMD kernel is not deterministic: LDS race"


@pytest.mark.parametrize("M,N,K", [(7072, 3072, 768), (7072, 3072, 128), (6000, 3072, 192), (7168, 2688, 256),
                                   (7001, 3072, 3072)])
@pytest.mark.parametrize("mode", ["bf16", "f16"])
def test_gemm_one_round_224x384_bit_identical_and_race_screen(mode, M, N, K):
    """The 224x384 one-wave-per-SIMD kernel (tile_config 10; generated k-loop, per-wave epilogue; what the heuristic
    picks for c_fc at B = 32): against float64, bit-identical to the 128x128 kernel, rows-per-tile that do not divide
    into 32 (221 of 224, 219, ...), M tails, K from two 64-deep tiles up, race screen of 25 launches."""
    from rpo_amd import _lib as L
    o = ops()
    if mode == "f16" and K == 3072:
        pytest.skip("one long-K case per format is enough (CPU reference matmul time)")
    a, w = rnd((M, K), 1, 0.5), rnd((N, K), 2, K ** -0.5)
    bias = rnd((N,), 3)
    acc = q(a, mode) @ q(w, mode).t()
    ad, wd, bd = a.to(dev(), DT[mode]), w.to(dev(), DT[mode]), bias.to(dev())
    row0 = max(M - 800, 0)
    for epi, ref in ((L.EPI_BIAS, acc + bias.double()), (L.EPI_BIAS_QGELU, R.qgelu(acc + bias.double()))):
        outs = {}
        for cfg in (10, 2):
            out = torch.full((M, N), float("nan"), dtype=DT[mode], device=dev())
            kw = dict(bias=bd)
            if epi == L.EPI_BIAS_QGELU:
                kw.update(aux_row0=row0, aux=torch.full((M - row0, N), float("nan"), device=dev()))
            o.gemm_nt(ad, wd, out, epi, tile_config=cfg, **kw)
            outs[cfg] = (out, kw.get("aux"))
        close(outs[10][0], ref, mode, f"224x384 gemm epi {epi}")
        assert torch.equal(outs[10][0], outs[2][0]), "128x128 tiles differ from the 224x384 kernel"
        if epi == L.EPI_BIAS_QGELU:
            assert torch.equal(outs[10][1], outs[2][1])
        if epi == L.EPI_BIAS:
            for _ in range(25):
                out = torch.empty((M, N), dtype=DT[mode], device=dev())
                o.gemm_nt(ad, wd, out, epi, tile_config=10, bias=bd)
                assert torch.equal(out, outs[10][0]), "224x384 kernel is not deterministic: LDS race"


@pytest.mark.parametrize("mode", ["bf16", "f16"])
@pytest.mark.parametrize("n0,n1,units,N", [(197, 24, 32, 3072), (197, 4, 32, 3072), (197, 16, 32, 3072), (190, 34, 30, 3072),
                                           (224, 0, 28, 3072), (197, 24, 64, 3072),      # 64 units: two rounds of 256 tiles
                                           (257, 24, 16, 4096), (270, 18, 15, 4096)])    # ViT-L/14 rows: 288 x 256 tiles
def test_gemm_row_unit_hint_changes_tiling_not_results(mode, n0, n1, units, N):
    """rpo_gemm_args.seg_rows0 / seg_rows1 / seg1_row0 (include/rpo_amd.h): the 224x384 kernel then builds one tile
    from one unit's rows of BOTH row segments (an image's frozen rows + its prompt rows).  Output and saved
    pre-activations must be the bits of the contiguous tiling and of the 128x128 kernel; a hint that does not describe
    the matrix is ignored."""
    from rpo_amd import _lib as L
    o = ops()
    K = 256
    seg1 = n0 * units
    M = seg1 + n1 * units
    a, w, bias = rnd((M, K), 1, 0.5), rnd((N, K), 2, K ** -0.5), rnd((N,), 3)
    ad, wd, bd = a.to(dev(), DT[mode]), w.to(dev(), DT[mode]), bias.to(dev())
    row0 = seg1 if n1 else M - 300
    res = {}
    for name, cfg, hint in (("units", 10, (n0, n1, seg1)), ("contiguous", 10, None), ("128x128", 2, None),
                            ("bad hint", 10, (n0 + 1, n1, seg1)), ("auto", 0, (n0, n1, seg1))):
        out = torch.full((M, N), float("nan"), dtype=DT[mode], device=dev())
        aux = torch.full((M - row0, N), float("nan"), device=dev())
        o.gemm_nt(ad, wd, out, L.EPI_BIAS_QGELU, bias=bd, aux=aux, aux_row0=row0, tile_config=cfg, row_units=hint)
        res[name] = (out, aux)
    pre = q(a, mode) @ q(w, mode).t() + bias.double()
    close(res["units"][0], R.qgelu(pre), mode, "row-unit tiles")
    close(res["units"][1], pre[row0:], mode, "row-unit tiles, saved u")
    for name in ("contiguous", "128x128", "bad hint", "auto"):
        assert torch.equal(res[name][0], res["units"][0]) and torch.equal(res[name][1], res["units"][1]), name
    # the derivative form of the saved operand (16-bit aux): every tiling stores the same bits, close to float64
    d_res = {}
    for name, cfg, hint in (("units", 10, (n0, n1, seg1)), ("128x128", 2, None), ("64x64", 5, None)):
        out = torch.empty((M, N), dtype=DT[mode], device=dev())
        aux16 = torch.full((M - row0, N), float("nan"), dtype=DT[mode], device=dev())
        o.gemm_nt(ad, wd, out, L.EPI_BIAS_QGELU, bias=bd, aux=aux16, aux_row0=row0, tile_config=cfg, row_units=hint)
        assert torch.equal(out, res["units"][0])
        d_res[name] = aux16
    close(d_res["units"], R.qgelu_grad(pre[row0:]), mode, "row-unit tiles, saved derivative")
    assert torch.equal(d_res["128x128"], d_res["units"]) and torch.equal(d_res["64x64"], d_res["units"])
    for _ in range(10):
        out = torch.empty((M, N), dtype=DT[mode], device=dev())
        aux = torch.empty((M - row0, N), device=dev())
        o.gemm_nt(ad, wd, out, L.EPI_BIAS_QGELU, bias=bd, aux=aux, aux_row0=row0, tile_config=10, row_units=(n0, n1, seg1))
        assert torch.equal(out, res["units"][0]) and torch.equal(aux, res["units"][1]), "LDS race"


@pytest.mark.parametrize("mode", ["bf16", "f16"])
@pytest.mark.parametrize("n0,n1,units,K", [(197, 24, 32, 768), (197, 24, 32, 3072), (197, 4, 32, 256), (190, 34, 30, 512),
                                           (224, 0, 28, 1024), (197, 24, 64, 256),    # 64 units: two rounds
                                           (197, 48, 32, 768), (197, 48, 32, 3072), (230, 26, 30, 256)])   # 256x96 tiles (K = 48)
def test_gemm_one_round_224x96_split_k(mode, n0, n1, units, K):
    """The 224x96 kernel of the N = 768 residual GEMMs (tile_config 11; out-proj / c_proj of the image tower: one row
    unit x 96 columns per workgroup, the four waves split the contraction): C, the 16-bit copy and the 96-column row
    statistics against float64; deterministic over 25 launches (fixed reduction order, LDS race screen); a consumer
    given ln_group = 96 reproduces quickgelu(LN(C) W^T + b); layouts that do not fit are refused, not approximated."""
    from rpo_amd import _lib as L
    from rpo_amd._lib import RPOLibraryError
    o = ops()
    N = 768
    seg1 = n0 * units
    M = seg1 + n1 * units
    hint = (n0, n1, seg1)
    dt = DT[mode]
    a, w, bias = rnd((M, K), 1, 0.5), rnd((N, K), 2, K ** -0.5), rnd((N,), 3)
    resid = rnd((M, N), 4, 2.0) + 0.4
    ad, wd, bd, rd = a.to(dev(), dt), w.to(dev(), dt), bias.to(dev()), resid.to(dev())

    def run(cfg, group, units_hint=hint):
        c = torch.full((M, N), float("nan"), device=dev())
        c2 = torch.full((M, N), float("nan"), dtype=dt, device=dev())
        st = torch.full((M, N // group, 2), float("nan"), device=dev())
        o.gemm_nt(ad, wd, c, L.EPI_BIAS_RESID, bias=bd, resid=rd, out2=c2, ln_stats=st, tile_config=cfg,
                  row_units=units_hint, ln_group=group)
        return c, c2, st

    c, c2, st = run(11, 96)
    ref = q(a, mode) @ q(w, mode).t() + bias.double() + resid.double()
    close(c, ref, "f32", "224x96 split-k C", tol=1e-4)
    assert torch.equal(c2.cpu(), c.cpu().to(dt)), "out2 must be the RNE act-dtype copy of C"
    grp = c.double().cpu().reshape(M, N // 96, 96)
    ref_st = torch.stack([grp.mean(-1), ((grp - grp.mean(-1, keepdim=True)) ** 2).sum(-1)], -1)
    close(st, ref_st, "f32", "96-column partial row statistics", tol=2e-5)
    c0, c20, st0 = run(0, 96)                      # the heuristic picks the same kernel
    assert torch.equal(c0, c) and torch.equal(c20, c2) and torch.equal(st0, st)
    for _ in range(25):
        cr, c2r, str_ = run(11, 96)
        assert torch.equal(cr, c) and torch.equal(c2r, c2) and torch.equal(str_, st), "not deterministic: LDS race"
    # the generic tiles agree to fp32 summation-order noise and keep their own 64-column layout
    cg, _, stg = run(6, 64)
    close(cg, c.double().cpu(), "f32", "64x128 tiles vs 224x96", tol=2e-5)
    assert stg.shape[1] == N // 64
    with pytest.raises(RPOLibraryError):
        run(6, 96)                                 # a generic kernel cannot write 96-column statistics
    with pytest.raises(RPOLibraryError):
        run(11, 96, units_hint=None)               # no row units: the kernel does not apply
    with pytest.raises(RPOLibraryError):
        run(11, 64)                                # ... and it cannot write 64-column statistics
    # consumer side with ln_group = 96
    if K == 768:
        Nc = 3072
        wc, bc = rnd((Nc, N), 5, N ** -0.5), rnd((Nc,), 6)
        gamma, beta = rnd((N,), 7, 0.1) + 1.0, rnd((N,), 8, 0.05)
        wq = (wc.double() * gamma.double()[None, :]).float().to(dt)
        sc = wq.double().sum(1).float()
        bq = (bc.double() + wc.double() @ beta.double()).float()
        x64 = c.double().cpu()
        mu = x64.mean(1, keepdim=True)
        rstd = (x64.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
        pre = ((x64 - mu) * rstd * gamma.double() + beta.double()) @ wc.double().t() + bc.double()
        outs = []
        for cfg in (0, 10, 8, 2):
            y = torch.full((M, Nc), float("nan"), dtype=dt, device=dev())
            o.gemm_nt(c2, wq.to(dev()), y, L.EPI_LN_BIAS_QGELU, bias=bq.to(dev()), ln_stats=st, ln_colsum=sc.to(dev()),
                      tile_config=cfg, row_units=hint, ln_group=96)
            outs.append(y)
        close(outs[0], R.qgelu(pre), mode, "LN-folded c_fc from 96-column statistics", tol=1.5 * TOL[mode])
        for y in outs[1:]:
            assert torch.equal(y, outs[0])


@experimental
@pytest.mark.parametrize("mode", ["bf16", "f16"])
@pytest.mark.parametrize("safe", [False, True])
@pytest.mark.parametrize("n0,n1,units", [(197, 24, 32), (197, 4, 32), (190, 34, 30)])
def test_mlp_fused_equals_the_two_launches(mode, safe, n0, n1, units):
    """rpo_mlp_fused (round 4, experiment): c_fc (LN-folded, QuickGELU, saved derivative of the prompt rows) and c_proj
    (residual as hi / lo halves, in place; 16-bit copy + 96-column statistics) of an image block in ONE launch -- the 8
    workgroups of a row unit hand g over at a counter.  Every output must be the bits of the two rpo_gemm_nt launches,
    with the XCD-local hand-off and with the placement-independent one, over repeated launches on the same counters
    (they are never reset); shapes the kernel does not cover are refused, not approximated."""
    from rpo_amd import _lib as L
    o = ops()
    d, Nf = 768, 3072
    dt = DT[mode]
    seg1 = n0 * units
    M = seg1 + n1 * units
    hint = (n0, n1, seg1)
    x = (rnd((M, d), 1, 2.0) + 0.4)
    hi = x.to(dt); lo = (x - hi.float()).to(dt)
    xb = hi.to(dev())
    grp = x.double().reshape(M, d // 96, 96)
    stats = torch.stack([grp.mean(-1), ((grp - grp.mean(-1, keepdim=True)) ** 2).sum(-1)], -1).float().to(dev())
    w_fc, b_fc = rnd((Nf, d), 2, d ** -0.5).to(dev(), dt), rnd((Nf,), 3).to(dev())
    s_fc = w_fc.double().sum(1).float()
    w_pr, b_pr = rnd((d, Nf), 4, Nf ** -0.5).to(dev(), dt), rnd((d,), 5).to(dev())

    def buffers():
        return dict(g=torch.full((M, Nf), float("nan"), dtype=dt, device=dev()),
                    aux=torch.full((M - seg1, Nf), float("nan"), dtype=dt, device=dev()),
                    c=torch.full((M, d), float("nan"), device=dev()), h=hi.to(dev()).clone(), l=lo.to(dev()).clone(),
                    st=torch.full((M, d // 96, 2), float("nan"), device=dev()))

    def kws(b):
        fc = dict(a=xb, w=w_fc, out=b["g"], epilogue=L.EPI_LN_BIAS_QGELU, bias=b_fc, aux=b["aux"], aux_row0=seg1,
                  ln_stats=stats, ln_colsum=s_fc, row_units=hint, ln_group=96)
        pr = dict(a=b["g"], w=w_pr, out=b["c"], epilogue=L.EPI_BIAS_RESID, bias=b_pr, row_units=hint, resid_hi=b["h"],
                  resid_lo=b["l"], out2=b["h"], out_lo=b["l"], c_row0=seg1, ln_stats=b["st"], ln_group=96)
        return fc, pr
    ref = buffers()
    fc, pr = kws(ref)
    o.gemm_nt(**fc); o.gemm_nt(**pr)
    cnt = torch.zeros(units + 1, dtype=torch.int32, device=dev())            # + the give-up word
    applies = units * 8 <= torch.cuda.get_device_properties(0).multi_processor_count
    for rep in range(4):
        got = buffers()
        fc, pr = kws(got)
        done = o.mlp_fused(fc, pr, cnt, safe=safe)
        assert done == applies
        if not done:
            return
        torch.cuda.synchronize()
        for k in ("g", "aux", "h", "l", "st"):
            assert torch.equal(got[k], ref[k]), (k, rep)
        assert torch.equal(got["c"][seg1:], ref["c"][seg1:]) and torch.isnan(got["c"][:seg1]).all()
    assert cnt.cpu().tolist() == [32] * units + [0]            # 4 launches x 8 arrivals, never reset; no poll gave up
    # not the row-unit forms: refused
    fc, pr = kws(buffers())
    fc2 = dict(fc, row_units=None); pr2 = dict(pr, row_units=None, resid_hi=None, resid_lo=None, out_lo=None, c_row0=0,
                                               resid=torch.zeros(M, d, device=dev()), ln_group=64,
                                               ln_stats=torch.empty(M, d // 64, 2, device=dev()))
    assert o.mlp_fused(fc2, pr2, cnt) is False


@pytest.mark.parametrize("mode", ["bf16", "f16"])
@pytest.mark.parametrize("n0,n1,units,N,K", [(197, 24, 32, 768, 768), (197, 24, 32, 768, 3072), (257, 24, 16, 1024, 1024),
                                             (197, 48, 32, 768, 768), (197, 48, 32, 768, 3072)])      # 256x96 tiles (K = 48)
def test_gemm_residual_stream_as_hi_lo_halves(mode, n0, n1, units, N, K):
    """rpo_gemm_args.resid_hi / resid_lo / out_lo / c_row0: the residual stream of the one-round residual GEMMs as two
    16-bit halves.  With a residual that IS exactly hi + lo the result must equal the fp32-residual launch bit for bit
    (C on the rows >= c_row0, out2, statistics); rows below c_row0 of C stay untouched; out2 + out_lo reproduces the fp32
    value to the 16 (bf16) / 21 (fp16) mantissa bits two halves carry; in place (resid_hi == out2, resid_lo == out_lo)
    gives the same bits; kernels that do not implement it refuse."""
    from rpo_amd import _lib as L
    from rpo_amd._lib import RPOLibraryError
    o = ops()
    seg1 = n0 * units
    M = seg1 + n1 * units
    hint = (n0, n1, seg1)
    dt = DT[mode]
    grp = 96 if N == 768 else 64
    a, w, bias = rnd((M, K), 1, 0.5), rnd((N, K), 2, K ** -0.5), rnd((N,), 3)
    r32 = rnd((M, N), 4, 2.0) + 0.4
    hi = r32.to(dt)
    lo = (r32 - hi.float()).to(dt)
    resid = hi.float() + lo.float()                 # exactly representable: what the halves decode to
    assert o.gemm_hilo_ok(M, N, K, dt, hint, grp) and not o.gemm_hilo_ok(M, N, K, dt, None, grp)
    ad, wd, bd = a.to(dev(), dt), w.to(dev(), dt), bias.to(dev())

    def outs():
        return (torch.full((M, N), float("nan"), device=dev()), torch.full((M, N), float("nan"), dtype=dt, device=dev()),
                torch.full((M, N // grp, 2), float("nan"), device=dev()))
    c, c2, st = outs()
    o.gemm_nt(ad, wd, c, L.EPI_BIAS_RESID, bias=bd, resid=resid.to(dev()), out2=c2, ln_stats=st, row_units=hint, ln_group=grp)
    ch, c2h, sth = outs()
    clo = torch.full((M, N), float("nan"), dtype=dt, device=dev())
    o.gemm_nt(ad, wd, ch, L.EPI_BIAS_RESID, bias=bd, resid_hi=hi.to(dev()), resid_lo=lo.to(dev()), out2=c2h, out_lo=clo,
              c_row0=seg1, ln_stats=sth, row_units=hint, ln_group=grp)
    assert torch.equal(ch[seg1:], c[seg1:]) and torch.equal(c2h, c2) and torch.equal(sth, st)
    assert bool(torch.isnan(ch[:seg1]).all()), "rows below c_row0 of C must not be written"
    rec = c2h.float() + clo.float()
    # (fp16: lo of a small value is a subnormal, resolution 6e-8)
    bound = c.abs() * (2.0 ** -15 if mode == "bf16" else 2.0 ** -19) + (0.0 if mode == "bf16" else 6e-8)
    assert bool(((rec - c).abs() <= bound).all()), f"hi + lo off by {((rec - c).abs() - bound).max().item():.2e} beyond the bound"
    # in place: the halves of the input are overwritten by the halves of the output
    hh, ll = hi.to(dev()).clone(), lo.to(dev()).clone()
    cp, _, stp = outs()
    o.gemm_nt(ad, wd, cp, L.EPI_BIAS_RESID, bias=bd, resid_hi=hh, resid_lo=ll, out2=hh, out_lo=ll, c_row0=seg1,
              ln_stats=stp, row_units=hint, ln_group=grp)
    assert torch.equal(hh, c2h) and torch.equal(ll, clo) and torch.equal(cp[seg1:], ch[seg1:]) and torch.equal(stp, sth)
    with pytest.raises(RPOLibraryError):            # the generic tiles do not implement it
        o.gemm_nt(ad, wd, outs()[0], L.EPI_BIAS_RESID, bias=bd, resid_hi=hi.to(dev()), resid_lo=lo.to(dev()))


@pytest.mark.parametrize("mode", ["bf16", "f16"])
@pytest.mark.parametrize("n0,n1,units,K", [(257, 24, 16, 1024), (257, 24, 16, 4096), (270, 18, 15, 768), (257, 4, 16, 256)])
def test_gemm_one_round_288x64_split_k(mode, n0, n1, units, K):
    """The 288x64 geometry of the split-k kernel (ViT-L/14: 257 + K rows per image, 1024 = 16 x 64 columns; LDS ring of 3,
    unrolled body of 6 iterations entered at position 5 (K = 1024, 4096, 256) or 3 (K = 768)): C, the 16-bit copy and
    the 64-column row statistics against float64 and against the 64x128 tiles; deterministic over 25 launches."""
    from rpo_amd import _lib as L
    from rpo_amd._lib import RPOLibraryError
    o = ops()
    N = 1024
    seg1 = n0 * units
    M = seg1 + n1 * units
    hint = (n0, n1, seg1)
    dt = DT[mode]
    a, w, bias = rnd((M, K), 1, 0.5), rnd((N, K), 2, K ** -0.5), rnd((N,), 3)
    resid = rnd((M, N), 4, 2.0) + 0.4
    ad, wd, bd, rd = a.to(dev(), dt), w.to(dev(), dt), bias.to(dev()), resid.to(dev())

    def run(cfg, units_hint=hint):
        c = torch.full((M, N), float("nan"), device=dev())
        c2 = torch.full((M, N), float("nan"), dtype=dt, device=dev())
        st = torch.full((M, N // 64, 2), float("nan"), device=dev())
        o.gemm_nt(ad, wd, c, L.EPI_BIAS_RESID, bias=bd, resid=rd, out2=c2, ln_stats=st, tile_config=cfg, row_units=units_hint)
        return c, c2, st

    assert o.gemm_stats_group(M, N, K, dt, hint) == 64
    c, c2, st = run(11)
    ref = q(a, mode) @ q(w, mode).t() + bias.double() + resid.double()
    close(c, ref, "f32", "288x64 split-k C", tol=1e-4)
    assert torch.equal(c2.cpu(), c.cpu().to(dt)), "out2 must be the RNE act-dtype copy of C"
    grp = c.double().cpu().reshape(M, N // 64, 64)
    ref_st = torch.stack([grp.mean(-1), ((grp - grp.mean(-1, keepdim=True)) ** 2).sum(-1)], -1)
    close(st, ref_st, "f32", "64-column partial row statistics", tol=2e-5)
    c0, c20, st0 = run(0)                          # the heuristic picks the same kernel
    assert torch.equal(c0, c) and torch.equal(c20, c2) and torch.equal(st0, st)
    for _ in range(25):
        cr, c2r, str_ = run(11)
        assert torch.equal(cr, c) and torch.equal(c2r, c2) and torch.equal(str_, st), "not deterministic: LDS race"
    cg, _, stg = run(6, units_hint=None)           # generic 64x128 tiles
    close(cg, c.double().cpu(), "f32", "64x128 tiles vs 288x64", tol=2e-5)
    close(stg, st.double().cpu(), "f32", "statistics, 64x128 tiles vs 288x64", tol=2e-5)
    with pytest.raises(RPOLibraryError):
        run(11, units_hint=None)


@pytest.mark.parametrize("mode", ["bf16", "f16"])
@pytest.mark.parametrize("M,d,N", [(300, 768, 3072), (4200, 768, 2304), (1000, 1024, 4096), (7072, 768, 3072)])
def test_gemm_layernorm_fold(mode, M, d, N):
    """LayerNorm folded into the GEMM around it (include/rpo_amd.h RPO_EPI_LN_*): the producer (BIAS_RESID) leaves the
    act-dtype copy of its result and per-row 64-column partial statistics; the consumer takes that copy as A, the
    gamma-scaled weight and s / b' and must reproduce  quickgelu(LN(x) W^T + b)  (clip/model.py:156-159 feeding
    :174-175) to the tolerance of the unfolded path.  Every tile shape of the consumer gives the same bits."""
    from rpo_amd import _lib as L
    o = ops()
    att, w_out, b_out = rnd((M, d), 1, 0.5), rnd((d, d), 2, d ** -0.5), rnd((d,), 3)
    resid = rnd((M, d), 4, 2.0) + 0.4
    w, b = rnd((N, d), 5, d ** -0.5), rnd((N,), 6)
    gamma, beta = rnd((d,), 7, 0.1) + 1.0, rnd((d,), 8, 0.05)
    dt = DT[mode]
    xm = torch.full((M, d), float("nan"), device=dev())
    xb = torch.full((M, d), float("nan"), dtype=dt, device=dev())
    stats = torch.full((M, d // 64, 2), float("nan"), device=dev())
    o.gemm_nt(att.to(dev(), dt), w_out.to(dev(), dt), xm, L.EPI_BIAS_RESID, bias=b_out.to(dev()), resid=resid.to(dev()),
              out2=xb, ln_stats=stats)
    xm64 = xm.double().cpu()
    close(xm, q(att, mode) @ q(w_out, mode).t() + b_out.double() + resid.double(), "f32", "producer result", tol=1e-4)
    assert torch.equal(xb.cpu(), xm.cpu().to(dt)), "out2 must be the RNE act-dtype copy of C"
    grp = xm64.reshape(M, d // 64, 64)
    ref_stats = torch.stack([grp.mean(-1), ((grp - grp.mean(-1, keepdim=True)) ** 2).sum(-1)], -1)
    close(stats, ref_stats, "f32", "partial row statistics", tol=2e-5)
    # consumer: fold on the host exactly as Engine._fold does
    wq = (w.double() * gamma.double()[None, :]).float().to(dt)
    s = wq.double().sum(1).float()
    bq = (b.double() + w.double() @ beta.double()).float()
    mu = xm64.mean(1, keepdim=True)
    rstd = (xm64.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    pre = ((xm64 - mu) * rstd * gamma.double() + beta.double()) @ w.double().t() + b.double()
    row0 = M - 100
    outs = {}
    one_round = (M, N) == (7072, 3072)
    cfgs = (0, 8, 2, 5) + ((10, "units") if one_round else ())
    for cfg in cfgs:
        y = torch.full((M, N), float("nan"), dtype=dt, device=dev())
        aux = torch.full((M - row0, N), float("nan"), device=dev())
        hint = dict(tile_config=10, row_units=(197, 24, 6304)) if cfg == "units" else dict(tile_config=cfg)
        o.gemm_nt(xb, wq.to(dev()), y, L.EPI_LN_BIAS_QGELU, bias=bq.to(dev()), aux=aux, aux_row0=row0,
                  ln_stats=stats, ln_colsum=s.to(dev()), **hint)
        outs[cfg] = (y, aux)
    close(outs[0][0], R.qgelu(pre), mode, "LN-folded c_fc", tol=1.5 * TOL[mode])
    close(outs[0][1], pre[row0:], mode, "LN-folded saved u", tol=1.5 * TOL[mode])
    for cfg in cfgs[1:]:
        assert torch.equal(outs[cfg][0], outs[0][0]) and torch.equal(outs[cfg][1], outs[0][1]), f"tile_config {cfg}"
    # the form the engine's 16-bit modes save: d quickgelu / du in the act dtype (the 224x384 kernel forms it in its block
    # loop from the activation's own sigmoid) -- the same bits from every tiling, and with a row stride that sends the
    # one-round kernel down its separate fp32 pass
    d16 = {}
    for cfg in cfgs:
        for pad in (0, 4):
            y = torch.full((M, N), float("nan"), dtype=dt, device=dev())
            aux16 = torch.full((M - row0, N + pad), float("nan"), dtype=dt, device=dev())[:, :N]
            hint = dict(tile_config=10, row_units=(197, 24, 6304)) if cfg == "units" else dict(tile_config=cfg)
            o.gemm_nt(xb, wq.to(dev()), y, L.EPI_LN_BIAS_QGELU, bias=bq.to(dev()), aux=aux16, aux_row0=row0,
                      ln_stats=stats, ln_colsum=s.to(dev()), **hint)
            assert torch.equal(y, outs[0][0]), f"tile_config {cfg}, aux16 pad {pad}"
            d16[cfg, pad] = aux16
    close(d16[0, 0], R.qgelu_grad(pre[row0:]), mode, "LN-folded saved derivative", tol=1.5 * TOL[mode])
    for key, v in d16.items():
        assert torch.equal(v, d16[0, 0]), f"saved derivative, tile_config / pad {key}"
    y = torch.full((M, N), float("nan"), dtype=dt, device=dev())
    o.gemm_nt(xb, wq.to(dev()), y, L.EPI_LN_BIAS, bias=bq.to(dev()), ln_stats=stats, ln_colsum=s.to(dev()))
    close(y, pre, mode, "LN-folded in-proj", tol=1.5 * TOL[mode])


@pytest.mark.parametrize("mode", ["f32", "bf16", "f16"])
def test_gemm_split_k_feeds_layernorm_bwd(mode):
    """split-K slabs (deterministic, no atomics) are summed by rpo_layernorm_bwd in slab order."""
    from rpo_amd import _lib as L
    o = ops()
    M, N, K, S = 200, 512, 2048, 8
    a, w = rnd((M, K), 1, 0.5), rnd((N, K), 2, K ** -0.5)
    x, gam, dres = rnd((M, N), 3, 2.0), rnd((N,), 4, 0.1) + 1.0, rnd((M, N), 5)
    slabs = torch.full((S, M + 8, N), float("nan"), device=dev())[:, :M]      # non-contiguous slabs
    o.gemm_nt(a.to(dev(), DT[mode]), w.to(dev(), DT[mode]), slabs, L.EPI_NONE, split_k=S)
    acc = q(a, mode) @ q(w, mode).t()
    close(slabs.sum(0), acc, mode, "split-K slab sum", tol=TOL["f32"] if mode == "f32" else 1e-4)
    dx = torch.empty(M, N, device=dev())
    o.layernorm_bwd(slabs, x.to(dev()), gam.to(dev()), dres.to(dev()), dx, None)
    ref = dres.double() + R.ln_bwd(acc, x.double(), gam.double())
    close(dx, ref, "f32", "ln bwd over split-K slabs", tol=2e-5 if mode == "f32" else 2e-4)


@pytest.mark.parametrize("mode", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("patch,size", [(16, 64), (14, 56)])
def test_patch_embed_matches_conv(mode, patch, size):
    from rpo_amd import _lib as L
    o = ops()
    B, d = 3, 256
    g = size // patch
    npatch = g * g
    img, w = rnd((B, 3, size, size), 1), rnd((d, 3, patch, patch), 2, 0.05)
    pos, cls, prm = rnd((npatch + 1, d), 3), rnd((d,), 4), rnd((5, d), 5)
    kmult = 32 if mode == "f32" else 64
    kp = (3 * patch * patch + kmult - 1) // kmult * kmult
    cols = torch.full((B * npatch, kp), float("nan"), dtype=DT[mode], device=dev())
    o.im2col_patches(img.to(dev()), cols, patch)
    wp = torch.zeros(d, kp)
    wp[:, :3 * patch * patch] = w.reshape(d, -1)
    N, Kp = npatch + 1, 5
    x = torch.full((B * (N + Kp), d), float("nan"), device=dev())
    o.gemm_nt(cols, wp.to(dev(), DT[mode]), x, L.EPI_PATCH, resid=pos.to(dev()), group=npatch)
    o.img_assemble(x, cls.to(dev()), pos.to(dev()), prm.to(dev()), B, N, Kp)
    conv = torch.nn.functional.conv2d(q(img, mode), q(w, mode), stride=patch)
    emb = conv.reshape(B, d, -1).permute(0, 2, 1)
    ref = torch.cat([cls.double().repeat(B, 1, 1), emb], 1) + pos.double()
    ref = torch.cat([ref.reshape(B * N, d), prm.double().repeat(B, 1)], 0)
    close(x, ref, mode, "patch embed + assemble", tol=TOL["f32"] if mode == "f32" else 1e-4)


@pytest.mark.parametrize("d", [512, 768, 1024])
def test_layernorm_fwd_bwd(d):
    o = ops()
    rows = 37
    x, w, b = rnd((rows, d), 1, 2.0) + 0.3, rnd((d,), 2, 0.1) + 1.0, rnd((d,), 3, 0.05)
    dy, dres = rnd((rows, d), 4), rnd((rows, d), 5)
    xd = x.to(dev())
    for mode in ("f32", "bf16", "f16"):
        y = torch.empty(rows, d, dtype=DT[mode], device=dev())
        o.layernorm_fwd(xd, w.to(dev()), b.to(dev()), y)
        close(y, R.ln_fwd(x.double(), w.double(), b.double()), mode, f"ln fwd d={d}",
              tol={"f32": 2e-6, "bf16": 8e-3, "f16": 1e-3}[mode])
        dyq = dy.to(DT[mode])
        dx = torch.empty(rows, d, device=dev())
        dxc = torch.empty(rows, d, dtype=DT[mode], device=dev())
        o.layernorm_bwd(dyq.to(dev()), xd, w.to(dev()), dres.to(dev()), dx, dxc)
        ref = dres.double() + R.ln_bwd(dyq.double(), x.double(), w.double())
        close(dx, ref, "f32", f"ln bwd d={d} dy={mode}", tol=5e-6)
        close(dxc, ref, mode, f"ln bwd cast d={d}", tol={"f32": 5e-6, "bf16": 8e-3, "f16": 1e-3}[mode])
        o.layernorm_bwd(dyq.to(dev()), xd, w.to(dev()), None, dx, None)
        close(dx, R.ln_bwd(dyq.double(), x.double(), w.double()), "f32", "ln bwd no-resid", tol=5e-6)
    # in-place fp32 (ln_pre)
    xin = xd.clone()
    o.layernorm_fwd(xin, w.to(dev()), b.to(dev()), xin)
    close(xin, R.ln_fwd(x.double(), w.double(), b.double()), "f32", "ln in-place", tol=2e-6)


def _img_rows(B, N, Kp, d, seed):
    """token matrix in the engine's row layout + per-image views"""
    t = rnd((B * (N + Kp), 3 * d), seed, 1.0)
    return t


@pytest.mark.parametrize("mode", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("B,H,N,Kp", [(2, 3, 197, 24), (1, 2, 50, 7), (2, 1, 257, 48), (1, 1, 5, 0), (1, 2, 224, 40)])
def test_attn_readonly_fwd(mode, B, H, N, Kp):
    o = ops()
    d = 64 * H
    qkv = _img_rows(B, N, Kp, d, 11)
    qkv[:, :d] *= 1.5                      # sharper softmax
    t = qkv.to(dev(), DT[mode])
    out = torch.full((B * (N + Kp), d), float("nan"), dtype=DT[mode], device=dev())
    o.attn_readonly_fwd(t[:, :d], t[:, d:2 * d], t[:, 2 * d:], out, B, H, N, Kp)
    q64 = q(qkv, mode)
    ref = torch.empty(B * (N + Kp), d, dtype=torch.float64)
    for b in range(B):
        fr = slice(b * N, (b + 1) * N)
        pr = slice(B * N + b * Kp, B * N + (b + 1) * Kp)
        k, v = q64[fr, d:2 * d], q64[fr, 2 * d:]
        ref[fr] = R.attn_rows_fwd(q64[fr, :d], k, v, H)
        if Kp:
            ref[pr] = R.attn_rows_fwd(q64[pr, :d], k, v, H)
    close(out, ref, mode, f"attn fwd B{B} H{H} N{N} K{Kp}")


@pytest.mark.parametrize("mode", ["bf16", "f16"])
@pytest.mark.parametrize("B,H,N,Kp", [(32, 12, 197, 24), (24, 12, 197, 4), (20, 16, 257, 24)])
def test_attn_readonly_fwd_split_units_change_nothing(mode, B, H, N, Kp):
    """A query's arithmetic does not depend on which workgroup computes it or on how many units a launch holds: the
    whole-batch launch must give the bits of image-by-image launches.  (Also the check of the -DRPO_ATTN_SPLIT build,
    which splits some units' query tiles over two workgroups when there are more units than CUs: attn_image.hip.)"""
    o = ops()
    d = 64 * H
    qkv = _img_rows(B, N, Kp, d, 21)
    t = qkv.to(dev(), DT[mode])
    Rf, S = B * N, N + Kp
    out = torch.full((B * S, d), float("nan"), dtype=DT[mode], device=dev())
    o.attn_readonly_fwd(t[:, :d], t[:, d:2 * d], t[:, 2 * d:], out, B, H, N, Kp)
    assert not torch.isnan(out.float()).any()
    for b in range(0, B, 5):
        one = torch.cat([t[b * N:(b + 1) * N], t[Rf + b * Kp:Rf + (b + 1) * Kp]]).contiguous()
        o1 = torch.full((S, d), float("nan"), dtype=DT[mode], device=dev())
        o.attn_readonly_fwd(one[:, :d], one[:, d:2 * d], one[:, 2 * d:], o1, 1, H, N, Kp)
        assert torch.equal(o1[:N], out[b * N:(b + 1) * N]) and torch.equal(o1[N:], out[Rf + b * Kp:Rf + (b + 1) * Kp]), b
    again = torch.empty_like(out)
    o.attn_readonly_fwd(t[:, :d], t[:, d:2 * d], t[:, 2 * d:], again, B, H, N, Kp)
    assert torch.equal(again, out)


@pytest.mark.parametrize("mode", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("B,H,N,Kp", [(2, 3, 197, 24), (1, 2, 50, 7), (2, 1, 257, 48), (1, 1, 197, 33)])
def test_attn_readonly_bwd(mode, B, H, N, Kp):
    o = ops()
    d = 64 * H
    qkv = _img_rows(B, N, Kp, d, 12)
    da = rnd((B * Kp, d), 13)
    t = qkv.to(dev(), DT[mode])
    dq = torch.full((B * Kp, d), float("nan"), dtype=DT[mode], device=dev())
    Rf = B * N
    o.attn_readonly_bwd(t[Rf:, :d], t[:Rf, d:2 * d], t[:Rf, 2 * d:], da.to(dev(), DT[mode]), dq, B, H, N, Kp)
    q64, da64 = q(qkv, mode), q(da, mode)
    ref = torch.empty(B * Kp, d, dtype=torch.float64)
    for b in range(B):
        fr = slice(b * N, (b + 1) * N)
        pr = slice(Rf + b * Kp, Rf + (b + 1) * Kp)
        ref[b * Kp:(b + 1) * Kp] = R.attn_rows_bwd(q64[pr, :d], q64[fr, d:2 * d], q64[fr, 2 * d:],
                                                   da64[b * Kp:(b + 1) * Kp], H)
    close(dq, ref, mode, f"attn bwd B{B} H{H} N{N} K{Kp}", tol=None if mode == "f32" else 3e-2)


@pytest.mark.parametrize("mode", ["bf16", "f16"])
@pytest.mark.parametrize("B,H,N,Kp", [(2, 12, 197, 24), (3, 8, 50, 7), (1, 12, 257, 32), (2, 8, 197, 1),
                                      (2, 16, 257, 24), (1, 16, 197, 32),      # d = 1024 (ViT-L/14): four weight slots per wave
                                      (2, 12, 197, 48), (1, 16, 257, 33), (2, 8, 60, 64)])   # Kp > 32: a workgroup per query tile
def test_attn_readonly_bwd_with_out_proj_folded_in(mode, B, H, N, Kp):
    """rpo_attn_readonly_bwd_proj: dq from the gradient of the out-proj OUTPUT -- every (image, head) workgroup forms
    da = dx . W_out[:, head] itself.  Against float64 (da rounded to the act dtype as the separate GEMM would), against
    the two-launch path it replaces, and refused where it does not apply."""
    from rpo_amd import _lib as L
    from rpo_amd._lib import RPOLibraryError
    o = ops()
    d = 64 * H
    dt = DT[mode]
    qkv = _img_rows(B, N, Kp, d, 12)
    dx, w_out = rnd((B * Kp, d), 13), rnd((d, d), 14, d ** -0.5)          # w_out[out, in] as nn.Linear
    t = qkv.to(dev(), dt)
    Rf = B * N
    w_t = w_out.t().contiguous().to(dev(), dt)                             # [in, out]: the dX-GEMM packing
    dq = torch.full((B * Kp, d), float("nan"), dtype=dt, device=dev())
    o.attn_readonly_bwd_proj(t[Rf:, :d], t[:Rf, d:2 * d], t[:Rf, 2 * d:], dx.to(dev(), dt), w_t, dq, B, H, N, Kp)
    # the path it replaces
    da = torch.empty((B * Kp, d), dtype=dt, device=dev())
    o.gemm_nt(dx.to(dev(), dt), w_t, da, L.EPI_NONE)
    dq2 = torch.full((B * Kp, d), float("nan"), dtype=dt, device=dev())
    o.attn_readonly_bwd(t[Rf:, :d], t[:Rf, d:2 * d], t[:Rf, 2 * d:], da, dq2, B, H, N, Kp)
    q64 = q(qkv, mode)
    da64 = q((q(dx, mode) @ q(w_out, mode)).float(), mode)                 # da[r, j] = sum_o dx[r, o] W[o, j], rounded
    ref = torch.empty(B * Kp, d, dtype=torch.float64)
    for b in range(B):
        fr = slice(b * N, (b + 1) * N)
        pr = slice(Rf + b * Kp, Rf + (b + 1) * Kp)
        ref[b * Kp:(b + 1) * Kp] = R.attn_rows_bwd(q64[pr, :d], q64[fr, d:2 * d], q64[fr, 2 * d:],
                                                   da64[b * Kp:(b + 1) * Kp], H)
    close(dq, ref, mode, f"attn bwd + d out-proj B{B} H{H} N{N} K{Kp}", tol=3e-2)
    close(dq, dq2.double().cpu(), mode, "fused vs GEMM + attn bwd", tol=3e-2)
    for _ in range(10):
        dq3 = torch.empty_like(dq)
        o.attn_readonly_bwd_proj(t[Rf:, :d], t[:Rf, d:2 * d], t[:Rf, 2 * d:], dx.to(dev(), dt), w_t, dq3, B, H, N, Kp)
        assert torch.equal(dq3, dq), "not deterministic: LDS race"
    if Kp == 24:
        with pytest.raises(RPOLibraryError):                               # 65 query rows: three query tiles
            big = torch.empty((B * 65, d), dtype=dt, device=dev())
            o.attn_readonly_bwd_proj(big, t[:Rf, d:2 * d], t[:Rf, 2 * d:], big, w_t, big.clone(), B, H, N, 65)


@experimental
@pytest.mark.parametrize("mode", ["bf16", "f16"])
@pytest.mark.parametrize("H,lens,Kr,Lmax", [(8, [3, 71, 20, 8, 10, 77, 1], 24, 77), (12, [5, 96, 33], 7, 96), (8, [9, 4], 32, 20)])
def test_text_attn_bwd_with_out_proj_folded_in(mode, H, lens, Kr, Lmax):
    """rpo_attn_bwd_proj_pair with ONE problem and per-group key counts: the text tower's attention backward (per-class
    K / V cache, trainers/rpo.py:144-151) on the image tower's MFMA kernel with the d out-proj GEMM folded in.  Against
    float64 and against the GEMM + VALU-kernel pair it replaces; rows of the cache past len_c hold garbage (NaN) that
    must never be read into the result."""
    from rpo_amd import _lib as L
    o = ops()
    n, d, dt = len(lens), 64 * H, DT[mode]
    kv = rnd((n * Lmax, 2 * d), 31)
    qr, dx, w_out = rnd((n * Kr, d), 32, 1.5), rnd((n * Kr, d), 33), rnd((d, d), 34, d ** -0.5)
    kvd = kv.to(dev(), dt)
    kv_poison = kvd.clone()
    for c, Lc in enumerate(lens):
        kv_poison[c * Lmax + Lc:(c + 1) * Lmax] = float("nan")
    len_d = torch.tensor(lens, dtype=torch.int32, device=dev())
    w_t = w_out.t().contiguous().to(dev(), dt)
    prob = lambda kvt, dq: dict(q_rows=qr.to(dev(), dt), k=kvt[:, :d], v=kvt[:, d:], dx=dx.to(dev(), dt), w_out_t=w_t, dq=dq,
                                groups=n, H=H, keys=Lmax, Kp=Kr, scale=0.125, key_len=len_d, key_stride=Lmax)
    dq = torch.full((n * Kr, d), float("nan"), dtype=dt, device=dev())
    o.attn_bwd_proj_pair(prob(kv_poison, dq))
    assert bool(torch.isfinite(dq.float()).all()), "rows past len_c leaked into the result"
    da = torch.empty((n * Kr, d), dtype=dt, device=dev())
    o.gemm_nt(dx.to(dev(), dt), w_t, da, L.EPI_NONE)
    dq2 = torch.full((n * Kr, d), float("nan"), dtype=dt, device=dev())
    o.text_attn_bwd(qr.to(dev(), dt), kvd[:, :d], kvd[:, d:], da, dq2, len_d, n, Kr, Lmax, H)
    kv64, q64 = q(kv, mode), q(qr, mode)
    da64 = q((q(dx, mode) @ q(w_out, mode)).float(), mode)
    ref = torch.empty(n * Kr, d, dtype=torch.float64)
    for c, Lc in enumerate(lens):
        sl = slice(c * Kr, (c + 1) * Kr)
        ref[sl] = R.attn_rows_bwd(q64[sl], kv64[c * Lmax:c * Lmax + Lc, :d], kv64[c * Lmax:c * Lmax + Lc, d:], da64[sl], H)
    close(dq, ref, mode, f"text attn bwd + d out-proj H{H} K{Kr}", tol=3e-2)
    close(dq, dq2.double().cpu(), mode, "fused vs GEMM + text attn bwd", tol=3e-2)
    for _ in range(5):
        dq3 = torch.empty_like(dq)
        o.attn_bwd_proj_pair(prob(kvd, dq3))
        assert torch.equal(dq3, dq), "not deterministic / depends on the rows past len_c"


@experimental
@pytest.mark.parametrize("mode", ["bf16", "f16"])
def test_paired_launches_equal_the_two_separate_launches(mode):
    """rpo_gemm_nt_pair / rpo_layernorm_bwd_pair / rpo_attn_bwd_proj_pair: one launch for the same stage of the image
    tower's and the text tower's prompt-row chain.  Every problem must come out bit-identical to its own launch (shapes of
    the B = 32 step: 768 rows x 768 / 3072 and 456 rows x 512 / 2048)."""
    from rpo_amd import _lib as L
    from rpo_amd._lib import RPOLibraryError
    o = ops()
    dt = DT[mode]
    D = lambda t: t.to(dev(), dt)
    shapes = [(768, 768), (456, 512)]                      # (rows, d) of the two chains
    # -- GEMMs: d c_proj + QuickGELU' (16-bit out), split-K d c_fc and d q-proj (fp32 slabs) --------------------------
    for epi, split, kmul, nmul, odt in ((L.EPI_QGELU_BWD, 1, 1, 4, dt), (L.EPI_NONE, 3, 4, 1, torch.float32),
                                        (L.EPI_NONE, 2, 1, 1, torch.float32), (L.EPI_NONE, 1, 1, 1, torch.float32)):
        calls, singles = [], []
        for i, (M, d) in enumerate(shapes):
            K, N = d * kmul, d * nmul
            a, w = D(rnd((M, K), 40 + i)), D(rnd((N, K), 42 + i, K ** -0.5))
            aux = D(rnd((M, N), 44 + i)) if epi == L.EPI_QGELU_BWD else None
            mk = lambda: torch.full((split, M, N) if split > 1 else (M, N), float("nan"), dtype=odt, device=dev())
            pf = D(rnd((64, 64), 46))
            kw = dict(a=a, w=w, epilogue=epi, aux=aux, split_k=split, prefetch=pf)
            out_p, out_s = mk(), mk()
            calls.append(dict(out=out_p, **kw)); singles.append(dict(out=out_s, **kw))
        o.gemm_nt_pair(*calls)
        for c, s_ in zip(calls, singles):
            o.gemm_nt(**s_)
            assert torch.equal(c["out"], s_["out"]), f"paired GEMM epi {epi} split {split} differs from its own launch"
    with pytest.raises(RPOLibraryError):                    # different epilogues cannot share a launch
        bad = dict(calls[1]); bad["epilogue"] = L.EPI_QGELU_BWD; bad["aux"] = D(rnd((456, 512), 47))
        o.gemm_nt_pair(calls[0], bad)
    # -- LayerNorm backward ------------------------------------------------------------------------------------------
    calls, singles = [], []
    for i, (M, d) in enumerate(shapes):
        dy, x = rnd((3, M, d), 50 + i).to(dev()), rnd((M, d), 52 + i, 2.0).to(dev())
        g, dres = rnd((d,), 54 + i).to(dev()), rnd((M, d), 56 + i).to(dev())
        mk = lambda: dict(dx=torch.full((M, d), float("nan"), device=dev()),
                          dx_cast=torch.full((M, d), float("nan"), dtype=dt, device=dev()))
        kw = dict(dy=dy[:2 + i], x=x, gamma=g, dres=dres if i == 0 else None)
        calls.append(dict(**kw, **mk())); singles.append(dict(**kw, **mk()))
    o.layernorm_bwd_pair(*calls)
    for c, s_ in zip(calls, singles):
        o.layernorm_bwd(**s_)
        assert torch.equal(c["dx"], s_["dx"]) and torch.equal(c["dx_cast"], s_["dx_cast"]), "paired ln_bwd differs"
    # -- attention backward + d out-proj: image problem (197 keys, d 768) with text problem (per-class key counts, d 512)
    # (B = 32: 384 + 152 workgroups exceed the 512 resident ones, so the text problem's workgroups walk two items each)
    for B in (4, 32):
        _paired_attention_case(o, dt, D, B)


def _paired_attention_case(o, dt, D, B):
    N, Kp, n, Lmax = 197, 24, 19, 77
    lens = torch.tensor([(7 * c) % 70 + 5 for c in range(n)], dtype=torch.int32, device=dev())
    qkv = D(_img_rows(B, N, Kp, 768, 60))
    Rf = B * N
    img = lambda dq: dict(q_rows=qkv[Rf:, :768], k=qkv[:Rf, 768:1536], v=qkv[:Rf, 1536:], dx=D(rnd((B * Kp, 768), 61)),
                          w_out_t=D(rnd((768, 768), 62, 768 ** -0.5)), dq=dq, groups=B, H=12, keys=N, Kp=Kp, scale=0.125)
    kv, qt = D(rnd((n * Lmax, 1024), 63)), D(rnd((n * Kp, 512), 64))
    txt = lambda dq: dict(q_rows=qt, k=kv[:, :512], v=kv[:, 512:], dx=D(rnd((n * Kp, 512), 65)),
                          w_out_t=D(rnd((512, 512), 66, 512 ** -0.5)), dq=dq, groups=n, H=8, keys=Lmax, Kp=Kp, scale=0.125,
                          key_len=lens, key_stride=Lmax)
    mk = lambda r, d: torch.full((r, d), float("nan"), dtype=dt, device=dev())
    dqi, dqt, dqi2, dqt2 = mk(B * Kp, 768), mk(n * Kp, 512), mk(B * Kp, 768), mk(n * Kp, 512)
    o.attn_bwd_proj_pair(img(dqi), txt(dqt))
    o.attn_bwd_proj_pair(img(dqi2))
    o.attn_bwd_proj_pair(txt(dqt2))
    assert torch.equal(dqi, dqi2) and torch.equal(dqt, dqt2), "paired attention backward differs from the single launches"
    dqi3 = mk(B * Kp, 768)
    a = img(dqi3)
    o.attn_readonly_bwd_proj(a["q_rows"], a["k"], a["v"], a["dx"], a["w_out_t"], dqi3, B, 12, N, Kp)
    assert torch.equal(dqi, dqi3), "rpo_attn_bwd_proj_pair differs from rpo_attn_readonly_bwd_proj"


# (16-bit modes, <= 64 query rows, <= 96 keys: the one-wave MFMA kernel with 3 / 1 / 2 key tiles and 1 / 1 / 2 query tiles;
#  f32 and the causal pass: the VALU kernel)
@pytest.mark.parametrize("lens,Kr", [([3, 71, 20, 8, 10], 6), ([10, 10, 14, 11, 8, 1], 24), ([33, 40, 5], 48)])
@pytest.mark.parametrize("mode", ["f32", "bf16", "f16"])
def test_text_attn_fwd_bwd(mode, lens, Kr):
    o = ops()
    n, H, Lmax = len(lens), 2, max(lens)
    d = 64 * H
    kv = rnd((n * Lmax, 2 * d), 21)
    qr, da = rnd((n * Kr, d), 22, 1.5), rnd((n * Kr, d), 23)
    kvd = kv.to(dev(), DT[mode])
    len_d = torch.tensor(lens, dtype=torch.int32, device=dev())
    out = torch.full((n * Kr, d), float("nan"), dtype=DT[mode], device=dev())
    dq = torch.full((n * Kr, d), float("nan"), dtype=DT[mode], device=dev())
    o.text_attn_fwd(qr.to(dev(), DT[mode]), kvd[:, :d], kvd[:, d:], out, len_d, n, Kr, Lmax, H, causal=False)
    o.text_attn_bwd(qr.to(dev(), DT[mode]), kvd[:, :d], kvd[:, d:], da.to(dev(), DT[mode]), dq, len_d, n, Kr, Lmax, H)
    kv64, q64, da64 = q(kv, mode), q(qr, mode), q(da, mode)
    rf, rb = torch.empty(n * Kr, d, dtype=torch.float64), torch.empty(n * Kr, d, dtype=torch.float64)
    for c, L in enumerate(lens):
        k, v = kv64[c * Lmax:c * Lmax + L, :d], kv64[c * Lmax:c * Lmax + L, d:]
        sl = slice(c * Kr, (c + 1) * Kr)
        rf[sl] = R.attn_rows_fwd(q64[sl], k, v, H)
        rb[sl] = R.attn_rows_bwd(q64[sl], k, v, da64[sl], H)
    close(out, rf, mode, "text attn fwd", tol=None if mode == "f32" else 8e-3)
    close(dq, rb, mode, "text attn bwd", tol=None if mode == "f32" else 8e-3)
    # causal pass over the frozen tokens: packed [n*Lmax, 3d]
    qkv = rnd((n * Lmax, 3 * d), 24)
    t = qkv.to(dev(), DT[mode])
    outc = torch.full((n * Lmax, d), float("nan"), dtype=DT[mode], device=dev())
    o.text_attn_fwd(t[:, :d], t[:, d:2 * d], t[:, 2 * d:], outc, len_d, n, Lmax, Lmax, H, causal=True)
    q3 = q(qkv, mode)
    for c, L in enumerate(lens):
        sl = slice(c * Lmax, c * Lmax + L)
        ref = R.attn_causal_fwd(q3[sl, :d], q3[sl, d:2 * d], q3[sl, 2 * d:], H, torch.arange(1, L + 1))
        close(outc[sl], ref, mode, f"text causal attn class {c}", tol=None if mode == "f32" else 8e-3)


@pytest.mark.parametrize("B,C,K,e", [(4, 19, 24, 512), (32, 19, 24, 512), (100, 37, 16, 512), (5, 128, 8, 768),
                                     (3, 300, 4, 768), (1, 2, 1, 64), (7, 3, 5, 100),
                                     # the matrix-pipe path (C > 128) at the reference's ImageNet size and just past the fused path's 128
                                     (32, 1000, 24, 512), (2, 129, 1, 768),
                                     # ... whose kernels tile images and classes by 32: ragged tiles on both sides, odd pair counts
                                     (40, 397, 24, 512), (33, 131, 3, 64),
                                     # above 128 classes with a width the matrix-pipe kernels do not take: the three-launch path
                                     (3, 150, 2, 100)])
def test_head_fwd_bwd(B, C, K, e):
    o = ops()
    i_f, t_f = rnd((B, K, e), 31), rnd((C, K, e), 32)
    lab = torch.tensor([(7 * b + 1) % C for b in range(B)])
    lg, ls, di, dt = R.head_fwd_bwd(i_f.double(), t_f.double(), lab, 100.0)
    logits = torch.empty(B, C, device=dev())
    loss = torch.empty(1, device=dev())
    d_i, d_t = torch.empty(B, K, e, device=dev()), torch.empty(C, K, e, device=dev())
    ws = torch.empty(o.head_workspace_floats(B, C, K, e), device=dev())
    o.head_fwd_bwd(i_f.to(dev()), t_f.to(dev()), lab.to(dev()), 100.0, logits, loss, d_i, d_t, ws)
    close(logits, lg, "f32", "head logits", tol=5e-6)
    assert abs(loss.item() - ls.item()) <= 5e-6 * max(1.0, abs(ls.item()))
    close(d_i, di, "f32", "head d_img_f", tol=2e-5)
    close(d_t, dt, "f32", "head d_text_f", tol=2e-5)
    for adt in (torch.bfloat16, torch.float16):     # act-dtype copies for the projections' dX GEMMs: RNE of the fp32 result
        ia, ta = torch.empty(B, K, e, dtype=adt, device=dev()), torch.empty(C, K, e, dtype=adt, device=dev())
        o.head_fwd_bwd(i_f.to(dev()), t_f.to(dev()), lab.to(dev()), 100.0, logits, loss, d_i, d_t, ws, d_img_f_act=ia, d_text_f_act=ta)
        assert torch.equal(ia, d_i.to(adt)) and torch.equal(ta, d_t.to(adt))
    logits.zero_()
    o.head_fwd_bwd(i_f.to(dev()), t_f.to(dev()), None, 100.0, logits, None, None, None, ws)
    close(logits, lg, "f32", "head logits (eval)", tol=5e-6)
    # an out-of-range target: F.cross_entropy raises; the kernel reports NaN and touches nothing out of bounds
    bad = lab.clone(); bad[B // 2] = C
    o.head_fwd_bwd(i_f.to(dev()), t_f.to(dev()), bad.to(dev()), 100.0, logits, loss, d_i, d_t, ws)
    assert torch.isnan(loss).item()
    assert torch.isnan(d_i).any() and torch.isnan(d_t).any()        # gradients poisoned too, not only the loss
    close(logits, lg, "f32", "head logits (bad label)", tol=5e-6)


@pytest.mark.parametrize("mode", ["bf16", "f16"])
def test_gemm_prefetch_hint_changes_nothing_but_time(mode):
    """rpo_gemm_args.prefetch (the next launch's weights, touched into the memory-side cache by every workgroup before
    its k-loop) is a hint: every kernel that implements it -- the three one-round kernels at the bench's shapes and the
    generic tiles, plain and split-K -- must return the bits it returns without it, for hint ranges that are larger than,
    equal to and smaller than one line per thread, and ranges whose size is not a multiple of anything."""
    from rpo_amd import _lib as L
    o = ops()
    N0, K0, units = 197, 24, 32
    M = units * (N0 + K0)
    ru = (N0, K0, units * N0)
    hints = [torch.randn(n, device=dev()).to(DT[mode]) for n in (768 * 3072, 96 * 64 + 13, 257)]
    cases = [  # (M, N, K, epilogue, extra kwargs)
        (M, 2304, 768, L.EPI_BIAS, {}),                                              # gemm_w4
        (M, 3072, 768, L.EPI_BIAS_QGELU, dict(row_units=ru)),                        # gemm_w4g
        (M, 768, 3072, L.EPI_BIAS_RESID, dict(row_units=ru)),                        # gemm_w4k
        (768, 3072, 768, L.EPI_BIAS, {}),                                            # generic 64x64
        (768, 768, 3072, L.EPI_NONE, dict(split_k=3)),                               # generic, split-K (2-D grid)
    ]
    for M_, N_, K_, epi, kw in cases:
        a, w = rnd((M_, K_), 1, 0.5).to(dev(), DT[mode]), rnd((N_, K_), 2, K_ ** -0.5).to(dev(), DT[mode])
        kw = dict(kw)
        if epi != L.EPI_NONE:
            kw["bias"] = rnd((N_,), 3).to(dev())
        if epi == L.EPI_BIAS_RESID:
            kw["resid"] = rnd((M_, N_), 4).to(dev())
        odt = torch.float32 if epi in (L.EPI_BIAS_RESID, L.EPI_NONE) else DT[mode]
        shape = (kw["split_k"], M_, N_) if "split_k" in kw else (M_, N_)
        base = torch.full(shape, float("nan"), dtype=odt, device=dev())
        o.gemm_nt(a, w, base, epi, **kw)
        assert torch.isfinite(base.float()).all()
        for h in hints:
            out = torch.full(shape, float("nan"), dtype=odt, device=dev())
            o.gemm_nt(a, w, out, epi, prefetch=h, **kw)
            assert torch.equal(out, base), f"prefetch hint changed the result of {M_}x{N_}x{K_} epilogue {epi}"


@pytest.mark.parametrize("mode", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("B,N,Kp,d", [(3, 50, 7, 768), (2, 197, 24, 768), (1, 17, 1, 1024), (2, 10, 0, 512)])
def test_img_embed_norm_equals_assemble_plus_two_layernorms(mode, B, N, Kp, d):
    """rpo_img_embed_norm: CLS / prompt rows + ln_pre + the first ln_1 in one launch -- the bits of rpo_img_assemble ->
    rpo_layernorm_fwd -> rpo_layernorm_fwd, and float64-close (trainers/rpo.py:201-206, clip/model.py:189)."""
    o = ops()
    R_ = B * (N + Kp)
    x_pre = rnd((R_, d), 1, 2.0)
    cls, pos0 = rnd((d,), 2), rnd((d,), 3)
    prompt = rnd((max(Kp, 1), d), 4)
    gp, bp, g1, b1 = rnd((d,), 5, 0.1) + 1.0, rnd((d,), 6, 0.1), rnd((d,), 7, 0.1) + 1.0, rnd((d,), 8, 0.1)
    dv = lambda t: t.to(dev())
    # the three launches
    xa = dv(x_pre).clone()
    o.img_assemble(xa, dv(cls), dv(pos0), dv(prompt), B, N, Kp)
    x0a = torch.empty(R_, d, device=dev()); ha = torch.empty(R_, d, dtype=DT[mode], device=dev())
    o.layernorm_fwd(xa, dv(gp), dv(bp), x0a)
    o.layernorm_fwd(x0a, dv(g1), dv(b1), ha)
    # one launch
    xb = dv(x_pre).clone()
    x0b = torch.full((R_, d), float("nan"), device=dev()); hb = torch.full((R_, d), float("nan"), dtype=DT[mode], device=dev())
    o.img_embed_norm(xb, dv(cls), dv(pos0), dv(prompt) if Kp else None, dv(gp), dv(bp), x0b, dv(g1), dv(b1), hb, B, N, Kp)
    assert torch.equal(xb, xa), "x_pre rows (CLS, prompts) differ from rpo_img_assemble"
    assert torch.equal(x0b, x0a) and torch.equal(hb, ha), "fused launch differs from the three separate ones"
    # float64
    tok = x_pre.double().clone()
    for b in range(B):
        tok[b * N] = cls.double() + pos0.double()
        if Kp:
            tok[B * N + b * Kp:B * N + (b + 1) * Kp] = prompt.double()
    ln = lambda t, g, b_: (t - t.mean(1, keepdim=True)) / (t.var(1, unbiased=False, keepdim=True) + 1e-5).sqrt() * g.double() + b_.double()
    y1 = ln(tok, gp, bp)
    close(x0b, y1, "f32", "ln_pre of the assembled tokens", tol=2e-5)
    close(hb, ln(y1, g1, b1), mode, "ln_1 of that")


def test_sgd_broadcast_reduce_convert():
    o = ops()
    n = 30720
    p, g = rnd((n,), 41), rnd((n,), 42)
    pd, gd, buf = p.to(dev()), g.to(dev()), torch.zeros(n, device=dev())
    rp, rb = p.double(), None
    for step in range(3):
        o.sgd_step(pd, gd, buf, 0.01, 0.9, 5e-4, 0.5, first_step=(step == 0))
        rp, rb = R.sgd(rp, g.double(), rb, 0.01, 0.9, 5e-4, first=(step == 0), grad_scale=0.5)
    close(pd, rp, "f32", "sgd", tol=1e-6)
    src = rnd((24, 512), 43)
    dst = torch.empty(19 * 24, 512, device=dev())
    o.broadcast_rows(src.to(dev()), dst, 19)
    assert torch.equal(dst.cpu(), src.repeat(19, 1))
    big = rnd((32 * 24, 768), 44)
    out = torch.empty(24, 768, device=dev())
    o.reduce_groups(big.to(dev()), out, 32)
    acc = torch.zeros(24, 768)
    for gidx in range(32):                      # same fixed order as the kernel: bit-exact
        acc += big[gidx * 24:(gidx + 1) * 24]
    assert torch.equal(out.cpu(), acc)
    # >= 64 groups (the text gradient over hundreds of classes): 8 strided partial sums per element, added in thread-row
    # order -- the same fixed order here, bit-exact; odd widths and group counts that do not divide by 8
    for groups, rows, d in ((1000, 24, 512), (397, 5, 100), (64, 3, 33)):
        big = rnd((groups * rows, d), 45 + groups)
        out = torch.empty(rows, d, device=dev())
        o.reduce_groups(big.to(dev()), out, groups)
        parts = []
        for j in range(8):
            a = torch.zeros(rows, d)
            for gidx in range(j, groups, 8):
                a += big[gidx * rows:(gidx + 1) * rows]
            parts.append(a)
        acc = parts[0].clone()
        for j in range(1, 8):
            acc += parts[j]
        assert torch.equal(out.cpu(), acc), (groups, rows, d)
    big = rnd((32 * 24, 768), 44)
    c = torch.empty(24, 768, dtype=torch.bfloat16, device=dev())
    o.convert(big[:24].to(dev()), c)
    assert torch.equal(c.cpu(), big[:24].to(torch.bfloat16))


def test_sgd_step_guarded_skips_nonfinite_gradients():
    """rpo_sgd_step_guarded = torch.cuda.amp.GradScaler.step on unscaled fp32 gradients (trainers/rpo.py:298-304): the
    whole step is skipped when a gradient is Inf / NaN; otherwise the bits of rpo_sgd_step."""
    o = ops()
    n = 30720
    p, g = rnd((n,), 41), rnd((n,), 42)
    found = torch.zeros(2, dtype=torch.int32, device=dev())
    pa, ba = p.to(dev()), torch.zeros(n, device=dev())
    pb, bb = p.to(dev()), torch.zeros(n, device=dev())
    gd = g.to(dev())
    for step in range(3):
        o.sgd_step(pa, gd, ba, 0.01, 0.9, 5e-4, 0.5, first_step=(step == 0))
        o.sgd_step_guarded(pb, gd, bb, 0.01, 0.9, 5e-4, 0.5, first_step=(step == 0), found_inf=found)
    assert torch.equal(pa, pb) and torch.equal(ba, bb) and found.tolist() == [0, 0]
    for poison in (float("inf"), float("nan"), -float("inf")):
        gbad = gd.clone(); gbad[n - 7] = poison
        before_p, before_b = pb.clone(), bb.clone()
        o.sgd_step_guarded(pb, gbad, bb, 0.01, 0.9, 5e-4, 0.5, first_step=False, found_inf=found)
        assert torch.equal(pb, before_p) and torch.equal(bb, before_b) and found[0].item() == 1
    assert found[1].item() == 3
    o.sgd_step(pa, gd, ba, 0.01, 0.9, 5e-4, 0.5, first_step=False)
    o.sgd_step_guarded(pb, gd, bb, 0.01, 0.9, 5e-4, 0.5, first_step=False, found_inf=found)
    assert torch.equal(pa, pb) and found.tolist() == [0, 3]
    # a skipped FIRST step: the next one starts from the zero momentum buffer, as torch's first step does
    pc, bc = p.to(dev()), torch.zeros(n, device=dev())
    gbad = gd.clone(); gbad[0] = float("nan")
    o.sgd_step_guarded(pc, gbad, bc, 0.01, 0.9, 5e-4, 1.0, first_step=True, found_inf=found)
    o.sgd_step_guarded(pc, gd, bc, 0.01, 0.9, 5e-4, 1.0, first_step=False, found_inf=found)
    pd, bd = p.to(dev()), torch.zeros(n, device=dev())
    o.sgd_step(pd, gd, bd, 0.01, 0.9, 5e-4, 1.0, first_step=True)
    assert torch.equal(pc, pd) and torch.equal(bc, bd)


def test_argument_errors_are_reported():
    from rpo_amd._lib import RPOLibraryError
    o = ops()
    a = torch.zeros(8, 48, device=dev())
    with pytest.raises(RPOLibraryError):
        o.gemm_nt(a, torch.zeros(16, 48, device=dev()), torch.zeros(8, 16, device=dev()))   # K % 32 != 0
    with pytest.raises(RPOLibraryError):
        o.layernorm_fwd(torch.zeros(4, 30, device=dev()), torch.zeros(30, device=dev()),
                        torch.zeros(30, device=dev()), torch.zeros(4, 30, device=dev()))


@pytest.mark.gpu
def test_peak_probes_are_sane():
    """The empirical-peak probes bench.py uses as second roofline denominators: above what our own GEMM
    reaches, below the datasheet peaks (2.5 PFLOP/s bf16, 157 TFLOP/s f32, 8 TB/s)."""
    o = ops()
    bf = o.probe_peaks(dev(), 0)
    assert 800.0 < bf["mfma_tflops"] < 2600.0, bf
    assert 1500.0 < bf["copy_gbs"] < 8100.0, bf
    f32 = o.probe_peaks(dev(), 1)
    assert 80.0 < f32["mfma_tflops"] < 165.0, f32


# ------------------------------------------------------------------------------------------
# rpo_gemm_ws: the prompt-row GEMMs with the weight streamed in fragment order (csrc/gemm_ws.hip)
@pytest.mark.parametrize("mode", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K,cfg", [(768, 3072, 768, 0), (768, 768, 768, 0), (456, 512, 512, 0), (456, 2048, 512, 0),
                                       (456, 512, 2048, 0), (96, 768, 768, 0), (200, 160, 64, 110), (200, 160, 128, 120),
                                       (200, 160, 192, 220), (200, 160, 320, 330), (97, 96, 1024, 330), (33, 64, 256, 220)])
def test_gemm_ws_epilogues(mode, M, N, K, cfg):
    """Every epilogue of rpo_gemm_ws against float64 on the values the kernel sees, chain shapes of both towers and
    small ragged ones (row tails, partial n-tiles, waves without a k-chunk), every tile geometry; and against
    rpo_gemm_nt on the row-major weight (same inputs: the two differ only in summation order)."""
    from rpo_amd import _lib as L
    o = ops()
    dt = DT[mode]
    a, w = rnd((M, K), 1, 0.5), rnd((N, K), 2, K ** -0.5)
    bias, resid, u = rnd((N,), 3), rnd((M, N), 4), rnd((M, N), 5)
    acc = q(a, mode) @ q(w, mode).t()
    ad, wd = a.to(dev(), dt), w.to(dev(), dt)
    wp = o.gemm_ws_pack(wd)
    bd, rd, ud = bias.to(dev()), resid.to(dev()), u.to(dev())
    f32tol = 1e-4
    kw = dict(tile_config=cfg)
    out = torch.full((M, N), float("nan"), dtype=dt, device=dev())
    close(o.gemm_ws(ad, wp, out, L.EPI_NONE, **kw), acc, mode, "ws none")
    close(o.gemm_ws(ad, wp, out, L.EPI_BIAS, bias=bd, **kw), acc + bias.double(), mode, "ws bias")
    of = torch.full((M, N), float("nan"), device=dev())
    close(o.gemm_ws(ad, wp, of, L.EPI_NONE, **kw), acc, mode, "ws none f32-out", tol=f32tol)
    ref_nt = torch.empty(M, N, device=dev())
    o.gemm_nt(ad, wd, ref_nt, L.EPI_NONE)
    close(of, ref_nt.double().cpu(), "f32", "ws vs gemm_nt", tol=2e-5)
    close(o.gemm_ws(ad, wp, of, L.EPI_BIAS_RESID, bias=bd, resid=rd, **kw), acc + bias.double() + resid.double(), mode,
          "ws resid", tol=f32tol)
    # split-K slabs (fp32), summed in order by the consumer
    for S in (2, 4):
        if K // 64 >= S:
            slabs = torch.full((S, M + 3, N), float("nan"), device=dev())[:, :M]
            o.gemm_ws(ad, wp, slabs, L.EPI_NONE, split_k=S, **kw)
            close(slabs.sum(0), acc, mode, f"ws split-K {S}", tol=f32tol)
    # QuickGELU forward with the saved operand of the bottom rows (both forms), and its backward (both forms)
    row0 = M // 3
    pre = acc + bias.double()
    aux = torch.full((M - row0, N), float("nan"), device=dev())
    close(o.gemm_ws(ad, wp, out, L.EPI_BIAS_QGELU, bias=bd, aux=aux, aux_row0=row0, **kw), R.qgelu(pre), mode, "ws qgelu")
    close(aux, pre[row0:], mode, "ws qgelu saved u", tol=f32tol)
    aux16 = torch.full((M - row0, N), float("nan"), dtype=dt, device=dev())
    close(o.gemm_ws(ad, wp, out, L.EPI_BIAS_QGELU, bias=bd, aux=aux16, aux_row0=row0, **kw), R.qgelu(pre), mode, "ws qgelu (aux16)")
    close(aux16, R.qgelu_grad(pre[row0:]), mode, "ws qgelu saved derivative")
    close(o.gemm_ws(ad, wp, out, L.EPI_QGELU_BWD, aux=ud, **kw), acc * R.qgelu_grad(u.double()), mode, "ws qgelu bwd")
    d16 = R.qgelu_grad(u.double()).float().to(dev(), dt)
    close(o.gemm_ws(ad, wp, out, L.EPI_QGELU_BWD, aux=d16, **kw), acc * d16.double().cpu(), mode, "ws qgelu bwd (aux16)")
    # deterministic: the same launch twice gives the same bits
    out2 = torch.empty_like(out)
    o.gemm_ws(ad, wp, out2, L.EPI_QGELU_BWD, aux=d16, **kw)
    assert torch.equal(out, out2)
    # the prefetch hint changes nothing
    o.gemm_ws(ad, wp, out2, L.EPI_QGELU_BWD, aux=d16, prefetch=wp, **kw)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("mode", ["bf16", "f16"])
@pytest.mark.parametrize("M,d,N", [(456, 512, 2048), (768, 768, 3072), (100, 1024, 256)])
def test_gemm_ws_layernorm_fold(mode, M, d, N):
    """The LayerNorm fold on rpo_gemm_ws, both sides: BIAS_RESID leaves out2 and 64-column partial statistics, the
    LN_BIAS / LN_BIAS_QGELU consumer reproduces quickgelu(LN(x) W^T + b) (clip/model.py:156-159 feeding :174-175, :186);
    producer and consumer are interchangeable with rpo_gemm_nt's (same statistics layout)."""
    from rpo_amd import _lib as L
    o = ops()
    dt = DT[mode]
    att, w_out, b_out = rnd((M, d), 1, 0.5), rnd((d, d), 2, d ** -0.5), rnd((d,), 3)
    resid = rnd((M, d), 4, 2.0) + 0.4
    w, b = rnd((N, d), 5, d ** -0.5), rnd((N,), 6)
    gamma, beta = rnd((d,), 7, 0.1) + 1.0, rnd((d,), 8, 0.05)
    xm = torch.full((M, d), float("nan"), device=dev())
    xb = torch.full((M, d), float("nan"), dtype=dt, device=dev())
    stats = torch.full((M, d // 64, 2), float("nan"), device=dev())
    wo_d = w_out.to(dev(), dt)
    o.gemm_ws(att.to(dev(), dt), o.gemm_ws_pack(wo_d), xm, L.EPI_BIAS_RESID, bias=b_out.to(dev()), resid=resid.to(dev()),
              out2=xb, ln_stats=stats)
    xm64 = xm.double().cpu()
    close(xm, q(att, mode) @ q(w_out, mode).t() + b_out.double() + resid.double(), "f32", "ws producer result", tol=1e-4)
    assert torch.equal(xb.cpu(), xm.cpu().to(dt)), "out2 must be the RNE act-dtype copy of C"
    grp = xm64.reshape(M, d // 64, 64)
    ref_stats = torch.stack([grp.mean(-1), ((grp - grp.mean(-1, keepdim=True)) ** 2).sum(-1)], -1)
    close(stats, ref_stats, "f32", "ws partial row statistics", tol=2e-5)
    wq = (w.double() * gamma.double()[None, :]).float().to(dt)
    s = wq.double().sum(1).float()
    bq = (b.double() + w.double() @ beta.double()).float()
    mu = xm64.mean(1, keepdim=True)
    rstd = (xm64.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    pre = ((xm64 - mu) * rstd * gamma.double() + beta.double()) @ w.double().t() + b.double()
    row0 = M // 2
    wqp = o.gemm_ws_pack(wq.to(dev()))
    for cfg in (0, 110, 220, 330):
        y = torch.full((M, N), float("nan"), dtype=dt, device=dev())
        aux16 = torch.full((M - row0, N), float("nan"), dtype=dt, device=dev())
        o.gemm_ws(xb, wqp, y, L.EPI_LN_BIAS_QGELU, bias=bq.to(dev()), aux=aux16, aux_row0=row0, ln_stats=stats,
                  ln_colsum=s.to(dev()), tile_config=cfg)
        close(y, R.qgelu(pre), mode, f"ws LN-folded c_fc cfg {cfg}", tol=1.5 * TOL[mode])
        close(aux16, R.qgelu_grad(pre[row0:]), mode, f"ws LN-folded saved derivative cfg {cfg}", tol=1.5 * TOL[mode])
        o.gemm_ws(xb, wqp, y, L.EPI_LN_BIAS, bias=bq.to(dev()), ln_stats=stats, ln_colsum=s.to(dev()), tile_config=cfg)
        close(y, pre, mode, f"ws LN-folded in-proj cfg {cfg}", tol=1.5 * TOL[mode])
    # statistics written by rpo_gemm_nt's producer feed the ws consumer the same way
    stats2 = torch.full_like(stats, float("nan"))
    xb2 = torch.empty_like(xb)
    xm2 = torch.empty_like(xm)
    o.gemm_nt(att.to(dev(), dt), wo_d, xm2, L.EPI_BIAS_RESID, bias=b_out.to(dev()), resid=resid.to(dev()), out2=xb2, ln_stats=stats2)
    close(stats2, ref_stats, "f32", "nt partial statistics vs ws result", tol=1e-4)


def test_gemm_ws_refuses_what_it_does_not_cover():
    from rpo_amd import _lib as L
    o = ops()
    assert o.gemm_ws_ok(768, 3072, 768, torch.bfloat16, torch.bfloat16, L.EPI_QGELU_BWD)
    assert o.gemm_ws_ok(456, 512, 2048, torch.float16, torch.float32, L.EPI_NONE, split_k=4)
    assert not o.gemm_ws_ok(768, 3072, 768, torch.float32, torch.float32, L.EPI_NONE)          # f32 mode: rpo_gemm_nt
    assert not o.gemm_ws_ok(768, 3080, 768, torch.bfloat16, torch.bfloat16, L.EPI_NONE)        # N % 32
    assert not o.gemm_ws_ok(768, 3072, 800, torch.bfloat16, torch.bfloat16, L.EPI_NONE)        # K % 64
    assert not o.gemm_ws_ok(768, 768, 768, torch.bfloat16, torch.bfloat16, L.EPI_NONE, split_k=2)   # slabs are fp32
    assert not o.gemm_ws_ok(768, 768, 3072, torch.bfloat16, torch.bfloat16, L.EPI_LN_BIAS)     # 48 statistics groups
    a = torch.zeros(64, 64, dtype=torch.bfloat16, device=dev())
    wp = o.gemm_ws_pack(torch.zeros(64, 64, dtype=torch.bfloat16, device=dev()))
    with pytest.raises(Exception):
        o.gemm_ws(a, wp, torch.zeros(64, 64, dtype=torch.bfloat16, device=dev()), L.EPI_NONE, skip_row0=0, skip_col0=0)
