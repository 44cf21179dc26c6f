"""Thin torch-tensor front-end of the C ABI (include/rpo_amd.h).

PyTorch is plumbing here: it owns device memory and streams; every function
below only passes ``data_ptr()``s, sizes and the current HIP stream to
librpo_hip.so.  No computation happens in torch, and there is no fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import (EPI_BIAS, EPI_BIAS_QGELU, EPI_BIAS_RESID, EPI_NONE, EPI_PATCH, EPI_QGELU_BWD, RPO_BF16,
                   RPO_F16, RPO_F32, GemmArgs, check)

LN_EPS = 1e-5


def dtype_code(t: torch.dtype) -> int:
    if t == torch.float32:
        return RPO_F32
    if t == torch.bfloat16:
        return RPO_BF16
    if t == torch.float16:
        return RPO_F16
    raise TypeError(f"unsupported dtype {t}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _ld(t: torch.Tensor) -> int:
    assert t.dim() == 2 and t.stride(1) == 1, "row-major 2-D view required"
    return t.stride(0)


def version() -> int:
    return _lib.load().rpo_version()


def gemm_args(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, epilogue: int = EPI_NONE,
            bias: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None,
            aux: Optional[torch.Tensor] = None, aux_row0: int = 0, skip_row0: int = -1, skip_col0: int = -1,
            group: int = 0, m_rows: Optional[int] = None, split_k: int = 1, tile_config: int = 0,
            out2: Optional[torch.Tensor] = None, ln_stats: Optional[torch.Tensor] = None,
            ln_colsum: Optional[torch.Tensor] = None, ln_eps: float = LN_EPS,
            row_units: Optional[tuple] = None, ln_group: int = 0,
            prefetch: Optional[torch.Tensor] = None, resid_hi: Optional[torch.Tensor] = None,
            resid_lo: Optional[torch.Tensor] = None, out_lo: Optional[torch.Tensor] = None, c_row0: int = 0) -> GemmArgs:
    """The rpo_gemm_args of out = a @ w.T with a fused epilogue.  For EPI_PATCH ``out`` is the token matrix
    (more rows than ``a``); ``m_rows`` overrides M otherwise taken from ``a``.  With
    ``split_k`` = S > 1, ``out`` is [S, M, N] fp32 slabs to be summed by the consumer.
    LayerNorm fold: a BIAS_RESID call may also leave ``out2`` (act-dtype copy of its result) and ``ln_stats``
    ([M, N/64, 2] partial row statistics); an LN_BIAS / LN_BIAS_QGELU call reads such ``ln_stats`` for its A rows
    together with ``ln_colsum`` (see include/rpo_amd.h)."""
    M = a.shape[0] if m_rows is None else m_rows
    N, K = w.shape
    assert a.shape[1] == K and a.dtype == w.dtype
    split_stride = 0
    if split_k > 1:
        assert out.dim() == 3 and out.shape[0] == split_k and out.stride(2) == 1
        split_stride, ldc = out.stride(0), out.stride(1)
    else:
        ldc = _ld(out)
    args = GemmArgs(A=a.data_ptr(), lda=_ld(a), W=w.data_ptr(), ldw=_ld(w), C=out.data_ptr(), ldc=ldc,
                    M=M, N=N, K=K, in_dtype=dtype_code(a.dtype), out_dtype=dtype_code(out.dtype),
                    epilogue=epilogue, bias=_p(bias), resid=_p(resid), ldr=0 if resid is None else _ld(resid),
                    aux=_p(aux), ldaux=0 if aux is None else _ld(aux), aux_row0=aux_row0,
                    skip_row0=skip_row0, skip_col0=skip_col0, group=group, split_k=split_k,
                    split_stride=split_stride, tile_config=tile_config,
                    out2=_p(out2), ldout2=0 if out2 is None else _ld(out2), ln_stats=_p(ln_stats),
                    ln_colsum=_p(ln_colsum), ln_eps=ln_eps)
    args.ln_group = ln_group
    if resid_hi is not None:                # residual stream as 16-bit hi / lo halves (include/rpo_amd.h)
        assert resid_hi.dtype == a.dtype and (resid_lo is None or (resid_lo.dtype == a.dtype and _ld(resid_hi) == _ld(resid_lo)))
        args.resid_hi, args.resid_lo, args.ldr16 = resid_hi.data_ptr(), _p(resid_lo), _ld(resid_hi)
    if out_lo is not None:
        assert out2 is not None and out_lo.dtype == out2.dtype and _ld(out_lo) == _ld(out2)
        args.out_lo = out_lo.data_ptr()
    args.c_row0 = c_row0
    if prefetch is not None:                # hint: what the next launch reads first (include/rpo_amd.h)
        assert prefetch.is_contiguous()
        args.prefetch, args.prefetch_bytes = prefetch.data_ptr(), prefetch.numel() * prefetch.element_size()
    if aux is not None and aux.dtype != torch.float32:      # 16-bit aux: holds d quickgelu / du (include/rpo_amd.h)
        args.aux_dtype = dtype_code(aux.dtype)
    if row_units is not None:               # (rows per unit in segment 0, rows per unit in segment 1, first row of segment 1)
        args.seg_rows0, args.seg_rows1, args.seg1_row0 = row_units
    if ln_stats is not None:
        rows = a.shape[0] if epilogue in (_lib.EPI_LN_BIAS, _lib.EPI_LN_BIAS_QGELU) else M
        cols = K if epilogue in (_lib.EPI_LN_BIAS, _lib.EPI_LN_BIAS_QGELU) else N
        assert ln_stats.dtype == torch.float32 and ln_stats.is_contiguous() and ln_stats.numel() >= rows * (cols // (ln_group or 64)) * 2
    return args


def gemm_nt(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, epilogue: int = EPI_NONE, **kw) -> torch.Tensor:
    """out = a @ w.T with a fused epilogue (rpo_gemm_nt); keyword arguments as `gemm_args`."""
    args = gemm_args(a, w, out, epilogue, **kw)
    check(_lib.load().rpo_gemm_nt(C.byref(args), _stream()), "rpo_gemm_nt")
    return out


class PackedWeight:
    """A frozen [N, K] weight in rpo_gemm_ws's fragment-major order (rpo_gemm_ws_pack): what `gemm_ws` takes as W."""
    __slots__ = ("data", "N", "K", "dtype")

    def __init__(self, data: torch.Tensor, N: int, K: int):
        self.data, self.N, self.K, self.dtype = data, N, K, data.dtype

    # the prefetch hint names its bytes (ops.gemm_args: prefetch=...)
    def is_contiguous(self) -> bool:
        return True

    def data_ptr(self) -> int:
        return self.data.data_ptr()

    def numel(self) -> int:
        return self.data.numel()

    def element_size(self) -> int:
        return self.data.element_size()


def gemm_ws_pack(w: torch.Tensor) -> PackedWeight:
    """Fragment-major copy of a row-major 16-bit [N, K] weight (N % 32 == 0, K % 64 == 0)."""
    N, K = w.shape
    out = torch.empty(N * K, dtype=w.dtype, device=w.device)
    check(_lib.load().rpo_gemm_ws_pack(w.data_ptr(), _ld(w), out.data_ptr(), N, K, dtype_code(w.dtype), _stream()),
          "rpo_gemm_ws_pack")
    return PackedWeight(out, N, K)


class _WView:
    """stand-in for the weight tensor in gemm_args (shape / dtype / pointer of the packed copy)"""
    def __init__(self, pw: PackedWeight):
        self.shape, self.dtype, self._pw = (pw.N, pw.K), pw.dtype, pw

    def data_ptr(self) -> int:
        return self._pw.data_ptr()

    def dim(self) -> int:
        return 2

    def stride(self, i: int) -> int:
        return (self._pw.K, 1)[i]


def gemm_ws(a: torch.Tensor, w: PackedWeight, out: torch.Tensor, epilogue: int = EPI_NONE, **kw) -> torch.Tensor:
    """out = a @ W.T with a fused epilogue for the prompt rows (rpo_gemm_ws): `w` is the packed copy of W; keyword
    arguments as `gemm_args`."""
    args = gemm_args(a, _WView(w), out, epilogue, **kw)
    check(_lib.load().rpo_gemm_ws(C.byref(args), _stream()), "rpo_gemm_ws")
    return out


def gemm_ws_try(a: torch.Tensor, w: PackedWeight, out: torch.Tensor, epilogue: int = EPI_NONE, **kw) -> bool:
    """`gemm_ws`, but a problem the kernel does not take (RPO_E_SHAPE: nothing was enqueued) returns False instead of
    raising, so that the caller can issue rpo_gemm_nt on the row-major weight; every other error still raises."""
    args = gemm_args(a, _WView(w), out, epilogue, **kw)
    rc = int(_lib.load().rpo_gemm_ws(C.byref(args), _stream()))
    if rc == _lib.E_SHAPE:
        return False
    check(rc, "rpo_gemm_ws")
    return True


def gemm_ws_ok(M: int, N: int, K: int, dtype: torch.dtype, out_dtype: torch.dtype, epilogue: int, split_k: int = 1,
               stats: bool = False) -> bool:
    """Does rpo_gemm_ws take this problem (rpo_gemm_ws_ok)?"""
    if dtype == torch.float32:
        return False
    args = GemmArgs(M=M, N=N, K=K, lda=K, ldw=K, ldc=N, in_dtype=dtype_code(dtype), out_dtype=dtype_code(out_dtype),
                    epilogue=epilogue, split_k=split_k, split_stride=M * N, skip_row0=-1, skip_col0=-1)
    if stats:
        args.ln_stats = 4096
    return int(_lib.load().rpo_gemm_ws_ok(C.byref(args))) == 1


def gemm_nt_pair(g0: dict, g1: dict) -> None:
    """Two small-M GEMMs of the same kind in ONE launch (rpo_gemm_nt_pair): g0 / g1 are the keyword arguments of
    `gemm_nt` (a, w, out, epilogue, ...) of the two problems."""
    a0, a1 = gemm_args(**g0), gemm_args(**g1)
    check(_lib.experimental().rpo_gemm_nt_pair(C.byref(a0), C.byref(a1), _stream()), "rpo_gemm_nt_pair")


def mlp_fused(fc: dict, proj: dict, counters: torch.Tensor, safe: bool = False) -> bool:
    """c_fc -> c_proj in ONE launch (rpo_mlp_fused): fc / proj are the keyword arguments of the two `gemm_nt` calls it
    stands for (a, w, out, epilogue, ...).  Returns False where the kernel does not apply (RPO_E_SHAPE): the caller then
    issues the two calls."""
    a0, a1 = gemm_args(**fc), gemm_args(**proj)
    assert counters.dtype == torch.int32 and counters.is_contiguous()
    rc = _lib.experimental().rpo_mlp_fused(C.byref(a0), C.byref(a1), counters.data_ptr(), int(safe), _stream())
    if rc == -2:                                     # RPO_E_SHAPE
        return False
    check(rc, "rpo_mlp_fused")
    return True


def gemm_stats_group(M: int, N: int, K: int, dtype: torch.dtype, row_units: Optional[tuple]) -> int:
    """Columns per partial row statistic a BIAS_RESID producer [M, K] x [N, K]^T -> fp32 writes (rpo_gemm_stats_group)."""
    if dtype == torch.float32:
        return 64
    args = GemmArgs(M=M, N=N, K=K, lda=K, ldw=K, ldc=N, in_dtype=dtype_code(dtype), out_dtype=_lib.RPO_F32,
                    epilogue=_lib.EPI_BIAS_RESID, split_k=1)
    if row_units is not None:
        args.seg_rows0, args.seg_rows1, args.seg1_row0 = row_units
    g = int(_lib.load().rpo_gemm_stats_group(C.byref(args)))
    if g < 0:
        check(g, "rpo_gemm_stats_group")
    return g


def gemm_hilo_ok(M: int, N: int, K: int, dtype: torch.dtype, row_units: Optional[tuple], ln_group: int = 0) -> bool:
    """Would a BIAS_RESID GEMM of these shapes run on a kernel that implements the hi / lo residual stream
    (rpo_gemm_hilo_ok)?"""
    if dtype == torch.float32:
        return False
    args = GemmArgs(M=M, N=N, K=K, lda=K, ldw=K, ldc=N, in_dtype=dtype_code(dtype), out_dtype=_lib.RPO_F32,
                    epilogue=_lib.EPI_BIAS_RESID, split_k=1)
    args.ln_group = ln_group
    if row_units is not None:
        args.seg_rows0, args.seg_rows1, args.seg1_row0 = row_units
    return int(_lib.load().rpo_gemm_hilo_ok(C.byref(args))) == 1


def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, y: torch.Tensor,
                  eps: float = LN_EPS) -> torch.Tensor:
    assert x.dtype == torch.float32
    check(_lib.load().rpo_layernorm_fwd(x.data_ptr(), _ld(x), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                                        _ld(y), dtype_code(y.dtype), x.shape[0], x.shape[1], eps, _stream()),
          "rpo_layernorm_fwd")
    return y


def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, dres: Optional[torch.Tensor],
                  dx: torch.Tensor, dx_cast: Optional[torch.Tensor] = None, eps: float = LN_EPS) -> torch.Tensor:
    assert x.dtype == torch.float32 and dx.dtype == torch.float32
    splits, split_stride = 1, 0
    if dy.dim() == 3:                       # [S, rows, d] split-K slabs
        splits, split_stride, lddy = dy.shape[0], dy.stride(0), dy.stride(1)
    else:
        lddy = _ld(dy)
    check(_lib.load().rpo_layernorm_bwd(
        dy.data_ptr(), dtype_code(dy.dtype), lddy, x.data_ptr(), _ld(x), gamma.data_ptr(),
        _p(dres), 0 if dres is None else _ld(dres), dx.data_ptr(), _ld(dx), _p(dx_cast),
        RPO_F32 if dx_cast is None else dtype_code(dx_cast.dtype), 0 if dx_cast is None else _ld(dx_cast),
        x.shape[0], x.shape[1], eps, splits, split_stride, _stream()), "rpo_layernorm_bwd")
    return dx


def _ln_bwd_args(dy, x, gamma, dres, dx, dx_cast=None, eps: float = LN_EPS) -> "_lib.LnBwdArgs":
    assert dy.dtype == torch.float32 and x.dtype == torch.float32 and dx.dtype == torch.float32
    splits, split_stride = 1, 0
    if dy.dim() == 3:
        splits, split_stride, lddy = dy.shape[0], dy.stride(0), dy.stride(1)
    else:
        lddy = _ld(dy)
    return _lib.LnBwdArgs(dy=dy.data_ptr(), lddy=lddy, x=x.data_ptr(), ldx=_ld(x), gamma=gamma.data_ptr(),
                          dres=_p(dres), lddres=0 if dres is None else _ld(dres), dx=dx.data_ptr(), lddx=_ld(dx),
                          dx_cast=_p(dx_cast), cast_dtype=RPO_F32 if dx_cast is None else dtype_code(dx_cast.dtype),
                          ldcast=0 if dx_cast is None else _ld(dx_cast), rows=x.shape[0], d=x.shape[1], eps=eps,
                          dy_splits=splits, dy_split_stride=split_stride)


def layernorm_bwd_pair(l0: dict, l1: dict) -> None:
    """Two `layernorm_bwd` problems (keyword arguments dy, x, gamma, dres, dx, dx_cast) in ONE launch."""
    a0, a1 = _ln_bwd_args(**l0), _ln_bwd_args(**l1)
    check(_lib.experimental().rpo_layernorm_bwd_pair(C.byref(a0), C.byref(a1), _stream()), "rpo_layernorm_bwd_pair")


def _attn_bwd_args(q_rows, k, v, dx, w_out_t, dq, groups: int, H: int, keys: int, Kp: int, scale: float = 0.125,
                   key_len: Optional[torch.Tensor] = None, key_stride: int = 0) -> "_lib.AttnBwdArgs":
    assert q_rows.dtype == k.dtype == v.dtype == dx.dtype == w_out_t.dtype == dq.dtype and _ld(k) == _ld(v)
    if key_len is not None:
        assert key_len.dtype == torch.int32 and key_len.is_contiguous() and key_len.numel() >= groups
    return _lib.AttnBwdArgs(q_rows=q_rows.data_ptr(), ldq=_ld(q_rows), k=k.data_ptr(), v=v.data_ptr(), ldkv=_ld(k),
                            dx=dx.data_ptr(), lddx=_ld(dx), w_out_t=w_out_t.data_ptr(), ldw=_ld(w_out_t),
                            dq=dq.data_ptr(), lddq=_ld(dq), groups=groups, H=H, keys=keys, Kp=Kp,
                            key_len=_p(key_len), key_stride=key_stride, scale=scale)


def attn_bwd_proj_pair(p0: dict, p1: Optional[dict] = None) -> None:
    """Attention backward of the prompt rows with the d out-proj GEMM folded in, for one or two problems in ONE launch
    (rpo_attn_bwd_proj_pair).  p0 / p1: keyword arguments q_rows, k, v, dx, w_out_t, dq, groups, H, keys, Kp, scale and,
    for per-group key counts (the text tower's K / V cache), key_len (int32 [groups]) + key_stride."""
    a0 = _attn_bwd_args(**p0)
    a1 = None if p1 is None else _attn_bwd_args(**p1)
    check(_lib.experimental().rpo_attn_bwd_proj_pair(C.byref(a0), None if a1 is None else C.byref(a1),
                                             dtype_code(p0["dq"].dtype), _stream()), "rpo_attn_bwd_proj_pair")


def im2col_patches(img: torch.Tensor, out: torch.Tensor, patch: int) -> torch.Tensor:
    B, Cc, H, W = img.shape
    assert Cc == 3 and img.dtype == torch.float32 and img.is_contiguous()
    check(_lib.load().rpo_im2col_patches(img.data_ptr(), out.data_ptr(), dtype_code(out.dtype), _ld(out), B, H, W,
                                         patch, _stream()), "rpo_im2col_patches")
    return out


def img_embed_norm(x_pre, cls, pos0, img_prompt, g_pre, b_pre, x0, g1, b1, h, B: int, N: int, Kp: int,
                   eps: float = LN_EPS, rows: Optional[tuple] = None):
    """img_assemble + ln_pre (-> x0) + the first block's ln_1 (-> h) in one launch.  rows = (row0, row1): those token rows
    only (rpo_img_embed_norm_rows: the frozen rows [0, B*N) do not depend on the prompts)."""
    r0, r1 = (0, B * (N + Kp)) if rows is None else rows
    check(_lib.load().rpo_img_embed_norm_rows(x_pre.data_ptr(), _ld(x_pre), cls.data_ptr(), pos0.data_ptr(), _p(img_prompt),
                                              g_pre.data_ptr(), b_pre.data_ptr(), x0.data_ptr(), _ld(x0), g1.data_ptr(),
                                              b1.data_ptr(), h.data_ptr(), _ld(h), dtype_code(h.dtype), B, N, Kp,
                                              x_pre.shape[1], eps, r0, r1, _stream()), "rpo_img_embed_norm_rows")
    return h


def img_assemble(x: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor, img_prompt: torch.Tensor, B: int, N: int,
                 Kp: int) -> torch.Tensor:
    check(_lib.load().rpo_img_assemble(x.data_ptr(), _ld(x), cls.data_ptr(), pos.data_ptr(), img_prompt.data_ptr(),
                                       B, N, Kp, x.shape[1], _stream()), "rpo_img_assemble")
    return x


def broadcast_rows(src: torch.Tensor, dst: torch.Tensor, groups: int) -> torch.Tensor:
    rows, d = src.shape
    assert src.is_contiguous() and dst.shape[0] >= groups * rows
    check(_lib.load().rpo_broadcast_rows(src.data_ptr(), dst.data_ptr(), _ld(dst), groups, rows, d, _stream()),
          "rpo_broadcast_rows")
    return dst


def reduce_groups(src: torch.Tensor, out: torch.Tensor, groups: int) -> torch.Tensor:
    rows, d = out.shape
    assert out.is_contiguous()
    check(_lib.load().rpo_reduce_groups(src.data_ptr(), _ld(src), out.data_ptr(), groups, rows, d, _stream()),
          "rpo_reduce_groups")
    return out


def attn_readonly_fwd(q, k, v, out, B: int, H: int, N: int, Kp: int, scale: float = 0.125, q_first: int = 0):
    """q_first > 0: only queries q_first .. N+Kp-1 of every image (the prompt rows when q_first = N)."""
    assert _ld(q) == _ld(k) == _ld(v)
    check(_lib.load().rpo_attn_readonly_fwd_rows(q.data_ptr(), k.data_ptr(), v.data_ptr(), _ld(q), out.data_ptr(),
                                                 _ld(out), dtype_code(q.dtype), B, H, N, Kp, scale, q_first, _stream()),
          "rpo_attn_readonly_fwd_rows")
    return out


def attn_readonly_bwd_proj(q_rows, k, v, dx, w_out_t, dq, B: int, H: int, N: int, Kp: int, scale: float = 0.125):
    """attn_readonly_bwd with the d out-proj GEMM folded in: dx = gradient of the out-proj output (act dtype)."""
    assert q_rows.dtype == k.dtype == v.dtype == dx.dtype == w_out_t.dtype == dq.dtype
    check(_lib.load().rpo_attn_readonly_bwd_proj(q_rows.data_ptr(), _ld(q_rows), k.data_ptr(), v.data_ptr(), _ld(k),
                                                 dx.data_ptr(), _ld(dx), w_out_t.data_ptr(), _ld(w_out_t), dq.data_ptr(),
                                                 _ld(dq), dtype_code(dq.dtype), B, H, N, Kp, scale, _stream()),
          "rpo_attn_readonly_bwd_proj")
    return dq


def attn_readonly_bwd(q_rows, k, v, da, dq, B: int, H: int, N: int, Kp: int, scale: float = 0.125):
    assert _ld(k) == _ld(v)
    check(_lib.load().rpo_attn_readonly_bwd(q_rows.data_ptr(), _ld(q_rows), k.data_ptr(), v.data_ptr(), _ld(k),
                                            da.data_ptr(), _ld(da), dq.data_ptr(), _ld(dq),
                                            dtype_code(q_rows.dtype), B, H, N, Kp, scale, _stream()),
          "rpo_attn_readonly_bwd")
    return dq


def chain_state() -> int:
    """bytes of device scratch rpo_chain_bwd needs (counters, placement table)"""
    return int(_lib.experimental().rpo_chain_state_bytes())


def _chain_args(layers: list, units: int, Kp: int, d: int, H: int, keys: int, dtype: torch.dtype, key_len=None,
                key_stride: int = 0, ldx: int = 0, ldq: int = 0, ldkv: int = 0, dxa=None, dxb=None, dxc=None, du=None,
                dq=None, dy=None, scale: float = 0.125, eps: float = 1e-5, wgs_per_group: int = 0, state=None,
                timeline=None):
    arr = (_lib.ChainLayer * max(1, len(layers)))()
    for i, L in enumerate(layers):
        arr[i] = _lib.ChainLayer(*[_p(L[k]) for k in ("w_proj_t", "w_fc_t", "w_out_t", "w_q_t", "aux", "x_ln2", "x_ln1",
                                                       "ln2_w", "ln1_w", "q_rows", "k", "v")])
    a = _lib.ChainBwdArgs(layer=arr, layers=len(layers), units=units, Kp=Kp, d=d, H=H, keys=keys, dtype=dtype_code(dtype),
                          key_len=_p(key_len), key_stride=key_stride, ldx=ldx, ldq=ldq, ldkv=ldkv, dxa=_p(dxa), dxb=_p(dxb),
                          dxc=_p(dxc), du=_p(du), dq=_p(dq), dy=_p(dy), dy_stride=(dy.stride(0) if dy is not None else 0),
                          scale=scale, eps=eps, wgs_per_group=wgs_per_group, state=_p(state), timeline=_p(timeline))
    a._keep = arr
    return a


def chain_bwd_ok(layers: int, units: int, Kp: int, d: int, H: int, keys: int, dtype: torch.dtype) -> bool:
    if dtype == torch.float32:
        return False
    a = _lib.ChainBwdArgs(layers=layers, units=units, Kp=Kp, d=d, H=H, keys=keys, dtype=dtype_code(dtype))
    return bool(_lib.experimental().rpo_chain_bwd_ok(C.byref(a)))


def chain_bwd(layers: list, **kw) -> None:
    """The prompt-row backward chain of one tower as one persistent launch (include/rpo_amd.h: rpo_chain_bwd).
    layers: one dict per block (block 0 first) with the tensors of struct rpo_chain_layer; x_ln2 / x_ln1 / q_rows are
    views of the back-propagated rows; dy: fp32 [4, rows, d] slabs."""
    a = _chain_args(layers, **kw)
    check(_lib.experimental().rpo_chain_bwd(C.byref(a), _stream()), "rpo_chain_bwd")


def text_attn_fwd(q, kc, vc, out, len_i32, n_cls: int, rows: int, Lmax: int, H: int, causal: bool = False,
                  scale: float = 0.125):
    assert _ld(kc) == _ld(vc) and len_i32.dtype == torch.int32
    check(_lib.load().rpo_text_attn_fwd(q.data_ptr(), _ld(q), kc.data_ptr(), vc.data_ptr(), _ld(kc),
                                        out.data_ptr(), _ld(out), dtype_code(q.dtype), len_i32.data_ptr(), n_cls,
                                        rows, Lmax, H, int(causal), scale, _stream()), "rpo_text_attn_fwd")
    return out


def text_attn_bwd_dense(q, k, v, d_out, dq, dk, dv, len_i32, n_cls: int, Lmax: int, H: int, scale: float = 0.125):
    """dq, dk, dv of the causal text attention for every row (include/rpo_amd.h: rpo_text_attn_bwd_dense)."""
    assert _ld(q) == _ld(k) == _ld(v) and _ld(dq) == _ld(dk) == _ld(dv) and len_i32.dtype == torch.int32
    assert q.dtype == d_out.dtype == dq.dtype
    check(_lib.load().rpo_text_attn_bwd_dense(q.data_ptr(), k.data_ptr(), v.data_ptr(), _ld(q), d_out.data_ptr(),
                                              _ld(d_out), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), _ld(dq),
                                              dtype_code(q.dtype), len_i32.data_ptr(), n_cls, Lmax, H, scale, _stream()),
          "rpo_text_attn_bwd_dense")
    return dq


def text_attn_bwd(q, kc, vc, da, dq, len_i32, n_cls: int, rows: int, Lmax: int, H: int, scale: float = 0.125):
    assert _ld(kc) == _ld(vc) and len_i32.dtype == torch.int32
    check(_lib.load().rpo_text_attn_bwd(q.data_ptr(), _ld(q), kc.data_ptr(), vc.data_ptr(), _ld(kc),
                                        da.data_ptr(), _ld(da), dq.data_ptr(), _ld(dq), dtype_code(q.dtype),
                                        len_i32.data_ptr(), n_cls, rows, Lmax, H, scale, _stream()),
          "rpo_text_attn_bwd")
    return dq


def head_workspace_floats(B: int, Cc: int, K: int, e: int) -> int:
    return int(_lib.load().rpo_head_workspace_floats(B, Cc, K, e))


def head_fwd_bwd(img_f, text_f, label, scale_exp: float, logits, loss, d_img_f, d_text_f, ws,
                 d_img_f_act=None, d_text_f_act=None):
    """d_*_act: optional act-dtype copies of the two gradients (what the projections' dX GEMMs read)."""
    B, K, e = img_f.shape
    Cc = text_f.shape[0]
    assert img_f.is_contiguous() and text_f.is_contiguous() and logits.is_contiguous()
    assert label is None or label.dtype == torch.int64
    act = d_img_f_act if d_img_f_act is not None else d_text_f_act
    assert act is None or (act.is_contiguous() and (d_text_f_act is None or d_text_f_act.is_contiguous()))
    check(_lib.load().rpo_head_fwd_bwd_act(img_f.data_ptr(), text_f.data_ptr(), _p(label), scale_exp,
                                           logits.data_ptr(), _p(loss), _p(d_img_f), _p(d_text_f), _p(d_img_f_act),
                                           _p(d_text_f_act), _lib.RPO_F32 if act is None else dtype_code(act.dtype),
                                           B, Cc, K, e, ws.data_ptr(), _stream()), "rpo_head_fwd_bwd_act")
    return logits


def metanet_fwd(img_f, w1, b1, w2, b2, f_norm, hidden, bias):
    B, e = img_f.shape
    h, d = w1.shape[0], w2.shape[0]
    assert all(t.dtype == torch.float32 and t.is_contiguous() for t in (img_f, w1, b1, w2, b2, f_norm, hidden, bias))
    check(_lib.load().rpo_metanet_fwd(img_f.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                      f_norm.data_ptr(), hidden.data_ptr(), bias.data_ptr(), B, e, h, d, _stream()),
          "rpo_metanet_fwd")
    return bias


def metanet_bwd(d_bias, f_norm, hidden, w2, g_w1, g_b1, g_w2, g_b2):
    B, d = d_bias.shape
    e, h = f_norm.shape[1], hidden.shape[1]
    assert all(t.dtype == torch.float32 and t.is_contiguous() for t in (d_bias, f_norm, hidden, w2, g_w1, g_b1, g_w2, g_b2))
    check(_lib.load().rpo_metanet_bwd(d_bias.data_ptr(), f_norm.data_ptr(), hidden.data_ptr(), w2.data_ptr(),
                                      g_w1.data_ptr(), g_b1.data_ptr(), g_w2.data_ptr(), g_b2.data_ptr(), B, e, h, d,
                                      _stream()), "rpo_metanet_bwd")


def sgd_step(p, g, buf, lr: float, momentum: float, wd: float, grad_scale: float, first_step: bool):
    assert p.is_contiguous() and g.is_contiguous() and buf.is_contiguous()
    check(_lib.load().rpo_sgd_step(p.data_ptr(), g.data_ptr(), buf.data_ptr(), p.numel(), lr, momentum, wd,
                                   grad_scale, int(first_step), _stream()), "rpo_sgd_step")
    return p


def sgd_step_guarded(p, g, buf, lr: float, momentum: float, wd: float, grad_scale: float, first_step: bool, found_inf):
    """sgd_step, skipped as a whole when a gradient is Inf / NaN (GradScaler.step); found_inf: int32[2] device tensor."""
    assert p.is_contiguous() and g.is_contiguous() and buf.is_contiguous()
    assert found_inf.dtype == torch.int32 and found_inf.numel() >= 2 and found_inf.is_contiguous()
    check(_lib.load().rpo_sgd_step_guarded(p.data_ptr(), g.data_ptr(), buf.data_ptr(), p.numel(), lr, momentum, wd,
                                           grad_scale, int(first_step), found_inf.data_ptr(), _stream()),
          "rpo_sgd_step_guarded")
    return p


def convert(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    assert src.dtype == torch.float32 and src.shape == dst.shape
    check(_lib.load().rpo_convert(src.data_ptr(), _ld(src), dst.data_ptr(), dtype_code(dst.dtype), _ld(dst),
                                  src.shape[0], src.shape[1], _stream()), "rpo_convert")
    return dst


def probe_mfma(which: int, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    d = torch.empty(32, 32, dtype=torch.float32, device=a.device)
    check(_lib.load().rpo_probe_mfma(which, a.data_ptr(), b.data_ptr(), d.data_ptr(), _stream()),
          "rpo_probe_mfma")
    return d


def probe_peaks(device: torch.device, which: int = 0, copy: bool = True) -> dict:
    """Empirical peaks of this GPU: sustained MFMA TFLOP/s of a pure-MFMA loop (8 waves per SIMD-pair resident,
    non-zero operands) and GB/s of a 1 GiB stream copy (read + write bytes).  ~0.2 s."""
    import ctypes
    lib = _lib.load()
    sink = torch.zeros(4, dtype=torch.float32, device=device)
    flops = ctypes.c_double(0.0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    blocks, iters = 256 * 8, 4096 if which != 1 else 2048
    for timed in (False, True):
        if timed:
            ev[0].record()
        check(lib.rpo_probe_peak_mfma(which, blocks, iters, sink.data_ptr(), ctypes.byref(flops), _stream()),
              "rpo_probe_peak_mfma")
        if timed:
            ev[1].record()
    if not copy:
        torch.cuda.synchronize(device)
        return {"mfma_tflops": flops.value / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e12}
    n = 1 << 29
    src = torch.empty(n, dtype=torch.uint8, device=device).fill_(1)
    dst = torch.empty_like(src)
    reps = 5
    for timed in (False, True):
        if timed:
            ev[2].record()
        for _ in range(reps if timed else 1):
            check(lib.rpo_probe_peak_copy(src.data_ptr(), dst.data_ptr(), n, _stream()), "rpo_probe_peak_copy")
        if timed:
            ev[3].record()
    torch.cuda.synchronize(device)
    return {"mfma_tflops": flops.value / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e12,
            "copy_gbs": 2.0 * n * reps / (ev[2].elapsed_time(ev[3]) * 1e-3) / 1e9}
