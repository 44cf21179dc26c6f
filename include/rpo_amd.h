/*
 * rpo_amd.h -- C ABI of librpo_hip.so: the MI355X (gfx950) kernels behind RPO's
 * few-shot train step.
 *
 * The reference (mlvlab/RPO) has no FFI / plugin registry: its hot path is stock
 * torch.nn modules called from trainers/rpo.py:161-232 (CustomCLIP.forward) and
 * clip/model.py:181-207 (ResidualAttentionBlock / Transformer).  Each entry point
 * below replaces one stock-torch op sequence on that path; the comment on each
 * cites the reference lines it stands in for.  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer unless noted
 *   - the caller (PyTorch-ROCm) allocates and owns every buffer; nothing here
 *     allocates, frees or keeps state a caller could observe (the one internal cache: an atomic per-kernel
 *     "large-LDS attribute already set on device d" bit in front of an idempotent hipFuncSetAttribute)
 *   - every function only ENQUEUES work on `stream` (a hipStream_t passed as
 *     void*; 0 = the null stream) and returns immediately
 *   - return value: 0 = ok, > 0 = hipError_t of the launch, < 0 = RPO_E_* argument
 *     error (nothing was enqueued); rpo_error_string() decodes any of them
 *   - dtype arguments take RPO_F32 / RPO_BF16 / RPO_F16.  "act dtype" is the storage type of
 *     activations and frozen weights: RPO_F32 = parity mode (exact-f32 MFMA,
 *     v_mfma_f32_32x32x2_f32), RPO_BF16 = throughput mode (bf16 storage,
 *     v_mfma_f32_32x32x16_bf16, fp32 accumulate).  The residual stream, LayerNorm
 *     statistics, softmax, logits, loss, gradients of the prompts and the
 *     optimiser state are fp32 in both modes.
 *   - leading dimensions (ld*) are in ELEMENTS of the respective dtype
 *
 * Row layout of the image tower (B images, N = 1 + patches frozen tokens,
 * Kp read-only prompts per image):
 *     rows [0, B*N)          frozen tokens, image-major:  b*N + t
 *     rows [B*N, B*N + B*Kp) prompt tokens, image-major:  B*N + b*Kp + i
 * so the rows that are back-propagated form one contiguous sub-matrix.
 */
#ifndef RPO_AMD_H
#define RPO_AMD_H

#include <stddef.h>
#include <stdint.h>

/* The library is built with -fvisibility=hidden: only what this header declares is exported. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif
#ifdef __cplusplus
extern "C" {
#endif

#define RPO_ABI_VERSION 8

enum { RPO_F32 = 0, RPO_BF16 = 1, RPO_F16 = 2 };

enum {
  RPO_E_BADARG = -1,   /* null pointer / non-positive size */
  RPO_E_SHAPE = -2,    /* size not supported by the kernels (see each function) */
  RPO_E_DTYPE = -3,    /* dtype combination not supported */
  RPO_E_ALIGN = -4,    /* pointer / leading dimension not 16-byte aligned */
  RPO_E_WORKSPACE = -5 /* caller-provided workspace / table bounds too small for this call */
};

/* GEMM epilogues (fused into the MFMA kernel's store) */
enum {
  RPO_EPI_NONE = 0,        /* C = acc                                  (dX GEMMs of the backward)            */
  RPO_EPI_BIAS = 1,        /* C = acc + bias[n]                        (in-proj, clip/model.py:186)          */
  RPO_EPI_BIAS_QGELU = 2,  /* C = quickgelu(acc + bias); optionally saves u = acc + bias (fp32) for rows
                              >= aux_row0 (the rows that will be back-propagated)   (c_fc, clip/model.py:174-175) */
  RPO_EPI_BIAS_RESID = 3,  /* C(f32) = resid + acc + bias              (out_proj / c_proj + residual, :189-190) */
  RPO_EPI_QGELU_BWD = 4,   /* C = acc * quickgelu'(aux[m,n])           (backward through clip/model.py:162-164) */
  RPO_EPI_PATCH = 5,       /* C(f32)[m + m/group + 1, n] = acc + resid[(m % group) + 1, n]: patch embedding
                              written straight into the token matrix with the positional embedding added
                              (trainers/rpo.py:198-202)                                                      */
  /* LayerNorm folded into the GEMM that consumes it (clip/model.py:156-159 feeding :186 / :174).  With
     W'[n,k] = gamma[k] W[n,k], s[n] = sum_k W'[n,k], b'[n] = b[n] + sum_k beta[k] W[n,k]:
         LN(x) . W^T + b  =  rstd[m] * (x . W'^T - mu[m] * s[n]) + b'[n]
     so A is the RAW residual row (16-bit copy written by the producing GEMM, `out2` below), W = W', bias = b',
     ln_colsum = s, and mu / rstd come from the per-row partial statistics the producer left in ln_stats. */
  RPO_EPI_LN_BIAS = 6,       /* C = LN-fold(acc)                        (ln_1 + in-proj)                     */
  RPO_EPI_LN_BIAS_QGELU = 7  /* C = quickgelu(LN-fold(acc)), aux as BIAS_QGELU   (ln_2 + c_fc)               */
};

typedef struct rpo_gemm_args {
  const void* A;  int64_t lda;   /* [M, K] act dtype, row-major                                   */
  const void* W;  int64_t ldw;   /* [N, K] act dtype, row-major (nn.Linear weight layout [out,in]) */
  void* C;        int64_t ldc;   /* [M, N] out_dtype                                              */
  int32_t M, N, K;
  int32_t in_dtype, out_dtype, epilogue;
  const float* bias;             /* [N] fp32 or NULL                                              */
  const float* resid; int64_t ldr; /* fp32; RESID: [M, N]; PATCH: positional embedding [group+1, N] */
  void* aux; int64_t ldaux;      /* fp32; see epilogues                                           */
  int32_t aux_row0;              /* BIAS_QGELU: first row whose pre-activation is saved (aux row 0);
                                    pass M (or aux = NULL) to save nothing                        */
  int32_t skip_row0, skip_col0;  /* tiles with all rows >= skip_row0 AND all cols >= skip_col0 are not
                                    computed (K/V of prompt rows are never read); -1 disables      */
  int32_t group;                 /* PATCH: patches per image                                       */
  int32_t split_k;               /* <= 1: off.  S > 1 (EPI_NONE, fp32 C only): k-range split into S slices,
                                    slice s writes its partial product to C + s * split_stride; the
                                    consumer (rpo_layernorm_bwd's dy_splits) adds the slabs in order    */
  int64_t split_stride;          /* elements between slabs (>= M * ldc)                            */
  int32_t tile_config;           /* 0 = choose by shape.  For benchmarking / tests (a config whose conditions do not hold falls
                                    through to the shape heuristic unless noted): 2 = 128x128 tiles, 5 = 64x64, 6 = 64x128,
                                    9 = 128x128 with 4 waves; 16-bit in / out with a BIAS-type or LN epilogue: 3 = 256x256
                                    lock-step, 7 = 256x256 ping-pong, 8 = 256x256 one wave per SIMD (what 0 picks when the
                                    tiles fill a round of the CUs), 10 = 224x384 (N % 384 == 0) or 288x256 (N % 256 == 0)
                                    tiles in whole rounds of the CUs (what 0 picks for c_fc at 32 / 64 / 128 images of
                                    ViT-B/16, 16 of ViT-L/14); BIAS_RESID, 16-bit in, fp32 out: 11 = 224x96 or 288x64
                                    tiles with the waves splitting k (needs the row-unit hint below; RPO_E_SHAPE if it
                                    does not apply).  Configs 2 / 3 / 5 / 6 / 7 / 8 / 9 / 10 give bit-identical results;
                                    11 sums k in four parts and differs in the last bits                              */
  /* LayerNorm fold (all optional, 0 / NULL = off) */
  void* out2; int64_t ldout2;    /* BIAS_RESID: also store C in the act dtype (in_dtype) here: the A operand of the
                                    GEMM that consumes LN(C)                                        */
  float* ln_stats;               /* BIAS_RESID: OUT, [M][N/g][2] fp32: (mean, sum of squared deviations) of every
                                    g-column group of the row of C just written, g = ln_group (64 or 96).
                                    LN_BIAS*: IN, the same array for the rows of A (groups = K / g <= 16) */
  const float* ln_colsum;        /* LN_BIAS*: s[n], fp32 [N]                                         */
  float ln_eps;                  /* LN_BIAS*: epsilon of the folded LayerNorm                        */
  /* Optional tiling hint (results do not depend on it): the rows of A / C are two segments, [0, seg1_row0) and
     [seg1_row0, M), and unit u owns rows [u * seg_rows0, +seg_rows0) of the first and [seg1_row0 + u * seg_rows1,
     +seg_rows1) of the second -- the image tower's layout: N frozen rows + Kp prompt rows per image.  A kernel that
     tiles one unit per workgroup then spreads the rows with extra epilogue work (saved pre-activations) evenly. */
  int32_t seg_rows0, seg_rows1, seg1_row0;
  /* Columns per partial row statistic in ln_stats: 0 or 64 (default), or 96.  96 is what the one-round 224x96 kernel
     writes (BIAS_RESID with a row-unit hint, N % 96 == 0, K % 256 == 0: tile_config 11 / the heuristic's choice for the
     image tower's out-proj and c_proj of ViT-B/16; its 288x64 geometry for ViT-L/14 writes 64); a producer that cannot
     write the requested layout returns RPO_E_SHAPE, and the consuming LN_BIAS* GEMM must be given the same value.
     rpo_gemm_stats_group() tells which. */
  int32_t ln_group;
  /* What aux holds.  RPO_F32 (0, default): the fp32 pre-activation u -- *_QGELU epilogues write it, QGELU_BWD evaluates
     quickgelu'(u).  in_dtype (16-bit modes only): d quickgelu / du itself in the act dtype ([rows, ldaux] of that type) --
     *_QGELU epilogues write it, QGELU_BWD multiplies by it: half the bytes, no transcendental in the backward. */
  int32_t aux_dtype;
  /* Optional hint (results do not depend on it; may be NULL): `prefetch_bytes` bytes at `prefetch` are what the NEXT
     launch on this stream will read first -- in the image tower the next GEMM's frozen weight matrix, which no cache
     still holds a whole step after its last use.  The one-round kernels have every workgroup touch its 1 / grid share
     of those lines (one dword per 128-B line, issued before the k-loop) so that the next kernel finds them in the
     memory-side cache instead of HBM.  Kernels that do not implement the hint ignore it. */
  const void* prefetch;
  int64_t prefetch_bytes;
  /* Residual stream as 16-bit hi / lo halves (BIAS_RESID, 16-bit inputs, optional; ABI 4).  The fp32 residual stream of
     a transformer block (clip/model.py:188-190: x = x + attn(ln_1(x)); x = x + mlp(ln_2(x))) is written and re-read in
     full by every residual GEMM, next to the 16-bit copy the following GEMM consumes.  With these fields the stream lives
     as hi = round16(v) -- which IS that copy (out2) -- and lo = round16(v - hi): resid_hi / resid_lo replace `resid` as
     the input (leading dimension ldr16, in elements; resid_lo may be NULL: the stream is then the 16-bit hi alone, which
     is what the reference's own `PREC: fp16` run keeps -- clip/model.py:379-400 converts the model, :153-159 casts
     LayerNorm's fp32 result back), out_lo receives lo next to out2 (leading dimension ldout2; NULL: not kept), and the fp32 C is stored only for rows >= c_row0 (the back-propagated rows, whose LayerNorm backward reads
     fp32; pass 0 to store all rows).  hi + lo carries 16 mantissa bits (bf16) / 22 (fp16) of the fp32 value.  In place is
     allowed (resid_hi == out2, resid_lo == out_lo: every element is read and written by the same thread).  Only the
     one-round row-unit kernels implement it: rpo_gemm_hilo_ok() tells; anything else returns RPO_E_SHAPE. */
  const void* resid_hi; const void* resid_lo; int64_t ldr16;
  void* out_lo;
  int32_t c_row0;
} rpo_gemm_args;

int rpo_version(void);
const char* rpo_error_string(int code);

/* C = A . W^T with a fused epilogue.  Requires K % 64 == 0 (bf16) / K % 32 == 0 (f32),
 * N % 4 == 0, 16-byte aligned rows.  Replaces nn.Linear / F.linear / the matmuls of
 * nn.MultiheadAttention's packed in-proj and out-proj (clip/model.py:171-177,186) and,
 * in the backward, autograd's mm(dY, W) (trainers/rpo.py:308). */
int rpo_gemm_nt(const rpo_gemm_args* args, void* stream);

/* ---- rpo_gemm_nt for the PROMPT ROWS: a few hundred rows against a frozen weight (ABI 7) -------------------------------
 * Same contract as rpo_gemm_nt (C = A . W^T with the fused epilogues above; replaces nn.Linear's matmul,
 * clip/model.py:173-177,186, and autograd's mm(dY, W), trainers/rpo.py:308, for the B*K / n_cls*K prompt rows of the
 * two towers), except that W is the FRAGMENT-MAJOR copy rpo_gemm_ws_pack made of the [N, K] weight once at load time:
 * the weight operand then streams global -> VGPR in MFMA fragment order (1 KiB per wave-instruction, no LDS), the four
 * waves of a workgroup split the contraction, and the tile (32x32 .. 96x96) is chosen so that the launch covers the CUs
 * about once (rpo_amd/csrc/gemm_ws.hip).  args->W = the packed copy, args->ldw is ignored.
 *   16-bit inputs only (RPO_E_DTYPE otherwise); K % 64 == 0, N % 32 == 0; epilogues NONE (16-bit or fp32 C; split_k slabs
 *   as rpo_gemm_nt), QGELU_BWD, BIAS, BIAS_QGELU, LN_BIAS, LN_BIAS_QGELU (K / ln_group <= 16), BIAS_RESID (fp32 C;
 *   out2 / ln_stats over 64-column groups); no skip_*, row units or hi / lo residual (RPO_E_SHAPE: use rpo_gemm_nt).
 *   tile_config: 0 = choose; 100 * MT + 10 * NT forces MT x NT MFMA tiles per workgroup (110, 120, 220, 330).
 * Deterministic; the k sum is split in four, so the last bits differ from rpo_gemm_nt's. */
int rpo_gemm_ws(const rpo_gemm_args* args, void* stream);
/* Wp[N * K] (act dtype, 16-byte aligned) <- W[N, K] row-major (ldw in elements): piece ((nb * K/16 + ks) * 64 + lane) of 8
 * elements = W[nb * 32 + (lane & 31)][ks * 16 + (lane >> 5) * 8 ..].  N % 32 == 0, K % 64 == 0. */
int rpo_gemm_ws_pack(const void* W, int64_t ldw, void* Wp, int N, int K, int dtype, void* stream);
/* 1 if rpo_gemm_ws takes these shapes / dtypes / epilogue (pointers are not looked at), else 0 */
int rpo_gemm_ws_ok(const rpo_gemm_args* args);

/* The partial-statistics layout (rpo_gemm_args.ln_group: 64 or 96) a BIAS_RESID producer writes for these shapes,
 * dtypes and row units when the kernel choice is left to the library (tile_config 0).  Only M, N, K, lda, ldw, the
 * dtypes, split_k and seg_* are looked at.  Callers size ln_stats as [M, N / group, 2] floats and pass the same group to
 * the producer and to the LN_BIAS* consumer. */
int rpo_gemm_stats_group(const rpo_gemm_args* args);

/* 1 if rpo_gemm_nt would run these shapes / dtypes / row units on a kernel that implements the hi / lo residual stream
 * (resid_hi, resid_lo, out_lo, c_row0 above), else 0.  Looks at the same fields as rpo_gemm_stats_group plus ln_group. */
int rpo_gemm_hilo_ok(const rpo_gemm_args* args);

/* y = LayerNorm(x) * gamma + beta, statistics in fp32, eps as given (1e-5).
 * x fp32 [rows, d] (ldx), y in y_dtype.  d % 4 == 0, d <= 2048.  In-place (y == x, fp32) is allowed.
 * Replaces clip/model.py:153-159 (ln_1, ln_2, ln_pre, ln_post, ln_final). */
int rpo_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta,
                      void* y, int64_t ldy, int y_dtype, int rows, int d, float eps, void* stream);

/* dx = dres + dLN(dy; x, gamma)   (frozen affine: no dgamma/dbeta).
 * dy in dy_dtype [rows, d] -- or, with dy_splits = S > 1 (fp32 only), the sum of S slabs
 * dy + s * dy_split_stride written by a split-K rpo_gemm_nt, added in the order s = 0..S-1;
 * x fp32 = the forward input; dres fp32 or NULL;
 * dx fp32; dx_cast (optional, may be NULL) receives a copy of dx in cast_dtype
 * (the next GEMM's A operand).  Replaces autograd through clip/model.py:158. */
int rpo_layernorm_bwd(const void* dy, int dy_dtype, int64_t lddy, const float* x, int64_t ldx,
                      const float* gamma, const float* dres, int64_t lddres,
                      float* dx, int64_t lddx, void* dx_cast, int cast_dtype, int64_t ldcast,
                      int rows, int d, float eps, int dy_splits, int64_t dy_split_stride, void* stream);

/* Non-overlapping-patch im2col: img [B,3,H,W] fp32 -> out [B*(H/p)*(W/p), ldo] act dtype, column
 * order (c, ky, kx) = conv1.weight.reshape(d, -1); columns [3*p*p, ldo) are zero-filled.
 * With rpo_gemm_nt(RPO_EPI_PATCH) this replaces the stride-p Conv2d at trainers/rpo.py:198-200. */
int rpo_im2col_patches(const float* img, void* out, int out_dtype, int64_t ldo,
                       int B, int H, int W, int patch, void* stream);

/* Writes the rows the patch GEMM does not: CLS rows x[b*N] = cls + pos[0] and the prompt rows
 * x[B*N + b*Kp + i] = img_prompt[i] (no positional embedding: trainers/rpo.py:201-204). */
int rpo_img_assemble(float* x, int64_t ldx, const float* cls, const float* pos0,
                     const float* img_prompt, int B, int N, int Kp, int d, void* stream);

/* rpo_img_assemble + ln_pre + the first block's ln_1 in one launch (trainers/rpo.py:201-206, clip/model.py:189): per
 * token row -- CLS + pos[0], the patch row already in x_pre, or the image's prompt row -- x0 = ln_pre(token) (fp32) and
 * h = ln_1(x0) (h_dtype).  The CLS and prompt rows are also written to x_pre (the backward of ln_pre reads the prompt
 * rows).  Same bits as the three separate launches.  d % 4 == 0, d <= 1024. */
int rpo_img_embed_norm(float* x_pre, int64_t ldx, const float* cls, const float* pos0, const float* img_prompt,
                       const float* g_pre, const float* b_pre, float* x0, int64_t ldx0, const float* g1, const float* b1,
                       void* h, int64_t ldh, int h_dtype, int B, int N, int Kp, int d, float eps, void* stream);
/* (ABI 8) The same for the token rows [row0, row1) only.  The frozen rows [0, B*N) depend on the batch alone, not on the
 * prompts (trainers/rpo.py:198-203), so a trainer may form them for the NEXT batch while this step's backward is still
 * running and run [B*N, B*(N+Kp)) -- the prompt rows -- at the head of the next image forward.  Row by row the bits of
 * rpo_img_embed_norm. */
int rpo_img_embed_norm_rows(float* x_pre, int64_t ldx, const float* cls, const float* pos0, const float* img_prompt,
                            const float* g_pre, const float* b_pre, float* x0, int64_t ldx0, const float* g1,
                            const float* b1, void* h, int64_t ldh, int h_dtype, int B, int N, int Kp, int d,
                            float eps, int row0, int row1, void* stream);

/* dst[g*rows + i, :] = src[i, :]  (text prompts written into every class, trainers/rpo.py:176-177) */
int rpo_broadcast_rows(const float* src, float* dst, int64_t ld, int groups, int rows, int d, void* stream);

/* out[i, :] = sum_g src[g*rows + i, :] in fixed order g = 0..groups-1 (autograd of .repeat()) */
int rpo_reduce_groups(const float* src, int64_t ld, float* out, int groups, int rows, int d, void* stream);

/* Read-only masked attention of the image tower, all heads (head_dim 64), all B images.
 * q, k, v: act-dtype matrices in the row layout above with leading dimension ld (typically
 * three column slices of one packed [R, 3d] in-proj output).  Every query row (frozen and
 * prompt) reads ONLY the N frozen keys of its own image: the additive mask of
 * trainers/rpo.py:154-156 is -inf on the prompt columns for all rows, so those columns are
 * skipped, not computed.  out[R, ldo] act dtype.  N <= 288.
 * Replaces F.scaled_dot_product_attention inside nn.MultiheadAttention (clip/model.py:186). */
int rpo_attn_readonly_fwd(const void* q, const void* k, const void* v, int64_t ld,
                          void* out, int64_t ldo, int dtype, int B, int H, int N, int Kp,
                          float scale, void* stream);
/* Same, computing only the queries q_first .. N+Kp-1 of every image (earlier rows of `out` are left untouched).  The
 * last block of the image tower needs the prompt queries only: ln_post reads nothing else (trainers/rpo.py:210). */
int rpo_attn_readonly_fwd_rows(const void* q, const void* k, const void* v, int64_t ld,
                               void* out, int64_t ldo, int dtype, int B, int H, int N, int Kp, float scale,
                               int q_first, void* stream);

/* Backward of the above for the prompt rows only: dq[B*Kp, lddq] given da[B*Kp, ldda].
 * q_rows points at the first PROMPT row of q; k, v at the first frozen row.  dK/dV are not
 * produced: keys/values belong to frozen tokens.  Kp <= 128, N <= 288. */
int rpo_attn_readonly_bwd(const void* q_rows, int64_t ldq, const void* k, const void* v, int64_t ldkv,
                          const void* da, int64_t ldda, void* dq, int64_t lddq, int dtype,
                          int B, int H, int N, int Kp, float scale, void* stream);

/* The same with the d out-proj GEMM folded in (16-bit storage, d = H * 64 = 512, 768 or 1024, Kp <= 64): dx[B*Kp, lddx] is
 * the gradient of the out-proj OUTPUT (act dtype) and w_out_t[d, ldw] the transposed out-proj weight ([in, out], as
 * packed for the dX GEMMs); every (image, head, 32-query tile) workgroup forms its slice of da = dx . W_out itself.
 * Replaces the autograd of out_proj + SDPA (clip/model.py:186) for the prompt rows with one launch. */
int rpo_attn_readonly_bwd_proj(const void* q_rows, int64_t ldq, const void* k, const void* v, int64_t ldkv,
                               const void* dx, int64_t lddx, const void* w_out_t, int64_t ldw, void* dq, int64_t lddq,
                               int dtype, int B, int H, int N, int Kp, float scale, void* stream);

/* Text-tower attention for `rows` query rows per class against that class's cached keys /
 * values kc, vc [n_cls * Lmax, ldkv] (class c uses rows c*Lmax .. c*Lmax + len[c]).
 * causal = 0: every row reads keys [0, len[c])            (prompt rows, trainers/rpo.py:146-149:
 *             causal AND col < len_c, and prompts sit at positions >= len_c)
 * causal = 1: row t reads keys [0, min(t + 1, len[c]))     (the one-off pass over the frozen tokens)
 * len: int32 device array [n_cls].  Lmax <= 128.
 * Kernels: 16-bit storage, causal = 0, rows <= 64, Lmax <= 96, 16-byte aligned rows -- the training path's shapes -- run as
 * ONE WAVE per (class, head) on the matrix cores (fp32 accumulation and softmax; P and dS rounded to the storage type for
 * the second contraction, as in rpo_attn_readonly_fwd); everything else as fp32 VALU arithmetic.  Same meaning either way. */
int rpo_text_attn_fwd(const void* q, int64_t ldq, const void* kc, const void* vc, int64_t ldkv,
                      void* out, int64_t ldo, int dtype, const int32_t* len, int n_cls, int rows,
                      int Lmax, int H, int causal, float scale, void* stream);

/* dq = d(loss)/d(q) of the causal = 0 case given da = d(loss)/d(out); K and V are frozen (no dk, dv).  Replaces the autograd of
 * nn.MultiheadAttention's SDPA (clip/model.py:186, trainers/rpo.py:308) for the text tower's prompt rows. */
int rpo_text_attn_bwd(const void* q, int64_t ldq, const void* kc, const void* vc, int64_t ldkv,
                      const void* da, int64_t ldda, void* dq, int64_t lddq, int dtype,
                      const int32_t* len, int n_cls, int rows, int Lmax, int H, float scale, void* stream);

/* Dense backward of the text tower's causal attention: dq, dk, dv for ALL rows of every class (q, k, v, their
 * gradients: [n_cls * Lmax, ld / ldd], class c = rows c*Lmax .. c*Lmax + len[c]; row t reads keys [0, min(t + 1, len[c]))
 * as rpo_text_attn_fwd with causal = 1), given d_out = dL/d(attention output).  Replaces autograd of
 * nn.MultiheadAttention's SDPA (clip/model.py:186) under CLIP's plain causal mask (:332-338) for the sibling trainers
 * whose learned parameters sit in front of the class name (CoOp, trainers/coop.py:117-134,258-281).  Rows >= len[c]
 * get zero gradients.  Lmax <= 80 (CLIP's context is 77). */
int rpo_text_attn_bwd_dense(const void* q, const void* k, const void* v, int64_t ld, const void* d_out, int64_t lddo,
                            void* dq, void* dk, void* dv, int64_t ldd, int dtype, const int32_t* len, int n_cls,
                            int Lmax, int H, float scale, void* stream);

/* CoCoOp's meta-net (trainers/cocoop.py:93-97, PromptLearner.forward :137-143): bias[b] = linear2(relu(linear1(f[b] /
 * |f[b]|))) for the B raw image features img_f [B, e]; w1 [h, e], b1 [h], w2 [d, h], b2 [d], all fp32.  f_norm [B, e] and
 * hidden [B, h] are kept for rpo_metanet_bwd, which turns d_bias [B, d] into the gradients of the four tensors (batch sum
 * in fixed order).  (e + h) * 4 and B * h * 4 bytes must fit 48 KB of LDS. */
int rpo_metanet_fwd(const float* img_f, const float* w1, const float* b1, const float* w2, const float* b2,
                    float* f_norm, float* hidden, float* bias, int B, int e, int h, int d, void* stream);
int rpo_metanet_bwd(const float* d_bias, const float* f_norm, const float* hidden, const float* w2,
                    float* g_w1, float* g_b1, float* g_w2, float* g_b2, int B, int e, int h, int d, void* stream);

/* Cosine-logit head, cross-entropy and their backward (trainers/rpo.py:215-230):
 *   logits[b,c] = (scale_exp / K) * sum_i <img_f[b,i]/|.|, text_f[c,i]/|.|>
 *   loss = mean_b CE(logits[b], label[b])
 * img_f [B,K,e], text_f [C,K,e], logits [B,C], loss [1], d_img_f / d_text_f like the inputs, all fp32.
 * label: int64 device array [B], or NULL for eval (only logits are written).  A target outside [0, C) makes the loss
 * NaN (F.cross_entropy raises; a kernel cannot) and reads nothing out of bounds.
 * workspace: fp32, at least rpo_head_workspace_floats(B, C, K, e) elements; no initialisation needed.
 * Two launches per training step for class sets up to 128 (logits + norms over B*C blocks; backward with the softmax
 * recomputed per block).  Above 128 classes (e a multiple of 32, 16-byte aligned features) the K pairings run as K small
 * GEMMs on the fp32 matrix pipe, six launches, every feature element read once; their scratch lies inside the same
 * workspace.  Fixed summation orders in both; e <= 1024. */
int64_t rpo_head_workspace_floats(int B, int C, int K, int e);
int rpo_head_fwd_bwd(const float* img_f, const float* text_f, const int64_t* label, float scale_exp,
                     float* logits, float* loss, float* d_img_f, float* d_text_f,
                     int B, int C, int K, int e, float* workspace, void* stream);
/* The same, also leaving act-dtype copies (RPO_BF16 / RPO_F16, round to nearest even) of the two feature gradients for
 * the dX GEMMs of the projections that follow (either may be NULL). */
int rpo_head_fwd_bwd_act(const float* img_f, const float* text_f, const int64_t* label, float scale_exp,
                         float* logits, float* loss, float* d_img_f, float* d_text_f, void* d_img_f_act,
                         void* d_text_f_act, int act_dtype, int B, int C, int K, int e, float* workspace, void* stream);

/* torch.optim.SGD (dampening 0, no nesterov) on n fp32 scalars (trainers/rpo.py:274,309):
 *   g' = grad_scale * g + wd * p;  buf = first ? g' : momentum * buf + g';  p -= lr * buf
 * grad_scale = 1 / world_size after a sum all-reduce. */
int rpo_sgd_step(float* p, const float* g, float* buf, int64_t n, float lr, float momentum, float wd,
                 float grad_scale, int first_step, void* stream);

/* The step as torch.cuda.amp.GradScaler.step takes it (trainers/rpo.py:298-304, PREC "amp"): if any element of g is Inf or
 * NaN, nothing is updated.  found_inf: int32[2] on the device -- [0] = this step's flag, [1] += 1 per skipped step.  Here
 * gradients are fp32 and never scaled, so this skip is all that is left of the scaler.  buf must start at zero (a
 * skipped first step then needs no special case); first_step as in rpo_sgd_step.  One workgroup. */
int rpo_sgd_step_guarded(float* p, const float* g, float* buf, int64_t n, float lr, float momentum, float wd,
                         float grad_scale, int first_step, int32_t* found_inf, void* stream);

/* fp32 -> act dtype copy with leading dimensions (weight packing at load time) */
int rpo_convert(const float* src, int64_t lds, void* dst, int dst_dtype, int64_t ldd,
                int rows, int cols, void* stream);

/* Hardware probe used by the test-suite: one wave runs one MFMA on index-coded operands so the
 * fragment layouts the kernels assume can be checked on the device.  which: 0 = 32x32x16 bf16,
 * 1 = 32x32x2 f32.  a [32, kdim], b [32, kdim] fp32 host-layout inputs (device memory),
 * d [32,32] fp32 receives D[i][j] = sum_k a[i][k] * b[j][k] as the kernels' layout map decodes it. */
int rpo_probe_mfma(int which, const float* a, const float* b, float* d, void* stream);

/* ---- on-device input transforms -------------------------------------------------------
 * Replaces, for a batch of decoded uint8 RGB images of arbitrary sizes, the reference's per-sample CPU transforms
 * selected by configs/trainers/RPO/main_K24.yaml:8-13 (INTERPOLATION bicubic, PIXEL_MEAN/STD, TRANSFORMS
 * random_resized_crop + random_flip + normalize; test: resize + center crop + normalize), whose arithmetic is
 * Pillow's 8-bit bicubic resample reached through Dassl -> torchvision (both un-vendored).  Output is
 * BIT-IDENTICAL to PIL crop -> resize(BICUBIC) -> [crop window] -> [flip] -> ToTensor -> Normalize.
 *
 * One image = crop box (PIL crop), size the crop is resized to, window of the resized image that becomes the
 * size x size output (win = 0,0 and resize = size for the train transform; the center-crop offsets at test time),
 * and a flip flag.  All random decisions are made by the caller (rpo_amd/input_pipeline.py). */
typedef struct rpo_image_desc {
  int64_t src_offset;                      /* byte offset of the image in `src` (HWC uint8 RGB, rows packed) */
  int32_t width, height;                   /* source image size */
  int32_t crop_x, crop_y, crop_w, crop_h;  /* crop box, inside the image */
  int32_t resize_w, resize_h;              /* size the crop is resized to (>= size) */
  int32_t win_x, win_y;                    /* window origin in the resized image */
  int32_t flip;                            /* horizontal flip of the output */
  int32_t reserved;
} rpo_image_desc;

/* number of taps Pillow uses for in_size -> out_size (1 when the pass is skipped); kmax of a batch = max of these */
int rpo_preprocess_ksize(int in_size, int out_size);
/* bytes of caller-owned workspace for B images, output size `size`, crops of at most max_rows rows, kmax taps */
size_t rpo_preprocess_workspace_bytes(int B, int size, int max_rows, int kmax);
/* src: device buffer holding the B images; desc_host/desc_dev: the same B descriptors in host and device memory
 * (the host copy is validated -> RPO_E_SHAPE / RPO_E_WORKSPACE, the device copy is what the kernels read);
 * mean3/std3: host pointers; out: fp32 [B, 3, size, size] device.  Enqueues three kernels on `stream`. */
int rpo_preprocess_batch(const uint8_t* src, int64_t src_bytes, const rpo_image_desc* desc_host,
                         const rpo_image_desc* desc_dev, int B, int size, int max_rows, int kmax,
                         const float* mean3, const float* std3, float* out, void* workspace,
                         size_t workspace_bytes, void* stream);

/* Empirical peaks of the box, used as second denominators by bench.py.
 * rpo_probe_peak_mfma: `blocks` workgroups of 4 waves each run `iters` rounds of 4 independent 32x32 MFMAs
 * (which: 0 = 32x32x16 bf16 on non-zero operands, 1 = 32x32x2 f32, 2 = 32x32x16 bf16 on ZERO operands: the chip
 * clocks to its power budget, so 0 and 2 bracket the sustained and the datasheet rate); *flops receives the flop
 * count of the launch (time it with events).  rpo_probe_peak_copy: 16-B-per-lane grid-stride copy of `bytes` bytes. */
int rpo_probe_peak_mfma(int which, int blocks, int iters, float* sink, double* flops, void* stream);
int rpo_probe_peak_copy(const void* src, void* dst, int64_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#if defined(__GNUC__)
#pragma GCC visibility pop
#endif

/* Measured-and-not-adopted experiments of rounds 3 / 4 (paired launches, the fused MLP launch, the persistent backward
 * chain) are NOT part of this interface: they are declared in rpo_amd_experimental.h and compiled only into the
 * -DRPO_EXPERIMENTAL build of the library (python -m rpo_amd.build --experimental). */
#ifdef RPO_EXPERIMENTAL
#include "rpo_amd_experimental.h"
#endif
#endif /* RPO_AMD_H */
