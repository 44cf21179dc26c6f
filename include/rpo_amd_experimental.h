/*
 * rpo_amd_experimental.h -- entry points of experiments that were built, measured SLOWER than the default step and kept
 * for reproduction only (DESIGN.md sections 11c / 11d; profiles/r03_*, r04_*).  They are compiled into the library only
 * with -DRPO_EXPERIMENTAL (python -m rpo_amd.build --experimental -> rpo_amd/build/librpo_hip_exp.so) and reached from
 * Python only with RPO_EXPERIMENTAL=1.  Nothing the default train step, the eval path, CoOp / CoCoOp or bench.py calls
 * is declared here.  Include through rpo_amd.h.
 */
#ifndef RPO_AMD_EXPERIMENTAL_H
#define RPO_AMD_EXPERIMENTAL_H

/* The library is built with -fvisibility=hidden: only what this header declares is exported. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif
#ifdef __cplusplus
extern "C" {
#endif

/* rpo_gemm_nt for TWO problems in one launch: workgroups [0, tiles of a0) tile problem a0, the rest a1.  The image
 * tower's and the text tower's prompt-row chains (autograd of trainers/rpo.py:308 through clip/model.py:181-207 for the
 * K prompt rows of every image / class) have the same stages; issued pairwise they are ONE chain of launches on one queue.
 * Both problems must be what rpo_gemm_nt runs on its 64x64 tiles: 16-bit inputs of one format, the same out_dtype,
 * epilogue (NONE or QGELU_BWD) and split_k, M < 2048, tile_config 0, no skip_*; otherwise RPO_E_SHAPE / RPO_E_DTYPE and
 * nothing is launched.  Bit-identical to two rpo_gemm_nt calls. */
int rpo_gemm_nt_pair(const rpo_gemm_args* a0, const rpo_gemm_args* a1, void* stream);

/* ---- c_fc -> c_proj of an image block as ONE launch (ABI 6; experiment of round 4, off by default in the engine) --------
 * `fc` / `proj` are the arguments of the two rpo_gemm_nt calls it stands for: x += c_proj(QuickGELU(c_fc(ln_2 x))),
 * clip/model.py:173-177,190 -- fc with a BIAS_QGELU / LN_BIAS_QGELU epilogue, proj with BIAS_RESID and proj->A == fc->C.
 * Applies (else RPO_E_SHAPE: issue the two calls) when both run on the one-round row-unit kernels with the same row units
 * and every workgroup is resident at once (units * 8 <= CUs): a workgroup then runs its c_fc tile, waits for the 7 other
 * workgroups of its row unit at `counters[unit]`, and runs its c_proj tile.  `counters`: units + 1 uint32, zeroed
 * ONCE by the caller (never reset: the kernel counts in rounds of 8; the last word counts polls that gave up after
 * ~10 s -- it must stay 0).  `safe` = 0 relies on the 8 workgroups of a unit
 * sharing an XCD's L2 (what the dispatch order gives today, and only when the unit count is a multiple of 8: otherwise
 * the library switches to the safe form itself); `safe` = 1 adds an agent-scope release / acquire around the hand-off
 * and is placement-independent.  Results are those of the two launches, bit for bit. */
int rpo_mlp_fused(const rpo_gemm_args* fc, const rpo_gemm_args* proj, void* counters, int safe, void* stream);

/* rpo_layernorm_bwd for two problems in one launch (fp32 dy slabs; both casts, where present, of one dtype). */
typedef struct rpo_ln_bwd_args {
  const float* dy; int64_t lddy;            /* fp32 [rows, d], or dy_splits slabs dy + s * dy_split_stride */
  const float* x; int64_t ldx;              /* the forward input, fp32 */
  const float* gamma;
  const float* dres; int64_t lddres;        /* fp32 or NULL */
  float* dx; int64_t lddx;
  void* dx_cast; int32_t cast_dtype; int64_t ldcast;   /* optional copy of dx (NULL: none) */
  int32_t rows, d; float eps;
  int32_t dy_splits; int64_t dy_split_stride;
} rpo_ln_bwd_args;
int rpo_layernorm_bwd_pair(const rpo_ln_bwd_args* a0, const rpo_ln_bwd_args* a1, void* stream);

/* rpo_attn_readonly_bwd_proj for one problem (a1 == NULL) or two problems in ONE launch, each optionally with per-group
 * key counts: group g (an image, or a class of the text tower) reads keys [0, key_len[g]) of the key_stride rows that
 * k / v hold per group -- the text tower's mask (trainers/rpo.py:144-151: causal AND column < len_c) lets a prompt row of
 * class c read exactly the len_c frozen tokens of its class, so the per-class K / V cache [n_cls * Lmax, .] is such a
 * layout; key_len == NULL: every group has `keys` keys stored back to back (the image tower, as
 * rpo_attn_readonly_bwd_proj).  keys <= 288 (the maximum over groups), Kp <= 32, H * 64 in {512, 768}; 16-bit dtypes.
 * Pairs: problem 0 with 97..224 keys and d = 768, problem 1 with <= 96 keys (ViT-B/16 image tower + text tower). */
typedef struct rpo_attn_bwd_args {
  const void* q_rows; int64_t ldq;          /* [groups * Kp, .]: q of the back-propagated rows */
  const void* k; const void* v; int64_t ldkv;
  const void* dx; int64_t lddx;             /* d(out-proj output) of those rows, act dtype [groups * Kp, d] */
  const void* w_out_t; int64_t ldw;         /* out_proj.weight transposed, [d (in), d (out)] */
  void* dq; int64_t lddq;
  int32_t groups, H, keys, Kp;
  const int32_t* key_len; int32_t key_stride;
  float scale;
} rpo_attn_bwd_args;
int rpo_attn_bwd_proj_pair(const rpo_attn_bwd_args* a0, const rpo_attn_bwd_args* a1, int dtype, void* stream);

/* ---- the prompt-row backward chain of one tower as ONE persistent launch (ABI 5) --------------------------------------
 * Replaces, for all `layers` blocks, the six launches per block that autograd's backward through
 * clip/model.py:181-191 (ResidualAttentionBlock) costs for the back-propagated rows of trainers/rpo.py:308:
 *     d c_proj GEMM x QuickGELU'  ->  d c_fc GEMM  ->  LayerNorm (ln_2) backward + residual
 *     ->  d out-proj GEMM  ->  attention backward (dq only: keys / values belong to frozen tokens)
 *     ->  d q-proj GEMM  ->  LayerNorm (ln_1) backward + residual
 * Every back-propagated row depends only on rows of its own unit (an image of the image tower, a class of the text
 * tower), so the units are dealt to 8 groups of `wgs_per_group` workgroups (workgroup b belongs to group b % 8, which
 * the hardware's round-robin places on XCD b % 8) and each group walks the 6 x layers stages on its own rows, exchanging
 * tiles through its XCD's L2 and ONE counter per group instead of a kernel boundary per stage.  The kernel verifies
 * with HW_REG_XCC_ID that a group's workgroups really share an XCD; if not (or RPO_CHAIN_SAFE=1) it adds the agent-scope
 * release (L2 write-back) per hand-off that cross-XCD visibility needs -- results never depend on placement.
 * 16-bit storage only; d = H * 64 in {512, 768}; Kp <= 32; keys <= 224; at most 96 rows per group
 * (units = 8 * m or fewer than 8 ...: ceil(units / 8) * Kp <= 96); the saved QuickGELU operand must be the derivative
 * in the act dtype (rpo_gemm_args.aux_dtype = in_dtype); anything else returns RPO_E_SHAPE and nothing is enqueued.
 * On entry dxa / dxc hold dL/d(output of the last block) of the rows (fp32 / act dtype); on return dxa holds
 * dL/d(input of block 0).  state: caller-owned device scratch of rpo_chain_state_bytes() bytes, zeroed by a memset node
 * this call enqueues; after the launch state[0] != 0 means a bounded spin gave up (results undefined). */
typedef struct rpo_chain_layer {
  const void* w_proj_t;      /* [4d, d]: c_proj.weight^T  (the dX operand of c_proj)                      */
  const void* w_fc_t;        /* [d, 4d]: c_fc.weight^T                                                     */
  const void* w_out_t;       /* [d, d]:  out_proj.weight^T                                                 */
  const void* w_q_t;         /* [d, d]:  in_proj_weight[:d]^T                                              */
  const void* aux;           /* [rows, 4d] act dtype: d quickgelu / du saved by the forward                */
  const float* x_ln2;        /* [rows, ldx] fp32: input of ln_2 in the forward (x + attn)                  */
  const float* x_ln1;        /* [rows, ldx] fp32: input of ln_1 (the block's input)                        */
  const float* ln2_w; const float* ln1_w;   /* LayerNorm gains, fp32 [d]                                   */
  const void* q_rows;        /* q of the back-propagated rows, act dtype, leading dimension ldq            */
  const void* k; const void* v;   /* keys / values of the frozen tokens, leading dimension ldkv            */
} rpo_chain_layer;
typedef struct rpo_chain_bwd_args {
  const rpo_chain_layer* layer;   /* HOST array [layers], block 0 first (the chain walks it backwards)      */
  int32_t layers, units, Kp, d, H, keys, dtype;
  const int32_t* key_len; int32_t key_stride;   /* as rpo_attn_bwd_args (NULL: every unit has `keys` keys)  */
  int64_t ldx, ldq, ldkv;
  float* dxa; float* dxb;    /* fp32 [units * Kp, d]                                                        */
  void* dxc;                 /* act dtype [units * Kp, d]                                                   */
  void* du;                  /* act dtype [units * Kp, 4d]                                                  */
  void* dq;                  /* act dtype [units * Kp, d]                                                   */
  float* dy; int64_t dy_stride;   /* fp32, 4 slabs of [units * Kp, d], dy_stride elements apart             */
  float scale, eps;
  int32_t wgs_per_group;     /* 0 = 32 (one workgroup per CU and chain on a 256-CU part); 32 .. 64               */
  void* state;               /* device scratch, rpo_chain_state_bytes() bytes                               */
  uint64_t* timeline;        /* optional device buffer of >= 1 + 7 * layers entries: s_memrealtime (100 MHz) of
                                workgroup 0 at every stage boundary; NULL = off                            */
} rpo_chain_bwd_args;
size_t rpo_chain_state_bytes(void);
int rpo_chain_bwd(const rpo_chain_bwd_args* args, void* stream);
/* 1 if rpo_chain_bwd covers these sizes (only layers, units, Kp, d, H, keys, dtype are looked at), else 0 */
int rpo_chain_bwd_ok(const rpo_chain_bwd_args* args);

#ifdef __cplusplus
}
#endif
#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#endif /* RPO_AMD_EXPERIMENTAL_H */
