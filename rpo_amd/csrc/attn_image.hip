// Read-only masked attention of the image tower (forward for all rows, backward for the
// prompt rows), MFMA on gfx950.
//
// Reference semantics: nn.MultiheadAttention with the additive mask of
// trainers/rpo.py:154-156 -- the last K *columns* are -inf for every query row, so every
// query (CLS, patches, prompts) reads exactly the N = 1 + patches frozen keys of its own
// image and nothing reads a prompt.  exp(-inf) = 0 exactly, so the prompt columns are
// skipped instead of computed-and-masked.
//
// One workgroup per (image, head).  K and V of the N frozen tokens are staged ONCE into
// LDS (N <= 288, head_dim 64: <= 78 KB in bf16, <= 150 KB in f32) and every wave takes a
// 32-query tile.  The score tile is computed transposed, S^T = K . Q^T, so that by the
// 32x32 C/D map (col = lane&31, row = (reg&3)+8*(reg>>2)+4*(lane>>5)) a lane holds ONE
// query's scores in registers: the softmax row-reduce is an in-register reduction plus a
// single cross-half exchange (shuffle by 32), no LDS.  P stays in registers: its
// register->key order is taken as the contraction order of the P.V MFMAs and V is read
// from LDS in that same order (any k-permutation applied to both operands cancels), so no
// lane movement is needed between softmax and P.V.
//
// bf16: v_mfma_f32_32x32x16_bf16; K rows padded to 144 B.  The P.V (and, in the backward, dS.K) MFMAs
//       contract over KEYS, i.e. need the row-major [key][d] operand transposed.  It is transposed on the
//       matrix core itself: D = M_tile . I (A = 32 rows of M exactly as they sit in memory, B = identity)
//       lands in the C/D layout "lane = d, registers = keys", which after bf16 packing IS the A-operand
//       fragment of M^T in this kernel's key order -- exact, no 2-byte LDS scatter, no bank conflicts.
//       The forward does it once per key tile (one wave each) and parks the fragments in LDS
//       fragment-major (lane-linear 16-B writes/reads); the backward keeps them in registers.
// f32 : v_mfma_f32_32x32x2_f32 (exact f32 fma chain); one float per lane per operand, so
//       K and V stay row-major [Npad][65].
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;

template <typename T, int NT> struct AL {   // LDS layout, 16-bit storage (bf16 / f16)
  static constexpr int NPAD = NT * 32;
  static constexpr int KROW = 144;                 // bytes per K (or row-major V) row: 128 + 16 pad
  static constexpr int K_BYTES = NPAD * KROW;
  static constexpr int F_BYTES = NT * 4 * 1024;                  // V^T fragments: [NT][2 dt][2 g2][64 lanes][16 B]
  static constexpr int PART_BYTES = 4 * 16 * 64 * 16 + 4 * 64 * 4;  // backward cross-wave partials
  static constexpr int FWD_BYTES = K_BYTES + F_BYTES;            // Ks | Vfrag
  static constexpr int BWD_STAGE = 2 * K_BYTES;                  // Ks | Vs
  static constexpr int BWD_BYTES = BWD_STAGE > PART_BYTES ? BWD_STAGE : PART_BYTES;
};
template <int NT> struct AL<float, NT> {
  static constexpr int NPAD = NT * 32;
  static constexpr int ROWF = 65;                  // floats per row
  static constexpr int K_BYTES = NPAD * ROWF * 4;
  static constexpr int PART_BYTES = 4 * 16 * 64 * 16 + 4 * 64 * 4;
  static constexpr int FWD_BYTES = 2 * K_BYTES;    // Ks | Vs
  static constexpr int BWD_STAGE = 2 * K_BYTES;
  static constexpr int BWD_BYTES = BWD_STAGE > PART_BYTES ? BWD_STAGE : PART_BYTES;
};

template <typename T> __device__ __forceinline__ bf16x8_t pack8(const float* p) {
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  u32x4 u;
  u[0] = pack2<T>(p[0], p[1]); u[1] = pack2<T>(p[2], p[3]);
  u[2] = pack2<T>(p[4], p[5]); u[3] = pack2<T>(p[6], p[7]);
  return __builtin_bit_cast(bf16x8_t, u);
}
__device__ __forceinline__ bf16x8_t join8(uint2 lo, uint2 hi) {
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  u32x4 u;
  u[0] = lo.x; u[1] = lo.y; u[2] = hi.x; u[3] = hi.y;
  return __builtin_bit_cast(bf16x8_t, u);
}

// B-operand identity fragment for the k-step pair (ks2 = 0, 1) of a 32-wide d' tile: element jj of lane
// (j = l31, half) is I[k][j] with k = 16*ks2 + 8*half + jj
template <typename T> __device__ __forceinline__ bf16x8_t ident_frag(int ks2, int l31, int half) {
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  u32x4 u;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int k0 = 16 * ks2 + 8 * half + 2 * w;
    u[w] = (k0 == l31 ? One16<T>::lo : 0u) | (k0 + 1 == l31 ? One16<T>::hi : 0u);
  }
  return __builtin_bit_cast(bf16x8_t, u);
}
// rows[4] = the four 16-element A fragments (d = 0..63) of 32 rows of a row-major matrix M; returns the
// A fragments of M^T for d' tile dt: out[g2] covers this kernel's key slots 8*g2 .. 8*g2+7
template <typename T>
__device__ __forceinline__ void transpose_tile(const bf16x8_t (&rows)[4], int dt, const bf16x8_t& i0,
                                               const bf16x8_t& i1, bf16x8_t (&out)[2]) {
  f32x16_t d;
#pragma unroll
  for (int r = 0; r < 16; ++r) d[r] = 0.f;
  d = mfma16<T>(rows[2 * dt], i0, d);
  d = mfma16<T>(rows[2 * dt + 1], i1, d);
  float f[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) f[r] = d[r];
  out[0] = pack8<T>(f);
  out[1] = pack8<T>(f + 8);
}

// ---- staging ---------------------------------------------------------------------------
// rows [0, N) of one (image, head) slice: src + key*ld (already offset to the head), 64 elements.
// All global loads of a pass are issued before the first LDS write (fully unrolled, static
// register indices): a load-use-per-iteration loop serialises one HBM round trip per iteration
// and was the dominant cost of these kernels.
template <int NT, int NTHREADS>
__device__ __forceinline__ void stage_rows_bf16(char* dst, const bf16_t* src, int64_t ld, int N, int tid) {
  constexpr int CHUNKS = NT * 32 * 8;
  constexpr int ITERS = (CHUNKS + NTHREADS - 1) / NTHREADS;
  uint4 v[ITERS];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int id = tid + it * NTHREADS;
    const int key = id >> 3, c = id & 7;
    v[it] = make_uint4(0, 0, 0, 0);
    if (id < CHUNKS && key < N) v[it] = *reinterpret_cast<const uint4*>(src + (int64_t)key * ld + c * 8);
  }
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int id = tid + it * NTHREADS;
    const int key = id >> 3, c = id & 7;
    if (id < CHUNKS) *reinterpret_cast<uint4*>(dst + key * 144 + c * 16) = v[it];
  }
}
// K and V row-major staging with every global load of both matrices in flight before the first LDS write
template <int NT, int NTHREADS>
__device__ __forceinline__ void stage2_bf16(char* dk, char* dv, const bf16_t* sk, const bf16_t* sv, int64_t ld, int N,
                                            int tid) {
  constexpr int CHUNKS = NT * 32 * 8;
  constexpr int ITERS = (CHUNKS + NTHREADS - 1) / NTHREADS;
  uint4 a[ITERS], b[ITERS];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int id = tid + it * NTHREADS;
    const int key = id >> 3, c = id & 7;
    a[it] = make_uint4(0, 0, 0, 0);
    b[it] = make_uint4(0, 0, 0, 0);
    if (id < CHUNKS && key < N) {
      a[it] = *reinterpret_cast<const uint4*>(sk + (int64_t)key * ld + c * 8);
      b[it] = *reinterpret_cast<const uint4*>(sv + (int64_t)key * ld + c * 8);
    }
  }
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int id = tid + it * NTHREADS;
    const int key = id >> 3, c = id & 7;
    if (id < CHUNKS) {
      *reinterpret_cast<uint4*>(dk + key * 144 + c * 16) = a[it];
      *reinterpret_cast<uint4*>(dv + key * 144 + c * 16) = b[it];
    }
  }
}
template <int NT, int NTHREADS>
__device__ __forceinline__ void stage_rows_f32(float* dst, const float* src, int64_t ld, int N, int tid) {
  constexpr int CHUNKS = NT * 32 * 16;
  constexpr int ITERS = (CHUNKS + NTHREADS - 1) / NTHREADS;
  constexpr int BATCH = 7;
#pragma unroll 1
  for (int it0 = 0; it0 < ITERS; it0 += BATCH) {
    float4 v[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int id = tid + (it0 + u) * NTHREADS;
      const int key = id >> 4, c = id & 15;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (id < CHUNKS && key < N) v[u] = *reinterpret_cast<const float4*>(src + (int64_t)key * ld + c * 4);
    }
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int id = tid + (it0 + u) * NTHREADS;
      const int key = id >> 4, c = id & 15;
      if (id < CHUNKS) {
        float* d = dst + key * 65 + c * 4;
        d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
      }
    }
  }
}

// ---- per-lane operand fragments of one 32-row tile loaded straight from global ----------
template <typename T> struct RowFrag {        // 16-bit storage
  bf16x8_t f[4];
  __device__ __forceinline__ void load(const T* rowp, int half) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f[ks] = *reinterpret_cast<const bf16x8_t*>(rowp + ks * 16 + half * 8);
  }
};
template <> struct RowFrag<float> {
  float f[32];  // element half*32 + ks
  __device__ __forceinline__ void load(const float* rowp, int half) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(rowp + half * 32 + 4 * i);
      f[4 * i] = v.x; f[4 * i + 1] = v.y; f[4 * i + 2] = v.z; f[4 * i + 3] = v.w;
    }
  }
};

// one 32x32 tile:  D[i][q] = M[32t + i][:] . frag[q][:]   (M = K or V rows, row-major in LDS)
// SW (16-bit): rows of 128 B without padding, 16-B chunk c of row r at slot c ^ ((r >> 1) & 7) -- the layout an LDS-DMA
// (lane-linear 1 KiB per wave instruction) can write; conflict-free for these reads as in gemm.hip
template <typename T, bool SW = false>
__device__ __forceinline__ f32x16_t tile_times_frag(const char* lds, int t, const RowFrag<T>& fr, int l31, int half) {
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if constexpr (sizeof(T) == 2 && SW) {
    const int row = 32 * t + l31, sw = (row >> 1) & 7;
    const char* rowp = lds + row * 128;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(rowp + (((2 * ks + half) ^ sw) << 4));
      acc = mfma16<T>(a, fr.f[ks], acc);
    }
  } else if constexpr (sizeof(T) == 2) {
    const char* rowp = lds + (32 * t + l31) * 144 + half * 16;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(rowp + ks * 32);
      acc = mfma16<T>(a, fr.f[ks], acc);
    }
  } else {
    const float* rowp = reinterpret_cast<const float*>(lds) + (32 * t + l31) * 65 + half * 32;
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(rowp[ks], fr.f[ks], acc, 0, 0, 0);
  }
  return acc;
}

// Same product with the frag operand given in "C/D order": cd[ks] = pack8 of registers 8*(ks&1) .. +7 of the 32x32 MFMA
// result tile ks>>1 whose ROWS are this contraction's index, i.e. lane (q, half) holds elements
// j = 16*ks + 4*half + {0..3} and 16*ks + 8 + 4*half + {0..3}.  A k-order applied to both operands cancels, so the LDS
// rows are read as the same two 8-byte pieces -- no lane exchange between the MFMA that produced cd and this one.
template <typename T>
__device__ __forceinline__ f32x16_t tile_times_cdfrag(const char* lds, int t, const bf16x8_t (&cd)[4], int l31, int half) {
  typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const char* rowp = lds + (32 * t + l31) * 144 + half * 8;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const u32x2 lo = *reinterpret_cast<const u32x2*>(rowp + ks * 32), hi = *reinterpret_cast<const u32x2*>(rowp + ks * 32 + 16);
    u32x4 a; a[0] = lo[0]; a[1] = lo[1]; a[2] = hi[0]; a[3] = hi[1];
    acc = mfma16<T>(__builtin_bit_cast(bf16x8_t, a), cd[ks], acc);
  }
  return acc;
}

// keys >= N of tile t -> -inf (only tiles that can hold padding pay for the compare), then the tile's row
// maximum over both lane halves.  A lane holds one query's scores for 16 of the tile's 32 keys.
__device__ __forceinline__ float mask_and_max(f32x16_t& s, int t, int N, int half) {
  if (32 * t + 32 > N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
      s[r] = key < N ? s[r] : -INFINITY;
    }
  }
  float m = s[0];
#pragma unroll
  for (int r = 1; r < 16; ++r) m = fmaxf(m, s[r]);
  return fmaxf(m, __shfl_xor(m, 32, 64));
}

// s <- exp((s - m) * scale) element-wise; returns this lane's partial row sum.
// bf16 path: one fma + raw v_exp_f32 per element (arguments <= 0: no range fix-up needed).
template <typename T>
__device__ __forceinline__ float exp_tile(f32x16_t& s, float m, float scale) {
  float l = 0.f;
  if constexpr (sizeof(T) == 2) {
    const float c = scale * LOG2E, mc = m * c;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c, -mc)); l += s[r]; }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = expf((s[r] - m) * scale); l += s[r]; }
  }
  return l;
}
template <typename T> __device__ __forceinline__ float exp_scalar(float d, float scale) {
  if constexpr (sizeof(T) == 2) return __builtin_amdgcn_exp2f(d * scale * LOG2E);
  else return expf(d * scale);
}

// o[dt] += M^T-contraction:  D[d][q] += sum_key M[key][d] * w[q][key], keys of tile t.
// bf16: M^T fragments parked fragment-major in LDS (frag, [NT][2][2][64][16 B]); f32: M row-major (ms, [NPAD][65]).
template <typename T, int NT>
__device__ __forceinline__ void contract_keys(const char* m_lds, int t, const f32x16_t& w, f32x16_t (&o)[2],
                                              int l31, int half) {
  if constexpr (sizeof(T) == 2) {
    float wf[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) wf[r] = w[r];
    const int lane = l31 + 32 * half;
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
      const bf16x8_t b = pack8<T>(wf + 8 * g2);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(m_lds + (((t * 2 + dt) * 2 + g2) * 64 + lane) * 16);
        o[dt] = mfma16<T>(a, b, o[dt]);
      }
    }
  } else {
    const float* ms = reinterpret_cast<const float*>(m_lds);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ms[key * 65 + 32 * dt + l31], w[r], o[dt], 0, 0, 0);
    }
  }
}

// ---- forward, 16-bit storage, round 6: the key loop ---------------------------------------------------------------------
// What a wave does per key tile is one dependent chain (4 score MFMAs on one accumulator -> max -> cross-half exchange ->
// exp -> rescale of the output tile -> 4 P.V MFMAs) and the workgroup's time is staging + NT of them
// (profiles/r06_attn_timeline.txt: 4.3-8.9 k cycles of staging, 9.5-12 k of loop, with ONE or TWO workgroups on the CU alike).
//  * THE PARTIAL KEY TILE (N = 197 = 6 x 32 + 5; 257 = 8 x 32 + 1) is peeled: by the C/D map a lane's registers 4g .. 4g+3
//    are key offsets 8g + 4 half + {0..3}, so with rem valid keys only G = ceil(rem / 8) register groups are live -- the
//    exps, sums and conversions of the others and the P.V MFMAs of a 16-key half without live keys are not issued
//    (rem = 5: 4 of 16 v_exp, 2 of 4 P.V MFMAs); key tiles past N (N < 32 (NT - 1)) are skipped.  What is skipped is
//    arithmetic on exact zeros (exp2(-inf) = 0, 0 x v, alpha = 1), so the result is BIT-IDENTICAL to the round-5 kernel.
//  * -DRPO_ATTN_LAZY (A/B build, NOT the default): lazy rescale -- the running maximum m as a reference point that is moved
//    (and l and the output tile rescaled: 18 multiplies + one v_exp) only when some row of the wave has outgrown it by more
//    than TAU = 8 in log2 units.  Measured (profiles/r06_attn_variants.txt): kernel 15.6 vs 16.0 us, step -0.4 %; but the
//    largest weight of a row is then no longer exactly 1.0 in the 16-bit P, and the op's max abs error grows 1.7x (bf16
//    6.6e-3 vs 3.8e-3 on |out| <= 1.3; model-level logits error unchanged) -- not worth a looser op-level bound.
#ifdef RPO_ATTN_LAZY
constexpr float ATTN_TAU = 8.0f;
#endif

// moves the running maximum m to cover this tile's row maximum tm; returns the factor the running sum / output carry
template <typename T>
__device__ __forceinline__ float move_max(float tm, float scale, float& m, f32x16_t (&o)[2]) {
#ifdef RPO_ATTN_LAZY
  if (!__any((tm - m) * (scale * LOG2E) > ATTN_TAU)) return 1.0f;      // wave-uniform; (tm - m) = +inf on the first tile
#endif
  const float mn = fmaxf(m, tm);                            // finite from the first tile on (N >= 1)
  const float alpha = exp_scalar<T>(m - mn, scale);         // first tile: exp(-inf) = 0
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
  m = mn;
  return alpha;
}

// the PARTIAL key tile: rem = N - 32 t valid keys, G = ceil(rem / 8) live register groups of four (wave-uniform branches:
// one code path for every G -- four template instances of it spilled 35 registers around a switch)
template <typename T, int NT, bool SW = false>
__device__ __forceinline__ void fwd_tile_tail(const char* ks, const char* vs, int t, int rem, const RowFrag<T>& qf,
                                              float scale, float& m, float& l, f32x16_t (&o)[2], int l31, int half) {
  f32x16_t sc = tile_times_frag<T, SW>(ks, t, qf, l31, half);  // (K rows >= N hold zeros / a copy of row N-1: finite scores)
  const int G = (rem + 7) >> 3;
  float tm = -INFINITY;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (g < G) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * g + j, off = j + 8 * g + 4 * half;
        sc[r] = off < rem ? sc[r] : -INFINITY;
        tm = fmaxf(tm, sc[r]);
      }
    }
  }
  tm = fmaxf(tm, __shfl_xor(tm, 32, 64));                      // finite: offset 0 is valid and belongs to half 0
  const float alpha = move_max<T>(tm, scale, m, o);
  const float c = scale * LOG2E, mc = m * c;
  float p[16], lt = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) p[r] = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (g < G) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {                            // exp2(-inf) = 0 for the masked offsets of the last group
        p[4 * g + j] = __builtin_amdgcn_exp2f(fmaf(sc[4 * g + j], c, -mc));
        lt += p[4 * g + j];
      }
    }
  }
  l = l * alpha + lt;
  const int lane = l31 + 32 * half;
#pragma unroll
  for (int g2 = 0; g2 < 2; ++g2) {
    if (2 * g2 < G) {
      const bf16x8_t b = pack8<T>(p + 8 * g2);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(vs + (((t * 2 + dt) * 2 + g2) * 64 + lane) * 16);
        o[dt] = mfma16<T>(a, b, o[dt]);
      }
    }
  }
}

// ---- forward ---------------------------------------------------------------------------
template <typename T, int NT, bool TWO_PASS = false>
__global__ __launch_bounds__(512, 4) void attn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                       const T* __restrict__ v, int64_t ld, T* out,
                                                       int64_t ldo, int B, int H, int N, int Kp, float scale,
                                                       int q_first, int split_from) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using L = AL<T, NT>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  // Block bid runs on XCD bid % 8: every XCD takes a contiguous run of (image, head) pairs, i.e. whole images -- the
  // images whose rows the out-proj / c_fc / c_proj kernels around this one handle on the same XCD (row units).
  // split_from (round 4): the first `split_from` workgroups take one (image, head) unit each, the rest take HALF the query
  // tiles of a unit (two workgroups per unit, each staging K / V itself) -- the launcher splits just enough units for
  // the grid to fill both resident slots of every CU (B = 32: 256 whole + 2 x 128 halves = 512 workgroups), so that no
  // CU hosts two whole units while another hosts one.
  int unit, qpart = -1;
  {
    const int bid = blockIdx.x;
#ifdef RPO_ATTN_PLAIN_ORDER
    unit = bid;
#else
    auto remap = [&](int i, int n) {
      const int qd = n >> 3, rm = n & 7, xcd = i & 7;
      return (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (i >> 3);
    };
    if (bid < split_from) {
      unit = remap(bid, split_from);
    } else {
      const int j = remap(bid - split_from, (int)gridDim.x - split_from);
      unit = split_from + (j >> 1);
      qpart = j & 1;
    }
#endif
  }
  const int wgid = unit;
  const int b = wgid / H, h = wgid % H;
  const T* kb = k + (int64_t)b * N * ld + h * 64;
  const T* vb = v + (int64_t)b * N * ld + h * 64;
  char* ks = smem;
  char* vs = smem + L::K_BYTES;
  const int S = N + Kp;
  RPO_STAMP(0);
  // every HBM round trip of the prologue is issued up front: this wave's first query fragment, its V rows
  // (bf16: the key tile it transposes), then the K staging loads
  auto qrow = [&](int qt) -> int64_t {
    const int sc = min(qt * 32 + l31, S - 1);
    return sc < N ? (int64_t)b * N + sc : (int64_t)B * N + (int64_t)b * Kp + (sc - N);
  };
  RowFrag<T> qf;
  // q_first > 0: only queries q_first .. S-1 of every image are wanted (last block: the prompt rows); whole
  // 32-query tiles are skipped, the leading queries of the first computed tile are computed but not stored
  int qt0 = q_first >> 5, qt_end = (S + 31) >> 5;
  if (qpart >= 0) {                                // this workgroup's half of the query tiles
    const int mid = qt0 + ((qt_end - qt0 + 1) >> 1);
    if (qpart == 0) qt_end = mid; else qt0 = mid;
  }
  if constexpr (sizeof(T) == 2) qf.load(q + qrow(qt0 + wave) * ld + h * 64, half);   // (f32: 32 live VGPRs too many)
  if constexpr (sizeof(T) == 2) {
    bf16x8_t vrows[(NT + 7) / 8][4];
#pragma unroll
    for (int ti = 0; ti < (NT + 7) / 8; ++ti) {
      const int key = 32 * (wave + 8 * ti) + l31;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        uint4 z = make_uint4(0, 0, 0, 0);
        if (wave + 8 * ti < NT && key < N) z = *reinterpret_cast<const uint4*>(vb + (int64_t)key * ld + kk * 16 + half * 8);
        vrows[ti][kk] = __builtin_bit_cast(bf16x8_t, z);
      }
    }
    stage_rows_bf16<NT, 512>(ks, reinterpret_cast<const bf16_t*>(kb), ld, N, tid);
    RPO_STAMP(1);
    // V^T fragments: wave w transposes key tiles w, w+8 on the matrix core
    const bf16x8_t i0 = ident_frag<T>(0, l31, half), i1 = ident_frag<T>(1, l31, half);
#pragma unroll
    for (int ti = 0; ti < (NT + 7) / 8; ++ti) {
      const int t = wave + 8 * ti;
      if (t < NT) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          bf16x8_t fr[2];
          transpose_tile<T>(vrows[ti], dt, i0, i1, fr);
#pragma unroll
          for (int g2 = 0; g2 < 2; ++g2)
            *reinterpret_cast<bf16x8_t*>(vs + (((t * 2 + dt) * 2 + g2) * 64 + lane) * 16) = fr[g2];
        }
      }
    }
  } else {
    stage_rows_f32<NT, 512>(reinterpret_cast<float*>(ks), kb, ld, N, tid);
    stage_rows_f32<NT, 512>(reinterpret_cast<float*>(vs), vb, ld, N, tid);
  }
  RPO_STAMP(2);
  __syncthreads();
  RPO_STAMP(3);
  for (int qt = qt0 + wave; qt < qt_end; qt += 8) {
    // K/V fragments in LDS do not depend on the query tile; without this opaque copy the
    // compiler hoists all of their ds_reads out of the loop and spills them to scratch
    int l31v = l31;
    asm volatile("" : "+v"(l31v));
    const int s = qt * 32 + l31;
    const int64_t grow = qrow(qt);
    if (sizeof(T) != 2 || qt != qt0 + wave) qf.load(q + grow * ld + h * 64, half);
    // online softmax over the key tiles (running max m, running sum l, output rescaled when m grows): only one
    // score tile is live, so the kernel fits 128 VGPRs and TWO workgroups share a CU -- the 384 (image, head)
    // workgroups of a B=32 launch are then all resident at once instead of running in two rounds.
    float m = -INFINITY, l = 0.f;
    f32x16_t o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    if constexpr (TWO_PASS) {
      // EXPERIMENT (round 4): exact two-pass softmax at the SAME register budget (two workgroups per CU): pass 1
      // recomputes nothing but the row maximum (4 MFMAs + 10 VALU per key tile, one cross-half exchange per QUERY tile),
      // pass 2 recomputes the scores and has no running maximum, no rescale of the output tile.  VALU per key tile
      // 115 -> ~75, MFMAs 8 -> 12.
#pragma unroll 1
      for (int t = 0; t < NT; ++t) {
        f32x16_t sc = tile_times_frag<T>(ks, t, qf, l31v, half);
        if (32 * t + 32 > N) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
            sc[r] = key < N ? sc[r] : -INFINITY;
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, sc[r]);
      }
      m = fmaxf(m, __shfl_xor(m, 32, 64));
#pragma unroll 1
      for (int t = 0; t < NT; ++t) {
        f32x16_t sc = tile_times_frag<T>(ks, t, qf, l31v, half);
        if (32 * t + 32 > N) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
            sc[r] = key < N ? sc[r] : -INFINITY;
          }
        }
        l += exp_tile<T>(sc, m, scale);
        contract_keys<T, NT>(vs, t, sc, o, l31v, half);
      }
    } else
#ifndef RPO_ATTN_OLD_LOOP           // (A/B build: the round-2..5 loop below for the 16-bit modes too)
    if constexpr (sizeof(T) == 2) {
      // round 6 (see fwd_tile_tail): full key tiles as before, the partial one peeled, key tiles past N skipped
      const int nfull = min(N >> 5, NT), rem = N - 32 * nfull;
      int t0 = 0;
#ifdef RPO_ATTN_TPI2
      // A/B build: TWO key tiles per trip -- independent score accumulators, one row maximum / exchange / rescale per 64
      // keys: 4 dependent chains per query tile instead of 7 (not bit-identical: the maximum moves in other steps)
#pragma unroll 1
      for (; t0 + 1 < nfull; t0 += 2) {
        f32x16_t sa = tile_times_frag<T>(ks, t0, qf, l31v, half);
        f32x16_t sb = tile_times_frag<T>(ks, t0 + 1, qf, l31v, half);
        float tm = fmaxf(sa[0], sb[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) tm = fmaxf(tm, fmaxf(sa[r], sb[r]));
        tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
        const float alpha = move_max<T>(tm, scale, m, o);
        l = l * alpha + (exp_tile<T>(sa, m, scale) + exp_tile<T>(sb, m, scale));
        contract_keys<T, NT>(vs, t0, sa, o, l31v, half);
        contract_keys<T, NT>(vs, t0 + 1, sb, o, l31v, half);
      }
#endif
#pragma unroll 1
      for (int t = t0; t < nfull; ++t) {
        f32x16_t sc = tile_times_frag<T>(ks, t, qf, l31v, half);
        float tm = sc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tm = fmaxf(tm, sc[r]);
        tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
        const float alpha = move_max<T>(tm, scale, m, o);
        l = l * alpha + exp_tile<T>(sc, m, scale);
        contract_keys<T, NT>(vs, t, sc, o, l31v, half);
      }
      if (rem > 0 && nfull < NT) fwd_tile_tail<T, NT>(ks, vs, nfull, rem, qf, scale, m, l, o, l31v, half);
    } else
#endif
#pragma unroll 1
    for (int t = 0; t < NT; ++t) {
      // (issuing the score tile of key tile t+1 before the softmax arithmetic of tile t -- one more live tile, 118 VGPRs --
      //  measured nothing: loop 9.5 k vs 9.7 k cycles per workgroup, in-step 14.7 vs 13.6 us, step 3.008 vs 3.009 ms)
      f32x16_t sc = tile_times_frag<T>(ks, t, qf, l31v, half);
      const float mn = fmaxf(m, mask_and_max(sc, t, N, half));     // finite from tile 0 on (N >= 1)
      const float alpha = exp_scalar<T>(m - mn, scale);           // first tile: exp(-inf) = 0
      l = l * alpha + exp_tile<T>(sc, mn, scale);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
      contract_keys<T, NT>(vs, t, sc, o, l31v, half);
      m = mn;
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    if (s < S && s >= q_first) {
      T* orow = out + grow * ldo + h * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          ActIO<T>::st4(orow + 32 * dt + 8 * g + 4 * half, o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv,
                        o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
    }
    RPO_STAMP(6);
  }
#ifdef RPO_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  RPO_STAMP(7);
}

// ---- forward, 16-bit storage, round 6: the key loop STARTS while the tail of K is still in flight ------------------------
// profiles/r06_attn_timeline.txt: of a workgroup's 16.6 k (one per CU) .. 25 k (two per CU) cycles, 4.3 .. 8.9 k pass before
// the staging barrier -- all 384 workgroups of a launch pull their 57 KB of K / V at once, ~41 MB at the rate the fabric
// delivers -- and the key loop (7 dependent chains of ~1.4 k cycles) cannot start before the LAST byte has landed.  Here:
//  * K goes HBM -> LDS by DMA (global_load_lds_dwordx4: no staging registers; rows of 128 B, XOR-swizzled chunks as in
//    gemm.hip instead of 144-B padded rows), every request of the prologue issued up front in the order q fragment, V rows,
//    K rows [0, 128) (phase A), the rest of K (phase B);
//  * the waves retire them with COUNTED waits: q + V -> V^T fragments on the matrix core; phase A -> barrier A -> the key
//    loop over key tiles 0 .. 3; phase B -> barrier B -> the remaining tiles.  Left to hipcc the V loads sink into the
//    `tile < NT` branch that uses them and every wait merges to vmcnt(0) (ISA of the first version) -- hence inline asm
//    for the issue and the waits; the compiler issues no vector-memory instruction of its own before barrier B.
//  * 104 VGPRs as in round 5 (a register-staged phase B: 114): two workgroups per CU AND room for the side queue's waves
//    (same-box step time by VGPR budget of this kernel: profiles/r06_attn_variants.txt).
// Arithmetic per query row is that of attn_fwd_kernel: bit-identical results.
template <typename T, int NT, bool DMA> struct AL16 {
  static constexpr int NPAD = NT * 32;
  static constexpr int K_BYTES = NPAD * (DMA ? 128 : 144);       // DMA: swizzled, unpadded rows; else rows padded to 144 B
  static constexpr int F_BYTES = NT * 4 * 1024;                  // V^T fragments
  static constexpr int BYTES = K_BYTES + F_BYTES;
  static constexpr int NDMA = (NPAD / 8 + 7) / 8;                // K requests per wave / thread (= ceil(NPAD * 8 / 512))
};

template <typename T, int NT, bool DMA>
__global__ __launch_bounds__(512, 4) void attn_fwd16_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                         const T* __restrict__ v, int64_t ld, T* out,
                                                         int64_t ldo, int B, int H, int N, int Kp, float scale,
                                                         int q_first) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  extern __shared__ __attribute__((aligned(128))) char smem[];
  using L = AL16<T, NT, DMA>;
  // K requests per thread / wave: DMA instructions of 8 rows (wave w: rows 8 (w + 8 i) ..), or 16-B register loads
  // (thread t: chunk t + 512 i); phase A = the first PH of them = K rows [0, 128) either way
  constexpr int NDMA = L::NDMA, PH = 2;
  static_assert(NDMA > PH && NDMA <= 5, "the counted waits below are written out for 3 .. 5 requests");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  int unit;
  {   // block bid runs on XCD bid % 8: every XCD takes a contiguous run of (image, head) pairs, i.e. whole images
    const int i = blockIdx.x, n = gridDim.x;
    const int qd = n >> 3, rm = n & 7, xcd = i & 7;
    unit = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (i >> 3);
  }
  const int b = unit / H, h = unit % H;
  const T* kb = k + (int64_t)b * N * ld + h * 64;
  const T* vb = v + (int64_t)b * N * ld + h * 64;
  char* ks = smem;
  char* vs = smem + L::K_BYTES;
  const int S = N + Kp;
  RPO_STAMP(0);
  auto qrow = [&](int qt) -> int64_t {
    const int sc = min(qt * 32 + l31, S - 1);
    return sc < N ? (int64_t)b * N + sc : (int64_t)B * N + (int64_t)b * Kp + (sc - N);
  };
  const int qt0 = q_first >> 5, qt_end = (S + 31) >> 5;
  const int qt_mine = qt0 + wave;
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  constexpr int NVT = (NT + 7) / 8;
  RowFrag<T> qf;
  u32x4 qreg[4], vreg[NVT][4], kreg[DMA ? 1 : NDMA];
  {
    const T* qp = q + qrow(qt_mine) * ld + h * 64 + half * 8;      // (rows are clamped: waves without a tile load a valid row)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qreg[kk]) : "v"(qp + kk * 16) : "memory");
#pragma unroll
    for (int ti = 0; ti < NVT; ++ti) {
      const T* vp = vb + (int64_t)min(32 * (wave + 8 * ti) + l31, N - 1) * ld + half * 8;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(vreg[ti][kk]) : "v"(vp + kk * 16) : "memory");
    }
    // K: DMA instruction j of the workgroup = rows [8 j, 8 j + 8); wave w issues j = w, w + 8, ...; lane l -> row 8 j + (l >> 3),
    // physical 16-B slot l & 7, which must receive logical chunk (l & 7) ^ ((row >> 1) & 7).  Rows >= N re-read row N - 1
    // (their scores are masked or never formed); an instruction past the last one repeats the last one (same bytes).
    if constexpr (DMA) {
      const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ks;
#pragma unroll
      for (int i = 0; i < NDMA; ++i) {
        const int j = min(wave + 8 * i, L::NPAD / 8 - 1);
        const int row = 8 * j + (lane >> 3);
        const T* kp = kb + (int64_t)min(row, N - 1) * ld + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
        const uint32_t dst = lds0 + (uint32_t)j * 1024u;           // (uniform)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(kp), "s"(dst) : "memory", "m0");
      }
    } else {
#pragma unroll
      for (int i = 0; i < NDMA; ++i) {
        const int id = tid + i * 512;
        const T* kp = kb + (int64_t)min(id >> 3, N - 1) * ld + (id & 7) * 8;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(kreg[i]) : "v"(kp) : "memory");
      }
    }
  }
  // (register path) chunk id = tid + 512 i of the [NPAD][8]-chunk K panel -> its padded LDS row; rows >= N are zeroed
  auto commit_k = [&](int i) {
    if constexpr (!DMA) {
      const int id = tid + i * 512;
      const int key = id >> 3, c = id & 7;
      const u32x4 z = {0u, 0u, 0u, 0u};
      if (id < L::NPAD * 8) *reinterpret_cast<u32x4*>(ks + key * 144 + c * 16) = key < N ? kreg[i] : z;
    }
  };
  // q and V have landed once at most the NDMA K requests are outstanding (returns are in order)
  if constexpr (NVT == 1)
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(qreg[0]), "+v"(qreg[1]), "+v"(qreg[2]), "+v"(qreg[3]), "+v"(vreg[0][0]),
                 "+v"(vreg[0][1]), "+v"(vreg[0][2]), "+v"(vreg[0][3]) : "n"(NDMA) : "memory");
  else
    asm volatile("s_waitcnt vmcnt(%12)" : "+v"(qreg[0]), "+v"(qreg[1]), "+v"(qreg[2]), "+v"(qreg[3]), "+v"(vreg[0][0]),
                 "+v"(vreg[0][1]), "+v"(vreg[0][2]), "+v"(vreg[0][3]), "+v"(vreg[NVT - 1][0]), "+v"(vreg[NVT - 1][1]),
                 "+v"(vreg[NVT - 1][2]), "+v"(vreg[NVT - 1][3]) : "n"(NDMA) : "memory");
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) qf.f[kk] = __builtin_bit_cast(bf16x8_t, qreg[kk]);
  // V^T fragments: wave w transposes key tiles w, w + 8 on the matrix core (rows >= N contribute zeros)
  {
    const bf16x8_t i0 = ident_frag<T>(0, l31, half), i1 = ident_frag<T>(1, l31, half);
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int ti = 0; ti < NVT; ++ti) {
      const int t = wave + 8 * ti;
      if (t < NT) {
        bf16x8_t vrows[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) vrows[kk] = __builtin_bit_cast(bf16x8_t, 32 * t + l31 < N ? vreg[ti][kk] : z);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          bf16x8_t fr[2];
          transpose_tile<T>(vrows, dt, i0, i1, fr);
#pragma unroll
          for (int g2 = 0; g2 < 2; ++g2)
            *reinterpret_cast<bf16x8_t*>(vs + (((t * 2 + dt) * 2 + g2) * 64 + lane) * 16) = fr[g2];
        }
      }
    }
  }
  RPO_STAMP(1);
  if constexpr (DMA) {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA - PH) : "memory");   // this wave's phase-A rows are in LDS
  } else {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(kreg[0]) : "n"(NDMA - 1) : "memory");
    commit_k(0);
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(kreg[DMA ? 0 : 1]) : "n"(NDMA - 2) : "memory");
    commit_k(1);
  }
  RPO_STAMP(2);
  __syncthreads();                                               // barrier A: V^T of every tile, K rows [0, 128)
  RPO_STAMP(3);
  const int nfull = min(N >> 5, NT), rem = N - 32 * nfull;
  const int nA = min(nfull, 2 * PH);                             // full key tiles inside phase A's rows
  float m = -INFINITY, l = 0.f;
  f32x16_t o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  auto tiles = [&](int ta, int tb, int l31v) {
#pragma unroll 1
    for (int t = ta; t < tb; ++t) {
      f32x16_t sc = tile_times_frag<T, DMA>(ks, t, qf, l31v, half);
      float tm = sc[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) tm = fmaxf(tm, sc[r]);
      tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
      const float alpha = move_max<T>(tm, scale, m, o);
      l = l * alpha + exp_tile<T>(sc, m, scale);
      contract_keys<T, NT>(vs, t, sc, o, l31v, half);
    }
  };
  auto finish = [&](int qt, int l31v) {
    if (rem > 0 && nfull < NT) fwd_tile_tail<T, NT, DMA>(ks, vs, nfull, rem, qf, scale, m, l, o, l31v, half);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    const int s = qt * 32 + l31;
    if (s < S && s >= q_first) {
      T* orow = out + qrow(qt) * ldo + h * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          ActIO<T>::st4(orow + 32 * dt + 8 * g + 4 * half, o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv,
                        o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
    }
  };
  // K / V fragments in LDS do not depend on the query tile; without this opaque copy the compiler hoists their
  // ds_reads out of the loops and spills them
  int l31v = l31;
  asm volatile("" : "+v"(l31v));
  const bool have = qt_mine < qt_end;                            // (wave-uniform)
  if (have) tiles(0, nA, l31v);
  RPO_STAMP(4);
  if constexpr (DMA) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's phase-B rows are in LDS
  } else {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(kreg[DMA ? 0 : 2]) : "n"(NDMA - 3) : "memory");
    commit_k(2);
    if constexpr (NDMA > 3) {
      asm volatile("s_waitcnt vmcnt(%1)" : "+v"(kreg[DMA ? 0 : (NDMA > 3 ? 3 : 0)]) : "n"(NDMA > 3 ? NDMA - 4 : 0) : "memory");
      commit_k(3);
    }
    if constexpr (NDMA > 4) {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(kreg[DMA ? 0 : NDMA - 1]) : : "memory");
      commit_k(NDMA - 1);
    }
  }
  __syncthreads();                                               // barrier B: the rest of K
  RPO_STAMP(5);
  if (have) {
    tiles(nA, nfull, l31v);
    finish(qt_mine, l31v);
  }
  RPO_STAMP(6);
  for (int qt = qt_mine + 8; qt < qt_end; qt += 8) {             // (S > 256: ViT-L/14's ninth query tile)
    asm volatile("" : "+v"(l31v));
    qf.load(q + qrow(qt) * ld + h * 64, half);
    m = -INFINITY; l = 0.f;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    tiles(0, nfull, l31v);
    finish(qt, l31v);
  }
#ifdef RPO_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  RPO_STAMP(7);
}

// ---- forward, 16-bit storage, round 4: every score tile of a query tile in registers, exact two-pass softmax ----------
// The online-softmax loop above is one dependent chain per key tile (4 score MFMAs on one accumulator -> max -> cross-half
// exchange -> exp -> rescale of the output tile -> 4 P.V MFMAs): ~1.4 k cycles per key tile for a wave, 9.7 k per
// workgroup (profiles/r03_attn_timeline.txt), whatever else the CU does.  With N <= 288 keys a lane's whole score row is
// only NT x 16 registers, so here a wave first issues ALL NT score tiles (independent accumulators: the MFMAs pipeline),
// takes the row maximum ONCE (one cross-half exchange per query tile instead of one per key tile), then exponentiates
// and contracts tile by tile with nothing to rescale.  ~200 VGPRs: one 8-wave workgroup per CU instead of two.
template <typename T, int NT>
__global__ __launch_bounds__(512, 2) void attn_fwd2_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                        const T* __restrict__ v, int64_t ld, T* out,
                                                        int64_t ldo, int B, int H, int N, int Kp, float scale,
                                                        int q_first) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using L = AL<T, NT>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wgid = [&] {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7;
    return (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
  }();
  const int b = wgid / H, h = wgid % H;
  const T* kb = k + (int64_t)b * N * ld + h * 64;
  const T* vb = v + (int64_t)b * N * ld + h * 64;
  char* ks = smem;
  char* vs = smem + L::K_BYTES;
  const int S = N + Kp;
  RPO_STAMP(0);
  auto qrow = [&](int qt) -> int64_t {
    const int sc = min(qt * 32 + l31, S - 1);
    return sc < N ? (int64_t)b * N + sc : (int64_t)B * N + (int64_t)b * Kp + (sc - N);
  };
  RowFrag<T> qf;
  const int qt0 = q_first >> 5;
  qf.load(q + qrow(qt0 + wave) * ld + h * 64, half);
  {
    bf16x8_t vrows[(NT + 7) / 8][4];
#pragma unroll
    for (int ti = 0; ti < (NT + 7) / 8; ++ti) {
      const int key = 32 * (wave + 8 * ti) + l31;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        uint4 z = make_uint4(0, 0, 0, 0);
        if (wave + 8 * ti < NT && key < N) z = *reinterpret_cast<const uint4*>(vb + (int64_t)key * ld + kk * 16 + half * 8);
        vrows[ti][kk] = __builtin_bit_cast(bf16x8_t, z);
      }
    }
    stage_rows_bf16<NT, 512>(ks, reinterpret_cast<const bf16_t*>(kb), ld, N, tid);
    RPO_STAMP(1);
    const bf16x8_t i0 = ident_frag<T>(0, l31, half), i1 = ident_frag<T>(1, l31, half);
#pragma unroll
    for (int ti = 0; ti < (NT + 7) / 8; ++ti) {
      const int t = wave + 8 * ti;
      if (t < NT) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          bf16x8_t fr[2];
          transpose_tile<T>(vrows[ti], dt, i0, i1, fr);
#pragma unroll
          for (int g2 = 0; g2 < 2; ++g2)
            *reinterpret_cast<bf16x8_t*>(vs + (((t * 2 + dt) * 2 + g2) * 64 + lane) * 16) = fr[g2];
        }
      }
    }
  }
  RPO_STAMP(2);
  __syncthreads();
  RPO_STAMP(3);
  const float c = scale * LOG2E;
  for (int qt = qt0 + wave; qt * 32 < S; qt += 8) {
    int l31v = l31;
    asm volatile("" : "+v"(l31v));
    const int s = qt * 32 + l31;
    const int64_t grow = qrow(qt);
    if (qt != qt0 + wave) qf.load(q + grow * ld + h * 64, half);
    // pass 1: all score tiles (S^T = K . Q^T: a lane holds 16 of the 32 keys of every tile for ONE query)
    f32x16_t sc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) sc[t] = tile_times_frag<T>(ks, t, qf, l31v, half);
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (32 * t + 32 > N) {                       // (uniform) keys >= N of a tile that can hold padding -> -inf
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
          sc[t][r] = key < N ? sc[t][r] : -INFINITY;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, sc[t][r]);
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));           // finite: tile 0 always holds keys
    RPO_STAMP(4);
    // pass 2: p = exp((s - m) * scale), row sum, O += V^T-contraction -- nothing to rescale
    const float mc = m * c;
    float l = 0.f;
    f32x16_t o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { sc[t][r] = __builtin_amdgcn_exp2f(fmaf(sc[t][r], c, -mc)); l += sc[t][r]; }
      contract_keys<T, NT>(vs, t, sc[t], o, l31v, half);
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    RPO_STAMP(5);
    if (s < S && s >= q_first) {
      T* orow = out + grow * ldo + h * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          ActIO<T>::st4(orow + 32 * dt + 8 * g + 4 * half, o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv,
                        o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
    }
    RPO_STAMP(6);
  }
#ifdef RPO_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  RPO_STAMP(7);
}

// ---- backward (prompt rows only; dq) -----------------------------------------------------
// dq = scale * (U - delta * W),  U = sum_key (p*dp)[key] K[key],  W = sum_key p[key] K[key],
// delta = sum_key p*dp, dp = da . V^T.  One pass over the key tiles, no saved forward output.
// The 4 waves of the workgroup share one 32-query tile: each recomputes the (cheap) score tile row
// statistics, then takes the key tiles t = wave, wave+4, ... for dP / U / W; partial U, W, delta are
// combined through LDS (the staging area is dead by then).
//
// NCW > 0 (16-bit storage): the d out-proj GEMM is folded in.  `da` then is d(out-proj output) [B*Kp, d] (d = 256 * NCW)
// and `wo` the transposed out-proj weight [d (in), d (out)]: the workgroup first forms its head's slice
// da_h = da . W_out[:, 64h .. 64h+63] -- 32 x 64 x d on the matrix cores, the four waves taking d / 4 of the contraction
// each, operands straight from global memory in one batch with the K / V staging loads, partial tiles summed through LDS
// before K / V land there -- and feeds it to the dP MFMAs in the layout the result tile already has
// (tile_times_cdfrag).  One launch and one [B*Kp, d] round trip per block less on the backward chain.
template <typename T, int NT, int NCW>
// (No __restrict__ on the inputs: hipcc treats loads through restrict-const pointers as invariant and moves them across
// the asm memory barriers that pin the order of the fused prologue's loads -- its vmcnt waits are counted.)
// (bid, nwg): this problem's workgroup index / count (the whole grid in a plain launch, its share of the grid in a paired
// one).  key_len / key_stride: group b reads keys [0, key_len[b]) of the key_stride rows that k / v hold per group (the text
// tower's per-class K / V cache: trainers/rpo.py:144-151 -- a prompt row of class c reads exactly the len_c frozen tokens);
// key_len == nullptr: every group has N keys stored back to back (the image tower).
__device__ __forceinline__ void attn_bwd_body(char* smem, const int bid, const int nwg,
                                              const T* qr, int64_t ldq, const T* k, const T* v,
                                              int64_t ldkv, const T* da, int64_t ldda,
                                              T* dq, int64_t lddq, int B, int H, int N, int Kp,
                                              float scale, const T* wo, int64_t ldwo,
                                              const int32_t* key_len, int key_stride, const bool plain_order = false,
                                              const int nqt_wg = 1) {
  using L = AL<T, NT>;
  // fused variant: every wave owns WSLOTS 8-KiB landing slots for its W_out^T chunks (NCW <= 3: two, the third chunk re-uses
  // slot 0; NCW = 4, d = 1024: four, all requested up front)
  constexpr int WSLOTS = NCW == 4 ? 4 : 2, WSTRIDE = WSLOTS * 8192;
  constexpr int stat_off = NCW > 0 && L::BWD_BYTES < 4 * WSTRIDE ? 4 * WSTRIDE : L::BWD_BYTES;   // launchers allocate stat_off + 2048
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  // Block bid runs on XCD bid % 8: every XCD takes a contiguous run of (image, head) pairs, i.e. whole images -- the
  // images whose rows the out-proj / c_fc / c_proj kernels around this one handle on the same XCD (row units).
#ifdef RPO_ATTN_PLAIN_ORDER
  const int wgid = bid;
#else
  // (plain_order: bid IS the (group, head) index -- the persistent chain kernel, chain.hip, places its items itself)
  const int wgid = plain_order ? bid : [&] {
    const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7;
    return (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
  }();
#endif
  // fused variant with Kp > 32: nqt_wg workgroups per (group, head), one per 32-query tile (adjacent: same XCD)
  const int pair_id = wgid / nqt_wg, qt_wg = wgid - pair_id * nqt_wg;
  const int b = pair_id / H, h = pair_id % H;
  if (key_len != nullptr) N = max(1, min(key_len[b], min(key_stride, NT * 32)));   // keys of this group
  else key_stride = N;
  const T* kb = k + (int64_t)b * key_stride * ldkv + h * 64;
  const T* vb = v + (int64_t)b * key_stride * ldkv + h * 64;
  char* ks = smem;
  char* vs = smem + L::K_BYTES;
  float4* part_u = reinterpret_cast<float4*>(smem);                 // [4 waves][8 groups][64 lanes]
  float4* part_w = part_u + 4 * 8 * 64;
  float* part_d = reinterpret_cast<float*>(part_w + 4 * 8 * 64);    // [4 waves][64 lanes]
  bf16x8_t i0, i1;
  if constexpr (sizeof(T) == 2) { i0 = ident_frag<T>(0, l31, half); i1 = ident_frag<T>(1, l31, half); }

  // fused variant: one query tile (the launcher checks Kp <= 32) -- with a runtime trip count hipcc hoists the address
  // arithmetic of every load of the prologue out of the loop and spills it
  const int nqt = NCW > 0 ? 1 : (Kp + 31) / 32;
  for (int qt0 = 0; qt0 < nqt; ++qt0) {
    if (qt0 > 0) __syncthreads();                  // partials of the previous tile have been consumed
    const int qt = NCW > 0 ? qt_wg : qt0;
    const int i = qt * 32 + l31;
    const int64_t prow = (int64_t)b * Kp + min(i, Kp - 1);
    RowFrag<T> qf, df;                             // issued ahead of the staging loads: one HBM round trip
    RPO_STAMP(10);
    qf.load(qr + prow * ldq + h * 64, half);
    bf16x8_t dcd[4];                               // fused path: da_h of this lane's query in C/D order
    if constexpr (NCW > 0) {
      static_assert(sizeof(T) == 2, "the fused d out-proj needs 16-bit storage");
      // This wave's NCW 64-deep chunks of the contraction (chunks wave, wave + 4, ...).  The W_out^T slice (64 rows x 128 B
      // per chunk) comes in by LDS-DMA into two wave-private 8-KB slots -- whole 128-B lines per 8 lanes, XOR-swizzled
      // like the GEMM tiles; as per-lane operand loads (32 rows x 32 B per instruction) the same bytes took 16 k cycles --
      // the dx rows as per-lane fragments, and K / V are requested behind them so that their (cold) latency hides under
      // the MFMAs.  The order of the groups is pinned (counted vmcnt waits below).
      typedef const __attribute__((address_space(1))) void* gptr_t;
      typedef __attribute__((address_space(3))) void* lptr_t;
      char* myslot = smem + wave * WSTRIDE;
      const int dr = lane >> 3, dc = lane & 7;
      const char* wsrc = reinterpret_cast<const char*>(wo) + ((int64_t)(h * 64 + dr) * ldwo) * 2 + 128 * wave;
      auto dma_chunk = [&](int c, int slot) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int sw = ((8 * i + dr) >> 1) & 7;
          __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + (int64_t)(8 * i) * ldwo * 2 + 512 * c + ((dc ^ sw) << 4)),
                                           (lptr_t)(myslot + slot * 8192 + i * 1024), 16, 0, 0);
        }
      };
      dma_chunk(0, 0);
      asm volatile("" ::: "memory");
      dma_chunk(1, 1);
      asm volatile("" ::: "memory");
      if constexpr (NCW == 4) {
        dma_chunk(2, 2);
        asm volatile("" ::: "memory");
        dma_chunk(3, 3);
        asm volatile("" ::: "memory");
      }
      bf16x8_t xb[NCW][4];
      const T* xrow = da + prow * ldda + half * 8 + 64 * wave;
#pragma unroll
      for (int c = 0; c < NCW; ++c)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) xb[c][kk] = *reinterpret_cast<const bf16x8_t*>(xrow + (256 * c + 16 * kk));
      asm volatile("" ::: "memory");
      constexpr int CHUNKS = NT * 32 * 8, ITERS = (CHUNKS + 255) / 256;
      constexpr int NXQ = 4 * NCW;                 // the xb loads (qf was requested before the first DMA: older)
      f32x16_t dpart[2];
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dpart[jt][r] = 0.f;
      const int fsw = (l31 >> 1) & 7;
      auto mfma_chunk = [&](int c, int slot) {
        const char* base = myslot + slot * 8192 + l31 * 128;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int off = ((2 * kk + half) ^ fsw) << 4;
          const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(base + off);
          const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(base + 32 * 128 + off);
          dpart[0] = mfma16<T>(a0, xb[c][kk], dpart[0]);
          dpart[1] = mfma16<T>(a1, xb[c][kk], dpart[1]);
        }
      };
      // younger than chunk 0: chunk 1 (8) + xb / qf
#ifdef RPO_FUSED_SAFE
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NCW == 4 ? 24 : 8) + NXQ) : "memory");
#endif
      mfma_chunk(0, 0);
      if constexpr (NCW == 3) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // slot 0 has been read
        dma_chunk(2, 0);
        asm volatile("" ::: "memory");
      }
      uint4 ka[ITERS], va[ITERS];
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {         // unconditional (clamped) loads: a branch per load serialises them
        const int id = tid + it * 256, key = min(id >> 3, N - 1), c = id & 7;
        ka[it] = *reinterpret_cast<const uint4*>(kb + (int64_t)key * ldkv + c * 8);
        va[it] = *reinterpret_cast<const uint4*>(vb + (int64_t)key * ldkv + c * 8);
      }
#ifdef RPO_FUSED_SAFE
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NXQ + (NCW == 3 ? 8 : NCW == 4 ? 16 : 0) + 2 * ITERS) : "memory");   // younger than chunk 1
#endif
      mfma_chunk(1, 1);
      if constexpr (NCW == 4) {
#ifdef RPO_FUSED_SAFE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NXQ + 8 + 2 * ITERS) : "memory");                    // younger than chunk 2
#endif
        mfma_chunk(2, 2);
#ifndef RPO_FUSED_SAFE
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NXQ + 2 * ITERS) : "memory");                        // younger than chunk 3
#endif
        mfma_chunk(3, 3);
      }
      if constexpr (NCW == 3) {
#ifdef RPO_FUSED_SAFE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * ITERS) : "memory");                            // younger than chunk 2
#endif
        mfma_chunk(2, 0);
      }
      RPO_STAMP(11);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      float4* px = reinterpret_cast<float4*>(myslot);              // this wave's partial tiles: [8][64 lanes], in its own slot
#pragma unroll
      for (int i = 0; i < 8; ++i)
        px[i * 64 + lane] = make_float4(dpart[i >> 2][4 * (i & 3)], dpart[i >> 2][4 * (i & 3) + 1],
                                        dpart[i >> 2][4 * (i & 3) + 2], dpart[i >> 2][4 * (i & 3) + 3]);
      __syncthreads();
      float dfull[32];
      const float4* pall = reinterpret_cast<const float4*>(smem);   // wave w's partials start at byte WSTRIDE * w
      constexpr int PW = WSTRIDE / 16;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 a = pall[i * 64 + lane], b = pall[PW + i * 64 + lane], c = pall[2 * PW + i * 64 + lane],
                     d = pall[3 * PW + i * 64 + lane];
        dfull[4 * i] = (a.x + b.x) + (c.x + d.x); dfull[4 * i + 1] = (a.y + b.y) + (c.y + d.y);
        dfull[4 * i + 2] = (a.z + b.z) + (c.z + d.z); dfull[4 * i + 3] = (a.w + b.w) + (c.w + d.w);
      }
#pragma unroll
      for (int ks2 = 0; ks2 < 4; ++ks2) dcd[ks2] = pack8<T>(dfull + 8 * ks2);
      __syncthreads();                             // the partials have been read: K / V may land on them
      RPO_STAMP(12);
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int id = tid + it * 256, key = id >> 3, c = id & 7;
        if (id < CHUNKS) {
          const bool live = key < N;               // rows past the last key: zeros
          *reinterpret_cast<uint4*>(ks + key * 144 + c * 16) = live ? ka[it] : make_uint4(0, 0, 0, 0);
          *reinterpret_cast<uint4*>(vs + key * 144 + c * 16) = live ? va[it] : make_uint4(0, 0, 0, 0);
        }
      }
    } else if constexpr (sizeof(T) == 2) {
      df.load(da + prow * ldda + h * 64, half);
      stage2_bf16<NT, 256>(ks, vs, reinterpret_cast<const bf16_t*>(kb), reinterpret_cast<const bf16_t*>(vb), ldkv, N, tid);          // K and V loads in ONE round trip
    } else {
      df.load(da + prow * ldda + h * 64, half);
      stage_rows_f32<NT, 256>(reinterpret_cast<float*>(ks), kb, ldkv, N, tid);
      stage_rows_f32<NT, 256>(reinterpret_cast<float*>(vs), vb, ldkv, N, tid);
    }
    __syncthreads();
    RPO_STAMP(13);
    int l31v = l31;
    asm volatile("" : "+v"(l31v));
    // phase 1 (4 MFMAs per tile): row max and sum of the scores, online -- nothing else is kept, which keeps the
    // kernel at two workgroups per CU (all 384 (image, head) workgroups resident at once).  Each wave takes the key
    // tiles it will own in phase 2 (t = wave, wave + 4, ...); the four (max, sum) pairs are merged through 2 KB of LDS
    // behind the staging area (every wave scanning all tiles cost 6 k cycles of a 28 k-cycle kernel).
    float m = -INFINITY, l = 0.f;
#pragma unroll 1
    for (int t = wave; t < NT; t += 4) {
      f32x16_t sc = tile_times_frag<T>(ks, t, qf, l31v, half);
      const float mn = fmaxf(m, mask_and_max(sc, t, N, half));
      if (mn > -INFINITY) {                          // a wave whose tiles hold no key at all keeps (-inf, 0)
        l = l * exp_scalar<T>(m - mn, scale) + exp_tile<T>(sc, mn, scale);
        m = mn;
      }
    }
    l += __shfl_xor(l, 32, 64);
    float2* rowstat = reinterpret_cast<float2*>(smem + stat_off);   // [4 waves][64 lanes]
    rowstat[wave * 64 + lane] = make_float2(m, l);
    __syncthreads();
    {
      const float2 s0 = rowstat[lane], s1 = rowstat[64 + lane], s2 = rowstat[128 + lane], s3 = rowstat[192 + lane];
      m = fmaxf(fmaxf(s0.x, s1.x), fmaxf(s2.x, s3.x));              // tile 0 always holds keys: finite
      l = (s0.y * exp_scalar<T>(s0.x - m, scale) + s1.y * exp_scalar<T>(s1.x - m, scale)) +
          (s2.y * exp_scalar<T>(s2.x - m, scale) + s3.y * exp_scalar<T>(s3.x - m, scale));
    }
    const float inv = 1.0f / l;
    RPO_STAMP(14);
    // phase 2: this wave's key tiles t = wave, wave+4, ...
    f32x16_t u[2], w[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { u[dt][r] = 0.f; w[dt][r] = 0.f; }
    float delta = 0.f;
#pragma unroll 1
    for (int t = wave; t < NT; t += 4) {           // wave-uniform trip count
      f32x16_t pt = tile_times_frag<T>(ks, t, qf, l31v, half);
      (void)mask_and_max(pt, t, N, half);
      (void)exp_tile<T>(pt, m, scale);
      f32x16_t dp;
      if constexpr (NCW > 0) dp = tile_times_cdfrag<T>(vs, t, dcd, l31v, half);
      else dp = tile_times_frag<T>(vs, t, df, l31v, half);
      if constexpr (sizeof(T) == 2) {
        const char* krow = ks + (32 * t + l31v) * 144 + half * 16;
        bf16x8_t krows[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) krows[kk] = *reinterpret_cast<const bf16x8_t*>(krow + kk * 32);
        float pw[16], pp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pp[r] = pt[r] * inv;
          pw[r] = dp[r] * pp[r];                   // P * dP  (P = 0 on padded keys)
          delta += pw[r];
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          bf16x8_t ktf[2];
          transpose_tile<T>(krows, dt, i0, i1, ktf);  // K^T fragments of this key tile, in registers
#pragma unroll
          for (int g2 = 0; g2 < 2; ++g2) {
            u[dt] = mfma16<T>(ktf[g2], pack8<T>(pw + 8 * g2), u[dt]);
            w[dt] = mfma16<T>(ktf[g2], pack8<T>(pp + 8 * g2), w[dt]);
          }
        }
      } else {
        f32x16_t pn;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pn[r] = pt[r] * inv;
          dp[r] *= pn[r];
          delta += dp[r];
        }
        contract_keys<T, NT>(ks, t, dp, u, l31v, half);
        contract_keys<T, NT>(ks, t, pn, w, l31v, half);
      }
    }
    delta += __shfl_xor(delta, 32, 64);
    RPO_STAMP(15);
    __syncthreads();                               // everybody is done with the staged K / V
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        part_u[(wave * 8 + dt * 4 + g) * 64 + lane] = make_float4(u[dt][4 * g], u[dt][4 * g + 1], u[dt][4 * g + 2], u[dt][4 * g + 3]);
        part_w[(wave * 8 + dt * 4 + g) * 64 + lane] = make_float4(w[dt][4 * g], w[dt][4 * g + 1], w[dt][4 * g + 2], w[dt][4 * g + 3]);
      }
    part_d[wave * 64 + lane] = delta;
    __syncthreads();
    const float dl = (part_d[lane] + part_d[64 + lane]) + (part_d[128 + lane] + part_d[192 + lane]);
    if (i < Kp) {
      T* orow = dq + prow * lddq + h * 64;
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        const int gi = wave * 2 + gg, dt = gi >> 2, g = gi & 3;
        float4 us = make_float4(0.f, 0.f, 0.f, 0.f), ws = us;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) {
          const float4 a = part_u[(wv * 8 + gi) * 64 + lane], c = part_w[(wv * 8 + gi) * 64 + lane];
          us.x += a.x; us.y += a.y; us.z += a.z; us.w += a.w;
          ws.x += c.x; ws.y += c.y; ws.z += c.z; ws.w += c.w;
        }
        ActIO<T>::st4(orow + 32 * dt + 8 * g + 4 * half, scale * (us.x - dl * ws.x), scale * (us.y - dl * ws.y),
                      scale * (us.z - dl * ws.z), scale * (us.w - dl * ws.w));
      }
    }
    RPO_STAMP(16);
  }
}

template <typename T, int NT, int NCW = 0>
__global__ __launch_bounds__(256, 2) void attn_bwd_kernel(const T* qr, int64_t ldq, const T* k, const T* v,
                                                       int64_t ldkv, const T* da, int64_t ldda,
                                                       T* dq, int64_t lddq, int B, int H, int N, int Kp,
                                                       float scale, const T* wo = nullptr, int64_t ldwo = 0,
                                                       const int32_t* key_len = nullptr, int key_stride = 0,
                                                       int nqt_wg = 1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  attn_bwd_body<T, NT, NCW>(smem, blockIdx.x, gridDim.x, qr, ldq, k, v, ldkv, da, ldda, dq, lddq, B, H, N, Kp, scale, wo,
                            ldwo, key_len, key_stride, false, nqt_wg);
}

// Two attention-backward problems (d out-proj folded in) in one launch: workgroups [0, blocks0) are the (image, head)
// pairs of problem 0, the rest the (class, head) pairs of problem 1 -- the same stage of the two prompt-row chains.
struct AttnBwdProblem {
  const void* qr; int64_t ldq; const void* k; const void* v; int64_t ldkv; const void* da; int64_t ldda;
  void* dq; int64_t lddq; int B, H, N, Kp; float scale; const void* wo; int64_t ldwo; const int32_t* key_len; int key_stride;
};
// Problem 1 may get fewer workgroups than it has (group, head) items (each then walks items bid, bid + wgs1, ...): the
// kernel keeps two workgroups per CU resident, and a second round for a few dozen workgroups would double the launch.
template <typename T, int NT0, int NCW0, int NT1, int NCW1>
__global__ __launch_bounds__(256, 2) void attn_bwd_pair_kernel(const AttnBwdProblem a, const AttnBwdProblem b,
                                                            const int blocks0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)blockIdx.x < blocks0) {
    attn_bwd_body<T, NT0, NCW0>(smem, blockIdx.x, blocks0, static_cast<const T*>(a.qr), a.ldq, static_cast<const T*>(a.k),
                                static_cast<const T*>(a.v), a.ldkv, static_cast<const T*>(a.da), a.ldda,
                                static_cast<T*>(a.dq), a.lddq, a.B, a.H, a.N, a.Kp, a.scale, static_cast<const T*>(a.wo),
                                a.ldwo, a.key_len, a.key_stride);
  } else {
    const int items = b.B * b.H, first = (int)blockIdx.x - blocks0, wgs1 = (int)gridDim.x - blocks0;
    for (int t = first; t < items; t += wgs1) {
      if (t != first) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                           // the previous item's partials have been read
      }
      attn_bwd_body<T, NT1, NCW1>(smem, t, items, static_cast<const T*>(b.qr), b.ldq, static_cast<const T*>(b.k),
                                  static_cast<const T*>(b.v), b.ldkv, static_cast<const T*>(b.da), b.ldda,
                                  static_cast<T*>(b.dq), b.lddq, b.B, b.H, b.N, b.Kp, b.scale, static_cast<const T*>(b.wo),
                                  b.ldwo, b.key_len, b.key_stride);
    }
  }
}

template <typename T, int NT>
int launch_fwd2(const void* q, const void* k, const void* v, int64_t ld, void* out, int64_t ldo, int B, int H,
                int N, int Kp, float scale, int q_first, hipStream_t s) {
  static rpo_lds_mask_t lds_ok{0};
  auto kern = attn_fwd2_kernel<T, NT>;
  constexpr int bytes = AL<T, NT>::FWD_BYTES;
  if (int rc = rpo_allow_lds(reinterpret_cast<const void*>(kern), bytes, &lds_ok)) return rc;
  hipLaunchKernelGGL(kern, dim3(B * H), dim3(512), bytes, s, static_cast<const T*>(q),
                     static_cast<const T*>(k), static_cast<const T*>(v), ld, static_cast<T*>(out), ldo, B, H,
                     N, Kp, scale, q_first);
  return rpo_launch_status();
}

template <typename T, int NT>
int launch_fwd(const void* q, const void* k, const void* v, int64_t ld, void* out, int64_t ldo, int B, int H,
               int N, int Kp, float scale, int q_first, hipStream_t s) {
#ifdef RPO_ATTN_FWD_V2            // (A/B build: every score tile in registers, one workgroup per CU -- attn_fwd2_kernel)
  if constexpr (sizeof(T) == 2) return launch_fwd2<T, NT>(q, k, v, ld, out, ldo, B, H, N, Kp, scale, q_first, s);
#endif
#if !defined(RPO_ATTN_OLD_LOOP) && !defined(RPO_ATTN_ONE_BARRIER) && !defined(RPO_ATTN_SPLIT) && !defined(RPO_ATTN_FWD_2PASS)
  // round 6: two-phase staging where the launch has more (image, head) units than CUs and seven key tiles (ViT-B/16 from
  // batch 22 on) -- there the staging burst is long and the early start pays (batch 64: -0.5 % on the step); with fewer
  // units (or ViT-L/14's nine key tiles, 256 units at batch 16) the second barrier's wave skew costs more than the early start
  // returns (ViT-L/14 +0.8 %): profiles/r06_ab_attn_configs.txt.  A/B builds: -DRPO_ATTN_ONE_BARRIER never, -DRPO_ATTN_TWO_PHASE always.
#ifdef RPO_ATTN_TWO_PHASE
  const bool two_phase = true;
#else
  const bool two_phase = NT == 7 && B * H > rpo_cu_count();
#endif
  if constexpr (sizeof(T) == 2) if (two_phase) {
    static rpo_lds_mask_t lds16_ok{0};
#ifdef RPO_ATTN_DMA                 // (A/B build: K by LDS-DMA into swizzled rows; default: 16-B register loads, padded rows)
    constexpr bool kdma = true;
#else
    constexpr bool kdma = false;
#endif
    auto k16 = attn_fwd16_kernel<T, NT, kdma>;
    constexpr int bytes16 = AL16<T, NT, kdma>::BYTES;
    if (int rc = rpo_allow_lds(reinterpret_cast<const void*>(k16), bytes16, &lds16_ok)) return rc;
    hipLaunchKernelGGL(k16, dim3(B * H), dim3(512), bytes16, s, static_cast<const T*>(q), static_cast<const T*>(k),
                       static_cast<const T*>(v), ld, static_cast<T*>(out), ldo, B, H, N, Kp, scale, q_first);
    return rpo_launch_status();
  }
#endif
  static rpo_lds_mask_t lds_ok{0};
#ifdef RPO_ATTN_FWD_2PASS         // (A/B build: two-pass softmax with recomputed scores, two workgroups per CU)
  auto kern = attn_fwd_kernel<T, NT, sizeof(T) == 2>;
#else
  auto kern = attn_fwd_kernel<T, NT>;
#endif
  constexpr int bytes = AL<T, NT>::FWD_BYTES;
  if (int rc = rpo_allow_lds(reinterpret_cast<const void*>(kern), bytes, &lds_ok)) return rc;
  // Two workgroups fit a CU.  EXPERIMENT (round 4, off: -DRPO_ATTN_SPLIT): with more units than CUs but fewer than
  // slots, split units until the slots are full.  Measured same-box: the kernel's in-stream time drops 14.5 -> 12.5 us
  // (B = 32: no CU hosts two whole units any more), but the STEP is 0.5 % slower (2.958 vs 2.943 ms, three alternating
  // pairs) and the forward pair with the text tower beside it unchanged: a grid that fills every resident slot has no
  // room left for the side queue's workgroups, whose hosts then run a third half-unit (profiles/r04_ab_attn_fwd_variants.txt).
  const int units = B * H;
  int nsplit = 0;
#ifdef RPO_ATTN_SPLIT
  const int cus = rpo_cu_count();
  if (sizeof(T) == 2 && q_first == 0 && units > cus && units < 2 * cus) nsplit = 2 * cus - units < units ? 2 * cus - units : units;
#endif
  hipLaunchKernelGGL(kern, dim3(units + nsplit), dim3(512), bytes, s, static_cast<const T*>(q),
                     static_cast<const T*>(k), static_cast<const T*>(v), ld, static_cast<T*>(out), ldo, B, H,
                     N, Kp, scale, q_first, units - nsplit);
  return rpo_launch_status();
}

template <typename T, int NT>
int launch_bwd(const void* qr, int64_t ldq, const void* k, const void* v, int64_t ldkv, const void* da,
               int64_t ldda, void* dq, int64_t lddq, int B, int H, int N, int Kp, float scale, hipStream_t s) {
  static rpo_lds_mask_t lds_ok{0};
  auto kern = attn_bwd_kernel<T, NT>;
  constexpr int bytes = AL<T, NT>::BWD_BYTES + 2048;
  if (int rc = rpo_allow_lds(reinterpret_cast<const void*>(kern), bytes, &lds_ok)) return rc;
  hipLaunchKernelGGL(kern, dim3(B * H), dim3(256), bytes, s, static_cast<const T*>(qr), ldq,
                     static_cast<const T*>(k), static_cast<const T*>(v), ldkv, static_cast<const T*>(da), ldda,
                     static_cast<T*>(dq), lddq, B, H, N, Kp, scale, static_cast<const T*>(nullptr), (int64_t)0,
                     static_cast<const int32_t*>(nullptr), 0, 1);
  return rpo_launch_status();
}

constexpr int bwd_proj_lds(int bwd_bytes, int ncw) {
  const int slots = 4 * (ncw == 4 ? 4 : 2) * 8192;
  return (bwd_bytes > slots ? bwd_bytes : slots) + 2048;
}

template <typename T, int NT, int NCW>
int launch_bwd_proj(const void* qr, int64_t ldq, const void* k, const void* v, int64_t ldkv, const void* dx,
                    int64_t lddx, const void* wo, int64_t ldwo, void* dq, int64_t lddq, int B, int H, int N, int Kp,
                    float scale, hipStream_t s) {
  static rpo_lds_mask_t lds_ok{0};
  auto kern = attn_bwd_kernel<T, NT, NCW>;
  constexpr int bytes = bwd_proj_lds(AL<T, NT>::BWD_BYTES, NCW);   // 4 waves x 2 (4) weight slots of 8 KiB + row statistics
  if (int rc = rpo_allow_lds(reinterpret_cast<const void*>(kern), bytes, &lds_ok)) return rc;
  const int nqt = (Kp + 31) / 32;                  // one workgroup per (image, head, 32-query tile)
  hipLaunchKernelGGL(kern, dim3(B * H * nqt), dim3(256), bytes, s, static_cast<const T*>(qr), ldq,
                     static_cast<const T*>(k), static_cast<const T*>(v), ldkv, static_cast<const T*>(dx), lddx,
                     static_cast<T*>(dq), lddq, B, H, N, Kp, scale, static_cast<const T*>(wo), ldwo,
                     static_cast<const int32_t*>(nullptr), 0, nqt);
  return rpo_launch_status();
}

bool ok_ld(int64_t ld, int esz) { return (ld * esz) % 16 == 0; }

// one problem with per-group key counts (the text tower)
template <typename T, int NT, int NCW>
int launch_bwd_proj_var(const AttnBwdProblem& a, hipStream_t s) {
  static rpo_lds_mask_t lds_ok{0};
  auto kern = attn_bwd_kernel<T, NT, NCW>;
  constexpr int bytes = bwd_proj_lds(AL<T, NT>::BWD_BYTES, NCW);
  if (int rc = rpo_allow_lds(reinterpret_cast<const void*>(kern), bytes, &lds_ok)) return rc;
  hipLaunchKernelGGL(kern, dim3(a.B * a.H), dim3(256), bytes, s, static_cast<const T*>(a.qr), a.ldq,
                     static_cast<const T*>(a.k), static_cast<const T*>(a.v), a.ldkv, static_cast<const T*>(a.da), a.ldda,
                     static_cast<T*>(a.dq), a.lddq, a.B, a.H, a.N, a.Kp, a.scale, static_cast<const T*>(a.wo), a.ldwo,
                     a.key_len, a.key_stride, 1);
  return rpo_launch_status();
}
template <typename T, int NT0, int NCW0, int NT1, int NCW1>
int launch_bwd_proj_pair(const AttnBwdProblem& a, const AttnBwdProblem& b, hipStream_t s) {
  static rpo_lds_mask_t lds_ok{0};
  auto kern = attn_bwd_pair_kernel<T, NT0, NCW0, NT1, NCW1>;
  constexpr int b0 = bwd_proj_lds(AL<T, NT0>::BWD_BYTES, NCW0), b1 = bwd_proj_lds(AL<T, NT1>::BWD_BYTES, NCW1);
  constexpr int bytes = b0 > b1 ? b0 : b1;
  if (int rc = rpo_allow_lds(reinterpret_cast<const void*>(kern), bytes, &lds_ok)) return rc;
  const int items0 = a.B * a.H, items1 = b.B * b.H, slots = 2 * rpo_cu_count();
  int wgs1 = items1;
  for (int walk = 2; items0 + wgs1 > slots && walk <= 8; ++walk) wgs1 = (items1 + walk - 1) / walk;
  hipLaunchKernelGGL(kern, dim3(items0 + wgs1), dim3(256), bytes, s, a, b, items0);
  return rpo_launch_status();
}
// (key tiles, 256-column chunks of d) of one problem; 0 if the fused kernel does not cover it
inline int bwd_proj_geometry(const rpo_attn_bwd_args& a, int* nt, int* ncw) {
  const int d = a.H * 64;
  if (a.keys > 288 || a.Kp > 32 || (d != 512 && d != 768)) return 0;
  *nt = a.keys <= 96 ? 3 : (a.keys <= 224 ? 7 : 9);
  *ncw = d / 256;
  return 1;
}
inline int bwd_proj_check(const rpo_attn_bwd_args& a) {
  if (!a.q_rows || !a.k || !a.v || !a.dx || !a.w_out_t || !a.dq || a.groups <= 0 || a.H <= 0 || a.keys <= 0 || a.Kp <= 0)
    return RPO_E_BADARG;
  if (a.key_len != nullptr && a.key_stride < a.keys) return RPO_E_BADARG;
  if (!aligned16(a.q_rows) || !aligned16(a.k) || !aligned16(a.v) || !aligned16(a.dx) || !aligned16(a.w_out_t) ||
      !ok_ld(a.ldq, 2) || !ok_ld(a.ldkv, 2) || !ok_ld(a.lddx, 2) || !ok_ld(a.ldw, 2) ||
      reinterpret_cast<uintptr_t>(a.dq) % 8 || (a.lddq * 2) % 8) return RPO_E_ALIGN;
  return 0;
}
inline AttnBwdProblem bwd_problem(const rpo_attn_bwd_args& a) {
  return AttnBwdProblem{a.q_rows, a.ldq, a.k, a.v, a.ldkv, a.dx, a.lddx, a.dq, a.lddq, a.groups, a.H, a.keys, a.Kp,
                        a.scale, a.w_out_t, a.ldw, a.key_len, a.key_len ? a.key_stride : a.keys};
}
template <typename T>
int dispatch_bwd_proj_args(const rpo_attn_bwd_args* a0, const rpo_attn_bwd_args* a1, hipStream_t s) {
  int nt0, ncw0, nt1 = 0, ncw1 = 0;
  if (!bwd_proj_geometry(*a0, &nt0, &ncw0) || (a1 && !bwd_proj_geometry(*a1, &nt1, &ncw1))) return RPO_E_SHAPE;
  const AttnBwdProblem p0 = bwd_problem(*a0);
  if (a1 == nullptr) {
#define RPO_BV(NT_, NCW_) if (nt0 == NT_ && ncw0 == NCW_) return launch_bwd_proj_var<T, NT_, NCW_>(p0, s)
    RPO_BV(3, 2); RPO_BV(3, 3); RPO_BV(7, 2); RPO_BV(7, 3); RPO_BV(9, 2); RPO_BV(9, 3);
#undef RPO_BV
    return RPO_E_SHAPE;
  }
  const AttnBwdProblem p1 = bwd_problem(*a1);
  // the pairs a step of this repo's models forms: image tower (197 keys, d 768) with the text tower (<= 96 keys, d 512)
#define RPO_BPAIR(A, B, C, D) if (nt0 == A && ncw0 == B && nt1 == C && ncw1 == D) return launch_bwd_proj_pair<T, A, B, C, D>(p0, p1, s)
  RPO_BPAIR(7, 3, 3, 2);
  RPO_BPAIR(7, 3, 3, 3);
#undef RPO_BPAIR
  return RPO_E_SHAPE;
}
template <typename T>
int dispatch_bwd_proj(int ncw, bool big, const void* qr, int64_t ldq, const void* k, const void* v, int64_t ldkv,
                      const void* dx, int64_t lddx, const void* wo, int64_t ldwo, void* dq, int64_t lddq, int B, int H,
                      int N, int Kp, float scale, hipStream_t s) {
#define RPO_BP(NT_, NCW_) return launch_bwd_proj<T, NT_, NCW_>(qr, ldq, k, v, ldkv, dx, lddx, wo, ldwo, dq, lddq, B, H, N, Kp, scale, s)
  if (ncw == 2) { if (big) RPO_BP(9, 2); RPO_BP(7, 2); }
  if (ncw == 3) { if (big) RPO_BP(9, 3); RPO_BP(7, 3); }
  if (ncw == 4) { if (big) RPO_BP(9, 4); RPO_BP(7, 4); }
#undef RPO_BP
  return RPO_E_SHAPE;
}


// ---- text tower, prompt rows (round 5): one WAVE per (class, head) -------------------------------------------------------
// The prompt rows of a class read only that class's frozen tokens (trainers/rpo.py:144-151): K <= 64 queries against
// len_c keys -- 8 .. 14 for the Oxford-Pets prompts, <= 77 for any CLIP prompt -- per (class, head).  That is one or two
// 32x32 MFMA tiles, so a single wave holds the whole problem in registers: K and V rows come straight from the row-major
// K / V cache as A-operand fragments (16 B per lane), S^T = K . Q^T leaves a lane with ONE query's scores (softmax =
// in-register reduce + one cross-half exchange), V^T / K^T for the second contraction are formed on the matrix core
// (transpose_tile: exact), P / dS go from the C/D registers straight into the B operand.  No LDS, no barrier, 64 threads.
// It replaces the VALU kernel of attn_text.hip (912 workgroups of 256 threads for the same work) in the 16-bit modes.
//   forward : out = softmax(scale q K^T) V                                      (clip/model.py:186 under the mask above)
//   backward: dq = scale * sum_key dS[key] K[key],  dS = P (dP - sum P dP),  dP = da V^T   (K, V frozen: no dK, dV)
template <typename T, bool BWD, int NKT>
__global__ __launch_bounds__(64) void text_attn_wave_kernel(const T* q, int64_t ldq, const T* kc, const T* vc, int64_t ldkv,
                                                            const T* da, int64_t ldda, T* out, int64_t ldo,
                                                            const int32_t* __restrict__ len, int rows, int Lmax, int H,
                                                            float scale) {
  const int lane = threadIdx.x, half = lane >> 5, l31 = lane & 31;
  const int c = blockIdx.x / H, h = blockIdx.x % H;
  const int L = min(len[c], Lmax);
  RowFrag<T> kf[NKT], vf[NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t) {          // keys >= L: the last key again (scores masked, p = 0 exactly)
    const int64_t base = ((int64_t)c * Lmax + max(min(32 * t + l31, L - 1), 0)) * ldkv + h * 64;
    kf[t].load(kc + base, half);
    vf[t].load(vc + base, half);
  }
  const bf16x8_t i0 = ident_frag<T>(0, l31, half), i1 = ident_frag<T>(1, l31, half);
  for (int r0 = 0; r0 < rows; r0 += 32) {
    const int r = r0 + l31;
    const int64_t row = (int64_t)c * rows + min(r, rows - 1);
    RowFrag<T> qf, df;
    qf.load(q + row * ldq + h * 64, half);
    if constexpr (BWD) df.load(da + row * ldda + h * 64, half);
    f32x16_t s[NKT];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[t][e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s[t] = mfma16<T>(kf[t].f[ks], qf.f[ks], s[t]);
      m = fmaxf(m, mask_and_max(s[t], t, L, half));
    }
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t) l += exp_tile<T>(s[t], m, scale);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    f32x16_t o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
    float fin = inv;
    if constexpr (!BWD) {
#pragma unroll
      for (int t = 0; t < NKT; ++t) {
        float w[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) w[e] = s[t][e];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          bf16x8_t vt[2];
          transpose_tile<T>(vf[t].f, dt, i0, i1, vt);
#pragma unroll
          for (int g2 = 0; g2 < 2; ++g2) o[dt] = mfma16<T>(vt[g2], pack8<T>(w + 8 * g2), o[dt]);
        }
      }
    } else {
      f32x16_t dp[NKT];
      float delta = 0.f;
#pragma unroll
      for (int t = 0; t < NKT; ++t) {
#pragma unroll
        for (int e = 0; e < 16; ++e) dp[t][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) dp[t] = mfma16<T>(vf[t].f[ks], df.f[ks], dp[t]);
#pragma unroll
        for (int e = 0; e < 16; ++e) { s[t][e] *= inv; delta = fmaf(s[t][e], dp[t][e], delta); }
      }
      delta += __shfl_xor(delta, 32, 64);
#pragma unroll
      for (int t = 0; t < NKT; ++t) {
        float w[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) w[e] = s[t][e] * (dp[t][e] - delta);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          bf16x8_t kt[2];
          transpose_tile<T>(kf[t].f, dt, i0, i1, kt);
#pragma unroll
          for (int g2 = 0; g2 < 2; ++g2) o[dt] = mfma16<T>(kt[g2], pack8<T>(w + 8 * g2), o[dt]);
        }
      }
      fin = scale;
    }
    if (r < rows) {
      T* orow = out + row * ldo + h * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          ActIO<T>::st4(orow + 32 * dt + 8 * g + 4 * half, o[dt][4 * g] * fin, o[dt][4 * g + 1] * fin,
                        o[dt][4 * g + 2] * fin, o[dt][4 * g + 3] * fin);
    }
  }
}

template <typename T, bool BWD>
int launch_text_wave(const void* q, int64_t ldq, const void* kc, const void* vc, int64_t ldkv, const void* da, int64_t ldda,
                     void* out, int64_t ldo, const int32_t* len, int n_cls, int rows, int Lmax, int H, float scale,
                     hipStream_t s) {
#define RPO_TW(NKT)                                                                                                   \
  hipLaunchKernelGGL((text_attn_wave_kernel<T, BWD, NKT>), dim3(n_cls * H), dim3(64), 0, s, static_cast<const T*>(q), \
                     ldq, static_cast<const T*>(kc), static_cast<const T*>(vc), ldkv, static_cast<const T*>(da), ldda, \
                     static_cast<T*>(out), ldo, len, rows, Lmax, H, scale)
  if (Lmax <= 32) RPO_TW(1);
  else if (Lmax <= 64) RPO_TW(2);
  else RPO_TW(3);
#undef RPO_TW
  return rpo_launch_status();
}

}  // namespace

#ifndef RPO_DEVICE_ONLY     // (chain.hip includes this file for the device bodies above)
// 16-bit storage, rows <= 64 prompt queries per class, Lmax <= 96 keys, 16-byte aligned rows: the one-wave kernel above;
// RPO_E_SHAPE otherwise (the caller, attn_text.hip, then runs its VALU kernel).  bwd: `da` = d(attention output), `out` = dq.
int rpo_text_attn_wave(int bwd, const void* q, int64_t ldq, const void* kc, const void* vc, int64_t ldkv, const void* da,
                       int64_t ldda, void* out, int64_t ldo, int dtype, const int32_t* len, int n_cls, int rows, int Lmax,
                       int H, float scale, hipStream_t s) {
  if ((dtype != RPO_BF16 && dtype != RPO_F16) || rows > 64 || Lmax > 96) return RPO_E_SHAPE;
  if (!aligned16(q) || (ldq * 2) % 16 != 0 || (reinterpret_cast<uintptr_t>(out) & 7u) || (ldo * 2) % 8 != 0) return RPO_E_SHAPE;
  if (bwd && (!aligned16(da) || (ldda * 2) % 16 != 0)) return RPO_E_SHAPE;
  if (dtype == RPO_BF16)
    return bwd ? launch_text_wave<bf16_t, true>(q, ldq, kc, vc, ldkv, da, ldda, out, ldo, len, n_cls, rows, Lmax, H, scale, s)
               : launch_text_wave<bf16_t, false>(q, ldq, kc, vc, ldkv, da, ldda, out, ldo, len, n_cls, rows, Lmax, H, scale, s);
  return bwd ? launch_text_wave<f16_t, true>(q, ldq, kc, vc, ldkv, da, ldda, out, ldo, len, n_cls, rows, Lmax, H, scale, s)
             : launch_text_wave<f16_t, false>(q, ldq, kc, vc, ldkv, da, ldda, out, ldo, len, n_cls, rows, Lmax, H, scale, s);
}

extern "C" int rpo_attn_readonly_fwd_rows(const void* q, const void* k, const void* v, int64_t ld, void* out,
                                          int64_t ldo, int dtype, int B, int H, int N, int Kp, float scale,
                                          int q_first, void* stream);

extern "C" int rpo_attn_readonly_fwd(const void* q, const void* k, const void* v, int64_t ld, void* out,
                                     int64_t ldo, int dtype, int B, int H, int N, int Kp, float scale,
                                     void* stream) {
  return rpo_attn_readonly_fwd_rows(q, k, v, ld, out, ldo, dtype, B, H, N, Kp, scale, 0, stream);
}

extern "C" int rpo_attn_readonly_fwd_rows(const void* q, const void* k, const void* v, int64_t ld, void* out,
                                          int64_t ldo, int dtype, int B, int H, int N, int Kp, float scale,
                                          int q_first, void* stream) {
  if (!q || !k || !v || !out || B <= 0 || H <= 0 || N <= 0 || Kp < 0 || q_first < 0 || q_first >= N + Kp)
    return RPO_E_BADARG;
  if (N > 288) return RPO_E_SHAPE;
  if (dtype != RPO_F32 && dtype != RPO_BF16 && dtype != RPO_F16) return RPO_E_DTYPE;
  const int esz = dtype == RPO_F32 ? 4 : 2;
  if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !ok_ld(ld, esz) ||
      reinterpret_cast<uintptr_t>(out) % (4 * esz) || (ldo * esz) % (4 * esz)) return RPO_E_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == RPO_BF16) {
    if (N <= 224) return launch_fwd<bf16_t, 7>(q, k, v, ld, out, ldo, B, H, N, Kp, scale, q_first, s);
    return launch_fwd<bf16_t, 9>(q, k, v, ld, out, ldo, B, H, N, Kp, scale, q_first, s);
  }
  if (dtype == RPO_F16) {
    if (N <= 224) return launch_fwd<f16_t, 7>(q, k, v, ld, out, ldo, B, H, N, Kp, scale, q_first, s);
    return launch_fwd<f16_t, 9>(q, k, v, ld, out, ldo, B, H, N, Kp, scale, q_first, s);
  }
  if (N <= 224) return launch_fwd<float, 7>(q, k, v, ld, out, ldo, B, H, N, Kp, scale, q_first, s);
  return launch_fwd<float, 9>(q, k, v, ld, out, ldo, B, H, N, Kp, scale, q_first, s);
}

extern "C" int rpo_attn_readonly_bwd(const void* q_rows, int64_t ldq, const void* k, const void* v,
                                     int64_t ldkv, const void* da, int64_t ldda, void* dq, int64_t lddq,
                                     int dtype, int B, int H, int N, int Kp, float scale, void* stream) {
  if (!q_rows || !k || !v || !da || !dq || B <= 0 || H <= 0 || N <= 0 || Kp <= 0) return RPO_E_BADARG;
  if (N > 288 || Kp > 128) return RPO_E_SHAPE;
  if (dtype != RPO_F32 && dtype != RPO_BF16 && dtype != RPO_F16) return RPO_E_DTYPE;
  const int esz = dtype == RPO_F32 ? 4 : 2;
  if (!aligned16(q_rows) || !aligned16(k) || !aligned16(v) || !aligned16(da) || !ok_ld(ldq, esz) ||
      !ok_ld(ldkv, esz) || !ok_ld(ldda, esz) || reinterpret_cast<uintptr_t>(dq) % (4 * esz) ||
      (lddq * esz) % (4 * esz)) return RPO_E_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == RPO_BF16) {
    if (N <= 224) return launch_bwd<bf16_t, 7>(q_rows, ldq, k, v, ldkv, da, ldda, dq, lddq, B, H, N, Kp, scale, s);
    return launch_bwd<bf16_t, 9>(q_rows, ldq, k, v, ldkv, da, ldda, dq, lddq, B, H, N, Kp, scale, s);
  }
  if (dtype == RPO_F16) {
    if (N <= 224) return launch_bwd<f16_t, 7>(q_rows, ldq, k, v, ldkv, da, ldda, dq, lddq, B, H, N, Kp, scale, s);
    return launch_bwd<f16_t, 9>(q_rows, ldq, k, v, ldkv, da, ldda, dq, lddq, B, H, N, Kp, scale, s);
  }
  if (N <= 224) return launch_bwd<float, 7>(q_rows, ldq, k, v, ldkv, da, ldda, dq, lddq, B, H, N, Kp, scale, s);
  return launch_bwd<float, 9>(q_rows, ldq, k, v, ldkv, da, ldda, dq, lddq, B, H, N, Kp, scale, s);
}

extern "C" int rpo_attn_readonly_bwd_proj(const void* q_rows, int64_t ldq, const void* k, const void* v, int64_t ldkv,
                                          const void* dx, int64_t lddx, const void* w_out_t, int64_t ldw, void* dq,
                                          int64_t lddq, int dtype, int B, int H, int N, int Kp, float scale,
                                          void* stream) {
  if (!q_rows || !k || !v || !dx || !w_out_t || !dq || B <= 0 || H <= 0 || N <= 0 || Kp <= 0) return RPO_E_BADARG;
  if (dtype != RPO_BF16 && dtype != RPO_F16) return RPO_E_DTYPE;
  const int d = H * 64;
  // Kp > 32: one workgroup per 32-query tile (each stages the keys again); d = 1024: four weight slots per wave
  if (N > 288 || Kp > 64 || (d != 512 && d != 768 && d != 1024)) return RPO_E_SHAPE;
  if (!aligned16(q_rows) || !aligned16(k) || !aligned16(v) || !aligned16(dx) || !aligned16(w_out_t) || !ok_ld(ldq, 2) ||
      !ok_ld(ldkv, 2) || !ok_ld(lddx, 2) || !ok_ld(ldw, 2) || reinterpret_cast<uintptr_t>(dq) % 8 || (lddq * 2) % 8)
    return RPO_E_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == RPO_BF16)
    return dispatch_bwd_proj<bf16_t>(d / 256, N > 224, q_rows, ldq, k, v, ldkv, dx, lddx, w_out_t, ldw, dq, lddq, B, H, N, Kp, scale, s);
  return dispatch_bwd_proj<f16_t>(d / 256, N > 224, q_rows, ldq, k, v, ldkv, dx, lddx, w_out_t, ldw, dq, lddq, B, H, N, Kp, scale, s);
}

// rpo_attn_readonly_bwd_proj for one or two problems in ONE launch, each with optional per-group key counts
// (include/rpo_amd.h: rpo_attn_bwd_args).  a1 == NULL: one problem.
#ifdef RPO_EXPERIMENTAL   // measured-slower experiment: include/rpo_amd_experimental.h
extern "C" int rpo_attn_bwd_proj_pair(const rpo_attn_bwd_args* a0, const rpo_attn_bwd_args* a1, int dtype, void* stream) {
  if (a0 == nullptr) return RPO_E_BADARG;
  if (dtype != RPO_BF16 && dtype != RPO_F16) return RPO_E_DTYPE;
  if (int rc = bwd_proj_check(*a0)) return rc;
  if (a1 != nullptr) if (int rc = bwd_proj_check(*a1)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == RPO_BF16) return dispatch_bwd_proj_args<bf16_t>(a0, a1, s);
  return dispatch_bwd_proj_args<f16_t>(a0, a1, s);
}
#endif  // RPO_EXPERIMENTAL
#endif  // RPO_DEVICE_ONLY
