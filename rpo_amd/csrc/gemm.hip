// rpo_gemm_nt: C[M,N] = A[M,K] . W[N,K]^T with fused epilogues, MFMA on gfx950.
//
// Replaces nn.Linear / the packed in-proj and out-proj matmuls of nn.MultiheadAttention
// (reference clip/model.py:171-177,186) and, in the backward, autograd's dX = dY . W.
//
// Tiling: 128x128 block tile, 4 waves in 2x2, each wave a 64x64 sub-tile as 2x2 MFMA 32x32
// tiles.  bf16: v_mfma_f32_32x32x16_bf16, k-tile 64.  f32 (parity mode):
// v_mfma_f32_32x32x2_f32 (exact f32 fma chain), k-tile 32.  Both k-tiles are 128 B per row.
//
// Staging is direct HBM->LDS DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction): no
// staging VGPRs and, decisive on this kernel, no ds_write pass -- with register staging the
// eight ds_write_b128 per thread per k-tile (13 LDS cycles each) plus the fragment reads
// exceeded the MFMA time of the tile, i.e. the loop was LDS-bound.  The DMA writes LDS
// lane-linearly (wave-uniform base + lane*16), so rows are an unpadded 128 B and the
// bank-conflict fix is an XOR swizzle applied on the per-lane SOURCE address and again on the
// fragment read (guide rule 21): 16-B chunk c of row r lives at chunk c ^ ((r >> 1) & 7).
// For the 16-lane groups of ds_read_b128 this gives 16 distinct 16-B slots.
// Double-buffered, one barrier per k-tile; the DMA of tile t+1 is in flight during the MFMAs
// of tile t and is drained (vmcnt(0)) right before the barrier.
//
// The weight tile is the MFMA *A* operand and the activation tile the *B* operand, i.e. the
// wave computes D[n][m]: by the 32x32 C/D map (col = lane&31, row = (reg&3)+8*(reg>>2)+
// 4*(lane>>5)) each lane then owns 4 CONSECUTIVE n for one m per register quad, which
// turns the epilogue's bias / residual loads and the stores into 16-B (f32) / 8-B (bf16)
// vector accesses on row-major C.  The contraction index needs no particular lane order:
// both operands are read with the same (k-step, lane>>5) -> k mapping, so any hardware
// k-permutation cancels.
//
// Workgroup ids are remapped so that each XCD (block b runs on XCD b % 8) owns a contiguous
// range of tiles and re-reads its A / W panels from its own L2.  Optional split-K
// (gridDim.y slices, each writing its own fp32 slab; the consumer sums the slabs in a fixed
// order) keeps the long-K, few-tile dX GEMMs of the backward from running on 36 CUs.
#include "common.h"

namespace {

struct GemmParams {
  const char* A; int64_t lda;
  const char* W; int64_t ldw;
  char* C; int64_t ldc;
  int M, N, K;
  const float* bias;
  const float* resid; int64_t ldr;
  float* aux; int64_t ldaux; int aux_row0;
  int skip_row0, skip_col0, group;
  int split_k; int64_t split_stride;   // elements of C between slabs
};

constexpr int BM = 128, BN = 128;
constexpr int LROW = 128;                  // bytes per LDS row (one k-tile, unpadded: DMA is lane-linear)
constexpr int TILE_BYTES = 128 * LROW;     // one operand tile (16 KiB)
constexpr int SMEM_BYTES = 4 * TILE_BYTES; // 2 buffers x (A, W)

template <typename T> struct Tr;
template <> struct Tr<bf16_t> {
  static constexpr int BK = 64, KSTEPS = 4;
  using frag_t = bf16x8_t;
  // rowp = start of the LDS row, sw = (row >> 1) & 7
  static __device__ __forceinline__ frag_t ldfrag(const char* rowp, int sw, int ks, int half) {
    return *reinterpret_cast<const frag_t*>(rowp + (((ks * 2 + half) ^ sw) << 4));
  }
  static __device__ __forceinline__ f32x16_t mfma(frag_t a, frag_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Tr<float> {
  static constexpr int BK = 32, KSTEPS = 16;
  using frag_t = float;
  static __device__ __forceinline__ frag_t ldfrag(const char* rowp, int sw, int ks, int half) {
    const int e = ks * 2 + half;               // float index in the row; chunk = e >> 2
    return *reinterpret_cast<const float*>(rowp + ((((e >> 2) ^ sw) << 4) | ((e & 3) << 2)));
  }
  static __device__ __forceinline__ f32x16_t mfma(frag_t a, frag_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
};

template <typename TIn, typename TOut, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using T = Tr<TIn>;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;

  const int tiles_n = (p.N + BN - 1) / BN;
  // XCD-aware bijective remap (guide T1): XCD x = bid % 8 gets a contiguous run of tiles
  int wg;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7;
    wg = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
  }
  const int tile_m = wg / tiles_n, tile_n = wg % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (p.skip_row0 >= 0 && m0 >= p.skip_row0 && n0 >= p.skip_col0) return;

  // k-range of this split
  const int nk_all = p.K / T::BK;
  const int kt0 = (int)(((int64_t)nk_all * blockIdx.y) / p.split_k);
  const int kt1 = (int)(((int64_t)nk_all * (blockIdx.y + 1)) / p.split_k);
  const int nk = kt1 - kt0;
  constexpr int KT_BYTES = T::BK * sizeof(TIn);  // 128

  // DMA assignment: one operand tile = 16 wave-instructions of 1 KiB (8 rows); wave w issues
  // instructions i*4 + w, i = 0..3.  Lane l -> row 8*(i*4+w) + (l>>3), physical chunk l&7, which
  // must receive logical chunk (l&7) ^ ((row>>1)&7) of that row.
  const char* ga[4];
  const char* gw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * (i * 4 + wave) + (lane >> 3);
    const int cl = (lane & 7) ^ ((row >> 1) & 7);
    const int ra = min(m0 + row, p.M - 1);
    const int rw = min(n0 + row, p.N - 1);
    ga[i] = p.A + ((int64_t)ra * p.lda) * sizeof(TIn) + cl * 16 + (int64_t)kt0 * KT_BYTES;
    gw[i] = p.W + ((int64_t)rw * p.ldw) * sizeof(TIn) + cl * 16 + (int64_t)kt0 * KT_BYTES;
  }
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
#define RPO_DMA(BUF, KT)                                                                          \
  do {                                                                                            \
    char* b_ = smem + (BUF) * 2 * TILE_BYTES + wave * 1024;                                       \
    const int64_t ko = (int64_t)(KT) * KT_BYTES;                                                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
      __builtin_amdgcn_global_load_lds((gptr_t)(ga[i] + ko), (lptr_t)(b_ + i * 4096), 16, 0, 0);  \
      __builtin_amdgcn_global_load_lds((gptr_t)(gw[i] + ko), (lptr_t)(b_ + TILE_BYTES + i * 4096), 16, 0, 0); \
    }                                                                                             \
  } while (0)

  f32x16_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  // fragment rows of this lane: rows wm*64 + tm*32 + l31 (A side), wn*64 + tn*32 + l31 (W side);
  // +32 rows keeps (row >> 1) & 7 unchanged, so one swizzle term per operand
  const int rx = wm * 64 + l31, rwv = wn * 64 + l31;
  const int swx = (rx >> 1) & 7, sww = (rwv >> 1) & 7;

  RPO_DMA(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) RPO_DMA(cur ^ 1, kt + 1);
    const char* sx = smem + cur * 2 * TILE_BYTES + rx * LROW;
    const char* sw = smem + cur * 2 * TILE_BYTES + TILE_BYTES + rwv * LROW;
#pragma unroll
    for (int ks = 0; ks < T::KSTEPS; ++ks) {
      typename T::frag_t x0 = T::ldfrag(sx, swx, ks, half);
      typename T::frag_t x1 = T::ldfrag(sx + 32 * LROW, swx, ks, half);
      typename T::frag_t w0 = T::ldfrag(sw, sww, ks, half);
      typename T::frag_t w1 = T::ldfrag(sw + 32 * LROW, sww, ks, half);
      acc[0][0] = T::mfma(w0, x0, acc[0][0]);
      acc[0][1] = T::mfma(w0, x1, acc[0][1]);
      acc[1][0] = T::mfma(w1, x0, acc[1][0]);
      acc[1][1] = T::mfma(w1, x1, acc[1][1]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next tile landed (this wave's DMA)
    __syncthreads();                                   // ... and everybody else's; buffer `cur` is free
  }
#undef RPO_DMA

  // epilogue: acc[tn][tm] holds D[n][m]; lane: m = l31, n = 8*g + 4*half + j (reg = 4*g + j)
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const int m = m0 + wm * 64 + tm * 32 + l31;
    if (m >= p.M) continue;
    int64_t orow = m;
    int prow = 0;
    if (EPI == RPO_EPI_PATCH) {
      const int img = m / p.group;
      prow = m - img * p.group + 1;
      orow = (int64_t)m + img + 1;
    }
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 64 + tn * 32 + 8 * g + 4 * half;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[tn][tm][4 * g + j];
        if (EPI == RPO_EPI_BIAS || EPI == RPO_EPI_BIAS_QGELU || EPI == RPO_EPI_BIAS_RESID) {
          const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
          v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
        }
        if (EPI == RPO_EPI_BIAS_QGELU) {
          if (p.aux != nullptr && m >= p.aux_row0)
            *reinterpret_cast<float4*>(p.aux + (int64_t)(m - p.aux_row0) * p.ldaux + n) =
                make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = quick_gelu(v[j]);
        }
        if (EPI == RPO_EPI_BIAS_RESID) {
          const float4 r4 = *reinterpret_cast<const float4*>(p.resid + (int64_t)m * p.ldr + n);
          v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
        }
        if (EPI == RPO_EPI_QGELU_BWD) {
          const float4 u4 = *reinterpret_cast<const float4*>(p.aux + (int64_t)m * p.ldaux + n);
          v[0] *= quick_gelu_grad(u4.x); v[1] *= quick_gelu_grad(u4.y);
          v[2] *= quick_gelu_grad(u4.z); v[3] *= quick_gelu_grad(u4.w);
        }
        if (EPI == RPO_EPI_PATCH) {
          const float4 r4 = *reinterpret_cast<const float4*>(p.resid + (int64_t)prow * p.ldr + n);
          v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
        }
        ActIO<TOut>::st4(reinterpret_cast<TOut*>(p.C) + (int64_t)blockIdx.y * p.split_stride + orow * p.ldc + n, v[0],
                          v[1], v[2], v[3]);
      }
    }
  }
}

template <typename TIn, typename TOut, int EPI>
int launch(const GemmParams& p, hipStream_t s) {
  static bool attr_set = false;
  auto kern = gemm_nt_kernel<TIn, TOut, EPI>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  hipLaunchKernelGGL(kern, dim3(tiles, p.split_k), dim3(256), SMEM_BYTES, s, p);
  return rpo_launch_status();
}

template <typename TIn, typename TOut>
int dispatch_epi(int epi, const GemmParams& p, hipStream_t s) {
  switch (epi) {
    case RPO_EPI_NONE: return launch<TIn, TOut, RPO_EPI_NONE>(p, s);
    case RPO_EPI_BIAS: return launch<TIn, TOut, RPO_EPI_BIAS>(p, s);
    case RPO_EPI_BIAS_QGELU: return launch<TIn, TOut, RPO_EPI_BIAS_QGELU>(p, s);
    case RPO_EPI_QGELU_BWD: return launch<TIn, TOut, RPO_EPI_QGELU_BWD>(p, s);
    default: return RPO_E_DTYPE;
  }
}

template <typename TIn>
int dispatch_f32out(int epi, const GemmParams& p, hipStream_t s) {
  switch (epi) {
    case RPO_EPI_BIAS_RESID: return launch<TIn, float, RPO_EPI_BIAS_RESID>(p, s);
    case RPO_EPI_PATCH: return launch<TIn, float, RPO_EPI_PATCH>(p, s);
    default: return dispatch_epi<TIn, float>(epi, p, s);
  }
}

}  // namespace

extern "C" int rpo_gemm_nt(const rpo_gemm_args* a, void* stream) {
  if (a == nullptr || a->A == nullptr || a->W == nullptr || a->C == nullptr) return RPO_E_BADARG;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return RPO_E_BADARG;
  const bool in_bf16 = a->in_dtype == RPO_BF16, out_bf16 = a->out_dtype == RPO_BF16;
  if ((a->in_dtype != RPO_F32 && !in_bf16) || (a->out_dtype != RPO_F32 && !out_bf16)) return RPO_E_DTYPE;
  if (!in_bf16 && out_bf16) return RPO_E_DTYPE;
  const int bk = in_bf16 ? 64 : 32;
  const int esz = in_bf16 ? 2 : 4, osz = out_bf16 ? 2 : 4;
  if (a->K % bk != 0 || a->N % 4 != 0) return RPO_E_SHAPE;
  if (!aligned16(a->A) || !aligned16(a->W) || (a->lda * esz) % 16 != 0 || (a->ldw * esz) % 16 != 0)
    return RPO_E_ALIGN;
  if ((reinterpret_cast<uintptr_t>(a->C) % (4 * osz)) != 0 || (a->ldc * osz) % (4 * osz) != 0) return RPO_E_ALIGN;
  const int epi = a->epilogue;
  const bool needs_bias = epi == RPO_EPI_BIAS || epi == RPO_EPI_BIAS_QGELU || epi == RPO_EPI_BIAS_RESID;
  if (needs_bias && (a->bias == nullptr || !aligned16(a->bias))) return RPO_E_BADARG;
  if ((epi == RPO_EPI_BIAS_RESID || epi == RPO_EPI_PATCH) &&
      (a->resid == nullptr || !aligned16(a->resid) || a->ldr % 4 != 0 || out_bf16)) return RPO_E_BADARG;
  if (epi == RPO_EPI_QGELU_BWD && a->aux == nullptr) return RPO_E_BADARG;
  if ((epi == RPO_EPI_QGELU_BWD || (epi == RPO_EPI_BIAS_QGELU && a->aux != nullptr)) &&
      (!aligned16(a->aux) || a->ldaux % 4 != 0)) return RPO_E_ALIGN;
  if (epi == RPO_EPI_PATCH && a->group <= 0) return RPO_E_BADARG;

  GemmParams p;
  p.A = static_cast<const char*>(a->A); p.lda = a->lda;
  p.W = static_cast<const char*>(a->W); p.ldw = a->ldw;
  p.C = static_cast<char*>(a->C); p.ldc = a->ldc;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.bias = a->bias; p.resid = a->resid; p.ldr = a->ldr;
  p.aux = static_cast<float*>(a->aux); p.ldaux = a->ldaux; p.aux_row0 = a->aux_row0;
  p.skip_row0 = a->skip_row0; p.skip_col0 = a->skip_col0; p.group = a->group;
  p.split_k = a->split_k <= 1 ? 1 : a->split_k;
  p.split_stride = a->split_stride;
  if (p.split_k > 1 && (epi != RPO_EPI_NONE || out_bf16 || p.split_k > p.K / bk || p.split_stride % 4 != 0))
    return RPO_E_SHAPE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (in_bf16) {
    if (out_bf16) return dispatch_epi<bf16_t, bf16_t>(epi, p, s);
    return dispatch_f32out<bf16_t>(epi, p, s);
  }
  return dispatch_f32out<float>(epi, p, s);
}
