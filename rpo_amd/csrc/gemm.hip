// rpo_gemm_nt: C[M,N] = A[M,K] . W[N,K]^T with fused epilogues, MFMA on gfx950.
//
// Replaces nn.Linear / the packed in-proj and out-proj matmuls of nn.MultiheadAttention
// (reference clip/model.py:171-177,186) and, in the backward, autograd's dX = dY . W.
//
// Tiling (v1): 128x128 block tile, 4 waves in 2x2, each wave a 64x64 sub-tile as 2x2
// MFMA 32x32 tiles.  bf16: v_mfma_f32_32x32x16_bf16, k-tile 64.  f32 (parity mode):
// v_mfma_f32_32x32x2_f32 (exact f32 fma chain), k-tile 32.  Both k-tiles are 128 B per
// row; LDS rows are padded to 144 B so the 16-lane groups of ds_read_b128 hit 16 distinct
// 16-B slots (9*r mod 16 is a bijection on 16 consecutive rows).  Global->register->LDS
// staging, double-buffered, one barrier per k-tile; the next tile's global loads are issued
// before the MFMAs of the current one (guide T14).
//
// The weight tile is the MFMA *A* operand and the activation tile the *B* operand, i.e. the
// wave computes D[n][m]: by the 32x32 C/D map (col = lane&31, row = (reg&3)+8*(reg>>2)+
// 4*(lane>>5)) each lane then owns 4 CONSECUTIVE n for one m per register quad, which
// turns the epilogue's bias / residual loads and the stores into 16-B (f32) / 8-B (bf16)
// vector accesses on row-major C.  The contraction index needs no particular lane order:
// both operands are read with the same (k-step, lane>>5) -> k mapping, so any hardware
// k-permutation cancels.
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
};

constexpr int BM = 128, BN = 128;
constexpr int LROW = 144;                  // bytes per LDS row (128 payload + 16 pad)
constexpr int TILE_BYTES = 128 * LROW;     // one operand tile
constexpr int SMEM_BYTES = 4 * TILE_BYTES; // 2 buffers x (A, W)

template <typename T> struct Tr;
template <> struct Tr<bf16_t> {
  static constexpr int BK = 64, KSTEPS = 4;
  using frag_t = bf16x8_t;
  static __device__ __forceinline__ frag_t ldfrag(const char* rowp, int ks, int half) {
    return *reinterpret_cast<const frag_t*>(rowp + ks * 32 + half * 16);
  }
  static __device__ __forceinline__ f32x16_t mfma(frag_t a, frag_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Tr<float> {
  static constexpr int BK = 32, KSTEPS = 16;
  using frag_t = float;
  static __device__ __forceinline__ frag_t ldfrag(const char* rowp, int ks, int half) {
    return *reinterpret_cast<const float*>(rowp + (ks * 2 + half) * 4);
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
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (p.skip_row0 >= 0 && m0 >= p.skip_row0 && n0 >= p.skip_col0) return;

  // staging assignment: 1024 16-B chunks per operand tile, 4 per thread
  const char* ga[4];
  const char* gw[4];
  int soff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + 256 * i;
    const int row = id >> 3, cc = id & 7;
    const int ra = min(m0 + row, p.M - 1);
    const int rw = min(n0 + row, p.N - 1);
    ga[i] = p.A + ((int64_t)ra * p.lda) * sizeof(TIn) + cc * 16;
    gw[i] = p.W + ((int64_t)rw * p.ldw) * sizeof(TIn) + cc * 16;
    soff[i] = row * LROW + cc * 16;
  }

  f32x16_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  uint4 ra4[4], rw4[4];
  const int nk = p.K / T::BK;
  constexpr int KT_BYTES = T::BK * sizeof(TIn);  // 128

#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ra4[i] = *reinterpret_cast<const uint4*>(ga[i]);
    rw4[i] = *reinterpret_cast<const uint4*>(gw[i]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    *reinterpret_cast<uint4*>(smem + soff[i]) = ra4[i];
    *reinterpret_cast<uint4*>(smem + TILE_BYTES + soff[i]) = rw4[i];
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = (kt + 1) < nk;
    if (more) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra4[i] = *reinterpret_cast<const uint4*>(ga[i] + (int64_t)(kt + 1) * KT_BYTES);
        rw4[i] = *reinterpret_cast<const uint4*>(gw[i] + (int64_t)(kt + 1) * KT_BYTES);
      }
    }
    const char* sx = smem + cur * 2 * TILE_BYTES + (wm * 64 + l31) * LROW;
    const char* sw = smem + cur * 2 * TILE_BYTES + TILE_BYTES + (wn * 64 + l31) * LROW;
#pragma unroll
    for (int ks = 0; ks < T::KSTEPS; ++ks) {
      typename T::frag_t x0 = T::ldfrag(sx, ks, half);
      typename T::frag_t x1 = T::ldfrag(sx + 32 * LROW, ks, half);
      typename T::frag_t w0 = T::ldfrag(sw, ks, half);
      typename T::frag_t w1 = T::ldfrag(sw + 32 * LROW, ks, half);
      acc[0][0] = T::mfma(w0, x0, acc[0][0]);
      acc[0][1] = T::mfma(w0, x1, acc[0][1]);
      acc[1][0] = T::mfma(w1, x0, acc[1][0]);
      acc[1][1] = T::mfma(w1, x1, acc[1][1]);
    }
    if (more) {
      char* dst = smem + (cur ^ 1) * 2 * TILE_BYTES;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<uint4*>(dst + soff[i]) = ra4[i];
        *reinterpret_cast<uint4*>(dst + TILE_BYTES + soff[i]) = rw4[i];
      }
    }
    __syncthreads();
  }

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
        ActIO<TOut>::st4(reinterpret_cast<TOut*>(p.C) + orow * p.ldc + n, v[0], v[1], v[2], v[3]);
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
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), SMEM_BYTES, s, p);
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
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (in_bf16) {
    if (out_bf16) return dispatch_epi<bf16_t, bf16_t>(epi, p, s);
    return dispatch_f32out<bf16_t>(epi, p, s);
  }
  return dispatch_f32out<float>(epi, p, s);
}
