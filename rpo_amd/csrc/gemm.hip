// rpo_gemm_nt: C[M,N] = A[M,K] . W[N,K]^T with fused epilogues, MFMA on gfx950.
//
// Replaces nn.Linear / the packed in-proj and out-proj matmuls of nn.MultiheadAttention
// (reference clip/model.py:171-177,186) and, in the backward, autograd's dX = dY . W.
//
// bf16: v_mfma_f32_32x32x16_bf16, k-tile 64.  f32 (parity mode): v_mfma_f32_32x32x2_f32 (exact f32 fma
// chain), k-tile 32.  Both k-tiles are 128 B per row.  Tile shapes (Cfg below) are chosen per GEMM shape.
//
// Staging is direct HBM->LDS DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction): no staging VGPRs, no
// ds_write pass.  (A register-staged variant -- global_load_dwordx4 + swizzled ds_write_b128 -- measured
// 10-15 % slower at every forward shape.)  The DMA writes LDS lane-linearly (wave-uniform base + lane*16), so
// rows are an unpadded 128 B and the bank-conflict fix is an XOR swizzle applied on the per-lane SOURCE
// address and again on the fragment read (guide rule 21): 16-B chunk c of row r lives at chunk
// c ^ ((r >> 1) & 7), which gives the 16-lane groups of ds_read_b128 16 distinct 16-B slots
// (SQ_LDS_BANK_CONFLICT = 0 in the main loop).
// Two LDS stages, one raw s_barrier per k-tile with a counted vmcnt; the DMA of tile t+1 is issued piecewise
// between the MFMA groups of tile t and operand fragments are double-buffered in registers.  In-kernel
// s_memtime stamps (tools/gemm_timeline.py) show what bounds the loop: a wave spends ~1.1-1.6 k cycles issuing
// one k-tile's body for 256 cycles of its own MFMAs -- an LDS-DMA costs ~100 issue cycles -- so the matrix pipe
// runs at ~50 % inside the loop, and with K = 768 only 12 tiles deep ~35 % of a workgroup's life is prologue
// (first DMA round trip) and epilogue (store drain).  Deeper pipelines (3/4 stages at one workgroup per CU)
// were 20-25 % slower: the limit is in-wave issue, not DMA latency.
//
// The weight tile is the MFMA *A* operand and the activation tile the *B* operand, i.e. the
// wave computes D[n][m]: by the 32x32 C/D map (col = lane&31, row = (reg&3)+8*(reg>>2)+
// 4*(lane>>5)) each lane then owns 4 CONSECUTIVE n for one m per register quad.  The contraction index
// needs no particular lane order: both operands are read with the same (k-step, lane>>5) -> k mapping,
// so any hardware k-permutation cancels.
//
// Workgroup ids are remapped so that each XCD (block b runs on XCD b % 8) owns a contiguous
// range of tiles, walked in groups of 8 m-tiles so the co-resident workgroups of an XCD share their
// A / W panels in its 4 MiB L2.  Optional split-K (gridDim.y slices, each writing its own fp32 slab; the
// consumer sums the slabs in a fixed order, no atomics) keeps the long-K, few-tile dX GEMMs of the
// backward from running on a handful of CUs.
#include "common.h"

#include <type_traits>

namespace {

struct GemmParams {
  const char* A; int64_t lda;
  const char* W; int64_t ldw;
  char* C; int64_t ldc;
  int M, N, K;
  const float* bias;
  const float* resid; int64_t ldr;
  float* aux; int64_t ldaux; int aux_row0;
  int aux_mode;                             // 0: aux holds the fp32 pre-activation u; 1: d quickgelu / du in the act dtype
  int skip_row0, skip_col0, group;
  int split_k; int64_t split_stride;   // elements of C between slabs
  int force_cfg;                       // 0 = heuristic; 2/3/5/6 force a tile shape (benchmarking)
  // LayerNorm fold (include/rpo_amd.h, RPO_EPI_LN_*): producer side = out2 + ln_stats (written), consumer side =
  // ln_stats (read) + ln_colsum
  char* out2; int64_t ldout2;
  float* ln_stats;
  const float* ln_colsum;
  float ln_eps;
  int seg_rows0, seg_rows1, seg1_row0;      // tiling hint: see include/rpo_amd.h
  int ln_group;                             // columns per partial LayerNorm statistic (64, or 96: gemm_w4k.inc)
  const char* pf_ptr; int64_t pf_bytes;     // prefetch hint (include/rpo_amd.h)
  // residual stream as 16-bit hi / lo halves (include/rpo_amd.h): inputs, lo output next to out2, first row with fp32 C
  const char* resid_hi; const char* resid_lo; int64_t ldr16;
  char* out_lo; int c_row0;
};

// Prefetch hint: this workgroup's share of the lines at pf_ptr, one dword per 128-B line and lane, REQUESTED before the
// k-loop and consumed (by an empty asm) after it -- like the LayerNorm statistics the loads are older than every DMA of
// the loop, so its counted vmcnt waits only get stricter, and nothing waits for them before the loop's own first wait.
// (bid, nwg) = this problem's workgroup index / count along x: the whole grid for a plain launch, the problem's own share
// of it in a paired launch (rpo_gemm_nt_pair)
__device__ __forceinline__ uint32_t prefetch_range(const char* ptr, int64_t bytes, int bid, int nwg) {
  uint32_t v = 0;
  if (ptr != nullptr) {
    const int64_t lines = bytes >> 7;
    const int64_t nblk = (int64_t)nwg * gridDim.y;                    // (split-K launches are 2-D)
    const int64_t per = (lines + nblk - 1) / nblk;
    const int64_t l0 = ((int64_t)blockIdx.y * nwg + bid) * per;
    for (int64_t i = threadIdx.x; i < per; i += blockDim.x)
      if (l0 + i < lines) v ^= *reinterpret_cast<const uint32_t*>(ptr + ((l0 + i) << 7));
  }
  return v;
}
__device__ __forceinline__ uint32_t prefetch_touch(const GemmParams& p, int bid, int nwg) {
#ifndef RPO_NO_PREFETCH
  return prefetch_range(p.pf_ptr, p.pf_bytes, bid, nwg);
#else
  return 0;
#endif
}
__device__ __forceinline__ uint32_t prefetch_touch(const GemmParams& p) { return prefetch_touch(p, blockIdx.x, gridDim.x); }

constexpr int LN_GROUP = 64;               // columns per partial LayerNorm statistic written by the generic epilogues

constexpr int LROW = 128;                  // bytes per LDS row (one k-tile, unpadded: DMA is lane-linear)

template <typename T> struct Tr {              // 16-bit storage: bf16_t or f16_t
  static constexpr int BK = 64, KSTEPS = 4;
  using frag_t = bf16x8_t;
  // rowp = start of the LDS row, sw = (row >> 1) & 7
  static __device__ __forceinline__ frag_t ldfrag(const char* rowp, int sw, int ks, int half) {
    return *reinterpret_cast<const frag_t*>(rowp + (((ks * 2 + half) ^ sw) << 4));
  }
  static __device__ __forceinline__ f32x16_t mfma(frag_t a, frag_t b, f32x16_t c) { return mfma16<T>(a, b, c); }
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

// Tile configurations.  BM = WAVES_M * WM_T * 32 rows of A, BN = WAVES_N * WN_T * 32 rows of W,
// NSTAGE LDS buffers of one k-tile each (BM + BN rows x 128 B).
//   CfgMid   128x128, 8 waves (64x32 each), 2 stages (66 KiB, 2 WG/CU): the default.  Issuing one
//            1-KiB LDS-DMA costs a wave ~100 issue cycles, in order with its MFMAs; 8 waves halve the DMA
//            and MFMA share of each and let the 4 waves per SIMD overlap them (2-20 % over 4 waves)
//   CfgBig   256x256, 8 waves, 2 stages (128 KiB, 1 WG/CU): in-proj of the image forward.  A 128x128
//            tile needs 32 KiB per 512 MFMA-cycles = 64 B/clk/CU, which IS the L1/L2->CU rate, so it
//            cannot pass ~50 % MFMA; 256x256 halves the bytes per flop.
// (NSTAGE > 2 is supported by the counted-vmcnt loop below, but 3- and 4-stage 128x128 variants at one WG/CU
//  measured 20-25 % slower than 2 stages at two WG/CU: the loop is bound by in-wave issue, not DMA latency.
//  A persistent variant -- workgroups walking several tiles, next tile's first k-tile fetched under the current
//  epilogue -- was bit-identical and not faster: with two workgroups per CU the hardware already overlaps one
//  workgroup's prologue / epilogue with the other's main loop.)
template <int WAVES_M_, int WAVES_N_, int WM_T_, int WN_T_, int NSTAGE_, int SPREAD_ = 4>
struct Cfg {
  static constexpr int WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, WM_T = WM_T_, WN_T = WN_T_, NSTAGE = NSTAGE_;
  static constexpr int SPREAD = SPREAD_;
  static constexpr int NWAVES = WAVES_M * WAVES_N, THREADS = 64 * NWAVES;
  static constexpr int BM = WAVES_M * WM_T * 32, BN = WAVES_N * WN_T * 32;
  static constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE_BYTES = A_BYTES + W_BYTES;
  // epilogue staging (see the kernel): 256x256 stages the bf16 result, everything else the fp32 accumulators
  static constexpr bool PRECONV_EPI = BM * (BN * 4 + 16) > 140 * 1024;
  static constexpr int CROW_BYTES(bool preconv) { return preconv ? BN * 2 + 16 : BN * 4 + 16; }
  static constexpr int EPI_BYTES = BM * CROW_BYTES(PRECONV_EPI);
  static constexpr int SMEM = NSTAGE * STAGE_BYTES > EPI_BYTES ? NSTAGE * STAGE_BYTES : EPI_BYTES;
  static constexpr int DA = BM / 8 / NWAVES, DW = BN / 8 / NWAVES;   // DMA instructions per wave per k-tile
  static constexpr int DPT = DA + DW;
  static_assert(BM % (8 * NWAVES) == 0 && BN % (8 * NWAVES) == 0, "tile rows must split over the waves");
};
// SPREAD = k-steps (of 4) over which a k-tile's DMA pieces are issued.  Measured (B=32 shapes, MI355X): the 256x256
// and 64x64 tiles gain 6-8 % from issuing everything in the first two k-steps (the last two cover the DMA latency
// before the next vmcnt wait); 128x128 and 64x128 at 2-3 workgroups per CU are best with an even spread.
#ifndef RPO_SPREAD_MID
#define RPO_SPREAD_MID 4
#define RPO_SPREAD_BIG 2
#define RPO_SPREAD_TINY 2
#define RPO_SPREAD_TALL 4
#endif
#ifndef RPO_MID_STAGES
#define RPO_MID_STAGES 2
#endif
using CfgMid = Cfg<2, 4, 2, 1, RPO_MID_STAGES, RPO_SPREAD_MID>;
using CfgBig = Cfg<2, 4, 4, 2, 2, RPO_SPREAD_BIG>;
// The 64x64 tiles serve the prompt-row GEMMs (backward, text tower): latency chains of ~170 small kernels whose
// inputs were just written by the previous kernel, i.e. come from MALL / HBM, not L2.  A third LDS stage (one more
// k-tile of prefetch) is slightly SLOWER in the back-to-back micro-benchmark (L2-hot operands) but makes the whole
// train step 4 % faster (3.68 -> 3.53 ms, three A/B rounds on one box); a fourth stage adds nothing.
#ifndef RPO_TINY_STAGES
#define RPO_TINY_STAGES 3
#endif
#ifndef RPO_TALL_STAGES
#define RPO_TALL_STAGES 2
#endif
// 128x128 owned by FOUR waves (64x64 each: 4 fragment reads feed 4 MFMAs per k-step).  The 8-wave layouts above read
// 1.5 (128x128) / 2 (64x128) fragments per MFMA, i.e. 190 / 250 B/clk of LDS reads at full MFMA rate against the 256
// the LDS delivers: the long-K, N = 768 GEMMs (c_proj, out-proj) were bound by that, not by HBM or the matrix pipe.
#ifndef RPO_SPREAD_QUAD
#define RPO_SPREAD_QUAD 4
#endif
using CfgQuad = Cfg<2, 2, 2, 2, 2, RPO_SPREAD_QUAD>;                  // 128x128, 4 waves
using CfgTiny = Cfg<2, 2, 1, 1, RPO_TINY_STAGES, RPO_SPREAD_TINY>;    // 64x64, 4 waves
using CfgTall = Cfg<2, 4, 1, 1, RPO_TALL_STAGES, RPO_SPREAD_TALL>;    // 64x128, 8 waves
// (Measured and dropped for the prompt-row chains, round 3: 128x64 and 64x128 tiles with 8 waves and 3 stages -- 24 KB of
//  LDS-DMA per k-tile for twice the MACs of a 64x64 tile -- are 5-15 % SLOWER than 64x64 on every chain shape (d c_proj
//  768x3072x768: 12.6 vs 12.0 us; text c_proj 456x512x2048: 15.4 vs 13.1 us): fewer workgroups than CUs.  They win
//  only at M = 1768, i.e. the image forward of a batch of 8 (c_fc 16.9 vs 22.2 us).)

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Workgroup id -> tile origin.  XCD-aware bijective remap (guide T1): XCD x = bid % 8 gets a contiguous run of
// tiles; grouped order: consecutive ids sweep GM m-tiles of one n-tile, then the next n-tile, so the workgroups
// resident on one XCD form a compact super-tile whose A and W panels fit its 4 MiB L2.
#ifndef RPO_GM
#define RPO_GM 8
#endif
template <int BM, int BN, int GM = RPO_GM>
__device__ __forceinline__ void tile_origin(const GemmParams& p, int& m0, int& n0, const int bid, const int nwg) {
  const int tiles_n = (p.N + BN - 1) / BN;      // BM, BN are powers of two or constants: shifts / mul-shift
  int wg;
  {
    const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7;
    wg = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
  }
  const int tiles_m = (p.M + BM - 1) / BM;
  const int per_group = GM * tiles_n;
  // these divisions sit in front of the first DMA of every workgroup: a float reciprocal + one correction step
  // (exact for operands < 2^22) instead of the ~40-instruction integer sequence each
  auto fdiv = [](int a, int b) {
    int q = (int)((float)a * __builtin_amdgcn_rcpf((float)b));
    const int r = a - q * b;
    q += (r >= b) - (r < 0);
    return q;
  };
  const int grp = fdiv(wg, per_group);
  const int gm = min(GM, tiles_m - grp * GM);
  const int in_grp = wg - grp * per_group;
  const int tile_n = fdiv(in_grp, gm);
  const int tile_m = grp * GM + (in_grp - tile_n * gm);
  m0 = tile_m * BM; n0 = tile_n * BN;
}
template <int BM, int BN, int GM = RPO_GM>
__device__ __forceinline__ void tile_origin(const GemmParams& p, int& m0, int& n0) {
  tile_origin<BM, BN, GM>(p, m0, n0, blockIdx.x, gridDim.x);
}

// The saved operand of the QuickGELU backward (rpo_gemm_args.aux / aux_dtype): either the fp32 pre-activation u (the
// backward evaluates the derivative) or, in the 16-bit modes, the derivative itself in the act dtype -- half the bytes
// of the forward's store (the c_fc kernel ended 4.5 us later with the fp32 store than without any), and no
// transcendental in the backward epilogue.  `off` is in elements of the respective type.
template <typename TA>
__device__ __forceinline__ float4 aux_load4(const GemmParams& p, int64_t off) {
  if constexpr (sizeof(TA) == 2) {
    if (p.aux_mode) {
      const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const TA*>(p.aux) + off);
      return make_float4(unpack1<TA>((uint16_t)(u.x & 0xffffu)), unpack1<TA>((uint16_t)(u.x >> 16)),
                         unpack1<TA>((uint16_t)(u.y & 0xffffu)), unpack1<TA>((uint16_t)(u.y >> 16)));
    }
  }
  return *reinterpret_cast<const float4*>(p.aux + off);
}
// what the backward multiplies by, given what aux_load4 returned
__device__ __forceinline__ float4 aux_to_grad(const GemmParams& p, const float4 a) {
  if (p.aux_mode) return a;
  return make_float4(quick_gelu_grad(a.x), quick_gelu_grad(a.y), quick_gelu_grad(a.z), quick_gelu_grad(a.w));
}
// forward side: u = the pre-activation of four consecutive columns
template <typename TA>
__device__ __forceinline__ void aux_store4(const GemmParams& p, int64_t off, const float4 u) {
  if constexpr (sizeof(TA) == 2) {
    if (p.aux_mode) {
      *reinterpret_cast<uint2*>(reinterpret_cast<TA*>(p.aux) + off) =
          make_uint2(pack2<TA>(quick_gelu_grad(u.x), quick_gelu_grad(u.y)), pack2<TA>(quick_gelu_grad(u.z), quick_gelu_grad(u.w)));
      return;
    }
  }
  *reinterpret_cast<float4*>(p.aux + off) = u;
}

// Epilogues that read a second [M, N] operand (residual / saved pre-activation) can fetch it BEFORE the main loop when
// the row-major pass of a thread is a single group of 4 rows (64-row tiles: 4 float4 = 16 VGPRs): the HBM round trip
// then overlaps the k-loop instead of sitting in the epilogue.  Round 6: also the 8 rows of the 128x128 tiles (32 VGPRs
// of the 128 a wave has at two workgroups per CU; the kernel used 76) -- the 24 000-row text tower of the 1000-class
// workload runs its residual / d QuickGELU GEMMs on them with 8-tile k-loops (K = 512), where two exposed round trips
// per tile were a third of the tile's life (-DRPO_EPI_PRE4_ONLY: the round-5 behaviour).
template <int EPI, typename CF>
struct EpiPre {
  static constexpr int CPR = CF::BN / 4, RPP = CF::THREADS / CPR, ROWS = CF::BM / RPP;
#ifdef RPO_EPI_PRE4_ONLY
  static constexpr bool rows_ok = ROWS == 4;
#else
  static constexpr bool rows_ok = ROWS == 4 || ROWS == 8;
#endif
  static constexpr bool value = !CF::PRECONV_EPI && (EPI == RPO_EPI_BIAS_RESID || EPI == RPO_EPI_QGELU_BWD) && rows_ok;
  static constexpr int N = value ? ROWS : 4;
};
template <int EPI, typename CF, typename TA>
__device__ __forceinline__ void epi_preload(const GemmParams& p, int m0, int n0, float4 (&pre)[EpiPre<EPI, CF>::N]) {
  constexpr int CPR = EpiPre<EPI, CF>::CPR, RPP = EpiPre<EPI, CF>::RPP;
  const int tid = threadIdx.x;
  const int cc = tid % CPR, r0 = tid / CPR;
  const int n = n0 + cc * 4;
#pragma unroll
  for (int u = 0; u < EpiPre<EPI, CF>::N; ++u) {
    const int m = m0 + u * RPP + r0;
    pre[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < p.N && m < p.M) {
      if (EPI == RPO_EPI_BIAS_RESID) pre[u] = *reinterpret_cast<const float4*>(p.resid + (int64_t)m * p.ldr + n);
      else pre[u] = aux_load4<TA>(p, (int64_t)m * p.ldaux + n);
    }
  }
}

// LayerNorm fold, consumer side: (mu, rstd) of the tile's A rows from the producer's 64-column partials (mean_g, M2_g),
// combined the numerically safe way (Chan): mu = mean of means, M2 = sum M2_g + 64 (mean_g - mu)^2.  One row per
// thread; the result is parked behind the ring / staging image, which nothing else touches, and is read in the
// epilogue (behind a barrier).  Called before the k-loop so that the two dependent global round trips are hidden.
// (sum of group means, sum of group M2, sum of squared group means) -> (mu, rstd); every operation spelled out so that
// all kernels produce the same bits
__device__ __forceinline__ float2 ln_finish(float mu_sum, float m2_sum, float sq_sum, int G, int K, float eps) {
  const float mu = mu_sum / (float)G;
  const float between = fmaxf(fmaf(-(float)G * mu, mu, sq_sum), 0.f);
  const float m2 = fmaf((float)(K / G), between, m2_sum);          // K / G = columns per group
  return make_float2(mu, rsqrtf(m2 / (float)K + eps));
}
template <typename CF>
__device__ __forceinline__ void ln_row_stats(const GemmParams& p, char* smem, int m0) {
  float2* row_stats = reinterpret_cast<float2*>(smem + CF::SMEM);
  const int G = p.K / p.ln_group;
  for (int r = threadIdx.x; r < CF::BM; r += CF::THREADS) {
    const float2* ps = reinterpret_cast<const float2*>(p.ln_stats) + (int64_t)min(m0 + r, p.M - 1) * G;
    float mu = 0.f, m2 = 0.f, sq = 0.f;
    // single pass: sum of means, sum of M2, sum of squared means (the between-group term is
    // sum (mean_g - mu)^2 = sum mean_g^2 - G mu^2; the means of 64-column groups are O(1), no cancellation issue).
    // All loads of a row are issued before the first is used (a load-use loop costs one L2 round trip per group:
    // 12 in a row were +5 us on c_fc); groups beyond G are clamped loads that the sums mask out.
    for (int g0 = 0; g0 < G; g0 += 16) {
      float2 v[16];
#pragma unroll
      for (int g = 0; g < 16; ++g) v[g] = ps[min(g0 + g, G - 1)];
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const float mg = g0 + g < G ? v[g].x : 0.f;
        mu += mg; m2 += g0 + g < G ? v[g].y : 0.f; sq = fmaf(mg, mg, sq);
      }
    }
    row_stats[r] = ln_finish(mu, m2, sq, G, p.K, p.ln_eps);
  }
}

// TAct = the storage type of the activations (= TIn): what a BIAS_RESID epilogue writes its second copy in
template <typename TOut, int EPI, typename CF, typename TAct = TOut>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16_t (&acc)[CF::WN_T][CF::WM_T], char* smem,
                                              const int m0, const int n0, const float4* pre = nullptr,
                                              const float4* pre_bias = nullptr) {
  constexpr int BM = CF::BM, BN = CF::BN;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave / CF::WAVES_N, wn = wave % CF::WAVES_N;
  // ---- epilogue, staged through LDS -------------------------------------------------------------
  // acc[tn][tm] holds D[n][m] (lane: m = l31, n = 8*g + 4*half + j, reg = 4*g + j).  Storing straight from
  // that layout means 8-B pieces scattered over 32 rows per instruction; measured with s_memtime it cost
  // 9-24 k cycles per workgroup (25-33 % of its lifetime; the store tail is issue-bound, guide T21).  So the
  // tile is first written to LDS (free after the last k-tile), then walked row-major: every global load
  // (residual, saved pre-activation) and store is a fully coalesced 16-B-per-lane access.
  //   PRECONV (256x256, bf16 out, bias only): bias added in the fragment layout, tile staged as bf16
  //   otherwise: tile staged as fp32, bias / residual / QuickGELU applied in the row-major pass
  constexpr bool PRECONV = CF::PRECONV_EPI;
  constexpr int CROW = CF::CROW_BYTES(PRECONV);
  constexpr bool IS_LN = EPI == RPO_EPI_LN_BIAS || EPI == RPO_EPI_LN_BIAS_QGELU;
  constexpr bool IS_QG = EPI == RPO_EPI_BIAS_QGELU || EPI == RPO_EPI_LN_BIAS_QGELU;
  constexpr bool HAS_BIAS = EPI == RPO_EPI_BIAS || EPI == RPO_EPI_BIAS_QGELU || EPI == RPO_EPI_BIAS_RESID || IS_LN;
  // LayerNorm fold, consumer side: (mu, rstd) of this tile's A rows were parked behind the staging image by
  // ln_row_stats() at the start of the kernel (their global round trips hide under the k-loop);
  // v = rstd * (acc - mu * s[n]) + b'[n] below.
  const float2* row_stats = reinterpret_cast<const float2*>(smem + CF::SMEM);
  if constexpr (IS_LN) __syncthreads();      // (also: everybody is done reading the last stage)
  else __builtin_amdgcn_s_barrier();         // everybody is done reading the last stage
  if constexpr (PRECONV) {
    const int nb = n0 + wn * (CF::WN_T * 32) + 4 * half;
    // The pre-activation store of the back-propagated rows exists only in the code path of tiles that hold such rows
    // (uniform test): a per-quad `if` in the common path cuts it into 4-element basic blocks and leaves the
    // mul -> exp -> add -> rcp -> mul chains of QuickGELU without independent work to hide their latency.
    auto convert_and_stage = [&](auto save_u_tag) {
      constexpr bool SAVE_U = decltype(save_u_tag)::value;
  #pragma unroll
      for (int tn = 0; tn < CF::WN_T; ++tn) {      // tn outermost: 4 bias quads live at a time (the one-wave-per-SIMD
        float4 bias_r[4], sum_r[4];                // kernel arrives here with 256 accumulators)
  #pragma unroll
        for (int g = 0; g < 4; ++g) {
          bias_r[g] = HAS_BIAS ? *reinterpret_cast<const float4*>(p.bias + min(nb + tn * 32 + 8 * g, p.N - 4))
                               : make_float4(0.f, 0.f, 0.f, 0.f);
          if constexpr (IS_LN) sum_r[g] = *reinterpret_cast<const float4*>(p.ln_colsum + min(nb + tn * 32 + 8 * g, p.N - 4));
        }
  #pragma unroll
        for (int tm = 0; tm < CF::WM_T; ++tm) {
          float2 st = make_float2(0.f, 1.f);
          if constexpr (IS_LN) st = row_stats[wm * (CF::WM_T * 32) + tm * 32 + l31];
  #pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int row = wm * (CF::WM_T * 32) + tm * 32 + l31;
            const int col = wn * (CF::WN_T * 32) + tn * 32 + 8 * g + 4 * half;
            const float4 b4 = bias_r[g];
            float4 v;
            if constexpr (IS_LN) {
              const float4 s4 = sum_r[g];
              v = make_float4(fmaf(st.y, acc[tn][tm][4 * g] - st.x * s4.x, b4.x), fmaf(st.y, acc[tn][tm][4 * g + 1] - st.x * s4.y, b4.y),
                              fmaf(st.y, acc[tn][tm][4 * g + 2] - st.x * s4.z, b4.z), fmaf(st.y, acc[tn][tm][4 * g + 3] - st.x * s4.w, b4.w));
            } else {
              v = make_float4(acc[tn][tm][4 * g] + b4.x, acc[tn][tm][4 * g + 1] + b4.y,
                              acc[tn][tm][4 * g + 2] + b4.z, acc[tn][tm][4 * g + 3] + b4.w);
            }
            if (IS_QG) {
              const int m = m0 + row, n = n0 + col;   // pre-activation of the back-propagated rows (few)
              if (p.aux != nullptr && m >= p.aux_row0 && m < p.M && n < p.N)
                aux_store4<TAct>(p, (int64_t)(m - p.aux_row0) * p.ldaux + n, v);
              v = quick_gelu4(v);
            }
            if constexpr (sizeof(TOut) == 2)
              *reinterpret_cast<uint2*>(smem + row * CROW + col * 2) =
                  make_uint2(pack2<TOut>(v.x, v.y), pack2<TOut>(v.z, v.w));
          }
        }
      }
    };
    if (IS_QG && p.aux != nullptr && m0 + BM > p.aux_row0) convert_and_stage(std::true_type{});
    else convert_and_stage(std::false_type{});
    __syncthreads();
    constexpr int CPR = BN / 8;                       // 16-B chunks per row
    constexpr int RPP = CF::THREADS / CPR;            // rows per pass
    const int cc = tid % CPR, r0 = tid / CPR;
    const int n = n0 + cc * 8;
#pragma unroll 4
    for (int pass = 0; pass < BM / RPP; ++pass) {
      const int row = pass * RPP + r0;
      const int m = m0 + row;
      const uint4 v = *reinterpret_cast<const uint4*>(smem + row * CROW + cc * 16);
      if (m < p.M && n < p.N)
        store_out16(reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + (int64_t)m * p.ldc + n), v);
    }
  } else {
#pragma unroll
    for (int tm = 0; tm < CF::WM_T; ++tm)
#pragma unroll
      for (int tn = 0; tn < CF::WN_T; ++tn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = wm * (CF::WM_T * 32) + tm * 32 + l31;
          const int col = wn * (CF::WN_T * 32) + tn * 32 + 8 * g + 4 * half;
          *reinterpret_cast<float4*>(smem + row * CROW + col * 4) =
              make_float4(acc[tn][tm][4 * g], acc[tn][tm][4 * g + 1], acc[tn][tm][4 * g + 2], acc[tn][tm][4 * g + 3]);
        }
    __syncthreads();
    constexpr int CPR = BN / 4;                       // float4 chunks per row
    constexpr int RPP = CF::THREADS / CPR;
    constexpr int UNR = 4;                            // rows whose loads are issued together
    const int cc = tid % CPR, r0 = tid / CPR;
    const int n = n0 + cc * 4;
    const bool nok = n < p.N;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (HAS_BIAS && pre_bias != nullptr) b4 = *pre_bias;           // fetched before the main loop
    else if (HAS_BIAS && nok) b4 = *reinterpret_cast<const float4*>(p.bias + n);
    if (IS_LN && nok) s4 = *reinterpret_cast<const float4*>(p.ln_colsum + n);
    TOut* cbase = reinterpret_cast<TOut*>(p.C) + (int64_t)blockIdx.y * p.split_stride;
    auto row_major_pass = [&](auto save_u_tag) {        // (see convert_and_stage above for the tag)
      constexpr bool SAVE_U = decltype(save_u_tag)::value;
      constexpr int PASS_UNROLL = EpiPre<EPI, CF>::value ? EpiPre<EPI, CF>::N / UNR : 1;   // (static indices into pre[])
  #pragma unroll(PASS_UNROLL)
      for (int pass0 = 0; pass0 < BM / RPP; pass0 += UNR) {
        float4 ex[UNR];
        int64_t orow[UNR];
        bool ok[UNR];
  #pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int m = m0 + (pass0 + u) * RPP + r0;
          ok[u] = nok && m < p.M;
          const int mc = min(m, p.M - 1);
          orow[u] = mc;
          ex[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (EpiPre<EPI, CF>::value) {            // loaded before the main loop (see the kernel)
            ex[u] = pre[pass0 + u];
          } else if (EPI == RPO_EPI_PATCH) {
            const int img = mc / p.group;
            orow[u] = (int64_t)mc + img + 1;
            if (ok[u]) ex[u] = *reinterpret_cast<const float4*>(p.resid + (int64_t)(mc - img * p.group + 1) * p.ldr + n);
          } else if (EPI == RPO_EPI_BIAS_RESID) {
            if (ok[u]) ex[u] = *reinterpret_cast<const float4*>(p.resid + (int64_t)mc * p.ldr + n);
          } else if (EPI == RPO_EPI_QGELU_BWD) {
            if (ok[u]) ex[u] = aux_load4<TAct>(p, (int64_t)mc * p.ldaux + n);
          }
        }
  #pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int row = (pass0 + u) * RPP + r0;
          const int m = m0 + row;
          float4 v = *reinterpret_cast<const float4*>(smem + row * CROW + cc * 16);
          if constexpr (IS_LN) {
            const float2 st = row_stats[row];
            v.x = fmaf(st.y, v.x - st.x * s4.x, b4.x); v.y = fmaf(st.y, v.y - st.x * s4.y, b4.y);
            v.z = fmaf(st.y, v.z - st.x * s4.z, b4.z); v.w = fmaf(st.y, v.w - st.x * s4.w, b4.w);
          } else {
            v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
          }
          if (IS_QG) {
            if (SAVE_U && ok[u] && m >= p.aux_row0) aux_store4<TAct>(p, (int64_t)(m - p.aux_row0) * p.ldaux + n, v);
            v = quick_gelu4(v);
          }
          if (EPI == RPO_EPI_BIAS_RESID || EPI == RPO_EPI_PATCH) {
            v.x += ex[u].x; v.y += ex[u].y; v.z += ex[u].z; v.w += ex[u].w;
          }
          if (EPI == RPO_EPI_QGELU_BWD) {
            const float4 gq = aux_to_grad(p, ex[u]);
            v.x *= gq.x; v.y *= gq.y; v.z *= gq.z; v.w *= gq.w;
          }
          if (ok[u]) ActIO<TOut>::st4(cbase + orow[u] * p.ldc + n, v.x, v.y, v.z, v.w);
          if constexpr (EPI == RPO_EPI_BIAS_RESID && sizeof(TAct) == 2) {
            // LayerNorm fold, producer side: the 16-bit copy the consuming GEMM reads as its A operand, and the
            // (mean, sum of squared deviations) of each 64-column group of the row: 16 consecutive lanes hold one group
            if (p.out2 != nullptr && ok[u])
              ActIO<TAct>::st4(reinterpret_cast<TAct*>(p.out2) + orow[u] * p.ldout2 + n, v.x, v.y, v.z, v.w);
            if (p.ln_stats != nullptr) {
              const float sm = row16_sum((v.x + v.y) + (v.z + v.w));
              const float mean = sm * (1.0f / LN_GROUP);
              const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
              const float q = row16_sum((dx * dx + dy * dy) + (dz * dz + dw * dw));
              if (ok[u] && (cc & 15) == 0)
                reinterpret_cast<float2*>(p.ln_stats)[orow[u] * (p.N / LN_GROUP) + (n >> 6)] = make_float2(mean, q);
            }
          }
        }
      }
    };
    if (IS_QG && p.aux != nullptr && m0 + BM > p.aux_row0) row_major_pass(std::true_type{});
    else row_major_pass(std::false_type{});
  }
}

// The kernel body: workgroup `bid` of the `nwg` that tile problem p (the whole grid along x for a plain launch; a paired
// launch gives each of its two problems a contiguous share of the grid, rpo_gemm_nt_pair).
template <typename TIn, typename TOut, int EPI, typename CF>
__device__ __forceinline__ void gemm_nt_body(const GemmParams& p, char* smem, const int bid, const int nwg) {
  using T = Tr<TIn>;
  constexpr int BM = CF::BM, BN = CF::BN, NSTAGE = CF::NSTAGE;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave / CF::WAVES_N, wn = wave % CF::WAVES_N;

  RPO_STAMP(0);
  int m0, n0;
  tile_origin<BM, BN>(p, m0, n0, bid, nwg);
  if (p.skip_row0 >= 0 && m0 >= p.skip_row0 && n0 >= p.skip_col0) return;

  // k-range of this split
  const int nk_all = p.K / T::BK;
  const int kt0 = (int)(((int64_t)nk_all * blockIdx.y) / p.split_k);
  const int kt1 = (int)(((int64_t)nk_all * (blockIdx.y + 1)) / p.split_k);
  const int nk = kt1 - kt0;
  constexpr int KT_BYTES = T::BK * sizeof(TIn);  // 128

  // DMA assignment: an operand tile of R rows = R/8 wave-instructions of 1 KiB (8 rows each); wave w issues
  // instructions i*NWAVES + w.  Lane l -> row 8*(i*NWAVES+w) + (l>>3), physical chunk l&7, which must
  // receive logical chunk (l&7) ^ ((row>>1)&7) of that row.
  const char* ga[CF::DA];
  const char* gw[CF::DW];
#pragma unroll
  for (int i = 0; i < CF::DA; ++i) {
    const int row = 8 * (i * CF::NWAVES + wave) + (lane >> 3);
    const int cl = (lane & 7) ^ ((row >> 1) & 7);
    ga[i] = p.A + ((int64_t)min(m0 + row, p.M - 1) * p.lda) * sizeof(TIn) + cl * 16 + (int64_t)kt0 * KT_BYTES;
  }
#pragma unroll
  for (int i = 0; i < CF::DW; ++i) {
    const int row = 8 * (i * CF::NWAVES + wave) + (lane >> 3);
    const int cl = (lane & 7) ^ ((row >> 1) & 7);
    gw[i] = p.W + ((int64_t)min(n0 + row, p.N - 1) * p.ldw) * sizeof(TIn) + cl * 16 + (int64_t)kt0 * KT_BYTES;
  }
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // one DMA instruction (1 KiB): piece i < DA belongs to the A tile, the rest to the W tile
  auto dma_piece = [&](int stage, int kt, int i) {
    char* b_ = smem + stage * CF::STAGE_BYTES + wave * 1024;
    const int64_t ko = (int64_t)kt * KT_BYTES;
    if (i < CF::DA)
      __builtin_amdgcn_global_load_lds((gptr_t)(ga[i] + ko), (lptr_t)(b_ + i * CF::NWAVES * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((gptr_t)(gw[i - CF::DA] + ko),
                                       (lptr_t)(b_ + CF::A_BYTES + (i - CF::DA) * CF::NWAVES * 1024), 16, 0, 0);
  };
  auto dma = [&](int stage, int kt) {
#pragma unroll
    for (int i = 0; i < CF::DPT; ++i) dma_piece(stage, kt, i);
  };

  f32x16_t acc[CF::WN_T][CF::WM_T];
#pragma unroll
  for (int a = 0; a < CF::WN_T; ++a)
#pragma unroll
    for (int b = 0; b < CF::WM_T; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  // fragment rows of this lane (+32 rows per extra MFMA tile keeps (row >> 1) & 7 unchanged)
  const int rx = wm * (CF::WM_T * 32) + l31, rwv = wn * (CF::WN_T * 32) + l31;
  const int swx = (rx >> 1) & 7, sww = (rwv >> 1) & 7;

#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) dma(s, s);
  if constexpr (EPI == RPO_EPI_LN_BIAS || EPI == RPO_EPI_LN_BIAS_QGELU) ln_row_stats<CF>(p, smem, m0);
  // bias of this thread's 4 output columns (row-major epilogue pass), fetched now so its latency hides in the k-loop
  float4 pre_b = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (!CF::PRECONV_EPI && (EPI == RPO_EPI_BIAS || EPI == RPO_EPI_BIAS_QGELU || EPI == RPO_EPI_BIAS_RESID ||
                                     EPI == RPO_EPI_LN_BIAS || EPI == RPO_EPI_LN_BIAS_QGELU)) {
    const int n = n0 + (tid % (CF::BN / 4)) * 4;
    if (n < p.N) pre_b = *reinterpret_cast<const float4*>(p.bias + n);
  }
  float4 pre[EpiPre<EPI, CF>::N];
  if constexpr (EpiPre<EPI, CF>::value) {
    epi_preload<EPI, CF, TIn>(p, m0, n0, pre);     // plain loads: they count in vmcnt like the DMA, issued in order before the
                                              // in-loop DMA, so the counted waits below stay valid (conservative)
  }
  const uint32_t pf_touch = prefetch_touch(p, bid, nwg);   // the same holds for the prefetch hint's loads (GemmParams::pf_ptr)

#ifdef RPO_TIMELINE
  unsigned long long t_wait = 0, t_bar = 0, t_body = 0, t_a, t_b, t_c, t_d;
#endif
  for (int kt = 0; kt < nk; ++kt) {
#ifdef RPO_TIMELINE
    t_a = __builtin_amdgcn_s_memtime();
#endif
    // tiles already issued: up to kt + NSTAGE - 2; tile kt must have landed
    const int ahead = min(kt + NSTAGE - 2, nk - 1) - kt;
    if (NSTAGE == 2 || ahead <= 0) wait_vmcnt<0>();
    else if (ahead == 1) wait_vmcnt<CF::DPT>();
    else if (NSTAGE > 3 && ahead == 2) wait_vmcnt<2 * CF::DPT>();
    else wait_vmcnt<0>();
#ifdef RPO_TIMELINE
    t_b = __builtin_amdgcn_s_memtime();
#endif
    __builtin_amdgcn_s_barrier();   // tile kt landed for every wave; everybody is done with tile kt-1
#ifdef RPO_TIMELINE
    t_c = __builtin_amdgcn_s_memtime();
#endif
    RPO_STAMP(2 + min(kt, 50));
    // The DMA of tile kt+NSTAGE-1 is issued piecewise BETWEEN the MFMA groups of tile kt.  Issuing one LDS-DMA
    // costs the wave ~100 cycles; issued as a block right after the barrier, every wave of the workgroup pays
    // that before its first MFMA and the matrix pipe idles (measured, one workgroup alone on the chip:
    // ~1500 cycles per k-tile for 512 cycles of MFMA).  Interleaved, the issue hides behind running MFMAs.
    const bool more = kt + NSTAGE - 1 < nk;
    const int nstage = (kt + NSTAGE - 1) % NSTAGE, nkt = kt + NSTAGE - 1;
    // pieces per k-step: spread over the first SPREAD k-steps (bf16: of 4), the rest cover the DMA latency
    constexpr int SPR = CF::SPREAD < T::KSTEPS ? CF::SPREAD * (T::KSTEPS / 4) : T::KSTEPS;
    constexpr int PPS = (CF::DPT + SPR - 1) / SPR;
    const char* st = smem + (kt % NSTAGE) * CF::STAGE_BYTES;
    const char* sx = st + rx * LROW;
    const char* sw = st + CF::A_BYTES + rwv * LROW;
    // fragments are double-buffered in registers: the ds_reads of k-step ks+1 are in flight while the MFMAs
    // of k-step ks run (static indices after unrolling)
    typename T::frag_t xf[2][CF::WM_T], wf[2][CF::WN_T];
    auto ldfrags = [&](int buf, int ks) {
#pragma unroll
      for (int tm = 0; tm < CF::WM_T; ++tm) xf[buf][tm] = T::ldfrag(sx + tm * 32 * LROW, swx, ks, half);
#pragma unroll
      for (int tn = 0; tn < CF::WN_T; ++tn) wf[buf][tn] = T::ldfrag(sw + tn * 32 * LROW, sww, ks, half);
    };
    ldfrags(0, 0);
#pragma unroll
    for (int ks = 0; ks < T::KSTEPS; ++ks) {
      if (ks + 1 < T::KSTEPS) ldfrags((ks + 1) & 1, ks + 1);
#pragma unroll
      for (int tn = 0; tn < CF::WN_T; ++tn)
#pragma unroll
        for (int tm = 0; tm < CF::WM_T; ++tm) acc[tn][tm] = T::mfma(wf[ks & 1][tn], xf[ks & 1][tm], acc[tn][tm]);
      if (more) {
#pragma unroll
        for (int i = ks * PPS; i < (ks + 1) * PPS && i < CF::DPT; ++i) dma_piece(nstage, nkt, i);
      }
    }
#ifdef RPO_TIMELINE
    t_d = __builtin_amdgcn_s_memtime();
    t_wait += t_b - t_a; t_bar += t_c - t_b; t_body += t_d - t_c;
#endif
  }
#ifdef RPO_TIMELINE
  if (g_timeline != nullptr && tid == 0 && (blockIdx.x % 97) == 0 && blockIdx.x / 97 < 8) {
    g_timeline[(blockIdx.x / 97) * 64 + 53] = t_wait;
    g_timeline[(blockIdx.x / 97) * 64 + 54] = t_bar;
    g_timeline[(blockIdx.x / 97) * 64 + 55] = t_body;
  }
#endif
  asm volatile("" :: "v"(pf_touch));


  RPO_STAMP(60);
  gemm_epilogue<TOut, EPI, CF, TIn>(p, acc, smem, m0, n0, pre,
                                    CF::PRECONV_EPI ? nullptr : &pre_b);
#ifdef RPO_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  RPO_STAMP(61);
}

template <typename TIn, typename TOut, int EPI, typename CF>
__global__ __launch_bounds__(CF::THREADS) void gemm_nt_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm_nt_body<TIn, TOut, EPI, CF>(p, smem, blockIdx.x, gridDim.x);
}

// Two independent problems of the same kind (dtypes, epilogue, tile shape, split-K factor) in ONE launch: workgroups
// [0, tiles0) tile problem 0, the rest problem 1.  The two prompt-row chains of a step (image tower / text tower) have the
// same stages; launched pairwise they are one chain of kernels on one queue instead of two chains on two queues whose
// kernels delay each other (profiles/README.md, round 3).
// A problem may get fewer workgroups than it has tiles (its workgroups then walk tiles bid, bid + wgs, ...): the launcher
// trims the problem with the shorter k-loop until both fit the CUs in ONE round -- a second round for a few dozen
// workgroups doubles the duration of a launch that is one link of a latency chain.
struct GemmPair { GemmParams p[2]; int tiles[2]; int wgs0; };
template <typename TIn, typename TOut, int EPI, typename CF>
__global__ __launch_bounds__(CF::THREADS) void gemm_nt_pair_kernel(const GemmPair g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int second = (int)blockIdx.x >= g.wgs0;                    // uniform
  const int bid = second ? (int)blockIdx.x - g.wgs0 : (int)blockIdx.x;
  const int nwg = second ? (int)gridDim.x - g.wgs0 : g.wgs0;
  const int tiles = g.tiles[second];
  for (int t = bid; t < tiles; t += nwg) {
    if (t != bid) {                                                // the previous tile's stores are out, its staging image read
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    if (second) gemm_nt_body<TIn, TOut, EPI, CF>(g.p[1], smem, t, tiles);
    else gemm_nt_body<TIn, TOut, EPI, CF>(g.p[0], smem, t, tiles);
  }
}

// ---- ping-pong 256x256 kernel (bf16) ---------------------------------------------------------------------
// The lock-step loop above leaves the matrix pipe idle whenever a wave reads fragments, issues DMA or waits at the
// barrier, and both waves of a SIMD do that at the same time (s_memtime: ~3.6 k cycles per 64-deep k-tile for 2.05 k
// cycles of MFMA).  Here the 8 waves form two groups (wm = 0 / 1: the two 128-row halves of the tile), one wave of
// each group per SIMD, and time is cut into slots by raw barriers: in every slot ONE group issues nothing but the 16
// MFMAs of a 32-deep k-tile (priority raised) while the OTHER issues the 12 fragment reads of its next k-tile and its
// share of the LDS-DMA; the roles swap at each barrier.  Group 1 runs one slot behind group 0.
//
// LDS: ring of 4 slots x (256 + 256 rows x 64 B) = 128 KiB; k-tile t lives in ring slot t % 4.  A 64-B row holds 4
// 16-B chunks; chunk c of row r is stored at c ^ ((r >> 2) & 3): the 16-lane groups of ds_read_b128
// ({0-3,12-15,20-27}, {4-11,16-19,28-31} + 32) then hit 16 distinct 16-B bank groups.
//
// Schedule (slot j, k-tile t; both groups read the same tile sequence):
//   group 0: slot 2t   reads fragments of t, issues DMA of t+2;            slot 2t+1 MFMAs of t, then vmcnt
//   group 1: slot 2t+1 reads fragments of t, issues DMA of t+3, vmcnt;     slot 2t+2 MFMAs of t
// RAW: tile T is first read in slot 2T; every wave retires its own pieces of T with a counted vmcnt at the end of slot
//      2T-1, before the barrier that opens slot 2T (group 0 then has T+1 in flight -> vmcnt(4); group 1 has T+1, T+2
//      -> vmcnt(8)).  The read happens one slot AFTER the wait, never in the same slot.
// Measured (in-proj 7072x2304x768, MI355X; tools/gemm_timeline.py, tools/build_variant.sh ablations): bit-identical
// to the lock-step 256x256 kernel and 3-8 % faster (30.8-32.3 vs 32.8-34.7 us across boxes).  A slot takes ~830 cycles
// for 512 cycles of MFMA: the LOAD side is the long one (~700: 12 ds_read_b128 + 4 LDS-DMA per wave), the compute
// side ~600.  Ablations: without the in-loop DMA the kernel drops to 25.1 us (~600-cycle slots), without the fragment
// reads only to 30.8 us, so the 16 LDS-DMA instructions per slot (16 KiB into LDS) cost ~270 cycles of critical path
// wherever they are issued -- before the reads (slower), inside the MFMA stream (same), split over both groups
// (slower); hot (L2-resident) source addresses or dropping the vmcnt waits change nothing, i.e. it is the CU-side
// DMA path (TA at 64 B/clk plus its LDS writes), not memory latency.
// WAR: tile t is last read by group 1 in slot 2t+1 and those ds_reads have retired (lgkmcnt) inside slot 2t+2; its ring
//      slot is refilled with tile t+4 by group 1 in slot 2t+3 and by group 0 in slot 2t+4, both behind a barrier.
struct CfgPP {
  static constexpr int WAVES_M = 2, WAVES_N = 4, WM_T = 4, WN_T = 2, NWAVES = 8, THREADS = 512;
  static constexpr int BM = 256, BN = 256, BK = 32, NRING = 4, ROWB = 64;
  static constexpr int A_BYTES = BM * ROWB, SLOT_BYTES = (BM + BN) * ROWB;
  static constexpr bool PRECONV_EPI = true;
  static constexpr int CROW_BYTES(bool) { return BN * 2 + 16; }
  static constexpr int EPI_BYTES = BM * (BN * 2 + 16);
  static constexpr int SMEM = NRING * SLOT_BYTES > EPI_BYTES ? NRING * SLOT_BYTES : EPI_BYTES;
  static constexpr int PIECES = 4;     // DMA instructions per wave per k-tile: 2 A + 2 W (16 rows x 64 B each)
};

template <int N> __device__ __forceinline__ void pp_vmwait(int later) {   // `later` tiles of mine may stay in flight
  if (later >= 2) wait_vmcnt<2 * N>();
  else if (later == 1) wait_vmcnt<N>();
  else wait_vmcnt<0>();
}


template <typename TOut, int EPI>      // TOut = the 16-bit storage type of all three matrices
__global__ __launch_bounds__(CfgPP::THREADS) void gemm_pp_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using CF = CfgPP;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave / CF::WAVES_N, wn = wave % CF::WAVES_N;
  RPO_STAMP(0);
  int m0, n0;
  tile_origin<CF::BM, CF::BN>(p, m0, n0);
  if (p.skip_row0 >= 0 && m0 >= p.skip_row0 && n0 >= p.skip_col0) return;
  const int nk = p.K / CF::BK;

  // DMA: one instruction = 16 rows x 64 B; wave w owns pieces w and 8 + w of the A tile and of the W tile.
  // lane -> row 16 * piece + (lane >> 2), physical chunk lane & 3, which receives logical chunk (lane & 3) ^ ((row >> 2) & 3)
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const char* ga[2];
  const char* gw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 16 * (i * 8 + wave) + (lane >> 2);
    const int lc = (lane & 3) ^ ((row >> 2) & 3);
    ga[i] = p.A + (int64_t)min(m0 + row, p.M - 1) * p.lda * 2 + lc * 16;
    gw[i] = p.W + (int64_t)min(n0 + row, p.N - 1) * p.ldw * 2 + lc * 16;
  }
  auto dma = [&](int t) {
    char* b_ = smem + (t & 3) * CF::SLOT_BYTES + wave * 1024;
    const int64_t ko = (int64_t)t * (CF::BK * 2);
    __builtin_amdgcn_global_load_lds((gptr_t)(ga[0] + ko), (lptr_t)(b_), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(gw[0] + ko), (lptr_t)(b_ + CF::A_BYTES), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(ga[1] + ko), (lptr_t)(b_ + 8 * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(gw[1] + ko), (lptr_t)(b_ + CF::A_BYTES + 8 * 1024), 16, 0, 0);
  };

  f32x16_t acc[CF::WN_T][CF::WM_T];
#pragma unroll
  for (int a = 0; a < CF::WN_T; ++a)
#pragma unroll
    for (int b = 0; b < CF::WM_T; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  // fragment addresses inside a ring slot: rows wm*128 + tm*32 + l31 (A) / wn*64 + tn*32 + l31 (W); the swizzle term
  // (row >> 2) & 3 depends on l31 only.  k-step ks (16 deep) = chunks 2*ks + half.
  const int sw = (l31 >> 2) & 3;
  const int offx = (wm * 128 + l31) * CF::ROWB, offw = CF::A_BYTES + (wn * 64 + l31) * CF::ROWB;
  const int c0 = ((0 + half) ^ sw) << 4, c1 = ((2 + half) ^ sw) << 4;
  bf16x8_t xf[2][CF::WM_T], wf[2][CF::WN_T];
  auto ldfrags = [&](int t) {
    const char* st = smem + (t & 3) * CF::SLOT_BYTES;
#pragma unroll
    for (int tn = 0; tn < CF::WN_T; ++tn) {
      wf[0][tn] = *reinterpret_cast<const bf16x8_t*>(st + offw + tn * 32 * CF::ROWB + c0);
      wf[1][tn] = *reinterpret_cast<const bf16x8_t*>(st + offw + tn * 32 * CF::ROWB + c1);
    }
#pragma unroll
    for (int tm = 0; tm < CF::WM_T; ++tm) {
      xf[0][tm] = *reinterpret_cast<const bf16x8_t*>(st + offx + tm * 32 * CF::ROWB + c0);
      xf[1][tm] = *reinterpret_cast<const bf16x8_t*>(st + offx + tm * 32 * CF::ROWB + c1);
    }
  };
  auto mfmas = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int tn = 0; tn < CF::WN_T; ++tn)
#pragma unroll
        for (int tm = 0; tm < CF::WM_T; ++tm)
          acc[tn][tm] = mfma16<TOut>(wf[ks][tn], xf[ks][tm], acc[tn][tm]);
    __builtin_amdgcn_s_setprio(0);
  };
  auto slot_end = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

#ifdef RPO_TIMELINE
  // per-section cycle sums of this wave (no stores inside the loop: a store would count in vmcnt)
  unsigned long long tl_load = 0, tl_lbar = 0, tl_comp = 0, tl_cbar = 0, tl_0, tl_1;
#define PP_T0() tl_0 = __builtin_amdgcn_s_memtime()
#define PP_ACC(v) do { tl_1 = __builtin_amdgcn_s_memtime(); v += tl_1 - tl_0; tl_0 = tl_1; } while (0)
#else
#define PP_T0() do { } while (0)
#define PP_ACC(v) do { } while (0)
#endif
  if constexpr (EPI == RPO_EPI_LN_BIAS || EPI == RPO_EPI_LN_BIAS_QGELU) ln_row_stats<CF>(p, smem, m0);
  if (wm == 0) {
    // prologue: tiles 0, 1 in flight; tile 0 retired before the barrier that opens slot 0
    if (0 < nk) dma(0);
    if (1 < nk) dma(1);
    pp_vmwait<CF::PIECES>(min(nk - 1, 1));
    slot_end();
    RPO_STAMP(2);
    PP_T0();
    for (int t = 0; t < nk; ++t) {
      // slot 2t: load
      ldfrags(t);
      if (t + 2 < nk) dma(t + 2);
      PP_ACC(tl_load);
      slot_end();
      PP_ACC(tl_lbar);
      // slot 2t+1: compute, then retire my pieces of tile t+1 (tile t+2 may stay in flight)
      mfmas();
      pp_vmwait<CF::PIECES>(t + 2 < nk ? 1 : 0);
      PP_ACC(tl_comp);
      slot_end();
      PP_ACC(tl_cbar);
    }
    slot_end();                        // slot 2nk: group 1 computes its last tile
  } else {
    if (0 < nk) dma(0);
    if (1 < nk) dma(1);
    if (2 < nk) dma(2);
    pp_vmwait<CF::PIECES>(min(nk - 1, 2));
    slot_end();
    slot_end();                        // slot 0: group 0 loads its first tile
    PP_T0();
    for (int t = 0; t < nk; ++t) {
      // slot 2t+1: load, DMA, retire my pieces of tile t+1 (t+2, t+3 may stay in flight)
      ldfrags(t);
      if (t + 3 < nk) dma(t + 3);
      pp_vmwait<CF::PIECES>(min(nk - 1, t + 3) - min(nk - 1, t + 1));
      PP_ACC(tl_load);
      slot_end();
      PP_ACC(tl_lbar);
      // slot 2t+2: compute
      mfmas();
      PP_ACC(tl_comp);
      slot_end();
      PP_ACC(tl_cbar);
    }
  }
#ifdef RPO_TIMELINE
  if (g_timeline != nullptr && lane == 0 && (wave == 0 || wave == 4) && (blockIdx.x % 97) == 0 && blockIdx.x / 97 < 8) {
    unsigned long long* o = g_timeline + (blockIdx.x / 97) * 64 + 40 + (wave >> 2) * 4;
    o[0] = tl_load; o[1] = tl_lbar; o[2] = tl_comp; o[3] = tl_cbar;
  }
#endif
  RPO_STAMP(60);
  gemm_epilogue<TOut, EPI, CF>(p, acc, smem, m0, n0);
#ifdef RPO_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  RPO_STAMP(61);
}

#include "gemm_w4.inc"
#include "gemm_w4g.inc"
#include "gemm_w4k.inc"
#include "gemm_mlp.inc"

template <typename TOut, int EPI>
int launch_pp(const GemmParams& p, hipStream_t s) {
  static rpo_lds_mask_t lds_ok{0};
  auto kern = gemm_pp_kernel<TOut, EPI>;
  constexpr int smem_bytes = CfgPP::SMEM + CfgPP::BM * 8;     // + (mu, rstd) per row of the LayerNorm-fold epilogues
  if (int rc = rpo_allow_lds(reinterpret_cast<const void*>(kern), smem_bytes, &lds_ok)) return rc;
  const int tiles = ((p.M + CfgPP::BM - 1) / CfgPP::BM) * ((p.N + CfgPP::BN - 1) / CfgPP::BN);
  hipLaunchKernelGGL(kern, dim3(tiles, 1), dim3(CfgPP::THREADS), smem_bytes, s, p);
  return rpo_launch_status();
}

template <typename TIn, typename TOut, int EPI, typename CF>
int launch_cfg(const GemmParams& p, hipStream_t s) {
  static rpo_lds_mask_t lds_ok{0};
  auto kern = gemm_nt_kernel<TIn, TOut, EPI, CF>;
  constexpr int smem_bytes = CF::SMEM + CF::BM * 8;           // + (mu, rstd) per row of the LayerNorm-fold epilogues
  if (int rc = rpo_allow_lds(reinterpret_cast<const void*>(kern), smem_bytes, &lds_ok)) return rc;
  const int tiles = ((p.M + CF::BM - 1) / CF::BM) * ((p.N + CF::BN - 1) / CF::BN);
  hipLaunchKernelGGL(kern, dim3(tiles, p.split_k), dim3(CF::THREADS), smem_bytes, s, p);
  return rpo_launch_status();
}

// shape heuristic (measured on MI355X, tools/bench_gemm.py): see the Cfg comments
template <typename TIn, typename TOut, int EPI>
int launch(const GemmParams& p, hipStream_t s) {
  // N = 768-class GEMMs with a residual epilogue (out-proj, c_proj): one round of 224x96 split-k tiles when the caller's
  // row units allow it (gemm_w4k.inc).  Its row statistics are over 96 columns, the generic epilogues' over 64: the
  // caller says which it expects (rpo_gemm_args.ln_group) and gets an error instead of the other layout.
  if constexpr (EPI == RPO_EPI_BIAS_RESID && sizeof(TIn) == 2 && sizeof(TOut) == 4) {
    W4KPlan kplan;
    const bool fits32 = (int64_t)p.M * p.lda * 2 < (1ll << 31) && (int64_t)p.N * p.ldw * 2 < (1ll << 31);
    const int kgrp = w4k_plan(p, &kplan);                           // 0, or the geometry's statistics group (96 / 64)
    const bool k_ok = p.split_k == 1 && fits32 && kgrp != 0 && aligned16(p.C) && p.ldc % 4 == 0 &&
                      (p.ln_stats == nullptr || p.ln_group == kgrp);
    if (k_ok && (p.force_cfg == 11 || p.force_cfg == 0)) return launch_w4k<TIn>(p, s);
    if (p.force_cfg == 11 || (p.ln_stats != nullptr && p.ln_group != LN_GROUP)) return RPO_E_SHAPE;
  }
  if (p.resid_hi != nullptr || p.out_lo != nullptr || p.c_row0 > 0) return RPO_E_SHAPE;   // only the kernel above implements hi / lo
  // measured (tools/bench_gemm.py, profiles/): 256x256 wins for the wide-N forward GEMMs of the image tower
  // (in-proj 31.6 vs 36.1 us); a 4-stage 128x128 variant was slower than 2 stages on every shape
  constexpr bool big_ok = sizeof(TIn) == 2 && sizeof(TOut) == 2 &&
                          (EPI == RPO_EPI_BIAS || EPI == RPO_EPI_BIAS_QGELU || EPI == RPO_EPI_LN_BIAS || EPI == RPO_EPI_LN_BIAS_QGELU);
  if constexpr (big_ok) {
    // one 256x256 workgroup per CU: only worth it when the tiles fill whole rounds of the 256 CUs
    // (in-proj at B=32: 252 tiles; c_fc: 336 tiles = 1.3 rounds -> 65 us vs 55 us with 128x128; ViT-L/14 in-proj at
    //  B=16: 216 tiles = 84 % of one round, step 7.42 -> 7.28 ms with it)
    const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    const int rounds = (tiles + 255) / 256;
#ifndef RPO_FILL_PCT
#define RPO_FILL_PCT 80
#endif
#ifndef RPO_W4G_MOD
#define RPO_W4G_MOD 256
#endif
    const bool fills = tiles * 100 >= rounds * 256 * RPO_FILL_PCT;
    const bool ok = p.N % 8 == 0 && p.ldc % 8 == 0 && p.split_k == 1 && aligned16(p.C);
    // one-wave-per-SIMD kernel (gemm_w4.inc): 32-bit byte offsets inside the operand matrices, at least two 64-deep
    // k-tiles.  Bit-identical to the ping-pong kernel (tile_config 7) and the lock-step 256x256 one (3) and faster
    // than both (in-proj at B=32: 28.1 vs 34.3 / 29.5 us on one box), so it is what the heuristic picks.
    const bool fits32 = (int64_t)p.M * p.lda * 2 < (1ll << 31) && (int64_t)p.N * p.ldw * 2 < (1ll << 31);
    constexpr bool epi_ln = EPI == RPO_EPI_LN_BIAS || EPI == RPO_EPI_LN_BIAS_QGELU;
    const bool w4_ok = ok && fits32 && p.K >= 2 * CfgW4::BK && (!epi_ln || p.K <= 16 * p.ln_group);
    const bool wants_big = p.force_cfg == 0 && p.M >= 2048 && p.N >= 1536 && fills;
    // 224x384 tiles when they cover the output in exactly one round and 256x256 tiles do not (c_fc at B = 32)
    {
      W4GPlan gplan;
      const bool g_ok = w4_ok && w4g_plan(p, &gplan) != 0;
      const bool big_shape = p.force_cfg == 0 && p.M >= 2048 && p.N >= 1536;
      // ... or whole rounds of row-unit tiles (64 / 128 images: 2 / 4 rounds; 256x256 tiles would need 2.6 / 5.3)
      if (g_ok && (p.force_cfg == 10 || (big_shape && (!fills || (gplan.from_units && gplan.tiles_m * gplan.tiles_n % RPO_W4G_MOD == 0)))))
        return launch_w4g<TOut, EPI>(p, s);
    }
    // a single round that fills at least 55 % of the CUs still favours the one-wave-per-SIMD kernel (K / V projection of
    // the frozen rows in the last image block, 6304 x 1536 x 768 = 150 tiles: 22.1 vs 25.6 us ping-pong, 28.6 us 128x128)
    const bool one_round_ok = p.force_cfg == 0 && p.M >= 2048 && p.N >= 1536 && rounds == 1 && tiles * 100 >= 256 * 55;
    if (w4_ok && (p.force_cfg == 8 || wants_big || one_round_ok)) return launch_w4<TOut, EPI>(p, s);
    if (ok && (p.force_cfg == 7 || wants_big)) return launch_pp<TOut, EPI>(p, s);
    if (ok && p.force_cfg == 3) return launch_cfg<TIn, TOut, EPI, CfgBig>(p, s);
  }
  // small-M GEMMs (prompt rows: backward, text tower) are a latency chain on few CUs: 64x64 tiles give 4x
  // the workgroups and half the per-k-tile DMA issue per wave (da 9.2 -> 5.9 us, text c_proj 19.7 -> 11.3 us);
  // N <= 1024 at large M (out_proj, c_proj) prefers 64x128 (more workgroups than 128x128's 336)
#ifndef RPO_TINY_MAXN
#define RPO_TINY_MAXN (1 << 30)
#endif
  // ... except wide, short-K outputs from 1024 rows on (in-proj / c_fc of the image forward at a batch of 5 .. 9): 64x128
  // tiles with 8 waves, bit-identical (c_fc at M = 1768: 17.2 vs 22.2 us; the long-K c_proj stays on 64x64: 25 vs 31 us)
  if (p.force_cfg == 0 && p.M >= 1024 && p.M < 2048 && p.N >= 1536 && p.K <= 1024 && p.split_k == 1)
    return launch_cfg<TIn, TOut, EPI, CfgTall>(p, s);
  if (p.force_cfg == 5 || (p.force_cfg == 0 && p.M < 2048 && p.N <= RPO_TINY_MAXN)) return launch_cfg<TIn, TOut, EPI, CfgTiny>(p, s);
#ifndef RPO_TALL_N
#define RPO_TALL_N 1024
#endif
  if constexpr (sizeof(TIn) == 2) {
    if (p.force_cfg == 9) return launch_cfg<TIn, TOut, EPI, CfgQuad>(p, s);
  }
  // ... up to 8192 rows: from there on (the text tower over hundreds of classes: 24 000 prompt rows at ImageNet's 1000) the
  // 64x128 tiles' DMA bytes lose to 128x128 (tools/bench_gemm.py --only "t1k_*", profiles/r06_bench_gemm_t1k.txt: q-proj
  // 26.8 -> 23.6 us, c_proj 75.6 -> 68.4, d c_fc 71.2 -> 64.0 at 24 000 x 512)
  if (p.force_cfg == 6 || (p.force_cfg == 0 && p.N <= RPO_TALL_N && p.M < 8192)) return launch_cfg<TIn, TOut, EPI, CfgTall>(p, s);
  return launch_cfg<TIn, TOut, EPI, CfgMid>(p, s);
}

template <typename TIn, typename TOut>
int dispatch_epi(int epi, const GemmParams& p, hipStream_t s) {
  switch (epi) {
    case RPO_EPI_NONE: return launch<TIn, TOut, RPO_EPI_NONE>(p, s);
    case RPO_EPI_BIAS: return launch<TIn, TOut, RPO_EPI_BIAS>(p, s);
    case RPO_EPI_BIAS_QGELU: return launch<TIn, TOut, RPO_EPI_BIAS_QGELU>(p, s);
    case RPO_EPI_QGELU_BWD: return launch<TIn, TOut, RPO_EPI_QGELU_BWD>(p, s);
    case RPO_EPI_LN_BIAS:
      if constexpr (sizeof(TOut) == 2) return launch<TIn, TOut, RPO_EPI_LN_BIAS>(p, s);
      return RPO_E_DTYPE;
    case RPO_EPI_LN_BIAS_QGELU:
      if constexpr (sizeof(TOut) == 2) return launch<TIn, TOut, RPO_EPI_LN_BIAS_QGELU>(p, s);
      return RPO_E_DTYPE;
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

#ifdef RPO_TIMELINE
__device__ unsigned long long* g_timeline = nullptr;
extern "C" __attribute__((visibility("default"))) int rpo_debug_set_timeline(unsigned long long* buf) {   // (debug build only: not in the header)
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &buf, sizeof(buf));
}
#endif

// Which partial-statistics layout a BIAS_RESID producer would write for these shapes / dtypes / row units when the choice
// is left to the library: 96 when the one-round 224x96 kernel applies, else 64.  Looks at M, N, K, lda, ldw, the dtypes,
// split_k and the row-unit hint only; nothing is dereferenced.
extern "C" int rpo_gemm_stats_group(const rpo_gemm_args* a) {
  if (a == nullptr || a->M <= 0 || a->N <= 0 || a->K <= 0) return RPO_E_BADARG;
  const bool in16 = a->in_dtype == RPO_BF16 || a->in_dtype == RPO_F16;
  if (a->epilogue != RPO_EPI_BIAS_RESID || !in16 || a->out_dtype != RPO_F32 || a->split_k > 1) return LN_GROUP;
  GemmParams p{};
  p.M = a->M; p.N = a->N; p.K = a->K; p.lda = a->lda; p.ldw = a->ldw;
  p.seg_rows0 = a->seg_rows0; p.seg_rows1 = a->seg_rows1; p.seg1_row0 = a->seg1_row0;
  W4KPlan q;
  const bool fits32 = (int64_t)p.M * p.lda * 2 < (1ll << 31) && (int64_t)p.N * p.ldw * 2 < (1ll << 31);
  const int kgrp = fits32 ? w4k_plan(p, &q) : 0;
  return kgrp != 0 ? kgrp : LN_GROUP;
}

// Argument checks of rpo_gemm_nt and the kernel-side parameter block (shared with rpo_gemm_nt_pair)
static int gemm_prepare(const rpo_gemm_args* a, GemmParams& p) {
  if (a == nullptr || a->A == nullptr || a->W == nullptr || a->C == nullptr) return RPO_E_BADARG;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return RPO_E_BADARG;
  // "bf16" below = either 16-bit storage format; a 16-bit output must have the input's format
  const bool in_f16 = a->in_dtype == RPO_F16, out_f16 = a->out_dtype == RPO_F16;
  const bool in_bf16 = a->in_dtype == RPO_BF16 || in_f16, out_bf16 = a->out_dtype == RPO_BF16 || out_f16;
  if ((a->in_dtype != RPO_F32 && !in_bf16) || (a->out_dtype != RPO_F32 && !out_bf16)) return RPO_E_DTYPE;
  if (!in_bf16 && out_bf16) return RPO_E_DTYPE;
  if (out_bf16 && out_f16 != in_f16) return RPO_E_DTYPE;
  const int bk = in_bf16 ? 64 : 32;
  const int esz = in_bf16 ? 2 : 4, osz = out_bf16 ? 2 : 4;
  if (a->K % bk != 0 || a->N % 4 != 0) return RPO_E_SHAPE;
  if (!aligned16(a->A) || !aligned16(a->W) || (a->lda * esz) % 16 != 0 || (a->ldw * esz) % 16 != 0)
    return RPO_E_ALIGN;
  if ((reinterpret_cast<uintptr_t>(a->C) % (4 * osz)) != 0 || (a->ldc * osz) % (4 * osz) != 0) return RPO_E_ALIGN;
  const int epi = a->epilogue;
  const bool is_ln = epi == RPO_EPI_LN_BIAS || epi == RPO_EPI_LN_BIAS_QGELU;
  const bool needs_bias = epi == RPO_EPI_BIAS || epi == RPO_EPI_BIAS_QGELU || epi == RPO_EPI_BIAS_RESID || is_ln;
  if (is_ln && (!out_bf16 || a->ln_stats == nullptr || a->ln_colsum == nullptr || !aligned16(a->ln_colsum) ||
                reinterpret_cast<uintptr_t>(a->ln_stats) % 8 != 0 || !(a->ln_eps > 0.0f))) return RPO_E_BADARG;
  if (!is_ln && epi != RPO_EPI_BIAS_RESID && (a->ln_stats != nullptr || a->out2 != nullptr)) return RPO_E_BADARG;
  if (epi == RPO_EPI_BIAS_RESID && (a->ln_stats != nullptr || a->out2 != nullptr)) {
    if (!in_bf16 || a->N % 64 != 0) return RPO_E_SHAPE;
    if (a->out2 != nullptr && (reinterpret_cast<uintptr_t>(a->out2) % 8 != 0 || (a->ldout2 * 2) % 8 != 0)) return RPO_E_ALIGN;
    if (a->ln_stats != nullptr && reinterpret_cast<uintptr_t>(a->ln_stats) % 8 != 0) return RPO_E_ALIGN;
  }
  if (needs_bias && (a->bias == nullptr || !aligned16(a->bias))) return RPO_E_BADARG;
  const bool hilo_in = a->resid_hi != nullptr || a->resid_lo != nullptr;
  if (hilo_in || a->out_lo != nullptr || a->c_row0 != 0) {
    if (epi != RPO_EPI_BIAS_RESID || !in_bf16 || out_bf16) return RPO_E_BADARG;
    if (hilo_in && (a->resid_hi == nullptr || (a->ldr16 * 2) % 8 != 0 ||
                    reinterpret_cast<uintptr_t>(a->resid_hi) % 8 != 0 || reinterpret_cast<uintptr_t>(a->resid_lo) % 8 != 0))
      return RPO_E_ALIGN;
    if (a->out_lo != nullptr && (a->out2 == nullptr || reinterpret_cast<uintptr_t>(a->out_lo) % 8 != 0)) return RPO_E_BADARG;
    if (a->c_row0 < 0 || a->c_row0 > a->M) return RPO_E_BADARG;
  }
  if ((epi == RPO_EPI_BIAS_RESID || epi == RPO_EPI_PATCH) && !hilo_in &&
      (a->resid == nullptr || !aligned16(a->resid) || a->ldr % 4 != 0 || out_bf16)) return RPO_E_BADARG;
  if (hilo_in && out_bf16) return RPO_E_BADARG;
  if (epi == RPO_EPI_QGELU_BWD && a->aux == nullptr) return RPO_E_BADARG;
  if ((epi == RPO_EPI_QGELU_BWD || ((epi == RPO_EPI_BIAS_QGELU || epi == RPO_EPI_LN_BIAS_QGELU) && a->aux != nullptr)) &&
      (!aligned16(a->aux) || a->ldaux % 4 != 0)) return RPO_E_ALIGN;
  if (epi == RPO_EPI_PATCH && a->group <= 0) return RPO_E_BADARG;

  p.A = static_cast<const char*>(a->A); p.lda = a->lda;
  p.W = static_cast<const char*>(a->W); p.ldw = a->ldw;
  p.C = static_cast<char*>(a->C); p.ldc = a->ldc;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.bias = a->bias; p.resid = a->resid; p.ldr = a->ldr;
  p.aux = static_cast<float*>(a->aux); p.ldaux = a->ldaux; p.aux_row0 = a->aux_row0;
  if (a->aux_dtype != RPO_F32 && (a->aux == nullptr || a->aux_dtype != a->in_dtype || !in_bf16)) return RPO_E_DTYPE;
  p.aux_mode = a->aux_dtype != RPO_F32 ? 1 : 0;
  p.skip_row0 = a->skip_row0; p.skip_col0 = a->skip_col0; p.group = a->group;
  p.split_k = a->split_k <= 1 ? 1 : a->split_k;
  p.split_stride = a->split_stride;
  p.force_cfg = a->tile_config;
  p.out2 = static_cast<char*>(a->out2); p.ldout2 = a->ldout2;
  p.ln_stats = a->ln_stats; p.ln_colsum = a->ln_colsum; p.ln_eps = a->ln_eps;
  p.seg_rows0 = a->seg_rows0; p.seg_rows1 = a->seg_rows1; p.seg1_row0 = a->seg1_row0;
  p.ln_group = a->ln_group == 0 ? LN_GROUP : a->ln_group;
  p.resid_hi = static_cast<const char*>(a->resid_hi); p.resid_lo = static_cast<const char*>(a->resid_lo);
  p.ldr16 = a->ldr16; p.out_lo = static_cast<char*>(a->out_lo); p.c_row0 = a->c_row0;
  p.pf_ptr = static_cast<const char*>(a->prefetch);
  p.pf_bytes = a->prefetch == nullptr ? 0 : a->prefetch_bytes;
  if (p.pf_bytes < 0 || (p.pf_ptr != nullptr && reinterpret_cast<uintptr_t>(p.pf_ptr) % 4 != 0)) return RPO_E_BADARG;
  if (p.ln_group != 64 && p.ln_group != 96) return RPO_E_BADARG;
  if (is_ln && (p.K % p.ln_group != 0)) return RPO_E_SHAPE;
  if (p.seg_rows0 < 0 || p.seg_rows1 < 0 || p.seg1_row0 < 0 || p.seg1_row0 > p.M) return RPO_E_BADARG;
  if (p.split_k > 1 && (epi != RPO_EPI_NONE || out_bf16 || p.split_k > p.K / bk || p.split_stride % 4 != 0))
    return RPO_E_SHAPE;
  return 0;
}

extern "C" int rpo_gemm_hilo_ok(const rpo_gemm_args* a) {
  if (a == nullptr || a->M <= 0 || a->N <= 0 || a->K <= 0) return 0;
  const bool in16 = a->in_dtype == RPO_BF16 || a->in_dtype == RPO_F16;
  if (a->epilogue != RPO_EPI_BIAS_RESID || !in16 || a->out_dtype != RPO_F32 || a->split_k > 1 ||
      (a->tile_config != 0 && a->tile_config != 11)) return 0;
  GemmParams p{};
  p.M = a->M; p.N = a->N; p.K = a->K; p.lda = a->lda; p.ldw = a->ldw;
  p.seg_rows0 = a->seg_rows0; p.seg_rows1 = a->seg_rows1; p.seg1_row0 = a->seg1_row0;
  W4KPlan q;
  const bool fits32 = (int64_t)p.M * p.lda * 2 < (1ll << 31) && (int64_t)p.N * p.ldw * 2 < (1ll << 31);
  const int kgrp = fits32 ? w4k_plan(p, &q) : 0;
  const int want = a->ln_group == 0 ? LN_GROUP : a->ln_group;
  return kgrp != 0 && a->ldc % 4 == 0 && want == kgrp ? 1 : 0;
}

extern "C" int rpo_gemm_nt(const rpo_gemm_args* a, void* stream) {
  GemmParams p;
  if (int rc = gemm_prepare(a, p)) return rc;
  const bool in_f16 = a->in_dtype == RPO_F16;
  const bool in_bf16 = a->in_dtype == RPO_BF16 || in_f16, out_bf16 = a->out_dtype == RPO_BF16 || a->out_dtype == RPO_F16;
  const int epi = a->epilogue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (in_f16) {
    if (out_bf16) return dispatch_epi<f16_t, f16_t>(epi, p, s);
    return dispatch_f32out<f16_t>(epi, p, s);
  }
  if (in_bf16) {
    if (out_bf16) return dispatch_epi<bf16_t, bf16_t>(epi, p, s);
    return dispatch_f32out<bf16_t>(epi, p, s);
  }
  return dispatch_f32out<float>(epi, p, s);
}

#ifdef RPO_EXPERIMENTAL   // measured-slower experiments (include/rpo_amd_experimental.h): rpo_mlp_fused, rpo_gemm_nt_pair
// c_fc -> c_proj of an image block in ONE launch (gemm_mlp.inc; include/rpo_amd.h).  RPO_E_SHAPE where it does not apply
// (the caller then issues the two rpo_gemm_nt calls it stands for): both GEMMs must be the row-unit, one-round forms --
// 224x384 tiles with a QuickGELU epilogue feeding 224x96 split-k tiles with the residual epilogue, the same row units, 8
// column tiles each, every workgroup resident at once.
extern "C" int rpo_mlp_fused(const rpo_gemm_args* fc, const rpo_gemm_args* proj, void* counters, int safe, void* stream) {
  GemmParams pf, pp;
  if (counters == nullptr) return RPO_E_BADARG;
  if (int rc = gemm_prepare(fc, pf)) return rc;
  if (int rc = gemm_prepare(proj, pp)) return rc;
  const bool in16 = fc->in_dtype == RPO_BF16 || fc->in_dtype == RPO_F16;
  if (!in16 || fc->in_dtype != proj->in_dtype || fc->out_dtype != fc->in_dtype || proj->out_dtype != RPO_F32) return RPO_E_DTYPE;
  if ((fc->epilogue != RPO_EPI_LN_BIAS_QGELU && fc->epilogue != RPO_EPI_BIAS_QGELU) || proj->epilogue != RPO_EPI_BIAS_RESID)
    return RPO_E_SHAPE;
  if (pf.split_k != 1 || pp.split_k != 1 || pf.force_cfg != 0 || pp.force_cfg != 0 || pf.M != pp.M || pp.K != pf.N ||
      proj->A != fc->C || pp.lda != pf.ldc) return RPO_E_SHAPE;
  W4GPlan g;
  W4KPlan k;
  const bool fits32 = (int64_t)pf.M * pf.lda * 2 < (1ll << 31) && (int64_t)pf.N * pf.ldw * 2 < (1ll << 31) &&
                      (int64_t)pp.M * pp.lda * 2 < (1ll << 31) && (int64_t)pp.N * pp.ldw * 2 < (1ll << 31);
  if (!fits32 || w4g_plan(pf, &g) != 1 || !g.from_units || w4k_plan(pp, &k) != CfgW4K::BN || k.geo != CfgW4K::TM) return RPO_E_SHAPE;
  if (g.rows0 != k.rows0 || g.rows1 != k.rows1 || g.seg1_base != k.seg1_base || g.tiles_m != k.tiles_m || g.tiles_n != 8 ||
      k.tiles_n != 8 || g.tiles_m * 8 > rpo_cu_count()) return RPO_E_SHAPE;
  if (pf.K < 2 * CfgW4G::BK || pf.N % 8 != 0 || pf.ldc % 8 != 0 || !aligned16(pf.C) || !aligned16(pp.C) || pp.ldc % 4 != 0 ||
      (pp.ln_stats != nullptr && pp.ln_group != CfgW4K::BN)) return RPO_E_SHAPE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  unsigned int* cnt = static_cast<unsigned int*>(counters);
  // the XCD-local hand-off needs every row unit's 8 workgroups inside one XCD's contiguous share of the grid (nwg / 8
  // workgroups each): only when the unit count is a multiple of 8; otherwise units straddle two XCDs and the hand-off
  // must be the agent-scope one (caught by test_mlp_fused_equals_the_two_launches at 30 units)
  if (g.tiles_m % 8 != 0) safe = 1;
  const bool ln = fc->epilogue == RPO_EPI_LN_BIAS_QGELU;
  if (ln && pf.K > 16 * pf.ln_group) return RPO_E_SHAPE;
  if (fc->in_dtype == RPO_F16)
    return ln ? launch_mlp_fused<f16_t, RPO_EPI_LN_BIAS_QGELU>(pf, pp, g, cnt, safe, s)
              : launch_mlp_fused<f16_t, RPO_EPI_BIAS_QGELU>(pf, pp, g, cnt, safe, s);
  return ln ? launch_mlp_fused<bf16_t, RPO_EPI_LN_BIAS_QGELU>(pf, pp, g, cnt, safe, s)
            : launch_mlp_fused<bf16_t, RPO_EPI_BIAS_QGELU>(pf, pp, g, cnt, safe, s);
}

// Two GEMMs in one launch (see gemm_nt_pair_kernel): the same stage of the image tower's and of the text tower's
// prompt-row chain.  Both must be small-M problems of the same kind -- 16-bit inputs of one format, the same output
// dtype, epilogue and split-K factor, M < 2048, tile_config 0 -- i.e. what rpo_gemm_nt would run on its 64x64 tiles;
// anything else returns RPO_E_SHAPE / RPO_E_DTYPE (the caller then issues two rpo_gemm_nt calls).  Results are
// bit-identical to the two separate launches.
namespace {
template <typename TIn, typename TOut, int EPI>
int launch_pair(const GemmPair& g, int wgs1, int split_k, hipStream_t s) {
  static rpo_lds_mask_t lds_ok{0};
  using CF = CfgTiny;
  auto kern = gemm_nt_pair_kernel<TIn, TOut, EPI, CF>;
  constexpr int smem_bytes = CF::SMEM + CF::BM * 8;
  if (int rc = rpo_allow_lds(reinterpret_cast<const void*>(kern), smem_bytes, &lds_ok)) return rc;
  hipLaunchKernelGGL(kern, dim3(g.wgs0 + wgs1, split_k), dim3(CF::THREADS), smem_bytes, s, g);
  return rpo_launch_status();
}
template <typename TIn>
int dispatch_pair(int epi, bool out16, const GemmPair& g, int wgs1, int split_k, hipStream_t s) {
  if (out16) {
    if (epi == RPO_EPI_QGELU_BWD) return launch_pair<TIn, TIn, RPO_EPI_QGELU_BWD>(g, wgs1, split_k, s);
    if (epi == RPO_EPI_NONE) return launch_pair<TIn, TIn, RPO_EPI_NONE>(g, wgs1, split_k, s);
    return RPO_E_SHAPE;
  }
  if (epi == RPO_EPI_NONE) return launch_pair<TIn, float, RPO_EPI_NONE>(g, wgs1, split_k, s);
  return RPO_E_SHAPE;
}
}  // namespace

extern "C" int rpo_gemm_nt_pair(const rpo_gemm_args* a0, const rpo_gemm_args* a1, void* stream) {
  GemmPair g;
  if (int rc = gemm_prepare(a0, g.p[0])) return rc;
  if (int rc = gemm_prepare(a1, g.p[1])) return rc;
  const bool in16 = a0->in_dtype == RPO_BF16 || a0->in_dtype == RPO_F16;
  if (!in16 || a0->in_dtype != a1->in_dtype || a0->out_dtype != a1->out_dtype) return RPO_E_DTYPE;
  if (a0->epilogue != a1->epilogue || g.p[0].split_k != g.p[1].split_k || a0->tile_config != 0 || a1->tile_config != 0 ||
      a0->M >= 2048 || a1->M >= 2048 || g.p[0].skip_row0 >= 0 || g.p[1].skip_row0 >= 0 || g.p[0].resid_hi != nullptr ||
      g.p[1].resid_hi != nullptr || g.p[0].out_lo != nullptr || g.p[1].out_lo != nullptr) return RPO_E_SHAPE;
  using CF = CfgTiny;
  auto tiles = [](const GemmParams& p) { return ((p.M + CF::BM - 1) / CF::BM) * ((p.N + CF::BN - 1) / CF::BN); };
  g.tiles[0] = tiles(g.p[0]); g.tiles[1] = tiles(g.p[1]);
  // One round: the 64x64 kernel keeps RPO_PAIR_WG_PER_CU workgroups per CU resident (48 KiB of LDS each).  If the two
  // problems together need more, the one with the shorter k-loop gets half as many workgroups as tiles (each walks two),
  // then a third, ... until everything is resident at once.
#ifndef RPO_PAIR_WG_PER_CU
#define RPO_PAIR_WG_PER_CU 3
#endif
  int wgs[2] = {g.tiles[0], g.tiles[1]};
  {
    const int cus = rpo_cu_count();
    const int slots = RPO_PAIR_WG_PER_CU * cus / g.p[0].split_k;
    const int shorter = g.p[0].K <= g.p[1].K ? 0 : 1;
    for (int walk = 2; wgs[0] + wgs[1] > slots && walk <= 8; ++walk)
      wgs[shorter] = (g.tiles[shorter] + walk - 1) / walk;
  }
  g.wgs0 = wgs[0];
  const int tiles1 = wgs[1];
  const bool out16 = a0->out_dtype != RPO_F32;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (a0->in_dtype == RPO_F16) return dispatch_pair<f16_t>(a0->epilogue, out16, g, tiles1, g.p[0].split_k, s);
  return dispatch_pair<bf16_t>(a0->epilogue, out16, g, tiles1, g.p[0].split_k, s);
}
#endif  // RPO_EXPERIMENTAL
