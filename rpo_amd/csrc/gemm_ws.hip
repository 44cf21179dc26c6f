// rpo_gemm_ws: C[M,N] = A[M,K] . W[N,K]^T for the prompt-row chains -- a few hundred rows against a frozen weight
// matrix -- with the weight operand STREAMED global -> VGPR in MFMA fragment order (no LDS round trip for it).
//
// Replaces, for the B*K / n_cls*K prompt rows, nn.Linear's matmul (clip/model.py:173-177,186) and autograd's
// mm(dY, W) of trainers/rpo.py:308: the text tower's forward GEMMs, the last image block's prompt-row GEMMs and every
// dX GEMM of the two backward chains.  Until round 5 these ran on rpo_gemm_nt's generic 64x64 tiles
// (gemm_nt_kernel<Cfg<2,2,1,1,3,2>>): a third of the step's kernel time at 1.5-4.6 % MFMA utilisation, because at
// M = 768 / 456 rows a launch is a latency chain -- cold prologue, 8-12 barrier-paced k-tiles of 480 cycles, an epilogue
// staged through LDS -- and a 64x64 tile moves 16 KB through the LDS-DMA path per 4 MFMAs of each wave.
//
// What this kernel does instead:
//   * The frozen weight is packed ONCE at load (rpo_gemm_ws_pack) into fragment-major order: for every 32-row block nb
//     and 16-deep k-step ks the 64 lanes' 16-byte A-operand fragments of v_mfma_f32_32x32x16 (lane = 32 * khalf + row)
//     are stored back to back -- piece ((nb * K/16 + ks) * 64 + lane).  A wave's fragment load is then ONE fully
//     coalesced global_load_dwordx4 of 1 KiB, consecutive k-steps are consecutive KiBs, and the operand needs no LDS,
//     no swizzle, no ds_read and no barrier.
//   * The four waves of a workgroup SPLIT THE CONTRACTION (as gemm_w4k.inc does): wave w owns a contiguous range of the
//     workgroup's 64-deep k-chunks and the whole MT x NT block of 32x32 MFMA tiles, so nothing is shared between waves
//     until the end: the k-loop has NO s_barrier.  The four partial tiles are summed in a fixed order by the epilogue.
//   * The activation operand (row-major, written by the previous kernel of the chain) goes through a WAVE-PRIVATE pair of
//     LDS slots of 64-deep chunks: coalesced global_load_dwordx4 (8 lanes per 128-B row piece) into registers two chunks
//     ahead, ds_write_b128 one chunk ahead (XOR swizzle: 16-B chunk c of row r lives at c ^ ((r >> 1) & 7), conflict-free
//     for the write and for the fragment read), ds_read_b128 per k-step.  LDS is only the transposer that turns
//     row-major rows into per-lane fragments; no cross-wave synchronisation.  NOT LDS-DMA: hipcc's waitcnt pass files
//     global_load_lds and register loads under different event types of the same counter and then answers every
//     dependency with s_waitcnt vmcnt(0) (and puts a vmcnt(0) in front of every LDS read while a DMA is pending), which
//     serialises the weight stream behind each chunk; with plain loads every wait the compiler derives is exact.
//   * Tile = MT x NT MFMA tiles chosen so that M/32MT x N/32NT x split_k covers the CUs about once (ws_choose):
//     at M = 768, N = 3072, K = 768 (d c_proj) 96x96 tiles are exactly 256 workgroups, each streaming
//     (96 + 96) x 768 x 2 B = 295 KB; the 64x64 kernel's 576 workgroups streamed 442 KB per CU through LDS.
//   * XCD-aware order: consecutive workgroups of an XCD walk the m-tiles of one (n-tile, k-split), so a packed weight
//     range is fetched into ONE XCD's L2 and the activation panel (small) into all of them.
//
// Algorithmic work per launch: 2 M N K flops; bytes: N K 2 (weights, once) + M K 2 + M N x out size.
// Results: deterministic; the k sum is split in four (and by split_k), so the last bits differ from rpo_gemm_nt's.
#include "common.h"

#include <type_traits>

namespace {

struct WsParams {
  const char* A; int64_t lda;            // [M, K] act dtype, row-major; lda in elements
  const char* Wp;                        // packed weights (rpo_gemm_ws_pack)
  char* C; int64_t ldc;
  int M, N, K;
  int tiles_m, tiles_n, split_k; int64_t split_stride;
  const float* bias;
  const float* resid; int64_t ldr;
  char* aux; int64_t ldaux; int aux_row0; int aux_mode;   // as rpo_gemm_args (aux_mode 1: 16-bit derivative)
  char* out2; int64_t ldout2; float* stats_out;            // BIAS_RESID producer side of the LayerNorm fold (64-col groups)
  const float* stats_in; const float* ln_colsum; float ln_eps; int ln_group;
  const char* pf_ptr; int64_t pf_bytes;
};

template <int N> __device__ __forceinline__ void ws_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS hand-over between the waves of a workgroup WITHOUT __syncthreads()'s vmcnt(0): the epilogue's global stores (and
// the prefetch touches) stay in flight across it
__device__ __forceinline__ void ws_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int MT_, int NT_>
struct WsCfg {
  static constexpr int MT = MT_, NT = NT_, SLOTS = 2, THREADS = 256;
  static constexpr int BM = MT * 32, BN = NT * 32;
  static constexpr int SLOT_BYTES = BM * 128;                       // one 64-deep chunk of the activation tile
  static constexpr int WAVE_RING = SLOTS * SLOT_BYTES;
  static constexpr int RING_BYTES = 4 * WAVE_RING;
  static constexpr int FROW = BN * 4 + 16;                          // fp32 staging row of the reduction
  static constexpr int PART_BYTES = 32 * FROW;                      // one wave's 32-row partial block
  static constexpr int STAGE_BYTES = 2 * 4 * PART_BYTES;            // two buffers x four waves
  static constexpr int STATS_BYTES = BM * 8;                        // (mu, rstd) per tile row (LN fold, consumer side)
  static constexpr int SMEM = (RING_BYTES > STAGE_BYTES ? RING_BYTES : STAGE_BYTES) + STATS_BYTES;
  static constexpr int DA = MT * 4;                                 // activation loads per chunk (8 rows x 128 B each)
  static constexpr int DW = NT * 4;                                 // weight fragment loads per chunk
};

__device__ __forceinline__ float ws_row8_sum(float v) {            // sum over the 8 lanes (lane & ~7) .. (lane | 7)
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));   // row_half_mirror
  return v;
}

// (sum of group means, sum of group M2, sum of squared group means) -> (mu, rstd): the operations of gemm.hip's ln_finish
__device__ __forceinline__ float2 ws_ln_finish(float mu_sum, float m2_sum, float sq_sum, int G, int K, float eps) {
  const float mu = mu_sum / (float)G;
  const float between = fmaxf(fmaf(-(float)G * mu, mu, sq_sum), 0.f);
  const float m2 = fmaf((float)(K / G), between, m2_sum);
  return make_float2(mu, rsqrtf(m2 / (float)K + eps));
}

template <typename T, typename TOut, int EPI, typename CF>
__global__ __launch_bounds__(256, 1) void gemm_ws_kernel(const WsParams p) {
  extern __shared__ __attribute__((aligned(128))) char smem[];
  constexpr int MT = CF::MT, NT = CF::NT;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  RPO_STAMP(0); RPO_STAMP_RT(62);

  // ---- workgroup -> (m-tile, n-tile, k-split): XCD x = bid % 8 takes a contiguous run of the linear order
  //      ((split, n-tile), m-tile), m fastest
  int tile_m, tile_n, ksl;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7;
    const int wg = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    tile_m = wg % p.tiles_m;
    const int rest = wg / p.tiles_m;
    tile_n = rest % p.tiles_n;
    ksl = rest / p.tiles_n;
  }
  const int m0 = tile_m * CF::BM, n0 = tile_n * CF::BN;
  // k-chunks (64 deep) of this workgroup, then of this wave
  const int nch_all = p.K >> 6;
  const int wc0 = (int)(((int64_t)nch_all * ksl) / p.split_k), wc1 = (int)(((int64_t)nch_all * (ksl + 1)) / p.split_k);
  const int nwc = wc1 - wc0;
  const int c_lo = wc0 + (nwc * wave) / 4, c_hi = wc0 + (nwc * (wave + 1)) / 4;
  const int nc = c_hi - c_lo;

  // ---- addresses -----------------------------------------------------------------------------------------------
  // activation chunk: load j covers tile rows 8j .. 8j+7; lane -> row 8j + (lane >> 3), 16-B chunk lane & 7 of the
  // row's 128-B piece, parked at chunk (lane & 7) ^ ((row >> 1) & 7) of LDS row `row`
  uint32_t aoff[CF::DA];
  const int arow = lane >> 3;
  const uint32_t apark = (uint32_t)(arow * 128 + (((lane & 7) ^ ((arow >> 1) & 7)) << 4));   // (8j keeps (row >> 1) & 3; see below)
#pragma unroll
  for (int j = 0; j < CF::DA; ++j)
    aoff[j] = (uint32_t)min(m0 + 8 * j + arow, p.M - 1) * (uint32_t)(p.lda * 2) + (lane & 7) * 16;
  char* const ring = smem + wave * CF::WAVE_RING;
  // weight fragments: block nb = n0 / 32 + tn (clamped: a partial last n-tile recomputes the last block), k-step ks
  const int KS = p.K >> 4;
  const char* wb[NT];
#pragma unroll
  for (int tn = 0; tn < NT; ++tn) {
    const int nb = min(n0 / 32 + tn, p.N / 32 - 1);
    wb[tn] = p.Wp + ((int64_t)nb * KS * 64 + lane) * 16;
  }
  // fragment reads: tile row 32 tm + l31, k-step s of the chunk = 16-B chunks (2 s + half) ^ swizzle
  const int sw = (l31 >> 1) & 7;
  uint32_t frag_off[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) frag_off[s] = l31 * 128 + (((2 * s + half) ^ sw) << 4);

  f32x16_t acc[NT][MT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  // ---- prologue.  Everything the epilogue wants from memory is requested FIRST (it is then older than every load of the
  //      k-loop, whose counted waits retire it on the way), then the ring, then the first weight chunk.
  constexpr bool IS_LN = EPI == RPO_EPI_LN_BIAS || EPI == RPO_EPI_LN_BIAS_QGELU;
  constexpr bool IS_QG = EPI == RPO_EPI_BIAS_QGELU || EPI == RPO_EPI_LN_BIAS_QGELU;
  constexpr bool HAS_BIAS = EPI == RPO_EPI_BIAS || EPI == RPO_EPI_BIAS_QGELU || EPI == RPO_EPI_BIAS_RESID || IS_LN;
  float2* const row_stats = reinterpret_cast<float2*>(smem + CF::SMEM - CF::STATS_BYTES);
  // LayerNorm fold, consumer side: the producer's partial statistics of tile row `tid` (<= 16 groups), finished after the loop
  float2 lnp[IS_LN ? 16 : 1];
  const int G = IS_LN ? p.K / p.ln_group : 1;
  if constexpr (IS_LN) {
    if (tid < CF::BM) {
      const float2* ps = reinterpret_cast<const float2*>(p.stats_in) + (int64_t)min(m0 + tid, p.M - 1) * G;
#pragma unroll
      for (int g = 0; g < 16; ++g) lnp[g] = ps[min(g, G - 1)];
    }
  }
  // epilogue geometry: 32 rows at a time; wave w finishes rows 8w .. 8w+7 of a block, lane = (row8, 16-B column chunk c8
  // of every 32-column group j)
  const int row8 = lane >> 3, c8 = lane & 7;
  // second [M, N] operand of the epilogue (residual / saved QuickGELU operand)
  constexpr bool PRE_F32 = EPI == RPO_EPI_BIAS_RESID;
  constexpr bool PRE_AUX = EPI == RPO_EPI_QGELU_BWD;
  // raw bits: a float4 of fp32 values (residual / fp32 pre-activation: all four words) or four 16-bit derivatives (.x, .y)
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
  typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
  u32x4_t pre[(PRE_F32 || PRE_AUX) ? MT : 1][(PRE_F32 || PRE_AUX) ? NT : 1];
  if constexpr (PRE_F32 || PRE_AUX) {
    const bool wide = PRE_F32 || !p.aux_mode;
    const char* base = PRE_F32 ? reinterpret_cast<const char*>(p.resid) : p.aux;
    const int64_t ld = PRE_F32 ? p.ldr : p.ldaux;
#pragma unroll
    for (int tm = 0; tm < MT; ++tm) {
      const int m = min(m0 + tm * 32 + 8 * wave + row8, p.M - 1);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = min(n0 + 32 * j + 4 * c8, p.N - 4);
        if (wide) {
          pre[tm][j] = *reinterpret_cast<const u32x4_t*>(base + ((int64_t)m * ld + n) * 4);
        } else {
          const u32x2_t u = *reinterpret_cast<const u32x2_t*>(base + ((int64_t)m * ld + n) * 2);
          pre[tm][j] = u32x4_t{u.x, u.y, 0u, 0u};
        }
      }
    }
  }
  float4 b4[NT], s4[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = min(n0 + 32 * j + 4 * c8, p.N - 4);
    b4[j] = HAS_BIAS ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    s4[j] = IS_LN ? *reinterpret_cast<const float4*>(p.ln_colsum + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // ---- k-loop.  Per 64-deep chunk: 4 k-steps x (MT fragment reads, MT x NT MFMAs).  ONE set of weight registers: the
  //      fragments of k-step s are re-requested for the next chunk as soon as the MFMAs of k-step s have issued (the load
  //      lands a chunk later), so the weight stream runs one chunk ahead in the registers it is consumed from.  The
  //      activations run two chunks ahead: chunk i+1 is parked in the other LDS slot during chunk i (its loads were issued
  //      during chunk i-1) and chunk i+2 is requested into the registers the parking frees, piece by piece between the
  //      k-steps.  No barrier: everything is wave-private.  Chunk i lives in slot i & 1.
  //      The steady loop is branch-free and the last two chunks are peeled: with a run-time `if` around a load that sits
  //      between another load and its use, hipcc's waitcnt pass merges the two paths to s_waitcnt vmcnt(0).
  bf16x8_t areg[CF::DA];                           // (a clang vector type: HIP's uint4 class kept the array in scratch)
  bf16x8_t w[NT][4];
  constexpr int PPS = CF::DA / 4;                   // activation pieces parked / requested per k-step (= MT)
  // chunk 0 of both operands, chunk 0 of the activations parked, chunk 1 requested (NEXT); weights in k-step order, as
  // the loop re-requests them
#define WS_PROLOGUE(NEXT)                                                                                              \
  {                                                                                                                      \
    const char* src = p.A + (int64_t)c_lo * 128;                                                                         \
    _Pragma("unroll") for (int j = 0; j < CF::DA; ++j) areg[j] = *reinterpret_cast<const bf16x8_t*>(src + aoff[j]);      \
    _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                                        \
      _Pragma("unroll") for (int tn = 0; tn < NT; ++tn)                                                                  \
        w[tn][s] = *reinterpret_cast<const bf16x8_t*>(wb[tn] + (int64_t)(c_lo * 4 + s) * 1024);                          \
    _Pragma("unroll") for (int j = 0; j < CF::DA; ++j)                                                                   \
      *reinterpret_cast<bf16x8_t*>(ring + ((apark ^ (uint32_t)(((4 * j) & 7) << 4)) + j * 1024)) = areg[j];              \
    if (NEXT) {                                                                                                          \
      _Pragma("unroll") for (int j = 0; j < CF::DA; ++j)                                                                 \
        areg[j] = *reinterpret_cast<const bf16x8_t*>(src + 128 + aoff[j]);                                               \
    }                                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    RPO_STAMP(2);                                                                                                        \
  }
  // one chunk; REFRESH: re-request the weight fragments for chunk i+1; PARK: park chunk i+1 (in areg) into the other slot;
  // LOAD: request chunk i+2 into areg
#define WS_CHUNK(REFRESH, PARK, LOAD)                                                                                  \
  {                                                                                                                      \
    const char* sb = ring + (i & 1) * CF::SLOT_BYTES;                                                                    \
    char* so = ring + ((i + 1) & 1) * CF::SLOT_BYTES;                                                                    \
    const char* asrc = p.A + (int64_t)(c_lo + i + 2) * 128;                                                              \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                                      \
      bf16x8_t xf[MT];                                                                                                   \
      _Pragma("unroll") for (int tm = 0; tm < MT; ++tm)                                                                  \
        xf[tm] = *reinterpret_cast<const bf16x8_t*>(sb + tm * 4096 + frag_off[s]);                                       \
      _Pragma("unroll") for (int tn = 0; tn < NT; ++tn)                                                                  \
        _Pragma("unroll") for (int tm = 0; tm < MT; ++tm) acc[tn][tm] = mfma16<T>(w[tn][s], xf[tm], acc[tn][tm]);        \
      if (REFRESH) {                                                                                                     \
        _Pragma("unroll") for (int tn = 0; tn < NT; ++tn)                                                                \
          w[tn][s] = *reinterpret_cast<const bf16x8_t*>(wb[tn] + (int64_t)((c_lo + i + 1) * 4 + s) * 1024);              \
      }                                                                                                                  \
      _Pragma("unroll") for (int jj = 0; jj < PPS; ++jj) {                                                               \
        const int j = s * PPS + jj;                                                                                      \
        if (PARK) *reinterpret_cast<bf16x8_t*>(so + ((apark ^ (uint32_t)(((4 * j) & 7) << 4)) + j * 1024)) = areg[j];    \
        if (LOAD) areg[j] = *reinterpret_cast<const bf16x8_t*>(asrc + aoff[j]);                                          \
      }                                                                                                                  \
      __builtin_amdgcn_sched_barrier(0);   /* or hipcc sinks every request of the chunk behind its last MFMA */          \
    }                                                                                                                    \
    RPO_STAMP(3 + (i < 40 ? i : 40));                                                                                    \
  }
  RPO_STAMP(1);
  // Three separate code paths by chunk count, so that every join the waitcnt pass sees has the same requests pending on
  // all its edges (a shared prologue with `if (nc > 1)` around the chunk-1 request made the loop's waits those of nc = 1)
  if (nc >= 3) {
    WS_PROLOGUE(true)
    int i = 0;
    for (; i + 2 < nc; ++i) WS_CHUNK(true, true, true)
    WS_CHUNK(true, true, false)
    ++i;
    WS_CHUNK(false, false, false)
  } else if (nc == 2) {
    WS_PROLOGUE(true)
    int i = 0;
    WS_CHUNK(true, true, false)
    ++i;
    WS_CHUNK(false, false, false)
  } else if (nc == 1) {
    WS_PROLOGUE(false)
    const int i = 0;
    WS_CHUNK(false, false, false)
  }
#undef WS_PROLOGUE
#undef WS_CHUNK
  // Prefetch hint (rpo_gemm_args.prefetch): this workgroup's share of the lines the NEXT launch of the chain reads first,
  // one dword per 128-B line.  Issued HERE, behind the loop's last counted wait, and never waited for: the epilogue below
  // has no vmcnt wait (raw barriers, every operand already in registers), so the touches cost the chain nothing and have
  // the epilogue + the kernel boundary (~2-3 us) to pull the lines into the memory-side cache.  Branch-free, ONE asm
  // statement defining the register, which stays reserved to the end of the kernel (a copy on a control-flow edge would
  // free it while the loads are in flight).  Without a hint the four loads re-touch the kernel's own first weight line.
  // (Every register the prologue requested is NAMED here first: the compiler then retires those loads now -- they landed
  //  long ago -- instead of in the epilogue, where its s_waitcnt vmcnt(0) would also wait for the touches below.)
  if constexpr (PRE_F32 || PRE_AUX) {
#pragma unroll
    for (int tm = 0; tm < MT; ++tm)
#pragma unroll
      for (int j = 0; j < NT; ++j) asm volatile("" :: "v"(pre[tm][j]));
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    if constexpr (HAS_BIAS) asm volatile("" :: "v"(b4[j].x), "v"(b4[j].y), "v"(b4[j].z), "v"(b4[j].w));
    if constexpr (IS_LN) asm volatile("" :: "v"(s4[j].x), "v"(s4[j].y), "v"(s4[j].z), "v"(s4[j].w));
  }
  if constexpr (IS_LN) {
#pragma unroll
    for (int g = 0; g < 16; ++g) asm volatile("" :: "v"(lnp[g].x), "v"(lnp[g].y));
  }
  uint32_t pf_touch;
  {
    const bool has = p.pf_ptr != nullptr;
    const char* base = has ? p.pf_ptr : p.Wp;
    const int64_t lines = has ? max(p.pf_bytes >> 7, (int64_t)1) : 1;
    const int64_t stride = (int64_t)gridDim.x * CF::THREADS;
    const int64_t i0 = (int64_t)blockIdx.x * CF::THREADS + tid;
    const char* t0 = base + (min(i0, lines - 1) << 7);
    const char* t1 = base + (min(i0 + stride, lines - 1) << 7);
    const char* t2 = base + (min(i0 + 2 * stride, lines - 1) << 7);
    const char* t3 = base + (min(i0 + 3 * stride, lines - 1) << 7);
    asm volatile("global_load_dword %0, %1, off\n\tglobal_load_dword %0, %2, off\n\t"
                 "global_load_dword %0, %3, off\n\tglobal_load_dword %0, %4, off"
                 : "=&v"(pf_touch) : "v"(t0), "v"(t1), "v"(t2), "v"(t3) : "memory");
  }

  // ---- epilogue: sum the four partial tiles (fixed order), then the fused element-wise tail ---------------------------
  if constexpr (IS_LN) {
    if (tid < CF::BM) {
      float mu = 0.f, m2 = 0.f, sq = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const float mg = g < G ? lnp[g].x : 0.f;
        mu += mg; m2 += g < G ? lnp[g].y : 0.f; sq = fmaf(mg, mg, sq);
      }
      row_stats[tid] = ws_ln_finish(mu, m2, sq, G, p.K, p.ln_eps);
    }
  }
  RPO_STAMP(58);
  ws_lds_barrier();                                 // every wave is done with its ring: the staging image may overwrite it
  RPO_STAMP(59);
  TOut* const cbase = reinterpret_cast<TOut*>(p.C) + (int64_t)ksl * p.split_stride;
  auto park = [&](const int tm) {
    char* mine = smem + (tm & 1) * (4 * CF::PART_BYTES) + wave * CF::PART_BYTES;
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(mine + l31 * CF::FROW + (tn * 32 + 8 * g + 4 * half) * 4) =
            make_float4(acc[tn][tm][4 * g], acc[tn][tm][4 * g + 1], acc[tn][tm][4 * g + 2], acc[tn][tm][4 * g + 3]);
  };
  park(0);
#pragma unroll
  for (int tm = 0; tm < MT; ++tm) {
    const char* buf = smem + (tm & 1) * (4 * CF::PART_BYTES);
    const int row = tm * 32 + 8 * wave + row8;
    const int m = m0 + row;
    ws_lds_barrier();
    if (tm + 1 < MT) park(tm + 1);
    float4 v[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const char* src = buf + (8 * wave + row8) * CF::FROW + (32 * j + 4 * c8) * 4;
      const float4 p0 = *reinterpret_cast<const float4*>(src), p1 = *reinterpret_cast<const float4*>(src + CF::PART_BYTES);
      const float4 p2 = *reinterpret_cast<const float4*>(src + 2 * CF::PART_BYTES);
      const float4 p3 = *reinterpret_cast<const float4*>(src + 3 * CF::PART_BYTES);
      v[j] = make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y),
                         (p0.z + p1.z) + (p2.z + p3.z), (p0.w + p1.w) + (p2.w + p3.w));
    }
    float2 st = make_float2(0.f, 1.f);
    if constexpr (IS_LN) st = row_stats[row];
    float sum64 = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + 32 * j + 4 * c8;
      const bool ok = m < p.M && n < p.N;
      float4 x = v[j];
      if constexpr (IS_LN) {
        x.x = fmaf(st.y, x.x - st.x * s4[j].x, b4[j].x); x.y = fmaf(st.y, x.y - st.x * s4[j].y, b4[j].y);
        x.z = fmaf(st.y, x.z - st.x * s4[j].z, b4[j].z); x.w = fmaf(st.y, x.w - st.x * s4[j].w, b4[j].w);
      } else if constexpr (HAS_BIAS) {
        x.x += b4[j].x; x.y += b4[j].y; x.z += b4[j].z; x.w += b4[j].w;
      }
      if constexpr (IS_QG) {
        if (p.aux != nullptr && ok && m >= p.aux_row0) {     // what the QuickGELU backward wants of the back-propagated rows
          const int64_t off = (int64_t)(m - p.aux_row0) * p.ldaux + n;
          if (p.aux_mode)
            *reinterpret_cast<uint2*>(p.aux + off * 2) =
                make_uint2(pack2<T>(quick_gelu_grad(x.x), quick_gelu_grad(x.y)), pack2<T>(quick_gelu_grad(x.z), quick_gelu_grad(x.w)));
          else
            *reinterpret_cast<float4*>(p.aux + off * 4) = x;
        }
        x = quick_gelu4(x);
      }
      if constexpr (EPI == RPO_EPI_BIAS_RESID) {
        x.x += __uint_as_float(pre[tm][j].x); x.y += __uint_as_float(pre[tm][j].y);
        x.z += __uint_as_float(pre[tm][j].z); x.w += __uint_as_float(pre[tm][j].w);
      }
      if constexpr (EPI == RPO_EPI_QGELU_BWD) {
        float4 gq;
        if (p.aux_mode) {
          const uint32_t u0 = pre[tm][j].x, u1 = pre[tm][j].y;
          gq = make_float4(unpack1<T>((uint16_t)(u0 & 0xffffu)), unpack1<T>((uint16_t)(u0 >> 16)),
                           unpack1<T>((uint16_t)(u1 & 0xffffu)), unpack1<T>((uint16_t)(u1 >> 16)));
        } else {
          gq = make_float4(quick_gelu_grad(__uint_as_float(pre[tm][j].x)), quick_gelu_grad(__uint_as_float(pre[tm][j].y)),
                           quick_gelu_grad(__uint_as_float(pre[tm][j].z)), quick_gelu_grad(__uint_as_float(pre[tm][j].w)));
        }
        x.x *= gq.x; x.y *= gq.y; x.z *= gq.z; x.w *= gq.w;
      }
      if (ok) ActIO<TOut>::st4(cbase + (int64_t)m * p.ldc + n, x.x, x.y, x.z, x.w);
      if constexpr (EPI == RPO_EPI_BIAS_RESID) {
        // LayerNorm fold, producer side: the 16-bit copy the consuming GEMM reads, and (mean, sum of squared deviations)
        // of every 64-column group of the row -- two 32-column groups j of this thread x the 8 lanes of the row
        if (p.out2 != nullptr && ok) ActIO<T>::st4(reinterpret_cast<T*>(p.out2) + (int64_t)m * p.ldout2 + n, x.x, x.y, x.z, x.w);
        if (p.stats_out != nullptr) {
          sum64 += (x.x + x.y) + (x.z + x.w);
          v[j] = x;
          if (j & 1) {
            const float mean = ws_row8_sum(sum64) * (1.0f / 64.0f);
            float q = 0.f;
#pragma unroll
            for (int jj = j - 1; jj <= j; ++jj) {
              const float dx = v[jj].x - mean, dy = v[jj].y - mean, dz = v[jj].z - mean, dw = v[jj].w - mean;
              q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
            q = ws_row8_sum(q);
            if (ok && c8 == 0) reinterpret_cast<float2*>(p.stats_out)[(int64_t)m * (p.N / 64) + ((n0 + 32 * j) >> 6)] = make_float2(mean, q);
            sum64 = 0.f;
          }
        }
      }
    }
  }
#ifdef RPO_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  RPO_STAMP(61); RPO_STAMP_RT(63);
  asm volatile("" :: "v"(pf_touch));                // (keeps the touch register reserved to the end)
}

// fragment-major packing: piece ((nb * K/16 + ks) * 64 + lane) <- W[nb * 32 + (lane & 31), ks * 16 + (lane >> 5) * 8 .. +8]
__global__ void ws_pack_kernel(const char* W, int64_t ldw, char* Wp, int N, int K) {
  const int64_t pieces = (int64_t)N * K / 8;
  const int KS = K >> 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pieces; i += (int64_t)gridDim.x * blockDim.x) {
    const int l = (int)(i & 63);
    const int64_t blk = i >> 6;
    const int ks = (int)(blk % KS);
    const int nb = (int)(blk / KS);
    const int row = nb * 32 + (l & 31), k0 = ks * 16 + (l >> 5) * 8;
    *reinterpret_cast<uint4*>(Wp + i * 16) = *reinterpret_cast<const uint4*>(W + ((int64_t)row * ldw + k0) * 2);
  }
}

// ---- host side -----------------------------------------------------------------------------------------------------
struct WsPlan { int mt, nt; };

// Tile choice: the geometry whose busiest CU streams the fewest bytes (a CU's share = its workgroups x the operand bytes
// of one), ties to the bigger tile.  tile_config 100 * MT + 10 * NT forces one (benchmarks / tests).
static bool ws_geometry_ok(int mt, int nt) { return (mt == 1 && nt == 1) || (mt == 1 && nt == 2) || (mt == 2 && nt == 2) || (mt == 3 && nt == 3); }
static WsPlan ws_choose(int M, int N, int K, int split_k, int epilogue, bool wants_stats, int forced) {
  if (forced > 0) {
    const WsPlan f{forced / 100, (forced / 10) % 10};
    return ws_geometry_ok(f.mt, f.nt) ? f : WsPlan{0, 0};
  }
  static const WsPlan cand[] = {{3, 3}, {2, 2}, {1, 2}, {1, 1}};
  const int cus = rpo_cu_count();
  WsPlan best{0, 0};
  double best_cost = 0;
  for (const WsPlan& c : cand) {
    if (wants_stats && c.nt != 2) continue;                        // 64-column statistics groups = two 32-column groups of a row
    const int64_t wgs = (int64_t)((M + 32 * c.mt - 1) / (32 * c.mt)) * ((N + 32 * c.nt - 1) / (32 * c.nt)) * split_k;
    const int64_t per_cu = (wgs + cus - 1) / cus;
    const double bytes = (double)(c.mt + c.nt) * 32 * (K / split_k) * 2;
    // fixed cost per workgroup a CU runs (prologue round trip + reduction) in byte-equivalents at ~100 B/ns per CU
    const double cost = per_cu * (bytes + 60e3 + 12e3 * c.mt * c.nt);
    if (best.mt == 0 || cost < best_cost) { best = c; best_cost = cost; }
  }
  return best;
}

template <typename T, typename TOut, int EPI, typename CF>
int ws_launch_t(WsParams& p, hipStream_t s) {
  static rpo_lds_mask_t lds_ok{0};
  auto kern = gemm_ws_kernel<T, TOut, EPI, CF>;
  if (int rc = rpo_allow_lds(reinterpret_cast<const void*>(kern), CF::SMEM, &lds_ok)) return rc;
  p.tiles_m = (p.M + CF::BM - 1) / CF::BM;
  p.tiles_n = (p.N + CF::BN - 1) / CF::BN;
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n * p.split_k), dim3(CF::THREADS), CF::SMEM, s, p);
  return rpo_launch_status();
}

template <typename T, typename TOut, int EPI>
int ws_launch(WsParams& p, const WsPlan& q, hipStream_t s) {
  if (q.mt == 3 && q.nt == 3) return ws_launch_t<T, TOut, EPI, WsCfg<3, 3>>(p, s);
  if (q.mt == 2 && q.nt == 2) return ws_launch_t<T, TOut, EPI, WsCfg<2, 2>>(p, s);
  if (q.mt == 1 && q.nt == 2) return ws_launch_t<T, TOut, EPI, WsCfg<1, 2>>(p, s);
  if (q.mt == 1 && q.nt == 1) return ws_launch_t<T, TOut, EPI, WsCfg<1, 1>>(p, s);
  return RPO_E_SHAPE;
}

template <typename T>
int ws_dispatch(int epi, bool out16, WsParams& p, const WsPlan& q, hipStream_t s) {
  switch (epi) {
    case RPO_EPI_NONE:
      return out16 ? ws_launch<T, T, RPO_EPI_NONE>(p, q, s) : ws_launch<T, float, RPO_EPI_NONE>(p, q, s);
    case RPO_EPI_QGELU_BWD:
      return out16 ? ws_launch<T, T, RPO_EPI_QGELU_BWD>(p, q, s) : RPO_E_DTYPE;
    case RPO_EPI_BIAS:
      return out16 ? ws_launch<T, T, RPO_EPI_BIAS>(p, q, s) : RPO_E_DTYPE;
    case RPO_EPI_BIAS_QGELU:
      return out16 ? ws_launch<T, T, RPO_EPI_BIAS_QGELU>(p, q, s) : RPO_E_DTYPE;
    case RPO_EPI_LN_BIAS:
      return out16 ? ws_launch<T, T, RPO_EPI_LN_BIAS>(p, q, s) : RPO_E_DTYPE;
    case RPO_EPI_LN_BIAS_QGELU:
      return out16 ? ws_launch<T, T, RPO_EPI_LN_BIAS_QGELU>(p, q, s) : RPO_E_DTYPE;
    case RPO_EPI_BIAS_RESID:
      return out16 ? RPO_E_DTYPE : ws_launch<T, float, RPO_EPI_BIAS_RESID>(p, q, s);
    default: return RPO_E_SHAPE;
  }
}

// argument checks shared by rpo_gemm_ws and rpo_gemm_ws_ok; fills p and the plan
static int ws_prepare(const rpo_gemm_args* a, WsParams& p, WsPlan& q) {
  if (a == nullptr || a->M <= 0 || a->N <= 0 || a->K <= 0) return RPO_E_BADARG;
  const bool in16 = a->in_dtype == RPO_BF16 || a->in_dtype == RPO_F16;
  const bool out16 = a->out_dtype == RPO_BF16 || a->out_dtype == RPO_F16;
  if (!in16 || (out16 && a->out_dtype != a->in_dtype) || (!out16 && a->out_dtype != RPO_F32)) return RPO_E_DTYPE;
  if (a->K % 64 != 0 || a->N % 32 != 0) return RPO_E_SHAPE;
  if ((int64_t)a->M * a->lda * 2 >= (1ll << 32)) return RPO_E_SHAPE;             // 32-bit row offsets of the DMA
  const int epi = a->epilogue;
  const bool is_ln = epi == RPO_EPI_LN_BIAS || epi == RPO_EPI_LN_BIAS_QGELU;
  const bool needs_bias = epi == RPO_EPI_BIAS || epi == RPO_EPI_BIAS_QGELU || epi == RPO_EPI_BIAS_RESID || is_ln;
  if (epi == RPO_EPI_PATCH || epi < 0 || epi > RPO_EPI_LN_BIAS_QGELU) return RPO_E_SHAPE;
  if (a->skip_row0 >= 0 || a->resid_hi != nullptr || a->resid_lo != nullptr || a->out_lo != nullptr || a->c_row0 != 0 ||
      a->seg_rows0 != 0 || a->seg_rows1 != 0) return RPO_E_SHAPE;
  const int split = a->split_k <= 1 ? 1 : a->split_k;
  if (split > 1 && (epi != RPO_EPI_NONE || out16 || split > a->K / 64 || a->split_stride % 4 != 0)) return RPO_E_SHAPE;
  if (needs_bias && (a->bias == nullptr || !aligned16(a->bias))) return RPO_E_BADARG;
  if (is_ln) {
    const int g = a->ln_group == 0 ? 64 : a->ln_group;
    if (a->ln_stats == nullptr || a->ln_colsum == nullptr || !aligned16(a->ln_colsum) || !(a->ln_eps > 0.0f) ||
        reinterpret_cast<uintptr_t>(a->ln_stats) % 8 != 0) return RPO_E_BADARG;
    if ((g != 64 && g != 96) || a->K % g != 0 || a->K / g > 16) return RPO_E_SHAPE;
  }
  if (!is_ln && epi != RPO_EPI_BIAS_RESID && (a->ln_stats != nullptr || a->out2 != nullptr)) return RPO_E_BADARG;
  const bool wants_stats = epi == RPO_EPI_BIAS_RESID && a->ln_stats != nullptr;
  if (epi == RPO_EPI_BIAS_RESID) {
    if (a->resid == nullptr || !aligned16(a->resid) || a->ldr % 4 != 0) return RPO_E_BADARG;
    if (a->out2 != nullptr && (reinterpret_cast<uintptr_t>(a->out2) % 8 != 0 || (a->ldout2 * 2) % 8 != 0)) return RPO_E_ALIGN;
    if (wants_stats && (a->N % 64 != 0 || (a->ln_group != 0 && a->ln_group != 64) ||
                        reinterpret_cast<uintptr_t>(a->ln_stats) % 8 != 0)) return RPO_E_SHAPE;
  }
  if (epi == RPO_EPI_QGELU_BWD && a->aux == nullptr) return RPO_E_BADARG;
  if (a->aux != nullptr && (epi == RPO_EPI_QGELU_BWD || epi == RPO_EPI_BIAS_QGELU || epi == RPO_EPI_LN_BIAS_QGELU)) {
    if (!aligned16(a->aux) || a->ldaux % 4 != 0) return RPO_E_ALIGN;
    if (a->aux_dtype != RPO_F32 && a->aux_dtype != a->in_dtype) return RPO_E_DTYPE;
  }
  const int osz = out16 ? 2 : 4;
  if (!aligned16(a->A) || (a->lda * 2) % 16 != 0 || !aligned16(a->W)) return RPO_E_ALIGN;
  if ((reinterpret_cast<uintptr_t>(a->C) % (4 * osz)) != 0 || (a->ldc * osz) % (4 * osz) != 0) return RPO_E_ALIGN;
  q = ws_choose(a->M, a->N, a->K, split, epi, wants_stats, a->tile_config);
  if (q.mt == 0 || (wants_stats && q.nt != 2)) return RPO_E_SHAPE;
  p = WsParams{};
  p.A = static_cast<const char*>(a->A); p.lda = a->lda;
  p.Wp = static_cast<const char*>(a->W);
  p.C = static_cast<char*>(a->C); p.ldc = a->ldc;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.split_k = split; p.split_stride = a->split_stride;
  p.bias = a->bias; p.resid = a->resid; p.ldr = a->ldr;
  p.aux = static_cast<char*>(a->aux); p.ldaux = a->ldaux; p.aux_row0 = a->aux_row0;
  p.aux_mode = (a->aux != nullptr && a->aux_dtype != RPO_F32) ? 1 : 0;
  p.out2 = static_cast<char*>(a->out2); p.ldout2 = a->ldout2;
  p.stats_out = wants_stats ? a->ln_stats : nullptr;
  p.stats_in = is_ln ? a->ln_stats : nullptr;
  p.ln_colsum = a->ln_colsum; p.ln_eps = a->ln_eps; p.ln_group = a->ln_group == 0 ? 64 : a->ln_group;
  p.pf_ptr = static_cast<const char*>(a->prefetch);
  p.pf_bytes = a->prefetch == nullptr ? 0 : a->prefetch_bytes;
  if (p.pf_bytes < 0 || (p.pf_ptr != nullptr && reinterpret_cast<uintptr_t>(p.pf_ptr) % 4 != 0)) return RPO_E_BADARG;
  return 0;
}

}  // namespace

extern "C" int rpo_gemm_ws_pack(const void* W, int64_t ldw, void* Wp, int N, int K, int dtype, void* stream) {
  if (W == nullptr || Wp == nullptr || N <= 0 || K <= 0) return RPO_E_BADARG;
  if (dtype != RPO_BF16 && dtype != RPO_F16) return RPO_E_DTYPE;
  if (N % 32 != 0 || K % 64 != 0) return RPO_E_SHAPE;
  if (!aligned16(W) || !aligned16(Wp) || (ldw * 2) % 16 != 0) return RPO_E_ALIGN;
  const int64_t pieces = (int64_t)N * K / 8;
  const int blocks = (int)((pieces + 255) / 256 < 4096 ? (pieces + 255) / 256 : 4096);
  hipLaunchKernelGGL(ws_pack_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const char*>(W), ldw, static_cast<char*>(Wp), N, K);
  return rpo_launch_status();
}

extern "C" int rpo_gemm_ws_ok(const rpo_gemm_args* a) {
  WsParams p;
  WsPlan q;
  if (a == nullptr) return 0;
  rpo_gemm_args b = *a;
  // a query carries no pointers: stand-ins that pass the null / alignment checks
  static const uintptr_t dummy = 4096;
  auto fill = [](const void*& ptr) { if (ptr == nullptr) ptr = reinterpret_cast<const void*>(dummy); };
  fill(b.A); fill(b.W);
  if (b.C == nullptr) b.C = reinterpret_cast<void*>(dummy);
  const int epi = b.epilogue;
  const bool is_ln = epi == RPO_EPI_LN_BIAS || epi == RPO_EPI_LN_BIAS_QGELU;
  if (epi == RPO_EPI_BIAS || epi == RPO_EPI_BIAS_QGELU || epi == RPO_EPI_BIAS_RESID || is_ln)
    if (b.bias == nullptr) b.bias = reinterpret_cast<const float*>(dummy);
  if (epi == RPO_EPI_BIAS_RESID && b.resid == nullptr) { b.resid = reinterpret_cast<const float*>(dummy); if (b.ldr == 0) b.ldr = b.N; }
  if (epi == RPO_EPI_QGELU_BWD && b.aux == nullptr) { b.aux = reinterpret_cast<void*>(dummy); if (b.ldaux == 0) b.ldaux = b.N; }
  if (is_ln) {
    if (b.ln_stats == nullptr) b.ln_stats = reinterpret_cast<float*>(dummy);
    if (b.ln_colsum == nullptr) b.ln_colsum = reinterpret_cast<const float*>(dummy);
    if (!(b.ln_eps > 0.0f)) b.ln_eps = 1e-5f;
  }
  return ws_prepare(&b, p, q) == 0 ? 1 : 0;
}

extern "C" int rpo_gemm_ws(const rpo_gemm_args* a, void* stream) {
  WsParams p;
  WsPlan q;
  if (a == nullptr || a->A == nullptr || a->W == nullptr || a->C == nullptr) return RPO_E_BADARG;
  if (int rc = ws_prepare(a, p, q)) return rc;
  const bool out16 = a->out_dtype != RPO_F32;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (a->in_dtype == RPO_F16) return ws_dispatch<f16_t>(a->epilogue, out16, p, q, s);
  return ws_dispatch<bf16_t>(a->epilogue, out16, p, q, s);
}
