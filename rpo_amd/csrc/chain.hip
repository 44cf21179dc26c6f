// rpo_chain_bwd: the prompt-row backward chain of one tower as ONE persistent launch.
//
// Replaces, for every transformer block, the six launches that the backward through
// clip/model.py:181-191 costs for the back-propagated rows (autograd of trainers/rpo.py:308):
//     A  d c_proj GEMM x QuickGELU'   B  d c_fc GEMM (4 k-slices)   C  ln_2 backward + residual
//     D  d out-proj + attention backward (dq)   E  d q-proj GEMM (4 k-slices)   F  ln_1 backward + residual
// 72 stages for ViT-B/16 whose launch-per-stage form spent 5-19 us per kernel on 2-8 us of work
// (profiles/r03_kernel_trace_stats.txt: 774 us for the image tower's chain at B = 32).
//
// Structure.  A back-propagated row only ever meets rows of its own unit (image / class): its keys and values belong to
// frozen tokens.  So the units are dealt to 8 GROUPS, group g = workgroups {b : b % 8 == g} -- which the dispatcher
// places on XCD g -- and a group carries its <= 96 rows through all stages on its own: hand-offs between stages stay
// inside one XCD's L2, and a stage boundary is one arrive + poll on the group's counter (tools/ubench_group_handoff.hip,
// profiles/r04_group_handoff_ubench.txt: 2.1-2.5 us per all-to-all hand-off of a 32-workgroup group against 4.6-5.0 us
// for the placement-independent agent-scope recipe and ~1.9 us for a kernel boundary around the same work) -- the
// launch ramp, the cold instruction / operand fetch and the drain of a kernel per stage are gone.
//   * GEMM stages: one 96 x (32 NWC) tile per workgroup, k-tiles of 64 through a ring of LDS slots filled by LDS-DMA
//     (all four waves issue it, NWC of them multiply: 96 x 32 each, W as the MFMA A operand exactly as gemm.hip);
//     N = 4d: 32 column tiles; N = d: 8 column tiles x 4 k-slices whose fp32 slabs the LayerNorm stage sums in order.
//   * LayerNorm stages: one wave per row (the arithmetic of norm.hip's ln_bwd_row, same order).
//   * attention stage: attn_image.hip's attn_bwd_body (d out-proj folded in) per (unit, head) item.
// Placement: the fast hand-off (plain stores, drain, counter, ONE L1 invalidate) is valid only among workgroups of one
// XCD.  Every workgroup publishes its HW_REG_XCC_ID before the first (agent-scope, placement-independent) hand-off and
// the group switches to the fast form only if all its members reported the same XCD; otherwise (or RPO_CHAIN_SAFE=1)
// every hand-off carries the agent-scope release (buffer_wbl2) as well.  Results never depend on placement.
// Every spin is bounded; a give-up sets state[0] and the chain runs to its end without waiting (results undefined,
// nothing hangs).
#define RPO_DEVICE_ONLY
#include "attn_image.hip"

#include <stdlib.h>

namespace {

constexpr int CH_GROUPS = 8;
constexpr int CH_MAX_LAYERS = 24;
constexpr unsigned CH_SPIN_LIMIT = 1u << 21;

typedef __attribute__((address_space(1))) unsigned ch_gu32;

struct ChainLayerDev {
  const char* w_proj_t; const char* w_fc_t; const char* w_out_t; const char* w_q_t; const char* aux;
  const float* x_ln2; const float* x_ln1; const float* ln2_w; const float* ln1_w;
  const char* q_rows; const char* k; const char* v;
};
struct ChainParams {
  ChainLayerDev layer[CH_MAX_LAYERS];
  int layers, units, Kp, d, H, keys, key_stride;
  const int32_t* key_len;
  int64_t ldx, ldq, ldkv;
  float* dxa; float* dxb; char* dxc; char* du; char* dq; float* dy; int64_t dy_stride;
  float scale, eps;
  int gw;                       // workgroups per group
  int force_safe;
  unsigned* state;              // [0] give-ups, [1] groups that ran the safe protocol, [16 + 16 g] counter of group g,
                                // [256 + 64 g + slot] XCC id (+ 1) of the group's workgroups
  unsigned long long* timeline;
};
constexpr size_t CH_STATE_WORDS = 256 + 64 * CH_GROUPS;

// ---- hand-off ----------------------------------------------------------------------------------------------------
// A stage boundary is split in two so that everything the next stage reads that does NOT come from the previous stage
// (frozen weights by LDS-DMA, saved activations, keys / values) is requested between the two halves and travels while the
// group's last workgroups are still arriving:
//   hop_arrive: every wave drains its stores, ONE lane adds 1 to the group's counter (after the agent-scope release in
//               the safe protocol);   <the caller issues its independent loads>
//   hop_wait:   that lane polls the counter (relaxed, bounded), ONE L1 invalidate, workgroup barrier.
struct Hop {
  ch_gu32* cnt; ch_gu32* err; int* give_up; unsigned target; bool fast;
};
__device__ __forceinline__ void hop_arrive(const Hop& h) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every wave: its stores have been acknowledged by the L2
  __syncthreads();
  if (threadIdx.x == 0 && *h.give_up == 0) {
    if (!h.fast) {                                          // cross-XCD visibility: write the L2's dirty lines back
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (guide, G16 pitfall 12: the compiler may drop the fence's own wait)
    }
    __hip_atomic_fetch_add(h.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ __forceinline__ void hop_wait(const Hop& h) {
  if (threadIdx.x == 0 && *h.give_up == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(h.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < h.target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > CH_SPIN_LIMIT) {
        *h.give_up = 1;
        __hip_atomic_fetch_add(h.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // buffer_inv sc1: this CU's L1 forgets what it held
  }
  __syncthreads();
}

// ---- GEMM stage --------------------------------------------------------------------------------------------------
// (hand-off from the previous stage, then)  C[rows, n0 .. n0 + 32 NWC) = A[rows, k-range] . W[n0 .., k-range]^T for the tile
// this workgroup owns (`active`; the others only take part in the hand-off).  A, out point at the group's first row.
// EPI 0: fp32 slab (C = acc).  EPI 1: 16-bit C = acc * aux (aux = d quickgelu / du in the act dtype).
// Between arrive and wait: the saved derivative (EPI 1) and the W halves of the first NS - 1 k-tiles; after the wait the
// A halves, then the ring loop of gemm.hip (counted vmcnt, one barrier per k-tile).
template <typename T, int MT, int NWC, int NS, int EPI>
__device__ __attribute__((noinline)) void chain_gemm(char* smem, const Hop h, const bool active, const char* A, const int64_t lda,
                                                     const int arows, const char* W, const int64_t ldw, const int n0,
                                                     const int kt0, const int nk, char* out, const int64_t ldo,
                                                     const char* aux, const int64_t ldaux) {
  constexpr int BM = MT * 32, BN = NWC * 32, ROWS = BM + BN, SLOT = ROWS * 128, PPW = MT + NWC;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const bool computes = wave < NWC;

  hop_arrive(h);
  if (!active) { hop_wait(h); return; }

  uint2 auxr[MT][4];
  if constexpr (EPI == 1) {
    if (computes) {
#pragma unroll
      for (int tm = 0; tm < MT; ++tm)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int m = min(tm * 32 + l31, arows - 1);
          const int n = n0 + wave * 32 + 8 * g + 4 * half;
          auxr[tm][g] = *reinterpret_cast<const uint2*>(aux + ((int64_t)m * ldaux + n) * 2);
        }
    }
  }
  // DMA: a piece = 8 rows x 128 B.  Wave w owns A pieces w, w + 4, .. (MT of them) and W pieces w, w + 4, .. (NWC);
  // lane -> row 8 piece + (lane >> 3), physical chunk lane & 7 receives logical chunk (lane & 7) ^ ((row >> 1) & 7)
  const char* srca[MT];
  const char* srcw[NWC];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = 8 * (wave + 4 * i) + (lane >> 3);
    srca[i] = A + ((int64_t)min(row, arows - 1) * lda) * 2 + (((lane & 7) ^ ((row >> 1) & 7)) << 4) + (int64_t)kt0 * 128;
  }
#pragma unroll
  for (int i = 0; i < NWC; ++i) {
    const int row = 8 * (wave + 4 * i) + (lane >> 3);     // (BM is a multiple of 32: the swizzle term is that of row BM + row)
    srcw[i] = W + ((int64_t)(n0 + row) * ldw) * 2 + (((lane & 7) ^ ((row >> 1) & 7)) << 4) + (int64_t)kt0 * 128;
  }
  auto dma_a = [&](int slot, int kt) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(srca[i] + (int64_t)kt * 128), (lptr_t)(smem + slot * SLOT + (wave + 4 * i) * 1024), 16, 0, 0);
  };
  auto dma_w = [&](int slot, int kt) {
#pragma unroll
    for (int i = 0; i < NWC; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(srcw[i] + (int64_t)kt * 128),
                                       (lptr_t)(smem + slot * SLOT + BM * 128 + (wave + 4 * i) * 1024), 16, 0, 0);
  };
#pragma unroll
  for (int s_ = 0; s_ < NS - 1; ++s_)
    if (s_ < nk) dma_w(s_, s_);

  hop_wait(h);                                               // the previous stage's output is visible

#pragma unroll
  for (int s_ = 0; s_ < NS - 1; ++s_)
    if (s_ < nk) dma_a(s_, s_);

  f32x16_t acc[MT];
#pragma unroll
  for (int tm = 0; tm < MT; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f;

  const int sw = (l31 >> 1) & 7;
  const int rw = BM + wave * 32 + l31;                       // this wave's W rows in the slot
  for (int kt = 0; kt < nk; ++kt) {
    const int ahead = min(kt + NS - 2, nk - 1) - kt;         // tiles issued beyond kt
    if (ahead <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (kt == 0) {                                      // only the A halves of the tiles ahead are younger than tile 0's
      if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(MT) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * MT) : "memory");
    } else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PPW) : "memory");
    __builtin_amdgcn_s_barrier();                            // tile kt landed for every wave; tile kt - 1 has been read
    const bool more = kt + NS - 1 < nk;
    const int nslot = (kt + NS - 1) % NS, nkt = kt + NS - 1;
    const char* st = smem + (kt % NS) * SLOT;
    if (more) { dma_a(nslot, nkt); dma_w(nslot, nkt); }
    if (computes) {
      bf16x8_t xf[2][MT], wf[2];
      auto ldfrags = [&](int buf, int ks) {
        const int off = ((ks * 2 + half) ^ sw) << 4;
#pragma unroll
        for (int tm = 0; tm < MT; ++tm) xf[buf][tm] = *reinterpret_cast<const bf16x8_t*>(st + (tm * 32 + l31) * 128 + off);
        wf[buf] = *reinterpret_cast<const bf16x8_t*>(st + rw * 128 + off);
      };
      ldfrags(0, 0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks + 1 < 4) ldfrags((ks + 1) & 1, ks + 1);
#pragma unroll
        for (int tm = 0; tm < MT; ++tm) acc[tm] = mfma16<T>(wf[ks & 1], xf[ks & 1][tm], acc[tm]);
      }
    }
  }

  // epilogue straight from the C/D layout: lane (l31, half) holds row m = 32 tm + l31, columns 8 g + 4 half + {0..3}
  if (computes) {
#pragma unroll
    for (int tm = 0; tm < MT; ++tm) {
      const int m = tm * 32 + l31;
      if (m < arows) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + wave * 32 + 8 * g + 4 * half;
          float4 v = make_float4(acc[tm][4 * g], acc[tm][4 * g + 1], acc[tm][4 * g + 2], acc[tm][4 * g + 3]);
          if constexpr (EPI == 1) {
            const uint2 a = auxr[tm][g];
            v.x *= unpack1<T>((uint16_t)(a.x & 0xffffu)); v.y *= unpack1<T>((uint16_t)(a.x >> 16));
            v.z *= unpack1<T>((uint16_t)(a.y & 0xffffu)); v.w *= unpack1<T>((uint16_t)(a.y >> 16));
            *reinterpret_cast<uint2*>(out + ((int64_t)m * ldo + n) * 2) = make_uint2(pack2<T>(v.x, v.y), pack2<T>(v.z, v.w));
          } else {
            *reinterpret_cast<float4*>(out + ((int64_t)m * ldo + n) * 4) = v;
          }
        }
      }
    }
  }
}

// ---- LayerNorm stage ------------------------------------------------------------------------------------------------
// (hand-off, then) dx = dres + dLN(sum of the 4 slabs of dy; x, gamma), dxc = its 16-bit copy; one wave per row, the
// arithmetic of norm.hip's ln_bwd_row in the same order (no restrict qualifiers: the same buffers are rewritten stage
// after stage inside this launch).  x, gamma and dres do not come from the previous stage: requested before the wait.
template <typename TC, int NV>
__device__ __attribute__((noinline)) void chain_ln_stage(const Hop h, const float* dy, const int64_t dy_stride, char* dxc_, const int d,
                                                         const int64_t ldx, const float eps, const int gw, const int r0,
                                                         const int rg, const int slot, const float* x_, const float* gamma,
                                                         const float* dres_, float* dx_) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rpw = (rg + gw - 1) / gw;                        // rows per workgroup (3 of the 96 at 32 workgroups; <= 4)
  const int i_ = slot * rpw + wave;
  const bool mine = wave < rpw && i_ < rg;
  const int64_t r = r0 + (mine ? i_ : 0);
  const int nv4 = d >> 2;
  hop_arrive(h);
  float4 v[NV], w[NV], rr[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = min(lane + 64 * i, nv4 - 1);
    v[i] = *reinterpret_cast<const float4*>(x_ + r * ldx + 4 * c);
    w[i] = *reinterpret_cast<const float4*>(gamma + 4 * c);
    rr[i] = *reinterpret_cast<const float4*>(dres_ + r * d + 4 * c);
  }
  hop_wait(h);
  if (!mine) return;
  const float* dyr = dy + r * d;
  float4 g[NV];
  {
    float4 t[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = min(lane + 64 * i, nv4 - 1);
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) t[i][sp] = *reinterpret_cast<const float4*>(dyr + (int64_t)sp * dy_stride + 4 * c);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float4 a = t[i][0];
#pragma unroll
      for (int sp = 1; sp < 4; ++sp) {                       // four slabs, summed in order
        a.x = fmaf(1.0f, t[i][sp].x, a.x); a.y = fmaf(1.0f, t[i][sp].y, a.y);
        a.z = fmaf(1.0f, t[i][sp].z, a.z); a.w = fmaf(1.0f, t[i][sp].w, a.w);
      }
      const bool ok = lane + 64 * i < nv4;
      g[i] = ok ? make_float4(a.x * w[i].x, a.y * w[i].y, a.z * w[i].z, a.w * w[i].w) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (!ok) { v[i] = make_float4(0.f, 0.f, 0.f, 0.f); rr[i] = v[i]; }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mu = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (lane + 64 * i < nv4) {
      v[i].x -= mu; v[i].y -= mu; v[i].z -= mu; v[i].w -= mu;
      q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (lane + 64 * i < nv4) {
      v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;   // xhat
      sg += (g[i].x + g[i].y) + (g[i].z + g[i].w);
      sgx += (g[i].x * v[i].x + g[i].y * v[i].y) + (g[i].z * v[i].z + g[i].w * v[i].w);
    }
  }
  const float mg = wave_sum(sg) / (float)d;
  const float mgx = wave_sum(sgx) / (float)d;
  float* dx = dx_ + r * d;
  TC* dxc = reinterpret_cast<TC*>(dxc_) + r * d;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nv4) {
      float4 o = make_float4(rstd * (g[i].x - mg - v[i].x * mgx), rstd * (g[i].y - mg - v[i].y * mgx),
                             rstd * (g[i].z - mg - v[i].z * mgx), rstd * (g[i].w - mg - v[i].w * mgx));
      o.x += rr[i].x; o.y += rr[i].y; o.z += rr[i].z; o.w += rr[i].w;
      *reinterpret_cast<float4*>(dx + 4 * c) = o;
      ActIO<TC>::st4(dxc + 4 * c, o.x, o.y, o.z, o.w);
    }
  }
}

// ---- attention stage ------------------------------------------------------------------------------------------------
// (hand-off, then) dq of one (unit, head) item: attn_image.hip's attn_bwd_body without the folded d out-proj -- da comes
// from the 4 k-slice slabs of the GEMM stage in front (summed in order, rounded to the act dtype as the launch-per-stage
// chain's da matrix is).  q and the item's K / V rows (frozen tokens: written by the forward pass) are requested before
// the wait and land in registers while the group's last workgroups arrive.
struct ChainAttnArgs {
  const char* q_rows; const char* k; const char* v; const float* da; char* dq;
  const int32_t* key_len;
  int64_t ldq, ldkv, da_stride;
  int d, H, keys, Kp, key_stride, gw, u0, nu, slot;
  float scale;
};
template <typename T, int NT>
__device__ __attribute__((noinline)) void chain_attn_stage(char* smem, const Hop h, const ChainAttnArgs a) {
  using L = AL<T, NT>;
  constexpr int stat_off = L::BWD_BYTES;
  constexpr int CHUNKS = NT * 32 * 8, ITERS = (CHUNKS + 255) / 256;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  char* ks = smem;
  char* vs = smem + L::K_BYTES;
  float4* part_u = reinterpret_cast<float4*>(smem);                 // [4 waves][8 groups][64 lanes]
  float4* part_w = part_u + 4 * 8 * 64;
  float* part_d = reinterpret_cast<float*>(part_w + 4 * 8 * 64);    // [4 waves][64 lanes]
  const bf16x8_t i0 = ident_frag<T>(0, l31, half), i1 = ident_frag<T>(1, l31, half);

  hop_arrive(h);
  bool waited = false;
  for (int it = a.slot; it < a.nu * a.H || !waited; it += a.gw) {
    const bool live = it < a.nu * a.H;                      // (a workgroup without an item still takes part in the hand-off)
    const int b = a.u0 + (live ? it / a.H : 0), hh = live ? it % a.H : 0;
    int N = a.keys, kstride = a.keys;
    if (a.key_len != nullptr) { N = max(1, min(a.key_len[b], min(a.key_stride, NT * 32))); kstride = a.key_stride; }
    const T* kb = reinterpret_cast<const T*>(a.k) + (int64_t)b * kstride * a.ldkv + hh * 64;
    const T* vb = reinterpret_cast<const T*>(a.v) + (int64_t)b * kstride * a.ldkv + hh * 64;
    const int64_t prow = (int64_t)b * a.Kp + min(l31, a.Kp - 1);
    if (waited) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }   // the previous item's partials have been read
    RowFrag<T> qf, df;
    qf.load(reinterpret_cast<const T*>(a.q_rows) + prow * a.ldq + hh * 64, half);
    uint4 ka[ITERS], va[ITERS];
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {                        // unconditional (clamped) loads: a branch per load serialises them
      const int id = tid + i * 256, key = min(id >> 3, N - 1), c = id & 7;
      ka[i] = *reinterpret_cast<const uint4*>(kb + (int64_t)key * a.ldkv + c * 8);
      va[i] = *reinterpret_cast<const uint4*>(vb + (int64_t)key * a.ldkv + c * 8);
    }
    if (!waited) { hop_wait(h); waited = true; }
    if (!live) break;
    {                                                        // da of this lane's query: 4 slabs, summed in order
      const float* dar = a.da + prow * a.d + hh * 64 + half * 8;
      float4 t[4][4][2];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) {
          t[kk][sp][0] = *reinterpret_cast<const float4*>(dar + (int64_t)sp * a.da_stride + 16 * kk);
          t[kk][sp][1] = *reinterpret_cast<const float4*>(dar + (int64_t)sp * a.da_stride + 16 * kk + 4);
        }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float4 s4 = t[kk][0][e];
#pragma unroll
          for (int sp = 1; sp < 4; ++sp) { s4.x += t[kk][sp][e].x; s4.y += t[kk][sp][e].y; s4.z += t[kk][sp][e].z; s4.w += t[kk][sp][e].w; }
          f[4 * e] = s4.x; f[4 * e + 1] = s4.y; f[4 * e + 2] = s4.z; f[4 * e + 3] = s4.w;
        }
        df.f[kk] = pack8<T>(f);
      }
    }
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const int id = tid + i * 256, key = id >> 3, c = id & 7;
      if (id < CHUNKS) {
        const bool lv = key < N;                             // rows past the last key: zeros
        *reinterpret_cast<uint4*>(ks + key * 144 + c * 16) = lv ? ka[i] : make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(vs + key * 144 + c * 16) = lv ? va[i] : make_uint4(0, 0, 0, 0);
      }
    }
    __syncthreads();
    int l31v = l31;
    asm volatile("" : "+v"(l31v));
    // phase 1: row max and sum of the scores, each wave over the key tiles it owns in phase 2
    float m = -INFINITY, l = 0.f;
#pragma unroll 1
    for (int t = wave; t < NT; t += 4) {
      f32x16_t sc = tile_times_frag<T>(ks, t, qf, l31v, half);
      const float mn = fmaxf(m, mask_and_max(sc, t, N, half));
      if (mn > -INFINITY) {
        l = l * exp_scalar<T>(m - mn, a.scale) + exp_tile<T>(sc, mn, a.scale);
        m = mn;
      }
    }
    l += __shfl_xor(l, 32, 64);
    float2* rowstat = reinterpret_cast<float2*>(smem + stat_off);   // [4 waves][64 lanes]
    rowstat[wave * 64 + lane] = make_float2(m, l);
    __syncthreads();
    {
      const float2 s0 = rowstat[lane], s1 = rowstat[64 + lane], s2 = rowstat[128 + lane], s3 = rowstat[192 + lane];
      m = fmaxf(fmaxf(s0.x, s1.x), fmaxf(s2.x, s3.x));
      l = (s0.y * exp_scalar<T>(s0.x - m, a.scale) + s1.y * exp_scalar<T>(s1.x - m, a.scale)) +
          (s2.y * exp_scalar<T>(s2.x - m, a.scale) + s3.y * exp_scalar<T>(s3.x - m, a.scale));
    }
    const float inv = 1.0f / l;
    // phase 2: this wave's key tiles
    f32x16_t u[2], w[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { u[dt][r] = 0.f; w[dt][r] = 0.f; }
    float delta = 0.f;
#pragma unroll 1
    for (int t = wave; t < NT; t += 4) {
      f32x16_t pt = tile_times_frag<T>(ks, t, qf, l31v, half);
      (void)mask_and_max(pt, t, N, half);
      (void)exp_tile<T>(pt, m, a.scale);
      f32x16_t dp = tile_times_frag<T>(vs, t, df, l31v, half);
      const char* krow = ks + (32 * t + l31v) * 144 + half * 16;
      bf16x8_t krows[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) krows[kk] = *reinterpret_cast<const bf16x8_t*>(krow + kk * 32);
      float pw[16], pp[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pp[r] = pt[r] * inv;
        pw[r] = dp[r] * pp[r];                               // P * dP  (P = 0 on padded keys)
        delta += pw[r];
      }
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        bf16x8_t ktf[2];
        transpose_tile<T>(krows, dt, i0, i1, ktf);
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          u[dt] = mfma16<T>(ktf[g2], pack8<T>(pw + 8 * g2), u[dt]);
          w[dt] = mfma16<T>(ktf[g2], pack8<T>(pp + 8 * g2), w[dt]);
        }
      }
    }
    delta += __shfl_xor(delta, 32, 64);
    __syncthreads();                                         // everybody is done with the staged K / V
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
    if (l31 < a.Kp) {
      T* orow = reinterpret_cast<T*>(a.dq) + prow * a.d + hh * 64;
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        const int gi = wave * 2 + gg, dt = gi >> 2, g = gi & 3;
        float4 us = make_float4(0.f, 0.f, 0.f, 0.f), ws = us;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) {
          const float4 x = part_u[(wv * 8 + gi) * 64 + lane], c = part_w[(wv * 8 + gi) * 64 + lane];
          us.x += x.x; us.y += x.y; us.z += x.z; us.w += x.w;
          ws.x += c.x; ws.y += c.y; ws.z += c.z; ws.w += c.w;
        }
        ActIO<T>::st4(orow + 32 * dt + 8 * g + 4 * half, a.scale * (us.x - dl * ws.x), a.scale * (us.y - dl * ws.y),
                      a.scale * (us.z - dl * ws.z), a.scale * (us.w - dl * ws.w));
      }
    }
  }
}

// dynamic LDS: the larger of the GEMM ring and the attention stage's staging area, + 16 B of flags behind it (no static
// __shared__: it would shift the dynamic base off its 16-B alignment, guide G17)
template <typename T, int MT, int NWC, int NT>
constexpr int chain_lds_main() {
  constexpr int gemm_bytes = 3 * (MT * 32 + NWC * 32) * 128;
  constexpr int attn_bytes = AL<T, NT>::BWD_BYTES + 2048;
  return gemm_bytes > attn_bytes ? gemm_bytes : attn_bytes;
}

// NT: key tiles of the attention stage (7: <= 224 keys, 3: <= 96); NWC = d / 256; MT: 32-row tiles per group
template <typename T, int MT, int NWC, int NT>
__global__ __launch_bounds__(256, 2) void chain_bwd_kernel(const ChainParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int& give_up = *reinterpret_cast<int*>(smem + chain_lds_main<T, MT, NWC, NT>());
  int& same_xcd = *reinterpret_cast<int*>(smem + chain_lds_main<T, MT, NWC, NT>() + 4);
  constexpr int NS = 3, BN = NWC * 32, NV = NWC;             // NV float4 per lane and row: d = 256 NWC
  const int g = blockIdx.x % CH_GROUPS, slot = blockIdx.x / CH_GROUPS;
  const int base = p.units / CH_GROUPS, rem = p.units % CH_GROUPS;
  const int u0 = g * base + min(g, rem), nu = base + (g < rem ? 1 : 0);
  if (nu == 0) return;                                       // (whole groups leave: nobody waits for them)
  const int r0 = u0 * p.Kp, rg = nu * p.Kp;
  const int d = p.d;
  if (threadIdx.x == 0) { give_up = 0; same_xcd = 0; }
  Hop hop{(ch_gu32*)(p.state + 16 + 16 * g), (ch_gu32*)p.state, &give_up, 0u, false};
  auto next_hop = [&]() -> Hop { hop.target += (unsigned)p.gw; return hop; };
  // placement check: publish this workgroup's XCD, one placement-independent hand-off, compare
  if (threadIdx.x == 0)
    __hip_atomic_store((ch_gu32*)(p.state + 256 + 64 * g + slot), 1u + (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  {
    const Hop h0 = next_hop();
    hop_arrive(h0);
    hop_wait(h0);
    if (threadIdx.x == 0) {
      const unsigned first = __hip_atomic_load((ch_gu32*)(p.state + 256 + 64 * g), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int same = first != 0u;
      for (int i = 1; i < p.gw; ++i)
        same &= __hip_atomic_load((ch_gu32*)(p.state + 256 + 64 * g + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == first;
      same_xcd = same;
      if (!(same && !p.force_safe) && slot == 0)
        __hip_atomic_fetch_add((ch_gu32*)(p.state + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    hop.fast = same_xcd && !p.force_safe;
  }
  int tl = 0;
  auto stamp = [&]() {
    if (p.timeline != nullptr && blockIdx.x == 0 && threadIdx.x == 0) p.timeline[tl] = __builtin_amdgcn_s_memrealtime();
    ++tl;
  };
  stamp();

  float* dxa = p.dxa; float* dxb = p.dxb;
  const int64_t rb2 = (int64_t)r0 * d * 2;                   // byte offset of the group's rows in a [rows, d] 16-bit matrix
  const int tiles_d = d / BN;                                // column tiles of an N = d GEMM (8)
  const int tn = slot % tiles_d, ksl = slot / tiles_d;       // this workgroup's tile of the N = d GEMMs
  const bool act4 = slot < 4 * tiles_d;                      // (the GEMM stages have 4 * tiles_d = 32 tiles)
  char* dy_tile = reinterpret_cast<char*>(p.dy + (int64_t)(act4 ? ksl : 0) * p.dy_stride + (int64_t)r0 * d);
  for (int l = p.layers - 1; l >= 0; --l) {
    const ChainLayerDev L = p.layer[l];
    // (every stage function starts with the hand-off from the stage before it)
    // A: du = (dxc . Wproj) * quickgelu'     [rg, 4d], K = d
    chain_gemm<T, MT, NWC, NS, 1>(smem, next_hop(), act4, p.dxc + rb2, d, rg, L.w_proj_t, d, slot * BN, 0, d / 64,
                                  p.du + rb2 * 4, 4 * (int64_t)d, L.aux + rb2 * 4, 4 * (int64_t)d);
    stamp();
    // B: dy[s] = du[:, k-slice s] . Wfc[:, k-slice s]^T     [rg, d], K = 4d in 4 slices
    chain_gemm<T, MT, NWC, NS, 0>(smem, next_hop(), act4, p.du + rb2 * 4, 4 * (int64_t)d, rg, L.w_fc_t, 4 * (int64_t)d, tn * BN,
                                  ksl * (d / 64), d / 64, dy_tile, d, nullptr, 0);
    stamp();
    // C: ln_2 backward: dxb = dxa + dLN(sum of slabs), dxc = its 16-bit copy
    chain_ln_stage<T, NV>(next_hop(), p.dy, p.dy_stride, p.dxc, d, p.ldx, p.eps, p.gw, r0, rg, slot, L.x_ln2, L.ln2_w, dxa, dxb);
    stamp();
    // D0: da[s] = dxc[:, k-slice s] . Wout[:, k-slice s]^T   [rg, d], K = d in 4 slices (d / 256 k-tiles each)
    chain_gemm<T, MT, NWC, NS, 0>(smem, next_hop(), act4, p.dxc + rb2, d, rg, L.w_out_t, d, tn * BN, ksl * (d / 256), d / 256,
                                  dy_tile, d, nullptr, 0);
    stamp();
    // D1: dq = attention backward per (unit, head)
    chain_attn_stage<T, NT>(smem, next_hop(), ChainAttnArgs{L.q_rows, L.k, L.v, p.dy, p.dq, p.key_len, p.ldq, p.ldkv, p.dy_stride,
                                                            d, p.H, p.keys, p.Kp, p.key_stride, p.gw, u0, nu, slot, p.scale});
    stamp();
    // E: dy[s] = dq[:, k-slice s] . Wq[:, k-slice s]^T
    chain_gemm<T, MT, NWC, NS, 0>(smem, next_hop(), act4, p.dq + rb2, d, rg, L.w_q_t, d, tn * BN, ksl * (d / 256), d / 256,
                                  dy_tile, d, nullptr, 0);
    stamp();
    // F: ln_1 backward: dxa = dxb + dLN(sum of slabs), dxc = its 16-bit copy
    chain_ln_stage<T, NV>(next_hop(), p.dy, p.dy_stride, p.dxc, d, p.ldx, p.eps, p.gw, r0, rg, slot, L.x_ln1, L.ln1_w, dxb, dxa);
    stamp();
  }
}

template <typename T, int MT, int NWC, int NT>
int launch_chain(const ChainParams& p, hipStream_t s) {
  static rpo_lds_mask_t lds_ok{0};
  auto kern = chain_bwd_kernel<T, MT, NWC, NT>;
  constexpr int bytes = chain_lds_main<T, MT, NWC, NT>() + 16;
  static_assert(bytes <= 80 * 1024, "two workgroups per CU");
  static_assert(MT <= 4 && (MT * 32 + 3) / 4 <= 32, "rows per workgroup of the LayerNorm stages");
  if (int rc = rpo_allow_lds(reinterpret_cast<const void*>(kern), bytes, &lds_ok)) return rc;
  hipLaunchKernelGGL(kern, dim3(CH_GROUPS * p.gw), dim3(256), bytes, s, p);
  return rpo_launch_status();
}

template <typename T>
int dispatch_chain(const ChainParams& p, int mt, hipStream_t s) {
  if (p.d == 768 && p.keys > 96) {
    if (mt == 3) return launch_chain<T, 3, 3, 7>(p, s);
    if (mt == 2) return launch_chain<T, 2, 3, 7>(p, s);
    return launch_chain<T, 1, 3, 7>(p, s);
  }
  if (p.d == 512 && p.keys <= 96) {
    if (mt == 3) return launch_chain<T, 3, 2, 3>(p, s);
    if (mt == 2) return launch_chain<T, 2, 2, 3>(p, s);
    return launch_chain<T, 1, 2, 3>(p, s);
  }
  return RPO_E_SHAPE;
}

int chain_rows_per_group(const rpo_chain_bwd_args* a) { return ((a->units + CH_GROUPS - 1) / CH_GROUPS) * a->Kp; }

}  // namespace

extern "C" size_t rpo_chain_state_bytes(void) { return CH_STATE_WORDS * sizeof(unsigned); }

extern "C" int rpo_chain_bwd_ok(const rpo_chain_bwd_args* a) {
  if (!a || a->layers <= 0 || a->layers > CH_MAX_LAYERS || a->units <= 0 || a->Kp <= 0) return 0;
  if (a->dtype != RPO_BF16 && a->dtype != RPO_F16) return 0;
  if (a->d != a->H * 64 || a->Kp > 32 || chain_rows_per_group(a) > 96) return 0;
  if (a->d == 768) return a->keys > 96 && a->keys <= 224;
  if (a->d == 512) return a->keys > 0 && a->keys <= 96;
  return 0;
}

extern "C" int rpo_chain_bwd(const rpo_chain_bwd_args* a, void* stream) {
  if (!a || !a->layer || !a->dxa || !a->dxb || !a->dxc || !a->du || !a->dq || !a->dy || !a->state) return RPO_E_BADARG;
  if (a->dtype != RPO_BF16 && a->dtype != RPO_F16) return RPO_E_DTYPE;
  if (!rpo_chain_bwd_ok(a)) return RPO_E_SHAPE;
  int gw = a->wgs_per_group > 0 ? a->wgs_per_group : 32;
  if (const char* e = getenv("RPO_CHAIN_GW")) { if (atoi(e) > 0) gw = atoi(e); }       // A/B switch
  if (gw > 64 || gw < 32) return RPO_E_SHAPE;               // 32 GEMM tiles per stage and group; one row per wave in the LayerNorm stages
  const int64_t rows = (int64_t)a->units * a->Kp;
  if (a->dy_stride < rows * a->d || a->dy_stride % 4 || a->ldx % 4 || (a->ldq * 2) % 16 || (a->ldkv * 2) % 16) return RPO_E_ALIGN;
  if (a->key_len != nullptr && a->key_stride < a->keys) return RPO_E_BADARG;
  ChainParams p{};
  for (int l = 0; l < a->layers; ++l) {
    const rpo_chain_layer& s = a->layer[l];
    if (!s.w_proj_t || !s.w_fc_t || !s.w_out_t || !s.w_q_t || !s.aux || !s.x_ln2 || !s.x_ln1 || !s.ln2_w || !s.ln1_w ||
        !s.q_rows || !s.k || !s.v) return RPO_E_BADARG;
    if (!aligned16(s.w_proj_t) || !aligned16(s.w_fc_t) || !aligned16(s.w_out_t) || !aligned16(s.w_q_t) || !aligned16(s.aux) ||
        !aligned16(s.x_ln2) || !aligned16(s.x_ln1) || !aligned16(s.ln2_w) || !aligned16(s.ln1_w) || !aligned16(s.q_rows) ||
        !aligned16(s.k) || !aligned16(s.v)) return RPO_E_ALIGN;
    p.layer[l] = ChainLayerDev{static_cast<const char*>(s.w_proj_t), static_cast<const char*>(s.w_fc_t),
                               static_cast<const char*>(s.w_out_t), static_cast<const char*>(s.w_q_t),
                               static_cast<const char*>(s.aux), s.x_ln2, s.x_ln1, s.ln2_w, s.ln1_w,
                               static_cast<const char*>(s.q_rows), static_cast<const char*>(s.k), static_cast<const char*>(s.v)};
  }
  if (!aligned16(a->dxa) || !aligned16(a->dxb) || !aligned16(a->dxc) || !aligned16(a->du) || !aligned16(a->dq) || !aligned16(a->dy))
    return RPO_E_ALIGN;
  p.layers = a->layers; p.units = a->units; p.Kp = a->Kp; p.d = a->d; p.H = a->H; p.keys = a->keys;
  p.key_len = a->key_len; p.key_stride = a->key_len ? a->key_stride : a->keys;
  p.ldx = a->ldx; p.ldq = a->ldq; p.ldkv = a->ldkv;
  p.dxa = a->dxa; p.dxb = a->dxb; p.dxc = static_cast<char*>(a->dxc); p.du = static_cast<char*>(a->du);
  p.dq = static_cast<char*>(a->dq); p.dy = a->dy; p.dy_stride = a->dy_stride;
  p.scale = a->scale; p.eps = a->eps; p.gw = gw;
  { const char* e = getenv("RPO_CHAIN_SAFE"); p.force_safe = e && e[0] == '1'; }
  p.state = static_cast<unsigned*>(a->state);
  p.timeline = reinterpret_cast<unsigned long long*>(a->timeline);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipError_t e = hipMemsetAsync(a->state, 0, rpo_chain_state_bytes(), s); e != hipSuccess) return (int)e;
  const int rpg = chain_rows_per_group(a);
  const int mt = (rpg + 31) / 32;
  return a->dtype == RPO_BF16 ? dispatch_chain<bf16_t>(p, mt, s) : dispatch_chain<f16_t>(p, mt, s);
}
