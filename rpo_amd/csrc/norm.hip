// LayerNorm forward / backward-to-input, one wave per row (HBM-bound: each row is read
// once with 16-B loads, reduced with wave shuffles, written once).
//
// Replaces the reference's fp32-upcast LayerNorm (clip/model.py:153-159: ln_1, ln_2,
// ln_pre, ln_post, ln_final) and autograd through it.  Statistics are two-pass in
// registers (mean, then centred second moment), biased variance, eps inside the rsqrt --
// the same formula torch's CPU kernel evaluates.
#include "common.h"

// rows (= waves) per workgroup.  One row per workgroup spreads the 768-row LayerNorms of the backward chains over
// all CUs (step-level A/B: 4 -> 1 rows per workgroup = -0.6 % step time; the 7 072-row forward ones do not care).
#ifndef RPO_LN_RPB
#define RPO_LN_RPB 1
#endif

namespace {

constexpr int MAXV = 8;   // float4 per lane: d <= 64 * 4 * 8 = 2048 (kernels are templated on the count)

template <typename TY, int NV>
__global__ __launch_bounds__(64 * RPO_LN_RPB) void ln_fwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, TY* y, int64_t ldy,
                                                     int rows, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * RPO_LN_RPB + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv4 = d >> 2;
  const float* xr = x + (int64_t)row * ldx;
  float4 v[NV], gm[NV], bt[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < nv4) {
      v[i] = *reinterpret_cast<const float4*>(xr + 4 * c);
      gm[i] = *reinterpret_cast<const float4*>(gamma + 4 * c);   // issued with x: one memory round trip
      bt[i] = *reinterpret_cast<const float4*>(beta + 4 * c);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mu = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nv4) {
      const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, dd = v[i].w - mu;
      q += (a * a + b * b) + (cc * cc + dd * dd);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
  TY* yr = y + (int64_t)row * ldy;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nv4) {
      const float4 g = gm[i], b = bt[i];
      ActIO<TY>::st4(yr + 4 * c, (v[i].x - mu) * rstd * g.x + b.x, (v[i].y - mu) * rstd * g.y + b.y,
                     (v[i].z - mu) * rstd * g.z + b.z, (v[i].w - mu) * rstd * g.w + b.w);
    }
  }
}

// Image-tower input in one launch (trainers/rpo.py:201-206 + the first ln_1, clip/model.py:189): per row, the token
// (CLS + pos[0] | the patch row the patch GEMM wrote | the image's prompt row), ln_pre of it (-> x0, fp32) and ln_1 of
// that (-> h, act dtype).  Replaces assemble + two LayerNorm launches; every value is formed by the expressions of
// ln_fwd_kernel in the same order, so x0 and h carry the bits the three launches produce.
template <typename TY, int NV>
__global__ __launch_bounds__(64 * RPO_LN_RPB) void img_embed_norm_kernel(float* x_pre, int64_t ldx,
    const float* __restrict__ cls, const float* __restrict__ pos0, const float* __restrict__ prompt,
    const float* __restrict__ g_pre, const float* __restrict__ b_pre, float* x0, int64_t ldx0,
    const float* __restrict__ g1, const float* __restrict__ b1, TY* h, int64_t ldh, int B, int N, int Kp, int d,
    float eps, int row0, int row1) {
  const int lane = threadIdx.x & 63;
  const int row = row0 + blockIdx.x * RPO_LN_RPB + (threadIdx.x >> 6);      // rows [row0, row1) of the B * (N + Kp)
  if (row >= row1) return;
  const int nv4 = d >> 2;
  const bool is_prompt = row >= B * N;
  const bool is_cls = !is_prompt && row % N == 0;
  const float* src = is_prompt ? prompt + (int64_t)((row - B * N) % Kp) * d : (is_cls ? cls : x_pre + (int64_t)row * ldx);
  float4 v[NV], ga[NV], ba[NV], gb[NV], bb[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < nv4) {
      v[i] = *reinterpret_cast<const float4*>(src + 4 * c);
      if (is_cls) {
        const float4 p4 = *reinterpret_cast<const float4*>(pos0 + 4 * c);
        v[i].x += p4.x; v[i].y += p4.y; v[i].z += p4.z; v[i].w += p4.w;
      }
      ga[i] = *reinterpret_cast<const float4*>(g_pre + 4 * c); ba[i] = *reinterpret_cast<const float4*>(b_pre + 4 * c);
      gb[i] = *reinterpret_cast<const float4*>(g1 + 4 * c);    bb[i] = *reinterpret_cast<const float4*>(b1 + 4 * c);
      if (is_cls || is_prompt) *reinterpret_cast<float4*>(x_pre + (int64_t)row * ldx + 4 * c) = v[i];   // the backward reads these rows
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  auto norm = [&](float4 (&g)[NV], float4 (&b)[NV], float ssum) {
    const float mu = wave_sum(ssum) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv4) {
        const float a = v[i].x - mu, b2 = v[i].y - mu, cc = v[i].z - mu, dd = v[i].w - mu;
        q += (a * a + b2 * b2) + (cc * cc + dd * dd);
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
    float sn = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv4) {
        v[i] = make_float4((v[i].x - mu) * rstd * g[i].x + b[i].x, (v[i].y - mu) * rstd * g[i].y + b[i].y,
                           (v[i].z - mu) * rstd * g[i].z + b[i].z, (v[i].w - mu) * rstd * g[i].w + b[i].w);
        sn += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      }
    }
    return sn;
  };
  const float s1 = norm(ga, ba, s);                                   // ln_pre
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nv4) *reinterpret_cast<float4*>(x0 + (int64_t)row * ldx0 + 4 * c) = v[i];
  }
  (void)norm(gb, bb, s1);                                             // ln_1 of the first block
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nv4) ActIO<TY>::st4(h + (int64_t)row * ldh + 4 * c, v[i].x, v[i].y, v[i].z, v[i].w);
  }
}

template <typename T> __device__ __forceinline__ float4 load4f(const T* p);
template <> __device__ __forceinline__ float4 load4f<float>(const float* p) {
  return *reinterpret_cast<const float4*>(p);
}
template <> __device__ __forceinline__ float4 load4f<bf16_t>(const bf16_t* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                     __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}

template <> __device__ __forceinline__ float4 load4f<f16_t>(const f16_t* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(unpack1<f16_t>((uint16_t)(u.x & 0xffffu)), unpack1<f16_t>((uint16_t)(u.x >> 16)),
                     unpack1<f16_t>((uint16_t)(u.y & 0xffffu)), unpack1<f16_t>((uint16_t)(u.y >> 16)));
}

// dx = dres + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
template <typename TDY, typename TC, int NV>
__device__ __forceinline__ void ln_bwd_row(const TDY* __restrict__ dy, int64_t lddy,
                                           const float* __restrict__ x, int64_t ldx,
                                           const float* __restrict__ gamma,
                                           const float* __restrict__ dres, int64_t lddres,
                                           float* dx, int64_t lddx, TC* dxc, int64_t ldc,
                                           int rows, int d, float eps, int splits,
                                           int64_t split_stride, const int blk) {
  const int lane = threadIdx.x & 63;
  const int row = blk * RPO_LN_RPB + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv4 = d >> 2;
  const float* xr = x + (int64_t)row * ldx;
  const TDY* dyr = dy + (int64_t)row * lddy;
  // every global load of the row (x, dy slabs, gamma) is issued before the first reduction: the row's
  // lifetime is then ONE memory round trip plus four wave reductions (these launches are latency-bound)
  float4 v[NV], g[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < nv4) {
      v[i] = *reinterpret_cast<const float4*>(xr + 4 * c);
      const float4 w = *reinterpret_cast<const float4*>(gamma + 4 * c);
      // split-K slabs, summed in fixed order.  The first four are fetched unconditionally (clamped index, masked add):
      // a load inside a loop with a runtime trip count is one memory round trip per slab on a latency-bound chain
      float4 t = load4f<TDY>(dyr + 4 * c);
      float4 ts[3];
#pragma unroll
      for (int sp = 1; sp < 4; ++sp) ts[sp - 1] = load4f<TDY>(dyr + (int64_t)min(sp, splits - 1) * split_stride + 4 * c);
#pragma unroll
      for (int sp = 1; sp < 4; ++sp) {
        const float m = sp < splits ? 1.0f : 0.0f;
        t.x = fmaf(m, ts[sp - 1].x, t.x); t.y = fmaf(m, ts[sp - 1].y, t.y);
        t.z = fmaf(m, ts[sp - 1].z, t.z); t.w = fmaf(m, ts[sp - 1].w, t.w);
      }
      for (int sp = 4; sp < splits; ++sp) {
        const float4 t2 = load4f<TDY>(dyr + (int64_t)sp * split_stride + 4 * c);
        t.x += t2.x; t.y += t2.y; t.z += t2.z; t.w += t2.w;
      }
      g[i] = make_float4(t.x * w.x, t.y * w.y, t.z * w.z, t.w * w.w);
    }
  }
  float4 rr[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    rr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (dres != nullptr && c < nv4) rr[i] = *reinterpret_cast<const float4*>(dres + (int64_t)row * lddres + 4 * c);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mu = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nv4) {
      v[i].x -= mu; v[i].y -= mu; v[i].z -= mu; v[i].w -= mu;
      q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nv4) {
      v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;   // xhat
      sg += (g[i].x + g[i].y) + (g[i].z + g[i].w);
      sgx += (g[i].x * v[i].x + g[i].y * v[i].y) + (g[i].z * v[i].z + g[i].w * v[i].w);
    }
  }
  const float mg = wave_sum(sg) / (float)d;
  const float mgx = wave_sum(sgx) / (float)d;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nv4) {
      float4 o = make_float4(rstd * (g[i].x - mg - v[i].x * mgx), rstd * (g[i].y - mg - v[i].y * mgx),
                             rstd * (g[i].z - mg - v[i].z * mgx), rstd * (g[i].w - mg - v[i].w * mgx));
      o.x += rr[i].x; o.y += rr[i].y; o.z += rr[i].z; o.w += rr[i].w;
      *reinterpret_cast<float4*>(dx + (int64_t)row * lddx + 4 * c) = o;
      if (dxc != nullptr) ActIO<TC>::st4(dxc + (int64_t)row * ldc + 4 * c, o.x, o.y, o.z, o.w);
    }
  }
}

template <typename TDY, typename TC, int NV>
__global__ __launch_bounds__(64 * RPO_LN_RPB) void ln_bwd_kernel(const TDY* __restrict__ dy, int64_t lddy,
                                                     const float* __restrict__ x, int64_t ldx,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ dres, int64_t lddres,
                                                     float* dx, int64_t lddx, TC* dxc, int64_t ldc,
                                                     int rows, int d, float eps, int splits,
                                                     int64_t split_stride) {
  ln_bwd_row<TDY, TC, NV>(dy, lddy, x, ldx, gamma, dres, lddres, dx, lddx, dxc, ldc, rows, d, eps, splits, split_stride,
                          blockIdx.x);
}

// Two LayerNorm backward problems in one launch (fp32 dy slabs, the same cast dtype): blocks [0, blocks0) take the rows of
// problem 0, the rest those of problem 1 -- the same stage of the image tower's and the text tower's prompt-row chain.
struct LnBwdProblem {
  const float* dy; int64_t lddy; const float* x; int64_t ldx; const float* gamma; const float* dres; int64_t lddres;
  float* dx; int64_t lddx; void* dxc; int64_t ldc; int rows, d; float eps; int splits; int64_t split_stride;
};
struct LnBwdPair { LnBwdProblem p[2]; int blocks0; };
template <typename TC, int NV>
__global__ __launch_bounds__(64 * RPO_LN_RPB) void ln_bwd_pair_kernel(const LnBwdPair g) {
  const int second = (int)blockIdx.x >= g.blocks0;
  const LnBwdProblem& q = g.p[second];
  ln_bwd_row<float, TC, NV>(q.dy, q.lddy, q.x, q.ldx, q.gamma, q.dres, q.lddres, q.dx, q.lddx, static_cast<TC*>(q.dxc),
                            q.ldc, q.rows, q.d, q.eps, q.splits, q.split_stride,
                            second ? (int)blockIdx.x - g.blocks0 : (int)blockIdx.x);
}

}  // namespace

extern "C" int rpo_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta,
                                 void* y, int64_t ldy, int y_dtype, int rows, int d, float eps,
                                 void* stream) {
  if (!x || !gamma || !beta || !y || rows <= 0 || d <= 0) return RPO_E_BADARG;
  if (d % 4 != 0 || d > 64 * 4 * MAXV) return RPO_E_SHAPE;
  if (!aligned16(x) || !aligned16(gamma) || !aligned16(beta) || ldx % 4 != 0 || ldy % 4 != 0) return RPO_E_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((rows + RPO_LN_RPB - 1) / RPO_LN_RPB), block(64 * RPO_LN_RPB);
  const int nv = (d / 4 + 63) / 64;
#define RPO_LN_FWD(TY, NV)                                                                          \
  hipLaunchKernelGGL((ln_fwd_kernel<TY, NV>), grid, block, 0, s, x, ldx, gamma, beta, static_cast<TY*>(y), \
                     ldy, rows, d, eps)
#define RPO_LN_FWD_NV(TY)                                                                           \
  do {                                                                                              \
    if (nv <= 2) RPO_LN_FWD(TY, 2); else if (nv == 3) RPO_LN_FWD(TY, 3);                            \
    else if (nv == 4) RPO_LN_FWD(TY, 4); else RPO_LN_FWD(TY, 8);                                    \
  } while (0)
  if (y_dtype == RPO_F32) {
    if (!aligned16(y)) return RPO_E_ALIGN;
    RPO_LN_FWD_NV(float);
  } else if (y_dtype == RPO_BF16) {
    if (reinterpret_cast<uintptr_t>(y) % 8) return RPO_E_ALIGN;
    RPO_LN_FWD_NV(bf16_t);
  } else if (y_dtype == RPO_F16) {
    if (reinterpret_cast<uintptr_t>(y) % 8) return RPO_E_ALIGN;
    RPO_LN_FWD_NV(f16_t);
  } else {
    return RPO_E_DTYPE;
  }
#undef RPO_LN_FWD_NV
#undef RPO_LN_FWD
  return rpo_launch_status();
}

static int ln_bwd_check(const void* dy, int dy_dtype, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                        const float* dres, int64_t lddres, float* dx, int64_t lddx, void* dx_cast, int64_t ldcast, int rows,
                        int d, int& dy_splits, int64_t dy_split_stride) {
  if (!dy || !x || !gamma || !dx || rows <= 0 || d <= 0) return RPO_E_BADARG;
  if (d % 4 != 0 || d > 64 * 4 * MAXV) return RPO_E_SHAPE;
  if (dy_splits < 1) dy_splits = 1;
  if (dy_splits > 1 && (dy_dtype != RPO_F32 || dy_split_stride % 4 != 0)) return RPO_E_SHAPE;
  if (!aligned16(x) || !aligned16(gamma) || !aligned16(dx) || ldx % 4 || lddx % 4 || lddy % 4) return RPO_E_ALIGN;
  if (dres && (!aligned16(dres) || lddres % 4)) return RPO_E_ALIGN;
  if (dx_cast && (reinterpret_cast<uintptr_t>(dx_cast) % 8 || ldcast % 4)) return RPO_E_ALIGN;
  if (reinterpret_cast<uintptr_t>(dy) % (dy_dtype == RPO_F32 ? 16 : 8)) return RPO_E_ALIGN;
  return 0;
}

#ifdef RPO_EXPERIMENTAL   // measured-slower experiment: include/rpo_amd_experimental.h
extern "C" int rpo_layernorm_bwd_pair(const rpo_ln_bwd_args* a0, const rpo_ln_bwd_args* a1, void* stream) {
  if (!a0 || !a1) return RPO_E_BADARG;
  LnBwdPair g;
  int nv = 0;
  for (int i = 0; i < 2; ++i) {
    const rpo_ln_bwd_args* a = i ? a1 : a0;
    int splits = a->dy_splits;
    if (int rc = ln_bwd_check(a->dy, RPO_F32, a->lddy, a->x, a->ldx, a->gamma, a->dres, a->lddres, a->dx, a->lddx,
                              a->dx_cast, a->ldcast, a->rows, a->d, splits, a->dy_split_stride)) return rc;
    g.p[i] = LnBwdProblem{a->dy, a->lddy, a->x, a->ldx, a->gamma, a->dres, a->lddres, a->dx, a->lddx, a->dx_cast,
                          a->ldcast, a->rows, a->d, a->eps, splits, a->dy_split_stride};
    { const int n = (a->d / 4 + 63) / 64; if (n > nv) nv = n; }
  }
  const int cast = a0->dx_cast ? a0->cast_dtype : (a1->dx_cast ? a1->cast_dtype : RPO_F32);
  if ((a0->dx_cast && a0->cast_dtype != cast) || (a1->dx_cast && a1->cast_dtype != cast)) return RPO_E_DTYPE;
  if (cast != RPO_F32 && cast != RPO_BF16 && cast != RPO_F16) return RPO_E_DTYPE;
  g.blocks0 = (a0->rows + RPO_LN_RPB - 1) / RPO_LN_RPB;
  const dim3 grid(g.blocks0 + (a1->rows + RPO_LN_RPB - 1) / RPO_LN_RPB), block(64 * RPO_LN_RPB);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define RPO_LN_PAIR_(TC, NV) hipLaunchKernelGGL((ln_bwd_pair_kernel<TC, NV>), grid, block, 0, s, g)
#define RPO_LN_PAIR(TC)                                                                                    \
  do {                                                                                                     \
    if (nv <= 2) RPO_LN_PAIR_(TC, 2); else if (nv == 3) RPO_LN_PAIR_(TC, 3);                               \
    else if (nv == 4) RPO_LN_PAIR_(TC, 4); else RPO_LN_PAIR_(TC, 8);                                       \
  } while (0)
  if (cast == RPO_BF16) RPO_LN_PAIR(bf16_t); else if (cast == RPO_F16) RPO_LN_PAIR(f16_t); else RPO_LN_PAIR(float);
#undef RPO_LN_PAIR
#undef RPO_LN_PAIR_
  return rpo_launch_status();
}
#endif  // RPO_EXPERIMENTAL

extern "C" int rpo_layernorm_bwd(const void* dy, int dy_dtype, int64_t lddy, const float* x, int64_t ldx,
                                 const float* gamma, const float* dres, int64_t lddres, float* dx,
                                 int64_t lddx, void* dx_cast, int cast_dtype, int64_t ldcast, int rows,
                                 int d, float eps, int dy_splits, int64_t dy_split_stride, void* stream) {
  if (int rc = ln_bwd_check(dy, dy_dtype, lddy, x, ldx, gamma, dres, lddres, dx, lddx, dx_cast, ldcast, rows, d, dy_splits,
                            dy_split_stride)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((rows + RPO_LN_RPB - 1) / RPO_LN_RPB), block(64 * RPO_LN_RPB);
  const int nv = (d / 4 + 63) / 64;
#define RPO_LN_BWD_(TDY, TC, NV)                                                                          \
  hipLaunchKernelGGL((ln_bwd_kernel<TDY, TC, NV>), grid, block, 0, s, static_cast<const TDY*>(dy), lddy, \
                     x, ldx, gamma, dres, lddres, dx, lddx, static_cast<TC*>(dx_cast), ldcast, rows,     \
                     d, eps, dy_splits, dy_split_stride)
#define RPO_LN_BWD(TDY, TC)                                                                    \
  do {                                                                                         \
    if (nv <= 2) RPO_LN_BWD_(TDY, TC, 2); else if (nv == 3) RPO_LN_BWD_(TDY, TC, 3);           \
    else if (nv == 4) RPO_LN_BWD_(TDY, TC, 4); else RPO_LN_BWD_(TDY, TC, 8);                   \
  } while (0)
  const bool cast_bf16 = dx_cast != nullptr && cast_dtype == RPO_BF16;
  const bool cast_f16 = dx_cast != nullptr && cast_dtype == RPO_F16;
  if (dx_cast != nullptr && cast_dtype != RPO_BF16 && cast_dtype != RPO_F16 && cast_dtype != RPO_F32) return RPO_E_DTYPE;
  if (dy_dtype == RPO_F32) {
    if (cast_bf16) RPO_LN_BWD(float, bf16_t); else if (cast_f16) RPO_LN_BWD(float, f16_t); else RPO_LN_BWD(float, float);
  } else if (dy_dtype == RPO_BF16) {
    if (cast_bf16) RPO_LN_BWD(bf16_t, bf16_t); else RPO_LN_BWD(bf16_t, float);
  } else if (dy_dtype == RPO_F16) {
    if (cast_f16) RPO_LN_BWD(f16_t, f16_t); else RPO_LN_BWD(f16_t, float);
  } else {
    return RPO_E_DTYPE;
  }
#undef RPO_LN_BWD
#undef RPO_LN_BWD_
  return rpo_launch_status();
}

extern "C" int rpo_img_embed_norm(float* x_pre, int64_t ldx, const float* cls, const float* pos0, const float* img_prompt,
                                  const float* g_pre, const float* b_pre, float* x0, int64_t ldx0, const float* g1,
                                  const float* b1, void* h, int64_t ldh, int h_dtype, int B, int N, int Kp, int d,
                                  float eps, void* stream) {
  return rpo_img_embed_norm_rows(x_pre, ldx, cls, pos0, img_prompt, g_pre, b_pre, x0, ldx0, g1, b1, h, ldh, h_dtype, B, N, Kp,
                                 d, eps, 0, B * (N + Kp), stream);
}

extern "C" int rpo_img_embed_norm_rows(float* x_pre, int64_t ldx, const float* cls, const float* pos0,
                                       const float* img_prompt, const float* g_pre, const float* b_pre, float* x0,
                                       int64_t ldx0, const float* g1, const float* b1, void* h, int64_t ldh, int h_dtype,
                                       int B, int N, int Kp, int d, float eps, int row0, int row1, void* stream) {
  if (!x_pre || !cls || !pos0 || !g_pre || !b_pre || !x0 || !g1 || !b1 || !h || B <= 0 || N <= 0 || Kp < 0 || d <= 0 ||
      row0 < 0 || row1 > B * (N + Kp) || row0 >= row1 || (Kp > 0 && row1 > B * N && !img_prompt)) return RPO_E_BADARG;
  if (d % 4 != 0 || d > 64 * 4 * 4) return RPO_E_SHAPE;
  if (!aligned16(x_pre) || !aligned16(cls) || !aligned16(pos0) || (Kp > 0 && !aligned16(img_prompt)) || !aligned16(g_pre) ||
      !aligned16(b_pre) || !aligned16(x0) || !aligned16(g1) || !aligned16(b1) || ldx % 4 != 0 || ldx0 % 4 != 0 ||
      ldh % 4 != 0 || reinterpret_cast<uintptr_t>(h) % (h_dtype == RPO_F32 ? 16 : 8)) return RPO_E_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int rows = row1 - row0;
  const dim3 grid((rows + RPO_LN_RPB - 1) / RPO_LN_RPB), block(64 * RPO_LN_RPB);
  const int nv = (d / 4 + 63) / 64;
#define RPO_IEN(TY, NV)                                                                                       \
  hipLaunchKernelGGL((img_embed_norm_kernel<TY, NV>), grid, block, 0, s, x_pre, ldx, cls, pos0, img_prompt, g_pre, \
                     b_pre, x0, ldx0, g1, b1, static_cast<TY*>(h), ldh, B, N, Kp, d, eps, row0, row1)
#define RPO_IEN_NV(TY) do { if (nv <= 2) RPO_IEN(TY, 2); else if (nv == 3) RPO_IEN(TY, 3); else RPO_IEN(TY, 4); } while (0)
  if (h_dtype == RPO_F32) RPO_IEN_NV(float);
  else if (h_dtype == RPO_BF16) RPO_IEN_NV(bf16_t);
  else if (h_dtype == RPO_F16) RPO_IEN_NV(f16_t);
  else return RPO_E_DTYPE;
#undef RPO_IEN_NV
#undef RPO_IEN
  return rpo_launch_status();
}
