// Text-tower attention against per-class cached keys/values.
//
// Reference semantics: nn.MultiheadAttention with the per-class additive mask of
// trainers/rpo.py:144-151 (causal AND column < len_c).  The K prompt rows sit at positions
// len_c .. len_c+K-1 (:176-177), i.e. behind every allowed column, so each of them reads
// exactly keys [0, len_c) of its class; frozen token t reads keys [0, min(t+1, len_c)).
//
// Sizes are tiny (<= 77 keys, head_dim 64 = one wave): this is a VALU kernel.  One workgroup
// per (class, head) stages that class's K and V head slices in LDS once; each wave then
// walks query rows.  head_dim 64 == wave size, so "lane = key" for the score / dP passes
// and "lane = feature" for the P.V / dS.K passes; operands are broadcast with v_readlane.
#include "common.h"

namespace {

__device__ __forceinline__ float bcast(float v, int lane_uniform) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_uniform));
}

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void text_attn_kernel(const T* __restrict__ q, int64_t ldq,
                                                        const T* __restrict__ kc, const T* __restrict__ vc,
                                                        int64_t ldkv, const T* __restrict__ da, int64_t ldda,
                                                        T* out, int64_t ldo, const int32_t* __restrict__ len,
                                                        int rows, int Lmax, int H, int causal, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Kf = reinterpret_cast<float*>(smem);
  float* Vf = Kf + Lmax * 65;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = blockIdx.x / H, h = blockIdx.x % H;
  const int L = min(len[c], Lmax);
  // stage K and V head slices as fp32 [L][65]; 16-B global loads, all issued before the LDS writes
  constexpr int EPC = 16 / sizeof(T);          // elements per 16-B chunk
  constexpr int CPR = 64 / EPC;                // chunks per key row
  constexpr int ITERS = 128 * CPR / 256;       // Lmax <= 128
  uint4 kv[ITERS], vv[ITERS];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int id = tid + it * 256;
    const int j = id / CPR, cch = id % CPR;
    if (j < L) {
      const int64_t off = ((int64_t)c * Lmax + j) * ldkv + h * 64 + cch * EPC;
      kv[it] = *reinterpret_cast<const uint4*>(kc + off);
      vv[it] = *reinterpret_cast<const uint4*>(vc + off);
    }
  }
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int id = tid + it * 256;
    const int j = id / CPR, cch = id % CPR;
    if (j < L) {
      float* kd = Kf + j * 65 + cch * EPC;
      float* vd = Vf + j * 65 + cch * EPC;
      const uint32_t kw[4] = {kv[it].x, kv[it].y, kv[it].z, kv[it].w};
      const uint32_t vw[4] = {vv[it].x, vv[it].y, vv[it].z, vv[it].w};
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          kd[2 * e] = unpack1<T>((uint16_t)(kw[e] & 0xffffu)); kd[2 * e + 1] = unpack1<T>((uint16_t)(kw[e] >> 16));
          vd[2 * e] = unpack1<T>((uint16_t)(vw[e] & 0xffffu)); vd[2 * e + 1] = unpack1<T>((uint16_t)(vw[e] >> 16));
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { kd[e] = __uint_as_float(kw[e]); vd[e] = __uint_as_float(vw[e]); }
      }
    }
  }
  __syncthreads();
  const int j0 = min(lane, Lmax - 1), j1 = min(lane + 64, Lmax - 1);
  // one query row per wave; blockIdx.y walks the rows in chunks of 4 (each chunk re-stages the tiny K/V
  // slices): a row is ~1.5 k VALU instructions, so spreading rows over workgroups is what shortens the launch
  for (int r = blockIdx.y * 4 + wave; r < rows; r += 4 * gridDim.y) {
    const int64_t row = (int64_t)c * rows + r;
    const int nk = causal ? min(r + 1, L) : L;
    const float qv = ActIO<T>::ld(q + row * ldq + h * 64 + lane);
    float s0 = 0.f, s1 = 0.f;
#pragma unroll 8
    for (int d = 0; d < 64; ++d) {
      const float qd = bcast(qv, d);
      s0 = fmaf(qd, Kf[j0 * 65 + d], s0);
      s1 = fmaf(qd, Kf[j1 * 65 + d], s1);
    }
    s0 = lane < nk ? s0 * scale : -INFINITY;
    s1 = lane + 64 < nk ? s1 * scale : -INFINITY;
    const float m = wave_max(fmaxf(s0, s1));
    float p0 = expf(s0 - m), p1 = expf(s1 - m);
    const float inv = 1.0f / wave_sum(p0 + p1);
    p0 *= inv; p1 *= inv;
    if (!BWD) {
      float acc = 0.f;
      for (int j = 0; j < nk; ++j) {
        const float pj = j < 64 ? bcast(p0, j) : bcast(p1, j - 64);
        acc = fmaf(pj, Vf[j * 65 + lane], acc);
      }
      ActIO<T>::st(out + row * ldo + h * 64 + lane, acc);
    } else {
      const float dv = ActIO<T>::ld(da + row * ldda + h * 64 + lane);
      float dp0 = 0.f, dp1 = 0.f;
#pragma unroll 8
      for (int d = 0; d < 64; ++d) {
        const float dd = bcast(dv, d);
        dp0 = fmaf(dd, Vf[j0 * 65 + d], dp0);
        dp1 = fmaf(dd, Vf[j1 * 65 + d], dp1);
      }
      dp0 = lane < nk ? dp0 : 0.f;
      dp1 = lane + 64 < nk ? dp1 : 0.f;
      const float delta = wave_sum(p0 * dp0 + p1 * dp1);
      const float ds0 = p0 * (dp0 - delta), ds1 = p1 * (dp1 - delta);
      float acc = 0.f;
      for (int j = 0; j < nk; ++j) {
        const float dj = j < 64 ? bcast(ds0, j) : bcast(ds1, j - 64);
        acc = fmaf(dj, Kf[j * 65 + lane], acc);
      }
      ActIO<T>::st(out + row * ldo + h * 64 + lane, acc * scale);
    }
  }
}

// ---- dense backward (all rows, dQ / dK / dV) -------------------------------------------------------------------------
// The sibling trainers train parameters that sit IN FRONT of the class name (CoOp's context vectors, trainers/coop.py:
// 117-134), so their gradient flows through every token of the text tower under the plain causal mask
// (clip/model.py:332-338): autograd of nn.MultiheadAttention for q, k, v of ALL L = len[c] rows of a class.  Rows past
// the EOT token (>= len[c]) cannot reach the text feature (it is read at the EOT row, which sees columns <= itself) and
// get zero gradients.  One workgroup per (class, head); L <= 80 (CLIP's context is 77), head_dim 64 = one wave:
//   pass 1, one wave per query row r:   p = softmax(scale q_r K^T) over keys j <= r;  dp_j = <dO_r, v_j>;
//           ds_j = p_j (dp_j - sum_j p_j dp_j);   dq_r = scale sum_j ds_j k_j;   P[r][:], DS[r][:] parked in LDS
//   pass 2, one wave per key j:         dk_j = scale sum_{r >= j} DS[r][j] q_r;   dv_j = sum_{r >= j} P[r][j] dO_r
// fp32 arithmetic throughout, fixed summation orders (bit-reproducible).
template <typename T>
__global__ __launch_bounds__(256) void text_attn_bwd_dense_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                                  const T* __restrict__ v, int64_t ld,
                                                                  const T* __restrict__ dout, int64_t lddo, T* dq, T* dk,
                                                                  T* dv, int64_t ldd, const int32_t* __restrict__ len,
                                                                  int Lmax, int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int LP = Lmax + 1;
  float* Qf = reinterpret_cast<float*>(smem);
  float* Kf = Qf + Lmax * 65;
  float* Vf = Kf + Lmax * 65;
  float* Df = Vf + Lmax * 65;                   // dO
  float* Pm = Df + Lmax * 65;                   // [Lmax][LP]
  float* Sm = Pm + Lmax * LP;                   // dS, [Lmax][LP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = blockIdx.x / H, h = blockIdx.x % H;
  const int L = min(len[c], Lmax);
  for (int id = tid; id < Lmax * 64; id += 256) {
    const int j = id >> 6, d = id & 63;
    const int64_t row = (int64_t)c * Lmax + j;
    const bool ok = j < L;
    Qf[j * 65 + d] = ok ? ActIO<T>::ld(q + row * ld + h * 64 + d) : 0.f;
    Kf[j * 65 + d] = ok ? ActIO<T>::ld(k + row * ld + h * 64 + d) : 0.f;
    Vf[j * 65 + d] = ok ? ActIO<T>::ld(v + row * ld + h * 64 + d) : 0.f;
    Df[j * 65 + d] = ok ? ActIO<T>::ld(dout + row * lddo + h * 64 + d) : 0.f;
  }
  __syncthreads();
  const int j0 = min(lane, Lmax - 1), j1 = min(lane + 64, Lmax - 1);
  for (int r = wave; r < Lmax; r += 4) {
    const int64_t row = (int64_t)c * Lmax + r;
    if (r >= L) {                               // padding rows: zero gradients (the dX GEMM reads every row)
      ActIO<T>::st(dq + row * ldd + h * 64 + lane, 0.f);
      continue;
    }
    const int nk = r + 1;                       // causal, and r < L so every key j <= r is a real token
    const float qv = Qf[r * 65 + lane], dov = Df[r * 65 + lane];
    float s0 = 0.f, s1 = 0.f, dp0 = 0.f, dp1 = 0.f;
#pragma unroll 8
    for (int d = 0; d < 64; ++d) {
      const float qd = bcast(qv, d), dd = bcast(dov, d);
      s0 = fmaf(qd, Kf[j0 * 65 + d], s0);
      s1 = fmaf(qd, Kf[j1 * 65 + d], s1);
      dp0 = fmaf(dd, Vf[j0 * 65 + d], dp0);
      dp1 = fmaf(dd, Vf[j1 * 65 + d], dp1);
    }
    s0 = lane < nk ? s0 * scale : -INFINITY;
    s1 = lane + 64 < nk ? s1 * scale : -INFINITY;
    const float m = wave_max(fmaxf(s0, s1));
    float p0 = expf(s0 - m), p1 = expf(s1 - m);
    const float inv = 1.0f / wave_sum(p0 + p1);
    p0 *= inv; p1 *= inv;
    dp0 = lane < nk ? dp0 : 0.f;
    dp1 = lane + 64 < nk ? dp1 : 0.f;
    const float delta = wave_sum(p0 * dp0 + p1 * dp1);
    const float ds0 = p0 * (dp0 - delta), ds1 = p1 * (dp1 - delta);
    if (lane < Lmax) { Pm[r * LP + lane] = p0; Sm[r * LP + lane] = ds0; }
    if (lane + 64 < Lmax) { Pm[r * LP + lane + 64] = p1; Sm[r * LP + lane + 64] = ds1; }
    float acc = 0.f;
    for (int j = 0; j < nk; ++j) {
      const float dj = j < 64 ? bcast(ds0, j) : bcast(ds1, j - 64);
      acc = fmaf(dj, Kf[j * 65 + lane], acc);
    }
    ActIO<T>::st(dq + row * ldd + h * 64 + lane, acc * scale);
  }
  __syncthreads();
  for (int j = wave; j < Lmax; j += 4) {
    const int64_t row = (int64_t)c * Lmax + j;
    float ak = 0.f, av = 0.f;
    for (int r = j; r < L; ++r) {               // (empty for padding keys j >= L: zeros)
      ak = fmaf(Sm[r * LP + j], Qf[r * 65 + lane], ak);
      av = fmaf(Pm[r * LP + j], Df[r * 65 + lane], av);
    }
    ActIO<T>::st(dk + row * ldd + h * 64 + lane, ak * scale);
    ActIO<T>::st(dv + row * ldd + h * 64 + lane, av);
  }
}

template <typename T>
int launch_dense(const void* q, const void* k, const void* v, int64_t ld, const void* dout, int64_t lddo, void* dq,
                 void* dk, void* dv, int64_t ldd, const int32_t* len, int n_cls, int Lmax, int H, float scale,
                 hipStream_t s) {
  static rpo_lds_mask_t lds_ok{0};
  auto kern = text_attn_bwd_dense_kernel<T>;
  const int bytes = (4 * Lmax * 65 + 2 * Lmax * (Lmax + 1)) * 4;
  if (int rc = rpo_allow_lds(reinterpret_cast<const void*>(kern), (4 * 80 * 65 + 2 * 80 * 81) * 4, &lds_ok)) return rc;
  hipLaunchKernelGGL(kern, dim3(n_cls * H), dim3(256), bytes, s, static_cast<const T*>(q), static_cast<const T*>(k),
                     static_cast<const T*>(v), ld, static_cast<const T*>(dout), lddo, static_cast<T*>(dq),
                     static_cast<T*>(dk), static_cast<T*>(dv), ldd, len, Lmax, H, scale);
  return rpo_launch_status();
}

#ifndef RPO_TEXT_ROWS_PER_WG
#define RPO_TEXT_ROWS_PER_WG 4          // query rows per workgroup (one per wave and pass); A/B: tools/build_variant.sh
#endif
template <typename T, bool BWD>
int launch(const void* q, int64_t ldq, const void* kc, const void* vc, int64_t ldkv, const void* da,
           int64_t ldda, void* out, int64_t ldo, const int32_t* len, int n_cls, int rows, int Lmax, int H,
           int causal, float scale, hipStream_t s) {
  static rpo_lds_mask_t lds_ok{0};
  auto kern = text_attn_kernel<T, BWD>;
  const int bytes = 2 * Lmax * 65 * 4;
  if (int rc = rpo_allow_lds(reinterpret_cast<const void*>(kern), 2 * 128 * 65 * 4, &lds_ok)) return rc;
  hipLaunchKernelGGL(kern, dim3(n_cls * H, (rows + RPO_TEXT_ROWS_PER_WG - 1) / RPO_TEXT_ROWS_PER_WG), dim3(256), bytes, s, static_cast<const T*>(q), ldq,
                     static_cast<const T*>(kc), static_cast<const T*>(vc), ldkv, static_cast<const T*>(da), ldda,
                     static_cast<T*>(out), ldo, len, rows, Lmax, H, causal, scale);
  return rpo_launch_status();
}

}  // namespace

extern "C" int rpo_text_attn_fwd(const void* q, int64_t ldq, const void* kc, const void* vc, int64_t ldkv,
                                 void* out, int64_t ldo, int dtype, const int32_t* len, int n_cls, int rows,
                                 int Lmax, int H, int causal, float scale, void* stream) {
  if (!q || !kc || !vc || !out || !len || n_cls <= 0 || rows <= 0 || Lmax <= 0 || H <= 0) return RPO_E_BADARG;
  if (Lmax > 128) return RPO_E_SHAPE;
  {
    const int esz = dtype == RPO_F32 ? 4 : 2;
    if (!aligned16(kc) || !aligned16(vc) || (ldkv * esz) % 16 != 0) return RPO_E_ALIGN;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
#ifndef RPO_TEXT_ATTN_VALU
  if (!causal) {
    const int rc = rpo_text_attn_wave(0, q, ldq, kc, vc, ldkv, nullptr, 0, out, ldo, dtype, len, n_cls, rows, Lmax, H, scale, s);
    if (rc != RPO_E_SHAPE) return rc;
  }
#endif
  if (dtype == RPO_F16)
    return launch<f16_t, false>(q, ldq, kc, vc, ldkv, nullptr, 0, out, ldo, len, n_cls, rows, Lmax, H, causal, scale, s);
  if (dtype == RPO_BF16)
    return launch<bf16_t, false>(q, ldq, kc, vc, ldkv, nullptr, 0, out, ldo, len, n_cls, rows, Lmax, H, causal, scale, s);
  if (dtype == RPO_F32)
    return launch<float, false>(q, ldq, kc, vc, ldkv, nullptr, 0, out, ldo, len, n_cls, rows, Lmax, H, causal, scale, s);
  return RPO_E_DTYPE;
}

extern "C" int rpo_text_attn_bwd(const void* q, int64_t ldq, const void* kc, const void* vc, int64_t ldkv,
                                 const void* da, int64_t ldda, void* dq, int64_t lddq, int dtype,
                                 const int32_t* len, int n_cls, int rows, int Lmax, int H, float scale,
                                 void* stream) {
  if (!q || !kc || !vc || !da || !dq || !len || n_cls <= 0 || rows <= 0 || Lmax <= 0 || H <= 0) return RPO_E_BADARG;
  if (Lmax > 128) return RPO_E_SHAPE;
  {
    const int esz = dtype == RPO_F32 ? 4 : 2;
    if (!aligned16(kc) || !aligned16(vc) || (ldkv * esz) % 16 != 0) return RPO_E_ALIGN;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
#ifndef RPO_TEXT_ATTN_VALU
  {
    const int rc = rpo_text_attn_wave(1, q, ldq, kc, vc, ldkv, da, ldda, dq, lddq, dtype, len, n_cls, rows, Lmax, H, scale, s);
    if (rc != RPO_E_SHAPE) return rc;
  }
#endif
  if (dtype == RPO_F16)
    return launch<f16_t, true>(q, ldq, kc, vc, ldkv, da, ldda, dq, lddq, len, n_cls, rows, Lmax, H, 0, scale, s);
  if (dtype == RPO_BF16)
    return launch<bf16_t, true>(q, ldq, kc, vc, ldkv, da, ldda, dq, lddq, len, n_cls, rows, Lmax, H, 0, scale, s);
  if (dtype == RPO_F32)
    return launch<float, true>(q, ldq, kc, vc, ldkv, da, ldda, dq, lddq, len, n_cls, rows, Lmax, H, 0, scale, s);
  return RPO_E_DTYPE;
}

extern "C" int rpo_text_attn_bwd_dense(const void* q, const void* k, const void* v, int64_t ld, const void* d_out,
                                       int64_t lddo, void* dq, void* dk, void* dv, int64_t ldd, int dtype,
                                       const int32_t* len, int n_cls, int Lmax, int H, float scale, void* stream) {
  if (!q || !k || !v || !d_out || !dq || !dk || !dv || !len || n_cls <= 0 || Lmax <= 0 || H <= 0) return RPO_E_BADARG;
  if (Lmax > 80) return RPO_E_SHAPE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == RPO_F16) return launch_dense<f16_t>(q, k, v, ld, d_out, lddo, dq, dk, dv, ldd, len, n_cls, Lmax, H, scale, s);
  if (dtype == RPO_BF16) return launch_dense<bf16_t>(q, k, v, ld, d_out, lddo, dq, dk, dv, ldd, len, n_cls, Lmax, H, scale, s);
  if (dtype == RPO_F32) return launch_dense<float>(q, k, v, ld, d_out, lddo, dq, dk, dv, ldd, len, n_cls, Lmax, H, scale, s);
  return RPO_E_DTYPE;
}
