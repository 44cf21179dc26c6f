// Shared device helpers for librpo_hip.so (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>

#include "../../include/rpo_amd.h"
// Argument structs of the experiments are also what some default kernels take internally, so the DECLARATIONS are always
// visible to the library's own sources; the entry points are DEFINED only with -DRPO_EXPERIMENTAL.
#include "../../include/rpo_amd_experimental.h"

typedef uint16_t bf16_t;  // raw bf16 bits
// raw IEEE binary16 bits (TRAINER.RPO.PREC = fp16 / amp, trainers/rpo.py:247-249): a distinct type, so that templates
// can tell the two 16-bit storage formats apart
struct f16_t { uint16_t v; };
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // also the 128-bit container of 8 halves of either format
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define RPO_WAVE 64

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// f32 -> bf16, round-to-nearest-even, NaN-preserving: a plain fptrunc to __bf16, which hipcc lowers to
// gfx950's native v_cvt_pk_bf16_f32 (a hand-rolled integer rounding with a NaN branch costs an
// exec-mask branch per value in the GEMM epilogues).
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.0f) & 0xffffu); }

// ---- the two 16-bit storage formats behind one set of helpers (T = bf16_t or f16_t) ------------------------------
// Fragment registers are typed bf16x8_t for both (a 128-bit container); only conversion and the MFMA opcode differ.
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi);          // f32 x2 -> 16-bit x2, RNE
template <> __device__ __forceinline__ uint32_t pack2<bf16_t>(float lo, float hi) { return pack_bf16x2(lo, hi); }
template <> __device__ __forceinline__ uint32_t pack2<f16_t>(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));                    // v_cvt_f16_f32, RNE
}
template <typename T> __device__ __forceinline__ float unpack1(uint16_t bits);
template <> __device__ __forceinline__ float unpack1<bf16_t>(uint16_t bits) { return bf16_to_f32(bits); }
template <> __device__ __forceinline__ float unpack1<f16_t>(uint16_t bits) { return (float)__builtin_bit_cast(_Float16, bits); }
template <typename T> struct One16;                                                          // bit pattern of 1.0
template <> struct One16<bf16_t> { static constexpr uint32_t lo = 0x3F80u, hi = 0x3F800000u; };
template <> struct One16<f16_t> { static constexpr uint32_t lo = 0x3C00u, hi = 0x3C000000u; };
template <typename T> __device__ __forceinline__ f32x16_t mfma16(bf16x8_t a, bf16x8_t b, f32x16_t c);
template <> __device__ __forceinline__ f32x16_t mfma16<bf16_t>(bf16x8_t a, bf16x8_t b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16_t mfma16<f16_t>(bf16x8_t a, bf16x8_t b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

template <typename T> struct ActIO;
template <> struct ActIO<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ void st4(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
  }
};
template <> struct ActIO<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
  static __device__ __forceinline__ void st4(bf16_t* p, float a, float b, float c, float d) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
  }
};

template <> struct ActIO<f16_t> {
  static __device__ __forceinline__ float ld(const f16_t* p) { return unpack1<f16_t>(p->v); }
  static __device__ __forceinline__ void st(f16_t* p, float v) { p->v = (uint16_t)(pack2<f16_t>(v, 0.0f) & 0xffffu); }
  static __device__ __forceinline__ void st4(f16_t* p, float a, float b, float c, float d) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack2<f16_t>(a, b), pack2<f16_t>(c, d));
  }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over the 16-lane DPP row a lane belongs to, result in every lane: four rotate-and-add steps that hipcc folds into
// v_add_f32 ... row_ror (one VALU each; __shfl_xor would be a ds_bpermute round trip per step)
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// QuickGELU (clip/model.py:162-164) and its derivative.  sigmoid through the hardware transcendentals
// (v_exp_f32 + v_rcp_f32, ~1 ulp each): libm's expf plus an IEEE division cost ~40 VALU instructions per
// element, which made the c_fc epilogue ~20 % of that GEMM (s_memtime: 8-12 k cycles per 128x128 tile).
#define RPO_QG 1.702f
#define RPO_QG_L2 (-2.4554669595930156f)           // -1.702 * log2(e): sigmoid(1.702 u) = 1 / (1 + 2^(RPO_QG_L2 * u))
__device__ __forceinline__ float qg_sigmoid(float u) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(RPO_QG_L2 * u));
}
__device__ __forceinline__ float quick_gelu(float u) { return u * qg_sigmoid(u); }
__device__ __forceinline__ float quick_gelu_grad(float u) {
  float s = qg_sigmoid(u);
  return s * fmaf(RPO_QG * u, 1.0f - s, 1.0f);
}
// Four at a time, the full-rate part as packed fp32 (v_pk_mul_f32 / v_pk_add_f32): the epilogue of the 224x384 GEMM is
// VALU-bound (per element two quarter-rate transcendentals + the rest; disassembly: hipcc left the scaling multiplies
// and the +1 unpacked).  Same operations in the same order as quick_gelu, i.e. the same bits.
__device__ __forceinline__ float4 quick_gelu4(float4 v) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  const f32x2_t a = {v.x, v.y}, b = {v.z, v.w};
  const f32x2_t za = a * RPO_QG_L2, zb = b * RPO_QG_L2;
  f32x2_t ea = {__builtin_amdgcn_exp2f(za.x), __builtin_amdgcn_exp2f(za.y)};
  f32x2_t eb = {__builtin_amdgcn_exp2f(zb.x), __builtin_amdgcn_exp2f(zb.y)};
  ea = ea + 1.0f; eb = eb + 1.0f;
  const f32x2_t ra = {__builtin_amdgcn_rcpf(ea.x), __builtin_amdgcn_rcpf(ea.y)};
  const f32x2_t rb = {__builtin_amdgcn_rcpf(eb.x), __builtin_amdgcn_rcpf(eb.y)};
  const f32x2_t oa = a * ra, ob = b * rb;
  return make_float4(oa.x, oa.y, ob.x, ob.y);
}

// QuickGELU and its derivative from ONE sigmoid: o = the bits of quick_gelu4, d = the bits of quick_gelu_grad (same
// operations in the same order: s = rcp(1 + exp2(c u)), o = u s, d = s * fma(1.702 u, 1 - s, 1)).
__device__ __forceinline__ void quick_gelu4_du(const float4 v, float4& o, float4& d) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  const f32x2_t a = {v.x, v.y}, b = {v.z, v.w};
  const f32x2_t za = a * RPO_QG_L2, zb = b * RPO_QG_L2;
  f32x2_t ea = {__builtin_amdgcn_exp2f(za.x), __builtin_amdgcn_exp2f(za.y)};
  f32x2_t eb = {__builtin_amdgcn_exp2f(zb.x), __builtin_amdgcn_exp2f(zb.y)};
  ea = ea + 1.0f; eb = eb + 1.0f;
  const f32x2_t ra = {__builtin_amdgcn_rcpf(ea.x), __builtin_amdgcn_rcpf(ea.y)};
  const f32x2_t rb = {__builtin_amdgcn_rcpf(eb.x), __builtin_amdgcn_rcpf(eb.y)};
  const f32x2_t oa = a * ra, ob = b * rb;
  const f32x2_t qa = a * RPO_QG, qb = b * RPO_QG;
  const f32x2_t ta = 1.0f - ra, tb = 1.0f - rb;
  const f32x2_t wa = {fmaf(qa.x, ta.x, 1.0f), fmaf(qa.y, ta.y, 1.0f)};
  const f32x2_t wb = {fmaf(qb.x, tb.x, 1.0f), fmaf(qb.y, tb.y, 1.0f)};
  const f32x2_t da = ra * wa, db = rb * wb;
  o = make_float4(oa.x, oa.y, ob.x, ob.y);
  d = make_float4(da.x, da.y, db.x, db.y);
}

// 16-byte store of a finished output tile.  RPO_NT_STORE (experiment, off): non-temporal, i.e. streamed past the L2's
// write-back lines so that the end-of-kernel flush has nothing left to write.  Measured: the producers do not get
// shorter and the consumer of the tile then reads it from HBM (attention 13.7 -> 16.6 us after a streamed qkv): +0.7 %
// step time.  Plain stores are what keeps a layer's hand-offs in L2 / MALL.
// RPO_SC1_STORE (experiment): write-THROUGH stores (sc1) -- the line stays valid in this XCD's L2 for a consumer on the
// same XCD, but is not left dirty, so the end-of-kernel release has nothing to write back.
template <typename V>
__device__ __forceinline__ void store_out16(V* dst, const V v) {
#if defined(RPO_SC1_STORE)
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_wt;
  static_assert(sizeof(V) == 16, "16-byte stores only");
  asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(__builtin_bit_cast(u32x4_wt, v)) : "memory");
#elif defined(RPO_NT_STORE)
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_nt;
  static_assert(sizeof(V) == 16, "16-byte stores only");
  __builtin_nontemporal_store(__builtin_bit_cast(u32x4_nt, v), reinterpret_cast<u32x4_nt*>(dst));
#else
  *dst = v;
#endif
}

__device__ __forceinline__ void store_out8(void* dst, const uint2 v) {
#if defined(RPO_SC1_STORE)
  typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_wt;
  asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(dst), "v"(__builtin_bit_cast(u32x2_wt, v)) : "memory");
#else
  *reinterpret_cast<uint2*>(dst) = v;
#endif
}

// Kernels that need more than 64 KiB of dynamic LDS must raise the limit once per (kernel, device).  `mask` is the
// caller's function-local static: bit d = done on device d (devices >= 64 re-set it on every launch).
// Kernels that need more than 64 KiB of dynamic LDS must be told so once per device.  The per-kernel "already done on
// device d" bits are the library's only mutable state: an atomic cache of an idempotent driver call (a lost race repeats
// the call, nothing else), not something a caller can observe.
using rpo_lds_mask_t = std::atomic<unsigned long long>;
static inline int rpo_allow_lds(const void* kern, int bytes, rpo_lds_mask_t* mask) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  const unsigned long long bit = dev < 64 ? 1ull << dev : 0ull;
  if (bit && (mask->load(std::memory_order_relaxed) & bit)) return 0;
  e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  mask->fetch_or(bit, std::memory_order_relaxed);
  return 0;
}

static inline int rpo_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// CUs of the current device (the devices of a node are identical: asked once)
static inline int rpo_cu_count() {
  static std::atomic<int> cached{0};
  int n = cached.load(std::memory_order_relaxed);
  if (n > 0) return n;
  int dev = 0;
  n = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
    n = 256;
  cached.store(n, std::memory_order_relaxed);
  return n;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// attn_image.hip: the prompt-row attention of the text tower as one wave per (class, head) (16-bit modes); RPO_E_SHAPE
// where it does not apply.  Called by rpo_text_attn_fwd / rpo_text_attn_bwd (attn_text.hip) ahead of their VALU kernel.
int rpo_text_attn_wave(int bwd, const void* q, int64_t ldq, const void* kc, const void* vc, int64_t ldkv, const void* da,
                       int64_t ldda, void* out, int64_t ldo, int dtype, const int32_t* len, int n_cls, int rows, int Lmax,
                       int H, float scale, hipStream_t s);

// Optional in-kernel timeline (debug build with -DRPO_TIMELINE; tools/gemm_timeline.py, tools/attn_timeline.py):
// thread 0 of a few workgroups stamps s_memtime at phase boundaries into a global buffer.
#ifdef RPO_TIMELINE
extern __device__ unsigned long long* g_timeline;
#define RPO_STAMP(slot)                                                                               \
  do {                                                                                                \
    if (g_timeline != nullptr && threadIdx.x == 0 && (blockIdx.x % 97) == 0 && blockIdx.x / 97 < 8)   \
      g_timeline[(blockIdx.x / 97) * 64 + (slot)] = __builtin_amdgcn_s_memtime();                     \
  } while (0)
// the constant 100 MHz counter next to s_memtime: (delta s_memtime) / (delta s_memrealtime) * 100 MHz = the shader clock
#define RPO_STAMP_RT(slot)                                                                            \
  do {                                                                                                \
    if (g_timeline != nullptr && threadIdx.x == 0 && (blockIdx.x % 97) == 0 && blockIdx.x / 97 < 8)   \
      g_timeline[(blockIdx.x / 97) * 64 + (slot)] = __builtin_amdgcn_s_memrealtime();                 \
  } while (0)
#else
#define RPO_STAMP(slot) do { } while (0)
#define RPO_STAMP_RT(slot) do { } while (0)
#endif
