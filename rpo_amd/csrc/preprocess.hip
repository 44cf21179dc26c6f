// On-device input transforms: random_resized_crop / resize + center-crop, random_flip, normalize
// (configs/trainers/RPO/main_K24.yaml:8-13) for a batch of decoded uint8 RGB images of different sizes.
//
// The arithmetic reproduced is Pillow's 8-bit bicubic resample (the reference's transforms end in
// `PIL.Image.resize(..., BICUBIC)` through torchvision): separable, horizontal pass then vertical pass, both with
// uint8 results; coefficients computed in DOUBLE, normalised, rounded to 22-bit fixed point; support =
// 2 * max(scale, 1).  Results are bit-identical to Pillow (tests/test_gpu_preprocess.py), so this file is compiled
// with -ffp-contract=off: an FMA in the coefficient math would change the rounding.
//
// Three kernels per batch, all HBM/L2-bound byte work (no MFMA):
//   coeff_kernel    per (image, axis): bounds + int32 coefficients of the 224 output positions -> workspace
//   horiz_kernel    per (image, source row): 224 x 3 uint8 outputs from the cropped row       -> workspace temp
//   vert_norm_kernel per (image, output row): vertical pass, flip, ToTensor (/255) and Normalize -> fp32 NCHW
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

struct Layout {           // workspace carve-up, identical on host and device
  int B, size, max_rows, kmax;
  __host__ __device__ size_t bounds_off(int b, int axis) const { return ((size_t)(b * 2 + axis) * size) * 2 * sizeof(int); }
  __host__ __device__ size_t bounds_bytes() const { return (size_t)B * 2 * size * 2 * sizeof(int); }
  __host__ __device__ size_t coef_off(int b, int axis) const {
    return bounds_bytes() + ((size_t)(b * 2 + axis) * size) * kmax * sizeof(int);
  }
  __host__ __device__ size_t coef_bytes() const { return (size_t)B * 2 * size * kmax * sizeof(int); }
  __host__ __device__ size_t temp_off(int b) const {
    return bounds_bytes() + coef_bytes() + (size_t)b * max_rows * size * 3;
  }
  __host__ __device__ size_t total() const { return temp_off(B) + 16; }
};

__device__ __forceinline__ double bicubic_filter(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

// axis 0 = horizontal (crop_w -> resize_w, window win_x), axis 1 = vertical
__global__ __launch_bounds__(256) void coeff_kernel(const rpo_image_desc* __restrict__ desc, char* ws, Layout L) {
  const int b = blockIdx.x, axis = blockIdx.y;
  const rpo_image_desc d = desc[b];
  const int in_size = axis == 0 ? d.crop_w : d.crop_h;
  const int out_full = axis == 0 ? d.resize_w : d.resize_h;
  const int win = axis == 0 ? d.win_x : d.win_y;
  int* bounds = reinterpret_cast<int*>(ws + L.bounds_off(b, axis));
  int* coef = reinterpret_cast<int*>(ws + L.coef_off(b, axis));
  for (int o = threadIdx.x; o < L.size; o += blockDim.x) {
    int* k = coef + (size_t)o * L.kmax;
    const int xx = o + win;
    if (out_full == in_size) {           // Pillow skips the pass: identity tap reproduces the copy exactly
      bounds[2 * o] = xx; bounds[2 * o + 1] = 1;
      k[0] = 1 << PRECISION_BITS;
      continue;
    }
    // Resample.c:precompute_coeffs with box (0, in_size): in0 = 0.0f, in1 = (float)in_size
    const double scale = (double)((float)in_size - 0.0f) / out_full;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const double ss = 1.0 / filterscale;
    const double center = 0.0 + (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    if (xmax > L.kmax) xmax = L.kmax;    // host validated kmax; stay memory-safe regardless
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += bicubic_filter((x + xmin - center + 0.5) * ss);
    for (int x = 0; x < xmax; ++x) {
      double w = bicubic_filter((x + xmin - center + 0.5) * ss);
      if (ww != 0.0) w /= ww;
      k[x] = w < 0 ? (int)(-0.5 + w * (1 << PRECISION_BITS)) : (int)(0.5 + w * (1 << PRECISION_BITS));
    }
    bounds[2 * o] = xmin; bounds[2 * o + 1] = xmax;
  }
}

__device__ __forceinline__ int clip8(int v) {
  v >>= PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// grid (row chunk, image); each block walks ROWS source rows of the rows the vertical pass will read
constexpr int HROWS = 4;
__global__ __launch_bounds__(256) void horiz_kernel(const uint8_t* __restrict__ src, const rpo_image_desc* __restrict__ desc,
                                                    char* ws, Layout L) {
  const int b = blockIdx.y;
  const rpo_image_desc d = desc[b];
  const int* vb = reinterpret_cast<const int*>(ws + L.bounds_off(b, 1));
  const int first = vb[0], last = vb[2 * (L.size - 1)] + vb[2 * (L.size - 1) + 1];   // rows of the crop needed
  const int nrows = min(last - first, L.max_rows);
  const int* hb = reinterpret_cast<const int*>(ws + L.bounds_off(b, 0));
  const int* hk = reinterpret_cast<const int*>(ws + L.coef_off(b, 0));
  uint8_t* temp = reinterpret_cast<uint8_t*>(ws + L.temp_off(b));
  const int S = L.size;
  for (int r = blockIdx.x * HROWS; r < min(nrows, (int)(blockIdx.x + 1) * HROWS); ++r) {
    const uint8_t* row = src + d.src_offset + ((size_t)(d.crop_y + first + r) * d.width + d.crop_x) * 3;
    for (int o = threadIdx.x; o < S; o += blockDim.x) {
      const int xmin = hb[2 * o], xmax = hb[2 * o + 1];
      const int* k = hk + (size_t)o * L.kmax;
      int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
      const uint8_t* p = row + (size_t)xmin * 3;
      for (int x = 0; x < xmax; ++x) {
        const int kv = k[x];
        s0 += p[3 * x] * kv; s1 += p[3 * x + 1] * kv; s2 += p[3 * x + 2] * kv;
      }
      uint8_t* t = temp + ((size_t)r * S + o) * 3;
      t[0] = (uint8_t)clip8(s0); t[1] = (uint8_t)clip8(s1); t[2] = (uint8_t)clip8(s2);
    }
  }
}

// grid (output row, image); thread idx -> (c, ox): planar stores are contiguous in ox
__global__ __launch_bounds__(256) void vert_norm_kernel(const rpo_image_desc* __restrict__ desc, const char* ws, Layout L,
                                                        float m0, float m1, float m2, float sd0, float sd1, float sd2,
                                                        float* out) {
  const int oy = blockIdx.x, b = blockIdx.y;
  const int flip = desc[b].flip;
  const int S = L.size;
  const int* vb = reinterpret_cast<const int*>(ws + L.bounds_off(b, 1));
  const int* vk = reinterpret_cast<const int*>(ws + L.coef_off(b, 1)) + (size_t)oy * L.kmax;
  const uint8_t* temp = reinterpret_cast<const uint8_t*>(ws + L.temp_off(b));
  const int first = vb[0];
  const int ymin = vb[2 * oy] - first, ymax = vb[2 * oy + 1];
  for (int idx = threadIdx.x; idx < 3 * S; idx += blockDim.x) {
    const int c = idx / S, ox = idx - c * S;
    int s = 1 << (PRECISION_BITS - 1);
    const uint8_t* p = temp + ((size_t)ymin * S + ox) * 3 + c;
    for (int y = 0; y < ymax; ++y) s += p[(size_t)y * S * 3] * vk[y];
    const float v = (float)clip8(s) / 255.0f;                     // ToTensor
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? sd0 : (c == 1 ? sd1 : sd2);
    const int xo = flip ? S - 1 - ox : ox;
    out[(((size_t)b * 3 + c) * S + oy) * S + xo] = (v - mean) / sd;   // Normalize: sub then IEEE divide
  }
}

}  // namespace

extern "C" size_t rpo_preprocess_workspace_bytes(int B, int size, int max_rows, int kmax) {
  if (B <= 0 || size <= 0 || max_rows <= 0 || kmax <= 0) return 0;
  const Layout L{B, size, max_rows, kmax};
  return L.total();
}

extern "C" int rpo_preprocess_ksize(int in_size, int out_size) {
  if (in_size <= 0 || out_size <= 0) return RPO_E_BADARG;
  if (in_size == out_size) return 1;
  const double scale = (double)((float)in_size) / out_size;
  const double support = 2.0 * (scale < 1.0 ? 1.0 : scale);
  return (int)ceil(support) * 2 + 1;
}

extern "C" int rpo_preprocess_batch(const uint8_t* src, int64_t src_bytes, const rpo_image_desc* desc_host,
                                    const rpo_image_desc* desc_dev, int B, int size, int max_rows, int kmax,
                                    const float* mean3, const float* std3, float* out, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  if (!src || !desc_host || !desc_dev || !mean3 || !std3 || !out || !workspace || B <= 0 || size <= 0)
    return RPO_E_BADARG;
  const Layout L{B, size, max_rows, kmax};
  if (workspace_bytes < L.total()) return RPO_E_WORKSPACE;
  if (reinterpret_cast<uintptr_t>(workspace) % 16) return RPO_E_ALIGN;
  for (int b = 0; b < B; ++b) {        // the descriptors are the only untrusted input: validate on the host copy
    const rpo_image_desc& d = desc_host[b];
    if (d.width <= 0 || d.height <= 0 || d.crop_w <= 0 || d.crop_h <= 0 || d.crop_x < 0 || d.crop_y < 0 ||
        d.crop_x + d.crop_w > d.width || d.crop_y + d.crop_h > d.height)
      return RPO_E_SHAPE;
    if (d.resize_w < size || d.resize_h < size || d.win_x < 0 || d.win_y < 0 || d.win_x + size > d.resize_w ||
        d.win_y + size > d.resize_h)
      return RPO_E_SHAPE;
    if (d.src_offset < 0 || d.src_offset + (int64_t)d.width * d.height * 3 > src_bytes) return RPO_E_SHAPE;
    if (rpo_preprocess_ksize(d.crop_w, d.resize_w) > kmax || rpo_preprocess_ksize(d.crop_h, d.resize_h) > kmax)
      return RPO_E_WORKSPACE;
    if (d.crop_h > max_rows) return RPO_E_WORKSPACE;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(workspace);
  hipLaunchKernelGGL(coeff_kernel, dim3(B, 2), dim3(256), 0, s, desc_dev, ws, L);
  hipLaunchKernelGGL(horiz_kernel, dim3((max_rows + HROWS - 1) / HROWS, B), dim3(256), 0, s, src, desc_dev, ws, L);
  hipLaunchKernelGGL(vert_norm_kernel, dim3(size, B), dim3(256), 0, s, desc_dev, ws, L, mean3[0], mean3[1],
                     mean3[2], std3[0], std3[1], std3[2], out);
  return rpo_launch_status();
}
