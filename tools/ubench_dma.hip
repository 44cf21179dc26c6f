// Micro-benchmark: sustained HBM/L2 -> LDS rate of global_load_lds_dwordx4 per CU, GEMM-shaped access
// (a 256-row panel of a row-major [M, K] bf16 matrix walked along k), as a function of the k-tile depth
// (64-B vs 128-B row pieces) and of the bytes kept in flight.  hipcc --offload-arch=gfx950 -O3 tools/ubench_dma.hip -o tools/build/ubench_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ROWB = bytes per row piece (64 / 128 / 256), PIECES = DMA instructions per wave kept in flight, WAVES per workgroup
template <int ROWB, int PIECES, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void dma_kernel(const char* A, int lda_bytes, int M, int K_bytes, int iters, float* sink, int mode) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int LPR = ROWB / 16;                 // lanes per row piece
  constexpr int RPI = 64 / LPR;                  // rows per instruction
  // the workgroup walks a 512-row panel (A tile + W tile of a 256x256 GEMM tile): row r of instruction j of wave w
  // mode 0: every workgroup its own panel (the matrix streams through each XCD's L2: MALL / HBM rate)
  // mode 1: all workgroups of an XCD share ONE panel (786 KB for K = 768: L2-resident -> the L2 -> LDS rate of a CU)
  const int m0 = mode == 0 ? (blockIdx.x * 256) % (M - 512) : (blockIdx.x % 8) * 512;
  const int ktiles = K_bytes / ROWB;
  int issued = 0;
  for (int it = 0; it < iters; ++it) {
    const int kt = it % ktiles;
    // one "tile" = 512 rows x ROWB bytes = 512 * ROWB / 1024 instructions, spread over the waves
    constexpr int IPT = 512 * ROWB / 1024;
    for (int j = wave; j < IPT; j += WAVES) {
      const int row = j * RPI + lane / LPR;
      const char* g = A + (size_t)(m0 + row) * lda_bytes + kt * ROWB + (lane % LPR) * 16;
      char* l = smem + ((issued % PIECES) * WAVES + wave) * 1024;
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
      ++issued;
      if (issued >= PIECES) wait_vmcnt<PIECES - 1>();
    }
  }
  wait_vmcnt<0>();
  __syncthreads();
  if (sink != nullptr && smem[threadIdx.x] == 123 && iters < 0) sink[0] = 1.f;
}

template <int ROWB, int PIECES, int WAVES>
void run(const char* A, int lda, int M, int Kb, float* sink, const char* tag, int mode = 0, int grid = 256) {
  auto k = dma_kernel<ROWB, PIECES, WAVES>;
  const int smem = PIECES * WAVES * 1024;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  const int iters = 256 * 64 / ROWB;             // 256 tiles of 64-B depth worth of bytes: 8 MiB per workgroup
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(WAVES * 64), smem, 0, A, lda, M, Kb, iters, sink, mode);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)grid * iters * 512.0 * ROWB;
  printf("%-28s mode %d grid %3d row %3d B  in flight %3d KiB/CU  waves %d: %7.1f us  %6.2f TB/s  = %5.1f GB/s per CU\n", tag, mode, grid, ROWB,
         PIECES * WAVES, WAVES, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / grid);
}

int main() {
  const int M = 7072 + 512, K = 768;             // bf16 [M, 768]: 11.6 MB, L2/MALL resident after the first pass
  char* A; float* sink;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&sink, 16);
  hipMemset(A, 1, (size_t)M * K * 2);
  const int lda = K * 2;
  run<64, 4, 4>(A, lda, M, lda, sink, "qkv A panel (K=768)");
  run<64, 8, 4>(A, lda, M, lda, sink, "qkv A panel (K=768)");
  run<64, 16, 4>(A, lda, M, lda, sink, "qkv A panel (K=768)");
  run<64, 24, 4>(A, lda, M, lda, sink, "qkv A panel (K=768)");
  run<64, 32, 4>(A, lda, M, lda, sink, "qkv A panel (K=768)");
  run<128, 4, 4>(A, lda, M, lda, sink, "qkv A panel (K=768)");
  run<128, 8, 4>(A, lda, M, lda, sink, "qkv A panel (K=768)");
  run<128, 16, 4>(A, lda, M, lda, sink, "qkv A panel (K=768)");
  run<128, 24, 4>(A, lda, M, lda, sink, "qkv A panel (K=768)");
  run<128, 32, 4>(A, lda, M, lda, sink, "qkv A panel (K=768)");
  run<256, 16, 4>(A, lda, M, lda, sink, "qkv A panel (K=768)");
  run<256, 32, 4>(A, lda, M, lda, sink, "qkv A panel (K=768)");
  run<64, 8, 8>(A, lda, M, lda, sink, "qkv A panel (K=768)");
  run<64, 16, 8>(A, lda, M, lda, sink, "qkv A panel (K=768)");
  run<128, 8, 8>(A, lda, M, lda, sink, "qkv A panel (K=768)");
  run<128, 16, 8>(A, lda, M, lda, sink, "qkv A panel (K=768)");
  for (int grid : {256, 64, 8}) {
    run<64, 8, 4>(A, lda, M, lda, sink, "L2-resident panel", 1, grid);
    run<64, 24, 4>(A, lda, M, lda, sink, "L2-resident panel", 1, grid);
    run<128, 8, 4>(A, lda, M, lda, sink, "L2-resident panel", 1, grid);
    run<128, 24, 4>(A, lda, M, lda, sink, "L2-resident panel", 1, grid);
    run<64, 16, 8>(A, lda, M, lda, sink, "L2-resident panel", 1, grid);
    run<128, 16, 8>(A, lda, M, lda, sink, "L2-resident panel", 1, grid);
  }
  run<128, 24, 4>(A, lda, M, lda, sink, "own panel, few CUs", 0, 64);
  run<128, 24, 4>(A, lda, M, lda, sink, "own panel, few CUs", 0, 8);
  // a larger matrix that does not fit L2 + MALL residency as easily: [7584, 3072] bf16 = 46.6 MB (c_proj's A operand)
  char* B; hipMalloc(&B, (size_t)M * 3072 * 2); hipMemset(B, 1, (size_t)M * 3072 * 2);
  run<64, 16, 4>(B, 6144, M, 6144, sink, "c_proj A panel (K=3072)");
  run<64, 32, 4>(B, 6144, M, 6144, sink, "c_proj A panel (K=3072)");
  run<128, 16, 4>(B, 6144, M, 6144, sink, "c_proj A panel (K=3072)");
  run<128, 32, 4>(B, 6144, M, 6144, sink, "c_proj A panel (K=3072)");
  return 0;
}
