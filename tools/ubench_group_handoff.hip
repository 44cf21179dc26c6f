// Gate micro-benchmark for the persistent prompt-row chain (round-3 review, item 1): what does ONE all-to-all hand-off
// cost inside a launch when the workgroups that exchange data are the 32 workgroups of one XCD-sized group
// (8 groups x 32 workgroups = 256, one per CU), against the kernel boundary it would replace?
//
// Workload per phase and workgroup (the shape of a chain stage of the image tower's backward at B = 32, K = 24:
// 96 back-propagated rows per group): write one 96 x 96 16-bit tile (18 KB) of the group's [96, 3072] panel, hop,
// read a [96, 768] slice of the panel the group's workgroups wrote in the previous phase (147 KB, 16 B per lane), checking
// EVERY word against what the producer of that phase must have written (two alternating panels: the same addresses come
// back every other phase, so a consumer L1 that keeps a line sees stale data; one workgroup per phase is delayed).
//
// Modes
//   0 nosync     no hop at all (wrong results by construction): the store + load work of a phase alone
//   1 sc1        guide recipe R1: write-through (sc1) payload stores, every wave drains, ONE lane arrives on the group's
//                counter (relaxed, agent scope); one wave polls relaxed, ONE agent acquire (buffer_inv sc1), plain loads
//   2 release    plain payload stores, lane 0: agent release fence (buffer_wbl2 sc1) + asm vmcnt(0), arrive, poll, acquire
//   3 xcd        plain stores + drain, arrive with an L2-scope atomic (no sc1), poll with sc1 loads, payload read with sc1
//                loads (bypass L1, served by the XCD's L2): valid ONLY when the group's workgroups share an XCD, which the
//                run reports from HW_REG_XCC_ID
//   4 xcd+inv    as 3 but payload read with plain loads behind one buffer_inv sc1 (L1 invalidate)
//   launches     the same phases as one kernel launch each (stream order = the hop)
//
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_group_handoff tools/ubench_group_handoff.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int GROUPS = 8, MEMBERS = 32, ROWS = 96, PANEL_COLS = 3072, TILE_COLS = 96, SLICE_COLS = 768;
constexpr int ROW_BYTES = PANEL_COLS * 2;                 // 6144
constexpr int PANEL_BYTES = ROWS * ROW_BYTES;             // 589 824
constexpr unsigned SPIN_LIMIT = 1u << 20;

typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ uint32_t word_of(int phase, int group, int row, int w) {   // w = 32-bit word index in the row
  return (uint32_t)(phase * 2654435761u) ^ (uint32_t)(group * 40503u + 17) ^ (uint32_t)(row * 1536 + w) * 2246822519u;
}

__device__ __forceinline__ void store16(char* p, u32x4 v, bool sc1) {
  if (sc1) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  else *reinterpret_cast<u32x4*>(p) = v;
}
__device__ __forceinline__ u32x4 load16_sc1(const char* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}

struct Params {
  char* panels;          // [GROUPS][2][PANEL_BYTES]
  unsigned* counters;    // [GROUPS] (64-B apart), zeroed before every launch
  unsigned* errors;      // [0] mismatching words, [1] spin give-ups
  unsigned* xcc;         // [256] HW_REG_XCC_ID of every workgroup
  int phases, mode, delay;
};

__device__ __forceinline__ void write_tile(const Params& p, int phase, int group, int slot, bool sc1) {
  char* panel = p.panels + ((size_t)group * 2 + (phase & 1)) * PANEL_BYTES;
  // 96 rows x 12 chunks of 16 B
  for (int c = threadIdx.x; c < ROWS * 12; c += blockDim.x) {
    const int row = c / 12, ch = c % 12;
    const int w0 = slot * (TILE_COLS / 2) + ch * 4;
    u32x4 v = {word_of(phase, group, row, w0), word_of(phase, group, row, w0 + 1), word_of(phase, group, row, w0 + 2),
               word_of(phase, group, row, w0 + 3)};
    store16(panel + (size_t)row * ROW_BYTES + (size_t)w0 * 4, v, sc1);
  }
}

template <bool SC1LOAD>
__device__ __forceinline__ unsigned read_slice(const Params& p, int phase, int group, int slot) {
  const char* panel = p.panels + ((size_t)group * 2 + (phase & 1)) * PANEL_BYTES;
  const int slice = slot >> 3;                      // 4 slices of 768 columns = 8 producers each
  unsigned bad = 0;
  // 96 rows x 96 chunks of 16 B; 9 batches of 4 loads in flight per thread
  for (int c0 = threadIdx.x; c0 < ROWS * 96; c0 += blockDim.x * 4) {
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + u * blockDim.x;
      const int row = c / 96, ch = c % 96;
      const char* src = panel + (size_t)row * ROW_BYTES + (size_t)(slice * (SLICE_COLS / 2) + ch * 4) * 4;
      if (SC1LOAD) v[u] = load16_sc1(src); else v[u] = *reinterpret_cast<const u32x4*>(src);
    }
    if (SC1LOAD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + u * blockDim.x;
      const int row = c / 96, ch = c % 96;
      const int w0 = slice * (SLICE_COLS / 2) + ch * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) bad += v[u][k] != word_of(phase, group, row, w0 + k);
    }
  }
  return bad;
}

template <int MODE>
__global__ __launch_bounds__(256) void persistent_kernel(const Params p) {
  const int group = blockIdx.x % GROUPS, slot = blockIdx.x / GROUPS;
  if (threadIdx.x == 0) p.xcc[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID[3:0]
  gu32* cnt = (gu32*)(p.counters + group * 16);
  __shared__ int give_up;
  if (threadIdx.x == 0) give_up = 0;
  __syncthreads();
  unsigned bad = 0;
  for (int ph = 0; ph < p.phases; ++ph) {
    if (ph > 0) {
      if (MODE == 0 || MODE == 1 || MODE == 2 || MODE == 4) bad += read_slice<false>(p, ph - 1, group, slot);
      else bad += read_slice<true>(p, ph - 1, group, slot);
    }
    if (p.delay && slot == (ph * 7 + 3) % MEMBERS) {                 // uneven load: one late producer per phase
      for (int i = 0; i < p.delay; ++i) __builtin_amdgcn_s_sleep(64);
    }
    write_tile(p, ph, group, slot, MODE == 1);
    if (MODE == 0) continue;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // every storing wave drains
    __syncthreads();
    if (threadIdx.x == 0) {
      if (MODE == 2) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (MODE == 3 || MODE == 4) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)MEMBERS * (ph + 1);
      unsigned spins = 0;
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > SPIN_LIMIT) { give_up = 1; break; }
      }
      if (MODE == 1 || MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      if (MODE == 4) asm volatile("buffer_inv sc1" ::: "memory");
    }
    __syncthreads();
    if (give_up) break;
  }
  bad += 0;
  // block-level sum of the mismatch counts
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(p.errors, bad);
  if (threadIdx.x == 0 && give_up) atomicAdd(p.errors + 1, 1u);
}

// one phase per launch: read what the previous launch wrote, then write this phase's tile
__global__ __launch_bounds__(256) void phase_kernel(const Params p, int ph) {
  const int group = blockIdx.x % GROUPS, slot = blockIdx.x / GROUPS;
  unsigned bad = 0;
  if (ph > 0) bad = read_slice<false>(p, ph - 1, group, slot);
  write_tile(p, ph, group, slot, false);
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(p.errors, bad);
}

int main(int argc, char** argv) {
  const int phases = argc > 1 ? atoi(argv[1]) : 72;
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  Params p{};
  p.phases = phases;
  CHECK(hipMalloc(&p.panels, (size_t)GROUPS * 2 * PANEL_BYTES));
  CHECK(hipMalloc(&p.counters, GROUPS * 64));
  CHECK(hipMalloc(&p.errors, 8));
  CHECK(hipMalloc(&p.xcc, 256 * 4));
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const char* names[] = {"nosync", "sc1 (R1)", "release", "xcd", "xcd+inv"};
  printf("phases %d, reps %d; per phase and workgroup: 18 KB written, 147 KB read and checked\n", phases, reps);
  for (int delay = 0; delay <= 8; delay += 8) {
    p.delay = delay;
    for (int mode = 0; mode < 5; ++mode) {
      p.mode = mode;
      float best = 1e30f, sum = 0.f;
      unsigned err[2] = {0, 0};
      for (int r = 0; r < reps + 2; ++r) {
        CHECK(hipMemsetAsync(p.counters, 0, GROUPS * 64, s));
        CHECK(hipMemsetAsync(p.errors, 0, 8, s));
        CHECK(hipMemsetAsync(p.panels, 0xff, (size_t)GROUPS * 2 * PANEL_BYTES, s));
        CHECK(hipEventRecord(e0, s));
        switch (mode) {
          case 0: hipLaunchKernelGGL(persistent_kernel<0>, dim3(256), dim3(256), 0, s, p); break;
          case 1: hipLaunchKernelGGL(persistent_kernel<1>, dim3(256), dim3(256), 0, s, p); break;
          case 2: hipLaunchKernelGGL(persistent_kernel<2>, dim3(256), dim3(256), 0, s, p); break;
          case 3: hipLaunchKernelGGL(persistent_kernel<3>, dim3(256), dim3(256), 0, s, p); break;
          default: hipLaunchKernelGGL(persistent_kernel<4>, dim3(256), dim3(256), 0, s, p); break;
        }
        CHECK(hipEventRecord(e1, s));
        CHECK(hipStreamSynchronize(s));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned e[2];
        CHECK(hipMemcpy(e, p.errors, 8, hipMemcpyDeviceToHost));
        if (r >= 2) { best = ms < best ? ms : best; sum += ms; err[0] += e[0]; err[1] += e[1]; }
      }
      printf("delay %d  mode %d %-10s  %8.2f us/phase (best %.2f)  mismatching words %u  give-ups %u\n", delay, mode, names[mode],
             sum / reps * 1e3f / phases, best * 1e3f / phases, err[0], err[1]);
    }
  }
  {
    std::vector<unsigned> x(256);
    CHECK(hipMemcpy(x.data(), p.xcc, 1024, hipMemcpyDeviceToHost));
    int uniform = 1;
    for (int b = 0; b < 256; ++b) uniform &= x[b] == x[b % GROUPS];
    printf("XCC_ID of workgroups 0..15:");
    for (int b = 0; b < 16; ++b) printf(" %u", x[b]);
    printf("   groups (blockIdx %% 8) share an XCD: %s\n", uniform ? "yes" : "NO");
  }
  // the same phases as launches
  {
    p.delay = 0;
    float best = 1e30f, sum = 0.f;
    unsigned errs = 0;
    for (int r = 0; r < reps + 2; ++r) {
      CHECK(hipMemsetAsync(p.errors, 0, 8, s));
      CHECK(hipEventRecord(e0, s));
      for (int ph = 0; ph < phases; ++ph) hipLaunchKernelGGL(phase_kernel, dim3(256), dim3(256), 0, s, p, ph);
      CHECK(hipEventRecord(e1, s));
      CHECK(hipStreamSynchronize(s));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      unsigned e[2];
      CHECK(hipMemcpy(e, p.errors, 8, hipMemcpyDeviceToHost));
      if (r >= 2) { best = ms < best ? ms : best; sum += ms; errs += e[0]; }
    }
    printf("launches (one kernel per phase, eager)  %8.2f us/phase (best %.2f)  mismatching words %u\n",
           sum / reps * 1e3f / phases, best * 1e3f / phases, errs);
    // and replayed from a graph
    hipGraph_t g;
    hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int ph = 0; ph < phases; ++ph) hipLaunchKernelGGL(phase_kernel, dim3(256), dim3(256), 0, s, p, ph);
    CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    best = 1e30f; sum = 0.f;
    for (int r = 0; r < reps + 2; ++r) {
      CHECK(hipEventRecord(e0, s));
      CHECK(hipGraphLaunch(ge, s));
      CHECK(hipEventRecord(e1, s));
      CHECK(hipStreamSynchronize(s));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    printf("launches (graph replay)                 %8.2f us/phase (best %.2f)\n", sum / reps * 1e3f / phases, best * 1e3f / phases);
  }
  return 0;
}
