// Probe: how fast can ONE compute unit pull L2-resident operand tiles into its LDS with global_load_lds_dwordx4, as a
// function of workgroups per CU and ring depth (bytes in flight)?  Every tiled kernel of the repository that is not
// MFMA-bound stages at 40-46 GB/s per CU - is that a hardware ceiling or a property of the kernels?
// hipcc --offload-arch=gfx950 -O2 tools/probes/stage_rate.hip -o tools/probes/_bin/stage_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef const __attribute__((address_space(1))) void* gptr;
typedef __attribute__((address_space(3))) void* lptr;

// a stage = STAGE_KB of data = STAGE_KB / 4 dwordx4 DMA instructions per thread of a 256-thread workgroup
template <int DEPTH, int STAGE_KB>
__global__ __launch_bounds__(256) void dma_ring(const char* __restrict__ src, long region, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int PER = STAGE_KB / 4;                       // instructions per thread per stage
  const int tid = threadIdx.x;
  // every workgroup walks the SAME region (L2 / MALL resident), starting at its own offset
  long off = ((long)blockIdx.x * 65536) % region;
  auto issue = [&](int slot) {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const char* g = src + off + (long)j * 4096 + (tid >> 6) * 1024 + (tid & 63) * 16;
      char* l = lds + slot * STAGE_KB * 1024 + j * 4096 + (tid >> 6) * 1024;
      __builtin_amdgcn_global_load_lds((gptr)g, (lptr)l, 16, 0, 0);
    }
    off += STAGE_KB * 1024;
    if (off + STAGE_KB * 1024 > region) off = 0;
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(d);
  int slot = 0;
  for (int it = 0; it < iters; ++it) {
    if constexpr (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((DEPTH - 1) * PER) : "memory");
    __syncthreads();
    issue(slot);
    slot = slot + 1 == DEPTH ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0 && lds[17] == 123 && lds[4099] == 77) sink[0] = 1;
}
// the same ring through registers (global_load_dwordx4 -> VGPR, no LDS write)
template <int DEPTH, int STAGE_KB>
__global__ __launch_bounds__(256) void reg_ring(const char* __restrict__ src, long region, int iters, unsigned* sink) {
  constexpr int PER = STAGE_KB / 4;
  const int tid = threadIdx.x;
  long off = ((long)blockIdx.x * 65536) % region;
  uint4 r[DEPTH][PER];
  uint4 acc = {0, 0, 0, 0};
  auto issue = [&](uint4 (&dst)[PER]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < PER; ++j) dst[j] = *reinterpret_cast<const uint4*>(src + off + (long)j * 4096 + tid * 16);
    off += STAGE_KB * 1024;
    if (off + STAGE_KB * 1024 > region) off = 0;
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(r[d]);
  for (int it = 0; it < iters; it += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int j = 0; j < PER; ++j) { acc.x ^= r[d][j].x; acc.y ^= r[d][j].y; acc.z ^= r[d][j].z; acc.w ^= r[d][j].w; }
      issue(r[d]);
    }
  }
  if (acc.x == 0x1234567 && acc.y == 3) sink[0] = acc.z;
}

template <typename K>
static int run(const char* name, K kern, int depth, int stage_kb, int wg_per_cu, const char* src, long region, unsigned* sink, size_t lds_bytes) {
  const int iters = 2000, n_cu = 256;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(n_cu * wg_per_cu), dim3(256), lds_bytes, 0, src, region, iters, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double bytes = (double)n_cu * wg_per_cu * (iters + depth) * stage_kb * 1024.0;
  printf("%-4s depth %d x %2d KB, %d WG/CU (%3d KB in flight per CU): %7.1f us  %6.1f GB/s per CU  %5.2f TB/s chip\n", name, depth,
         stage_kb, wg_per_cu, depth * stage_kb * wg_per_cu, best * 1e3, bytes / n_cu / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e12);
  return 0;
}

int main() {
  unsigned* sink; CK(hipMalloc(&sink, 64));
  for (long region_mb : {2L, 32L}) {
    const long region = region_mb << 20;
    char* src; CK(hipMalloc(&src, region)); CK(hipMemset(src, 1, region));
    printf("# every workgroup walks the same %ld MB (%s)\n", region_mb, region_mb <= 4 ? "fits each XCD's 4 MB L2" : "MALL / HBM");
#define DMA(D, S) for (int w : {1, 2, 4}) if ((D) * (S) * w <= 128) run("dma", dma_ring<D, S>, D, S, w, src, region, sink, (size_t)(D) * (S) * 1024)
    DMA(1, 16); DMA(2, 16); DMA(4, 16); DMA(8, 16); DMA(2, 32); DMA(4, 32); DMA(2, 8); DMA(4, 8); DMA(8, 8);
#define REG(D, S) for (int w : {1, 2, 4}) run("reg", reg_ring<D, S>, D, S, w, src, region, sink, 0)
    REG(1, 16); REG(2, 16); REG(4, 16); REG(2, 32);
    CK(hipFree(src));
  }
  return 0;
}
