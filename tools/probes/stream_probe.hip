// Streaming-bandwidth probe for the flat-buffer kernels (BertAdam: 4 fp32 read streams + 3 fp32 / 1 bf16 / 1 zero write
// streams): which access shape reaches the float4-copy rate of this chip?  hipcc --offload-arch=gfx950 -O3 stream_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(2))) unsigned int u2;

template <bool NT> __device__ __forceinline__ f4 ld(const f4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(f4* p, f4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <bool NT> __device__ __forceinline__ void st2(u2* p, u2 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

__global__ __launch_bounds__(256) void copy_kernel(const f4* __restrict__ a, f4* __restrict__ b, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) b[i] = a[i];
}
__device__ __forceinline__ unsigned bf(float f) { unsigned u = __float_as_uint(f); u += 0x7fffu + ((u >> 16) & 1u); return u >> 16; }

// U chunks of 1024 elements per block iteration, all loads first.  WIDE: a block iteration covers U*1024 CONTIGUOUS elements
// and lane l of a wave takes float4 index (wave*64*U + u*64 + l) -> consecutive lanes consecutive 16 B, U passes 1 KB apart.
template <int U, bool NTL, bool NTS, bool ZERO>
__global__ __launch_bounds__(256) void adam_kernel(f4* __restrict__ p, f4* __restrict__ g, f4* __restrict__ m, f4* __restrict__ v,
                                                   u2* __restrict__ sh, long n_chunks, float lr) {
  const long per = (long)U;
  for (long c0 = (long)blockIdx.x * per; c0 < n_chunks; c0 += (long)gridDim.x * per) {
    f4 G[U], P[U], M[U], V[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long o = (c0 + u) * 256 + threadIdx.x;
      if (c0 + u < n_chunks) { G[u] = ld<NTL>(g + o); P[u] = ld<NTL>(p + o); M[u] = ld<NTL>(m + o); V[u] = ld<NTL>(v + o); }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (c0 + u >= n_chunks) continue;
      const long o = (c0 + u) * 256 + threadIdx.x;
      f4 gg = G[u] * 0.5f, mm = M[u] * 0.9f + gg * 0.1f, vv = V[u] * 0.999f + gg * gg * 0.001f, pp = P[u];
#pragma unroll
      for (int k = 0; k < 4; ++k) pp[k] -= lr * (mm[k] / (sqrtf(vv[k]) + 1e-6f) + 0.01f * pp[k]);
      st<NTS>(p + o, pp); st<NTS>(m + o, mm); st<NTS>(v + o, vv);
      u2 s = {bf(pp[0]) | (bf(pp[1]) << 16), bf(pp[2]) | (bf(pp[3]) << 16)};
      st2<false>(sh + o, s);
      if (ZERO) st<NTS>(g + o, f4{0.f, 0.f, 0.f, 0.f});
    }
  }
}
template <int U, bool NTL>
__global__ __launch_bounds__(256) void sqsum_kernel(const f4* __restrict__ g, long n_chunks, float* __restrict__ partial) {
  __shared__ float red[4];
  for (long c0 = (long)blockIdx.x * U; c0 < n_chunks; c0 += (long)gridDim.x * U) {
    f4 G[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (c0 + u < n_chunks) G[u] = ld<NTL>(g + (c0 + u) * 256 + threadIdx.x);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (c0 + u >= n_chunks) continue;
      float s = G[u][0] * G[u][0] + G[u][1] * G[u][1] + G[u][2] * G[u][2] + G[u][3] * G[u][3];
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
      __syncthreads();
      if (threadIdx.x == 0) partial[c0 + u] = red[0] + red[1] + red[2] + red[3];
      __syncthreads();
    }
  }
}
template <typename F> float timeit(F f, int reps = 7) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> t;
  for (int r = 0; r < reps; ++r) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms); }
  std::sort(t.begin(), t.end()); return t[t.size() / 2] * 1e3f;
}
int main() {
  const long n = 176L << 20;            // ~ the decoder's parameter count
  const long n_chunks = n / 1024;
  float *p, *g, *m, *v, *partial; unsigned short* sh;
  hipMalloc(&p, n * 4); hipMalloc(&g, n * 4); hipMalloc(&m, n * 4); hipMalloc(&v, n * 4); hipMalloc(&sh, n * 2); hipMalloc(&partial, n_chunks * 4);
  hipMemset(p, 0, n * 4); hipMemset(g, 0, n * 4); hipMemset(m, 0, n * 4); hipMemset(v, 0, n * 4);
  auto report = [&](const char* name, float us, double bytes) { printf("%-44s %8.1f us  %5.2f TB/s\n", name, us, bytes / us * 1e-6); fflush(stdout); };
  for (int grid : {2048, 4096, 8192, 16384}) {
    char nm[64]; snprintf(nm, 64, "copy float4 grid %d", grid);
    report(nm, timeit([&] { hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, (const f4*)p, (f4*)m, n / 4); }), 8.0 * n);
  }
  const double ab = 34.0 * n, abnz = 30.0 * n;
#define RUN(U, NTL, NTS, Z, GRID) { char nm[96]; snprintf(nm, 96, "adam U=%d ntl=%d nts=%d zero=%d grid %d", U, NTL, NTS, Z, GRID); \
    report(nm, timeit([&] { hipLaunchKernelGGL((adam_kernel<U, NTL, NTS, Z>), dim3(GRID), dim3(256), 0, 0, (f4*)p, (f4*)g, (f4*)m, (f4*)v, (u2*)sh, n_chunks, 1e-4f); }), Z ? ab : abnz); }
  RUN(1, false, false, true, 4096) RUN(1, false, false, true, 2048) RUN(1, false, false, true, 8192) RUN(1, false, false, true, 16384)
  RUN(2, false, false, true, 4096) RUN(4, false, false, true, 4096) RUN(4, false, false, true, 2048) RUN(4, false, false, true, 1024)
  RUN(1, true, false, true, 4096) RUN(1, false, true, true, 4096) RUN(1, true, true, true, 4096)
  RUN(2, true, true, true, 4096) RUN(4, true, true, true, 4096) RUN(4, true, true, true, 2048) RUN(2, true, true, true, 8192)
  RUN(1, false, false, false, 4096) RUN(2, true, true, false, 4096) RUN(4, true, true, false, 2048)
#define RUNS(U, NTL, GRID) { char nm[96]; snprintf(nm, 96, "sqsum U=%d ntl=%d grid %d", U, NTL, GRID); \
    report(nm, timeit([&] { hipLaunchKernelGGL((sqsum_kernel<U, NTL>), dim3(GRID), dim3(256), 0, 0, (const f4*)g, n_chunks, partial); }), 4.0 * n); }
  RUNS(1, false, 4096) RUNS(2, false, 4096) RUNS(4, false, 4096) RUNS(4, true, 4096) RUNS(4, true, 2048) RUNS(8, true, 2048)
  return 0;
}
