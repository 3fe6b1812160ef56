// Ablation study of the direct-to-LDS GEMM main loop (measurement aid, not part of the library):
// the same loop with one ingredient removed at a time tells which resource bounds the kernel.
//   ABL bit 0: no global->LDS loads      bit 1: no LDS fragment reads     bit 2: no MFMAs
//   bit 3: no epilogue stores            bit 4: no per-tile barrier (incorrect results, timing only)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/gemm_ablate.hip -o gpurun_out/gemm_ablate \
//        -Ltransform-and-tell_amd/csrc -ltell_hip -Wl,-rpath,$PWD/transform-and-tell_amd/csrc
#ifdef NT_STORE
#define TELL_PROBE_NT_STORE 1
#endif
#include "../../transform-and-tell_amd/csrc/gemm.hip"
#include <stdio.h>
#include <vector>
#include <stdlib.h>
#include <string.h>
#include <math.h>

template <int BM, int BN, int WAVES_M, int WAVES_N, int ABL>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void ablate_kernel(GemmArgs p) {
  constexpr int NW = WAVES_M * WAVES_N, BK = 64;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int IA = BM / 8 / NW, IB = BN / 8 / NW;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MI = WM / 32, NI = WN / 32;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int M = p.M, N = p.N, K = p.K;
  const int tiles_n = (N + BN - 1) / BN;
  const int nwg = ((M + BM - 1) / BM) * tiles_n;
  // bit 7: persistent - gridDim.x workgroups walk the tile list (v = blockIdx.x, += gridDim.x)
  for (int v = blockIdx.x; v < ((ABL & 128) ? nwg : (int)blockIdx.x + 1); v += gridDim.x) {
  int tile_id;
  {
    const int orig = v, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  int tm, tn;
  {
    constexpr int GROUP_M = 8;
    const int tiles_m = (M + BM - 1) / BM;
    const int per_group = GROUP_M * tiles_n;
    const int g = tile_id / per_group, first_m = g * GROUP_M;
    const int gm = tiles_m - first_m < GROUP_M ? tiles_m - first_m : GROUP_M;
    const int in_g = tile_id - g * per_group;
    tm = first_m + in_g % gm;
    tn = in_g / gm;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const uint16_t* A = static_cast<const uint16_t*>(p.A);
  const uint16_t* B = static_cast<const uint16_t*>(p.B);
  const uint16_t* asrc[IA];
  const uint16_t* bsrc[IB];
#pragma unroll
  for (int j = 0; j < IA; ++j) {
    const int s = (wave * IA + j) * 64 + lane, pr = s >> 4, l16 = (s & 15) ^ (pr & 15);
    asrc[j] = A + (long)(m0 + 2 * pr + (l16 >> 3)) * p.lda + (l16 & 7) * 8;
  }
#pragma unroll
  for (int j = 0; j < IB; ++j) {
    const int s = (wave * IB + j) * 64 + lane, pr = s >> 4, l16 = (s & 15) ^ (pr & 15);
    bsrc[j] = B + (long)(n0 + 2 * pr + (l16 >> 3)) * p.ldb + (l16 & 7) * 8;
  }
  auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
    if constexpr (ABL & 1) return;
    unsigned char* sa = smem + stage * STAGE + (wave * IA) * 1024;
    unsigned char* sb = smem + stage * STAGE + A_BYTES + (wave * IB) * 1024;
#pragma unroll
    for (int j = 0; j < IA; ++j)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(asrc[j] + kt * BK), (lds_ptr_t)(sa + j * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < IB; ++j)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(bsrc[j] + kt * BK), (lds_ptr_t)(sb + j * 1024), 16, 0, 0);
  };
  int a_base[MI], a_x[MI], a_hi[MI], b_base[NI], b_x[NI], b_hi[NI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int r = wm * WM + i * 32 + (lane & 31);
    a_base[i] = (r >> 1) * 256; a_x[i] = (r >> 1) & 15; a_hi[i] = (r & 1) << 3;
  }
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int r = wn * WN + j * 32 + (lane & 31);
    b_base[j] = (r >> 1) * 256; b_x[j] = (r >> 1) & 15; b_hi[j] = (r & 1) << 3;
  }
  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = K / BK;
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  bf16x8 a[2][MI], b[2][NI];
  auto rnd = [&](int salt) {
    bf16x8 v;
    unsigned x = (tid * 2654435761u) ^ (salt * 40503u) ^ (blockIdx.x * 97u);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      x = x * 1664525u + 1013904223u;
      const uint16_t bits = (uint16_t)(0x3c00 + ((x >> 16) & 0x3ff)) ^ (uint16_t)((x >> 3) & 0x8000);
      v[e] = __builtin_bit_cast(__bf16, bits);
    }
    return v;
  };
#pragma unroll
  for (int i = 0; i < MI; ++i) { a[0][i] = (ABL & 32) ? rnd(i) : bf16x8{}; a[1][i] = (ABL & 32) ? rnd(i + 8) : bf16x8{}; }
#pragma unroll
  for (int j = 0; j < NI; ++j) { b[0][j] = (ABL & 32) ? rnd(j + 16) : bf16x8{}; b[1][j] = (ABL & 32) ? rnd(j + 24) : bf16x8{}; }
  for (int kt = 0; kt < nk; ++kt) {
    const int st = kt & 1;
    if constexpr (!(ABL & 64)) { if (kt + 1 < nk) issue(kt + 1, st ^ 1); }
    auto issue_part = [&](int ks) __attribute__((always_inline)) {
      if constexpr ((ABL & 64) && !(ABL & 1)) {
        if (kt + 1 < nk) {
          unsigned char* sa = smem + (st ^ 1) * STAGE + (wave * IA) * 1024;
          unsigned char* sb = smem + (st ^ 1) * STAGE + A_BYTES + (wave * IB) * 1024;
#pragma unroll
          for (int j = 0; j < IA; ++j)
            if (j * 4 / IA == ks)
              __builtin_amdgcn_global_load_lds((glb_ptr_t)(asrc[j] + (kt + 1) * BK), (lds_ptr_t)(sa + j * 1024), 16, 0, 0);
#pragma unroll
          for (int j = 0; j < IB; ++j)
            if (j * 4 / IB == ks)
              __builtin_amdgcn_global_load_lds((glb_ptr_t)(bsrc[j] + (kt + 1) * BK), (lds_ptr_t)(sb + j * 1024), 16, 0, 0);
        }
      }
    };
    const unsigned char* ta = smem + st * STAGE;
    const unsigned char* tb = ta + A_BYTES;
    constexpr bool PREFETCH = MI * NI >= 8;
    auto ldfrag = [&](int ks, int buf) __attribute__((always_inline)) {
      if constexpr (ABL & 2) {
        asm volatile("" : "+v"(a[buf][0]), "+v"(b[buf][0]));     // opaque: keeps the MFMAs live without LDS traffic
        return;
      }
      const int c = ks * 2 + (lane >> 5);
#pragma unroll
      for (int i = 0; i < MI; ++i)
        a[buf][i] = *reinterpret_cast<const bf16x8*>(ta + a_base[i] + (((a_hi[i] | c) ^ a_x[i]) << 4));
#pragma unroll
      for (int j = 0; j < NI; ++j)
        b[buf][j] = *reinterpret_cast<const bf16x8*>(tb + b_base[j] + (((b_hi[j] | c) ^ b_x[j]) << 4));
    };
    if constexpr (PREFETCH) ldfrag(0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if constexpr (PREFETCH) {
        if (ks + 1 < 4) ldfrag(ks + 1, (ks + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
      } else {
        ldfrag(ks, ks & 1);
      }
      issue_part(ks);
      if constexpr (ABL & 4) {
#pragma unroll
        for (int i = 0; i < MI; ++i) asm volatile("" :: "v"(a[ks & 1][i]));
#pragma unroll
        for (int j = 0; j < NI; ++j) asm volatile("" :: "v"(b[ks & 1][j]));
      } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ks & 1][j], a[ks & 1][i], acc[i][j], 0, 0, 0);
      }
      if constexpr (PREFETCH) __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (!(ABL & 16)) __syncthreads();
  }
  if constexpr (ABL & 8) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) static_cast<uint16_t*>(p.C)[tid] = 1;
    if constexpr (ABL & 128) { __syncthreads(); continue; } else return;
  }
  constexpr int CS = (BM * (BN + 8) * 2 <= 2 * STAGE) ? BN + 8 : BN;
  __syncthreads();
  glds_store_tile<BM, BN, WM, WN, MI, NI, CS, 64 * NW>(acc, p, m0, n0, wm, wn, lane, tid,
                                                       reinterpret_cast<uint16_t*>(smem));
  if constexpr (ABL & 128) {            // LDS staging has been read into registers; the global stores drain behind the next tile
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  }
}


// ---- candidate: BK = 32 tiles, STAGES-deep direct-to-LDS pipeline (counted vmcnt, fence-free barrier)
template <int BM, int BN, int WAVES_M, int WAVES_N, int S, int ABL>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void deep_kernel(GemmArgs p) {
  constexpr int NW = WAVES_M * WAVES_N, BK = 32;
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;
  constexpr int IA = BM / 16 / NW, IB = BN / 16 / NW, L = IA + IB;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MI = WM / 32, NI = WN / 32;
  static_assert(IA >= 1 && IB >= 1, "tile too small");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[S * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int M = p.M, N = p.N, K = p.K;
  const int tiles_n = (N + BN - 1) / BN;
  const int nwg = gridDim.x;
  int tile_id;
  {
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  int tm, tn;
  {
    constexpr int GROUP_M = 8;
    const int tiles_m = (M + BM - 1) / BM;
    const int per_group = GROUP_M * tiles_n;
    const int g = tile_id / per_group, first_m = g * GROUP_M;
    const int gm = tiles_m - first_m < GROUP_M ? tiles_m - first_m : GROUP_M;
    const int in_g = tile_id - g * per_group;
    tm = first_m + in_g % gm;
    tn = in_g / gm;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const uint16_t* A = static_cast<const uint16_t*>(p.A);
  const uint16_t* B = static_cast<const uint16_t*>(p.B);
  // LDS image: row r = 64 bytes, logical 16-byte chunk c stored at position c ^ ((r>>2)&3)
  const uint16_t* asrc[IA];
  const uint16_t* bsrc[IB];
#pragma unroll
  for (int j = 0; j < IA; ++j) {
    const int s = (wave * IA + j) * 64 + lane, r = s >> 2, c = (s & 3) ^ ((r >> 2) & 3);
    asrc[j] = A + (long)(m0 + r) * p.lda + c * 8;
  }
#pragma unroll
  for (int j = 0; j < IB; ++j) {
    const int s = (wave * IB + j) * 64 + lane, r = s >> 2, c = (s & 3) ^ ((r >> 2) & 3);
    bsrc[j] = B + (long)(n0 + r) * p.ldb + c * 8;
  }
  auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
    if constexpr (ABL & 1) return;
    unsigned char* sa = smem + stage * STAGE + (wave * IA) * 1024;
    unsigned char* sb = smem + stage * STAGE + A_BYTES + (wave * IB) * 1024;
#pragma unroll
    for (int j = 0; j < IA; ++j)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(asrc[j] + kt * BK), (lds_ptr_t)(sa + j * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < IB; ++j)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(bsrc[j] + kt * BK), (lds_ptr_t)(sb + j * 1024), 16, 0, 0);
  };
  int a_off[MI], a_x[MI], b_off[NI], b_x[NI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int r = wm * WM + i * 32 + (lane & 31);
    a_off[i] = r * 64; a_x[i] = (r >> 2) & 3;
  }
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int r = wn * WN + j * 32 + (lane & 31);
    b_off[j] = r * 64; b_x[j] = (r >> 2) & 3;
  }
  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = K / BK;
#pragma unroll
  for (int s = 0; s < S - 1; ++s)
    if (s < nk) issue(s, s);
  if (nk >= S - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((S - 2) * L) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int kt = 0; kt < nk; ++kt) {
    const int st = kt % S;
    const bool more = kt + S - 1 < nk;
    if (more) issue(kt + S - 1, (kt + S - 1) % S);
    const unsigned char* ta = smem + st * STAGE;
    const unsigned char* tb = ta + A_BYTES;
    bf16x8 a[2][MI], b[2][NI];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = ks * 2 + (lane >> 5);
#pragma unroll
      for (int i = 0; i < MI; ++i)
        a[ks][i] = *reinterpret_cast<const bf16x8*>(ta + a_off[i] + ((c ^ a_x[i]) << 4));
#pragma unroll
      for (int j = 0; j < NI; ++j)
        b[ks][j] = *reinterpret_cast<const bf16x8*>(tb + b_off[j] + ((c ^ b_x[j]) << 4));
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ks][j], a[ks][i], acc[i][j], 0, 0, 0);
    if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((S - 2) * L) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if constexpr (ABL & 8) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) static_cast<uint16_t*>(p.C)[tid] = 1;
    return;
  }
  constexpr int CS = (BM * (BN + 8) * 2 <= S * STAGE) ? BN + 8 : BN;
  static_assert(BM * CS * 2 <= S * STAGE, "output tile must fit");
  __syncthreads();
  glds_store_tile<BM, BN, WM, WN, MI, NI, CS, 64 * NW>(acc, p, m0, n0, wm, wn, lane, tid,
                                                       reinterpret_cast<uint16_t*>(smem));
}

template <int BM, int BN, int WMs, int WNs, int S, int ABL>
static float run_deep(GemmArgs a, int iters) {
  const unsigned grid = (a.M / BM) * (a.N / BN);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((deep_kernel<BM, BN, WMs, WNs, S, ABL>), dim3(grid), dim3(64 * WMs * WNs), 0, 0, a);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((deep_kernel<BM, BN, WMs, WNs, S, ABL>), dim3(grid), dim3(64 * WMs * WNs), 0, 0, a);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) printf("launch error\n");
  return ms * 1e3f / iters;
}

// max |difference| between the product kernel's output and a candidate's (same inputs)
static double check(GemmArgs a, void* Cref, size_t n) {
  std::vector<uint16_t> x(n), y(n);
  hipMemcpy(x.data(), a.C, n * 2, hipMemcpyDeviceToHost);
  hipMemcpy(y.data(), Cref, n * 2, hipMemcpyDeviceToHost);
  double md = 0;
  for (size_t i = 0; i < n; ++i) {
    unsigned ux = (unsigned)x[i] << 16, uy = (unsigned)y[i] << 16;
    float fx, fy; memcpy(&fx, &ux, 4); memcpy(&fy, &uy, 4);
    double d = fabs((double)fx - fy);
    if (d > md) md = d;
  }
  return md;
}

template <int BM, int BN, int WMs, int WNs, int S>
static void study_deep(const char* name, GemmArgs a, void* Cref) {
  const double fl = 2.0 * a.M * a.N * a.K / 1e6;
  const float full = run_deep<BM, BN, WMs, WNs, S, 0>(a, 30);
  const double md = check(a, Cref, (size_t)a.M * a.N);
  printf("%s S%d M%d N%d K%d: full %.1f us (%.0f TF/s) maxdiff %.3g", name, S, a.M, a.N, a.K, full, fl / full, md);
  printf(" | no-gload %.1f", run_deep<BM, BN, WMs, WNs, S, 1>(a, 30));
  printf(" | no-store %.1f\n", run_deep<BM, BN, WMs, WNs, S, 8>(a, 30));
  fflush(stdout);
}

template <int BM, int BN, int WMs, int WNs, int ABL>
static float run(GemmArgs a, int iters) {
  const unsigned grid = (ABL & 128) ? 256 * (BM * BN <= 128 * 128 ? 2 : 1) : (a.M / BM) * (a.N / BN);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((ablate_kernel<BM, BN, WMs, WNs, ABL>), dim3(grid), dim3(64 * WMs * WNs), 0, 0, a);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((ablate_kernel<BM, BN, WMs, WNs, ABL>), dim3(grid), dim3(64 * WMs * WNs), 0, 0, a);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) printf("launch error\n");
  return ms * 1e3f / iters;
}

template <int BM, int BN, int WMs, int WNs>
static void study(const char* name, GemmArgs a) {
  const double fl = 2.0 * a.M * a.N * a.K / 1e6;
  const float full = run<BM, BN, WMs, WNs, 0>(a, 30);
  printf("%s M%d N%d K%d: full %.1f us (%.0f TF/s)", name, a.M, a.N, a.K, full, fl / full);
  printf(" | no-gload %.1f", run<BM, BN, WMs, WNs, 1>(a, 30));
  printf(" | no-ldsread %.1f", run<BM, BN, WMs, WNs, 2>(a, 30));
  printf(" | no-mfma %.1f", run<BM, BN, WMs, WNs, 4>(a, 30));
  printf(" | no-store %.1f", run<BM, BN, WMs, WNs, 8>(a, 30));
  printf(" | no-barrier %.1f", run<BM, BN, WMs, WNs, 16>(a, 30));
  printf(" | mfma-only %.1f", run<BM, BN, WMs, WNs, 1 | 2 | 8>(a, 30));
  printf(" | mfma-only(random operands) %.1f", run<BM, BN, WMs, WNs, 1 | 2 | 8 | 32>(a, 30));
  printf(" | mfma+gload(random) %.1f", run<BM, BN, WMs, WNs, 2 | 8 | 32>(a, 30));
  printf(" | store-only %.1f", run<BM, BN, WMs, WNs, 1 | 2 | 4>(a, 30));
  printf(" | spread-issue full %.1f", run<BM, BN, WMs, WNs, 64>(a, 30));
  printf(" | spread-issue no-store %.1f", run<BM, BN, WMs, WNs, 64 | 8>(a, 30));
  printf(" | persistent %.1f", run<BM, BN, WMs, WNs, 128>(a, 30));
  printf(" | persistent+spread %.1f", run<BM, BN, WMs, WNs, 128 | 64>(a, 30));
  printf(" | persistent+spread no-store %.1f", run<BM, BN, WMs, WNs, 128 | 64 | 8>(a, 30));
  printf(" | gload-only %.1f", run<BM, BN, WMs, WNs, 2 | 4 | 8>(a, 30));
  printf(" | lds-only %.1f\n", run<BM, BN, WMs, WNs, 1 | 4 | 8>(a, 30));
  fflush(stdout);
}


template <typename Kern>
static float run_kernel(Kern kern, GemmArgs a, unsigned grid, unsigned block, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, a);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, a);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) printf("launch error\n");
  return ms * 1e3f / iters;
}

int main() {
  const int shapes[][3] = {{8192, 4096, 1024}, {8192, 1024, 4096}, {8192, 3072, 1024}, {8192, 1024, 1024}, {8192, 2048, 1024}, {4096, 4096, 4096}, {8192, 8192, 8192}};
  for (auto& s : shapes) {
    const int M = s[0], N = s[1], K = s[2];
    std::vector<uint16_t> h((size_t)M * K);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 + ((x >> 16) & 0x3ff)) ^ (uint16_t)((x >> 3) & 0x8000); }
    void *A, *B, *C;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2);
    hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice);
    std::vector<uint16_t> hb((size_t)N * K);
    for (auto& v : hb) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 + ((x >> 16) & 0x3ff)) ^ (uint16_t)((x >> 3) & 0x8000); }
    hipMemcpy(B, hb.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
    GemmArgs a = {};
    a.A = A; a.B = B; a.C = C; a.lda = K; a.ldb = K; a.ldc = N; a.M = M; a.N = N; a.K = K; a.alpha = 1.f;
    void* Cref; hipMalloc(&Cref, (size_t)M * N * 2);
    if (getenv("ABLATE")) study<128, 128, 2, 2>("128x128/4w", a);
    run<128, 128, 2, 2, 0>(a, 1);
    hipMemcpy(Cref, C, (size_t)M * N * 2, hipMemcpyDeviceToDevice);
    if (getenv("ABLATE")) study<256, 256, 2, 4>("256x256/8w", a);
    {
      const double fl = 2.0 * M * N * K / 1e6;
      for (int rep = 0; rep < 2; ++rep) {
        const float t128 = run_kernel(gemm_nt_glds_kernel<uint16_t, 128, 128, 2, 2>, a, (M / 128) * (N / 128), 256, 30);
        const float t256 = run_kernel(gemm_nt_glds_kernel<uint16_t, 256, 256, 2, 4>, a, (M / 256) * (N / 256), 512, 30);
        hipMemset(C, 0, (size_t)M * N * 2);
        const float tpp = run_kernel(gemm_nt_pp_kernel<uint16_t>, a, (M / 256) * (N / 256), 512, 30);
        const double md = check(a, Cref, (size_t)M * N);
        printf("M%d N%d K%d: product 128x128 %.1f us (%.0f TF/s) | product 256x256 %.1f us (%.0f) | ping-pong %.1f us (%.0f TF/s) maxdiff %.3g\n",
               M, N, K, t128, fl / t128, t256, fl / t256, tpp, fl / tpp, md);
      }
      // with bias + GELU epilogue
      float* bias; hipMalloc(&bias, N * 4); hipMemset(bias, 0, N * 4);
      GemmArgs g = a; g.bias = bias; g.bias_mode = 1; g.act = 2;
      const float t256 = run_kernel(gemm_nt_glds_kernel<uint16_t, 256, 256, 2, 4>, g, (M / 256) * (N / 256), 512, 30);
      hipMemcpy(Cref, C, (size_t)M * N * 2, hipMemcpyDeviceToDevice);
      const float tpp = run_kernel(gemm_nt_pp_kernel<uint16_t>, g, (M / 256) * (N / 256), 512, 30);
      printf("   bias+gelu: product 256x256 %.1f us | ping-pong %.1f us maxdiff %.3g\n", t256, tpp, check(g, Cref, (size_t)M * N));
      hipFree(bias);
    }
    hipFree(Cref);
    hipFree(A); hipFree(B); hipFree(C);
  }
  return 0;
}
