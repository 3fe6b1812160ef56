// GEMMs on the CDNA4 matrix cores:  C[M,N] = epilogue(op(A) . op(B))
//
//   gemm_nt_glds_kernel  bf16, K % 64 == 0, >= 256 tiles: operands stream global -> LDS directly
//                        (global_load_lds_dwordx4), tiles 128x128 / 256x192 / 256x256.
//   gemm_nt_pp_kernel    bf16, whole rounds of full 256x256 tiles: the same DMA image consumed by two wave groups
//                        that alternate between LDS/DMA issue and MFMA segments (ping-pong), counted vmcnt.
//   gemm_nt_kernel       any dtype / K / size: register-staged 2-4 deep prefetch, tiles 64x64 .. 256x128.
//   gemm_tx_kernel       bf16 with K-major operands (A stored [K][M] and/or B stored [K][N]): the backward
//                        products dY^T.X and dY.W read activations and weights as the forward pass left them;
//                        fragments come from the LDS transpose read ds_read_b64_tr_b16.
//
//   bf16 : v_mfma_f32_32x32x16_bf16  (lane l supplies row l&31, k-chunk 8*(l>>5)..+8)
//   f32  : v_mfma_f32_32x32x2_f32    (exact-f32 parity mode; lane l: row l&31, k = l>>5)
// Every MFMA is issued with the operands SWAPPED (B fragment first), so the accumulator tile is C^T:
// lane l owns output row (l&31) and 4 consecutive output columns per register quad - see the epilogue.
//
// LDS rows are padded (bf16: 144 B stride -> conflict-free ds_read_b128 per 16-lane service group; f32:
// 33-dword stride; K-major tiles: +64 B skew for the transpose reads) or, for the lane-linear direct-to-LDS
// image, swizzled on the source address.
#include "common.h"
#include "options.h"
#include <atomic>
#include <mutex>
#include <vector>
#include "../../include/tell_hip.h"    // tell_gemm_tn_problem (and every prototype: a drifted definition fails to compile)
#include <type_traits>
#include <stdlib.h>
#include <alloca.h>
#include <stdio.h>

#include "gemm_common.h"

#include "gemm_epi.h"   // Mma, epilogues, staged stores, the BatchNorm statistics epilogue

// ------------------------------------------------------------- direct-to-LDS kernel (bf16, K % 64 == 0)
// global_load_lds_dwordx4: every lane's 16 bytes go straight from L2/HBM into LDS (no VGPR staging, no
// ds_write), the next K tile streams into the other LDS buffer while the MFMAs run on the current one.
// The LDS image of a tile must be lane-linear (wave-uniform base + lane*16), so the bank-conflict
// swizzle is applied on the SOURCE side: slot s (16 B) of the image holds row 2p + (l>>3), chunk l&7 with
// p = s>>4 and l = (s&15) ^ (p&15); fragment reads apply the same involution.  A 16-lane service group
// of ds_read_b128 then touches 16 distinct 16-byte slots of the 256-byte bank row (conflict-free).
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// s_waitcnt vmcnt(n) with a run-time n out of {0, PER, 2 PER, ...}: the count field is an immediate
template <int PER, int T>
__device__ __forceinline__ void wait_tiles_in_flight(int t) {
  if constexpr (T == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    if (t >= T) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER * T) : "memory");
    else wait_tiles_in_flight<PER, T - 1>(t);
  }
}

// NS = number of LDS stages.  NS = 2 keeps ONE K tile in flight while the current one is multiplied - enough when the
// MFMAs of a tile outlast a DMA round trip (128x64 wave tiles).  The small tiles do not: a 64x64 workgroup (one 32x32
// MFMA per wave and k-step) spends 128 clk of matrix time per K tile against 500-900 clk of L2 / HBM latency, so with
// two stages it advances one K tile per round trip (the decoder's M = 1024 GEMMs and the ResNet bottleneck convolutions:
// 16-64 dependent round trips per launch).  NS > 2 is a ring: NS - 1 tiles in flight, counted vmcnt (a wave waits for
// its own pieces of tile kt only), ONE raw s_barrier per K tile - after it every wave's pieces of tile kt have landed
// (RAW) and everyone has finished the MFMAs that read the stage the next issue overwrites (WAR).
template <typename OutT, int BM, int BN, int WAVES_M, int WAVES_N, bool CONV = false, int NS = 2>
__device__ __forceinline__ void gemm_nt_glds_body(const GemmArgs& p, const int block, const int n_blocks) {
  constexpr int NW = WAVES_M * WAVES_N, BK = 64;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int IA = BM / 8 / NW, IB = BN / 8 / NW;   // wave-instructions (1 KiB each) per wave per tile
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MI = WM / 32, NI = WN / 32;
  static_assert(IA >= 1 && IB >= 1, "tile too small for the wave count");
  static_assert(NS >= 2 && (IA + IB) * (NS - 2) <= 48, "ring depth against the 6-bit vmcnt field");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * STAGE];

  gemm_ts_enter(p);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  int M = p.M;
  if (p.m_dev) { int md = *p.m_dev; M = md < M ? md : M; }
  const int N = p.N, K = p.K;
  const int tiles_n = (N + BN - 1) / BN;
  const int nwg = n_blocks;
  int tile_id;
  {
    const int orig = block, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  // grouped traversal inside the XCD's contiguous range: GROUP_M m-tiles share one sweep over n, so the
  // ~64 workgroups resident on an XCD touch 8 A panels + 8 B panels (<= 4 MiB L2) instead of streaming
  // the whole B matrix once per pair of m-tiles (PMC: FETCH_SIZE 7.5x the operand bytes before).
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
  if (tile_id >= ((M + BM - 1) / BM) * tiles_n) return;

  const uint16_t* A = static_cast<const uint16_t*>(p.A);
  const uint16_t* B = static_cast<const uint16_t*>(p.B);
  // per-lane source pointers (row clamped: rows past M/N only feed outputs that are never stored)
  const uint16_t* asrc[IA];
  const uint16_t* bsrc[IB];
  constexpr bool conv = CONV;                               // implicit-convolution gather (tell_conv_bn_stats)
  int cv_ih[IA], cv_iw[IA];                                 // conv mode: top-left input pixel of the row's window
#pragma unroll
  for (int j = 0; j < IA; ++j) {
    const int s = (wave * IA + j) * 64 + lane, pr = s >> 4, l16 = (s & 15) ^ (pr & 15);
    int row = m0 + 2 * pr + (l16 >> 3);
    row = row < M ? row : M - 1;
    if (conv) {
      const int ow = row % p.conv_OW, t = row / p.conv_OW, oh = t % p.conv_OH, b = t / p.conv_OH;
      if (p.conv_cshift < 0) {
        // 7x7 / stride 2 / pad 3 stem over NHWC4 pixels (3 channels + a zero one = 8 bytes): a K tile is TWO kernel rows
        // of 8 pixels x 4 channels; the window starts at input column 2 ow - 4 (one column left of the first tap, whose
        // weight is zero) so that every 16-byte chunk = an aligned pixel pair, wholly inside the image or wholly outside
        // (W even).  This lane's chunk (l16 & 7): kernel row (chunk >> 2) of the tile's two, pixel pair chunk & 3.
        cv_ih[j] = oh * 2 - 3 + ((l16 & 7) >> 2);
        cv_iw[j] = ow * 2 - 4 + ((l16 & 7) & 3) * 2;
        asrc[j] = A + ((long)b * p.conv_H * p.conv_W << 2);
      } else {
        cv_ih[j] = oh * p.conv_stride - p.conv_pad;
        cv_iw[j] = ow * p.conv_stride - p.conv_pad;
        asrc[j] = A + ((long)b * p.conv_H * p.conv_W << (6 + p.conv_cshift)) + (l16 & 7) * 8;   // image base + chunk
      }
    } else {
      cv_ih[j] = cv_iw[j] = 0;
      asrc[j] = A + (long)row * p.lda + (l16 & 7) * 8;
    }
  }
  const uint16_t* zero_src = static_cast<const uint16_t*>(p.conv_zero);     // 16 zero bytes for a padding pixel
#pragma unroll
  for (int j = 0; j < IB; ++j) {
    const int s = (wave * IB + j) * 64 + lane, pr = s >> 4, l16 = (s & 15) ^ (pr & 15);
    int row = n0 + 2 * pr + (l16 >> 3);
    row = row < N ? row : N - 1;
    bsrc[j] = B + (long)row * p.ldb + (l16 & 7) * 8;
  }
  // quarter q (0..3) of tile kt's DMA instructions, or all of them (q < 0)
  auto issue = [&](int kt, int stage, int q) __attribute__((always_inline)) {
    unsigned char* sa = smem + stage * STAGE + (wave * IA) * 1024;
    unsigned char* sb = smem + stage * STAGE + A_BYTES + (wave * IB) * 1024;
    if (conv && p.conv_cshift < 0) {                      // stem: kernel rows 2 kt, 2 kt + 1 (row 7: zero weights)
#pragma unroll
      for (int j = 0; j < IA; ++j)
        if (q < 0 || j * 4 / IA == q) {
          const int ih = cv_ih[j] + 2 * kt, iw = cv_iw[j];
          const bool in = (unsigned)ih < (unsigned)p.conv_H && (unsigned)iw < (unsigned)p.conv_W;
          const uint16_t* src = in ? asrc[j] + ((long)(ih * p.conv_W + iw) << 2) : zero_src;
          __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(sa + j * 1024), 16, 0, 0);
        }
    } else if (conv) {
      // K tile kt = channels [c0, c0 + 64) of tap (kh, kw): scalar
      const int tap = kt >> p.conv_cshift, c0 = (kt & ((1 << p.conv_cshift) - 1)) << 6;
      const int kh = tap / p.conv_KW, kw = tap - kh * p.conv_KW;
#pragma unroll
      for (int j = 0; j < IA; ++j)
        if (q < 0 || j * 4 / IA == q) {
          const int ih = cv_ih[j] + kh, iw = cv_iw[j] + kw;
          const bool in = (unsigned)ih < (unsigned)p.conv_H && (unsigned)iw < (unsigned)p.conv_W;
          const uint16_t* src = in ? asrc[j] + ((long)(ih * p.conv_W + iw) << (6 + p.conv_cshift)) + c0 : zero_src;
          __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(sa + j * 1024), 16, 0, 0);
        }
    } else {
#pragma unroll
      for (int j = 0; j < IA; ++j)
        if (q < 0 || j * 4 / IA == q)
          __builtin_amdgcn_global_load_lds((glb_ptr_t)(asrc[j] + kt * BK), (lds_ptr_t)(sa + j * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < IB; ++j)
      if (q < 0 || j * 4 / IB == q)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(bsrc[j] + kt * BK), (lds_ptr_t)(sb + j * 1024), 16, 0, 0);
  };

  // fragment addressing: row r of a tile, 16-byte k-chunk c (0..7): byte = (r>>1)*256 + ((((r&1)<<3)|c) ^ ((r>>1)&15))*16
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
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) issue(s, s, -1);
  int st = 0, nst = NS - 1;                              // stage of tile kt / of tile kt + NS - 1
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed once at most min(NS - 2, tiles left) younger tiles of THIS wave are still in flight
    {
      const int left = nk - 1 - kt;
      wait_tiles_in_flight<IA + IB, NS - 2>(left < NS - 2 ? left : NS - 2);
    }
    __builtin_amdgcn_s_barrier();                        // (raw: __syncthreads() would drain the DMA queue, vmcnt(0))
    __builtin_amdgcn_sched_barrier(0);
    // tile kt+NS-1 streams in under the MFMAs below, a quarter of its DMA instructions per k-substep: the LDS port
    // is shared by these writes and the fragment reads, and a burst of 8 at the top of the iteration stalls both
    // (probe: tools/probes/gemm_ablate.hip, +3..9 %)
    const bool more = kt + NS - 1 < nk;
    const unsigned char* ta = smem + st * STAGE;
    const unsigned char* tb = ta + A_BYTES;
    // Large wave tiles (128x64: 8 MFMAs per k-substep): fragments of substep ks+1 are read into a second
    // register set BEFORE the MFMAs of substep ks issue, so the LDS latency sits behind this wave's own matrix
    // work (+5 % at 4096^3).  With 4 MFMAs per substep the compiler's own interleaving is as good (measured).
    constexpr bool PREFETCH = MI * NI >= 8;
    bf16x8 a[2][MI], b[2][NI];
    auto ldfrag = [&](int ks, int buf) __attribute__((always_inline)) {
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
        __builtin_amdgcn_sched_barrier(0);             // keep the prefetch ABOVE the MFMAs (the scheduler sinks it)
      } else {
        ldfrag(ks, ks & 1);
      }
      if (more) issue(kt + NS - 1, nst, ks);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ks & 1][j], a[ks & 1][i], acc[i][j], 0, 0, 0);
      if constexpr (PREFETCH) __builtin_amdgcn_sched_barrier(0);
    }
    st = st + 1 == NS ? 0 : st + 1;
    nst = nst + 1 == NS ? 0 : nst + 1;
  }
  __syncthreads();                                       // every wave is done with the stages: they become the staging area
  gemm_nt_glds_epilogue<OutT, BM, BN, WAVES_M, WAVES_N, CONV, NS>(acc, p, smem, m0, n0, tm, wm, wn, lane, tid, M, N);
}
template <typename OutT, int BM, int BN, int WAVES_M, int WAVES_N, bool CONV = false, int NS = 2>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_nt_glds_kernel(GemmArgs p) {
  gemm_nt_glds_body<OutT, BM, BN, WAVES_M, WAVES_N, CONV, NS>(p, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------- 256x256 ping-pong kernel (bf16, full tiles only)
// 8 waves as 2 groups x 4 (one wave of each group per SIMD).  The K tile (64) is staged as four half-tile images
// (A rows 0-127 / 128-255, B rows 0-127 / 128-255; 16 KB each = 2 DMA instructions per wave), two K tiles in LDS.
// A K tile is consumed in 4 phases, one 64x32 output quadrant per wave and phase (8 MFMAs):
//     phase 1: read A0, B0 -> C(0,0)   2: read B1 -> C(0,1)   3: read A1 -> C(1,1)   4: (B0 still held) -> C(1,0)
// Every phase is  { ds_reads + one half-tile of DMA prefetch | s_barrier | MFMAs | s_barrier }  and group 1 runs
// one barrier behind group 0, so on each SIMD one wave is in its MFMA segment while the other does its LDS reads
// and DMA issue - LDS port and matrix core are busy at the same time instead of taking turns (the lockstep 2-barrier
// kernels above lose ~40 % to that, tools/probes/gemm_ablate.hip).  Loads are never drained inside the loop:
// half-tile n (n = 4*tile + {A0,B0,B1,A1}) is issued in phase n-6 and retired by the counted vmcnt(4) of phase 4 of
// the PREVIOUS K tile, i.e. at least one barrier before its first read (RAW) and it overwrites a slot whose last
// read was >= 2 phases earlier (WAR).
// (A 256x128 variant of this schedule - 64x64 wave tiles, one uniform 32-k phase, ring of 6 LDS slots - was built and
//  measured: 923 TFLOP/s at 8192^3, the same as the lockstep 128x128 kernel.  With 64x64 wave tiles the LDS port
//  (fragment reads + DMA writes = 1.34x the MFMA time) is the wall whatever the schedule; only the 128x64 wave tile
//  of the 256x256 block gets under it.  N = 1024 outputs therefore stay on the 128x128 kernel.)
template <typename OutT>
__global__ __launch_bounds__(512) void gemm_nt_pp_kernel(GemmArgs p) {
  constexpr int BM = 256, BN = 256, BK = 64;
  constexpr int HALF = 128 * 128, TILE = 4 * HALF;        // bytes: one half-tile image, one K tile (A0 A1 B0 B1)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * TILE + 16];   // (+ the tile-queue broadcast word)
  gemm_ts_enter(p);
  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int M = p.M, N = p.N, K = p.K;
  const int tiles_n = N / BN, tiles_m = M / BM;
  const int n_tiles = tiles_m * tiles_n;
  // PERSISTENT when p.queue is set: gridDim.x <= #CUs workgroups pull tile indices from the launch's device counter
  // until it runs past n_tiles.  A 256x256 workgroup needs a whole CU (128 KB of LDS, 2 x 228 registers per SIMD); as
  // one workgroup per TILE it had to win a CU again for every tile against the small kernels of the other two streams
  // of the step, and inside the timed region this kernel ran 1.7x slower than alone.  A resident workgroup keeps its CU
  // for the whole GEMM; late starters simply take fewer tiles.  The very last fetch of a launch (value n_tiles +
  // gridDim.x - 1: every workgroup fetches once past the end) zeroes the counter for the next launch that uses it.
  int vb = blockIdx.x;
  volatile int* sq = reinterpret_cast<volatile int*>(smem + 2 * TILE);
  for (bool first = true;; first = false) {
  // every lane-derived value (fragment offsets, DMA source offsets) is re-derived per tile from an opaque copy of the
  // thread index: hoisted out of the tile loop they would have to survive the epilogue, whose staging needs every
  // register - the allocator then spills INSIDE the main loop (79 registers, 2x slower: measured)
  int tid = tid0;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, lh = lane >> 5;
  if (p.queue) {
    if (tid == 0) {
      const int got = atomicAdd(p.queue, 1);
      if (got == n_tiles + (int)gridDim.x - 1) atomicExch(p.queue, 0);
      *sq = got;
    }
    __syncthreads();
    vb = *sq;
    __syncthreads();                                       // (the word is rewritten by the next fetch)
    if (vb >= n_tiles) break;
  } else if (!first) {
    break;
  }
  int tile_id;
  {
    const int nwg = n_tiles, orig = vb, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  int tm, tn;
  {
    constexpr int GROUP_M = 8;
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
  // DMA sources (same swizzled lane-linear image as gemm_nt_glds_kernel, per half-tile of 128 rows)
  const uint16_t* asrc[2];
  const uint16_t* bsrc[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int s = (wave * 2 + jj) * 64 + lane, pr = s >> 4, l16 = (s & 15) ^ (pr & 15);
    const int row = 2 * pr + (l16 >> 3);
    asrc[jj] = A + (long)(m0 + row) * p.lda + (l16 & 7) * 8;
    bsrc[jj] = B + (long)(n0 + row) * p.ldb + (l16 & 7) * 8;
  }
  const long a_half = 128 * p.lda, b_half = 128 * p.ldb;
  // half-tile KIND of K tile tt: 0 = A0, 1 = B0, 2 = B1, 3 = A1 (the order the phases need them)
  auto stage = [&](int tt, auto kind_c) __attribute__((always_inline)) {
    constexpr int KIND = decltype(kind_c)::value;
    constexpr int H = KIND == 0 ? 0 : KIND == 3 ? 1 : KIND == 1 ? 2 : 3;      // LDS order A0 A1 B0 B1
    unsigned char* dst = smem + (tt & 1) * TILE + H * HALF + wave * 2048;
    const long off = (long)tt * BK + (KIND == 3 ? a_half : KIND == 2 ? b_half : 0);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(((KIND == 0 || KIND == 3) ? asrc[jj] : bsrc[jj]) + off),
                                       (lds_ptr_t)(dst + jj * 1024), 16, 0, 0);
  };
  // fragment addresses inside a half-tile image
  int a_off[2][4], b_off[4];
#pragma unroll
  for (int ii = 0; ii < 2; ++ii) {
    const int r = wr * 64 + ii * 32 + (lane & 31);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      a_off[ii][ks] = (r >> 1) * 256 + (((((r & 1) << 3) | (ks * 2 + lh)) ^ ((r >> 1) & 15)) << 4);
  }
  {
    const int r = wc * 32 + (lane & 31);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      b_off[ks] = (r >> 1) * 256 + (((((r & 1) << 3) | (ks * 2 + lh)) ^ ((r >> 1) & 15)) << 4);
  }
  bf16x8 af[2][4], bfr[2][4];
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto read_a = [&](int tt, int a) __attribute__((always_inline)) {
    const unsigned char* t = smem + (tt & 1) * TILE + a * HALF;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) af[ii][ks] = *reinterpret_cast<const bf16x8*>(t + a_off[ii][ks]);
  };
  auto read_b = [&](int tt, int bb) __attribute__((always_inline)) {
    const unsigned char* t = smem + (tt & 1) * TILE + (2 + bb) * HALF;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bfr[bb][ks] = *reinterpret_cast<const bf16x8*>(t + b_off[ks]);
  };
  auto mma = [&](int a, int bb) __attribute__((always_inline)) {
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
        acc[a * 2 + ii][bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[bb][ks], af[ii][ks], acc[a * 2 + ii][bb], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
  using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;

  const int nk = K / BK;
  stage(0, K0{}); stage(0, K1{}); stage(0, K2{}); stage(0, K3{});
  if (nk >= 2) {
    stage(1, K0{}); stage(1, K1{});
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();            // group 1 runs one barrier behind
  for (int t = 0; t < nk; ++t) {
    // phase 1
    read_a(t, 0); read_b(t, 0);
    if (t + 1 < nk) stage(t + 1, K2{});
    __builtin_amdgcn_sched_barrier(0);
    mma(0, 0);
    // phase 2
    read_b(t, 1);
    if (t + 1 < nk) stage(t + 1, K3{});
    __builtin_amdgcn_sched_barrier(0);
    mma(0, 1);
    // phase 3
    read_a(t, 1);
    if (t + 2 < nk) stage(t + 2, K0{});
    __builtin_amdgcn_sched_barrier(0);
    mma(1, 1);
    // phase 4: K tile t+1 must have landed before anyone reads it in the next phase 1
    if (t + 2 < nk) {
      stage(t + 2, K1{});
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    mma(1, 0);
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
  // every wave has passed its last phase: the tile buffers are free for the staged store
  glds_store_tile<BM, BN, 128, 64, 4, 2, BN, 512, 1>(acc, p, m0, n0, wr, wc, lane, tid,
                                                     reinterpret_cast<uint16_t*>(smem));
  __syncthreads();                                        // staging area read out: the next tile may stream in
  }
  gemm_ts_exit(p);
}

// ------------------------------------------------------------- register-staged kernel (any dtype, any K)
template <typename T, typename OutT, int BM, int BN, int WAVES_M, int WAVES_N, int PF>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (WAVES_M * WAVES_N) / 4 * (BM * BN >= 256 * 128 ? 1 : 2))
void gemm_nt_kernel(GemmArgs p) {
  using M_ = Mma<T>;
  constexpr int NT = 64 * WAVES_M * WAVES_N;    // threads per workgroup
  constexpr int BK = M_::BK, STRIDE = M_::STRIDE, VEC = Elem<T>::VEC;
  constexpr int CPR = BK / VEC;                 // 16-byte chunks per tile row (= 8)
  constexpr int CHA = BM * CPR / NT, CHB = BN * CPR / NT;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;   // per-wave tile
  constexpr int MI = WM / 32, NI = WN / 32;
  static_assert(PF == 2 || PF == 4, "prefetch depth 2 or 4 (even: LDS double-buffer parity)");

  __shared__ __attribute__((aligned(16))) T As[2][BM * STRIDE];
  __shared__ __attribute__((aligned(16))) T Bs[2][BN * STRIDE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  int M = p.M;
  if (p.m_dev) { int md = *p.m_dev; M = md < M ? md : M; }
  const int N = p.N, K = p.K;

  // XCD-aware tile mapping: hardware places workgroup b on XCD b % 8; give every XCD a
  // contiguous range of tiles (row-major over (m-tile, n-tile)) so that neighbouring tiles,
  // which share an A row-panel, hit the same per-XCD L2.  Bijective for any grid size.
  const int tiles_n = (N + BN - 1) / BN;
  const int nwg = gridDim.x;
  int tile_id;
  {
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  // grouped traversal inside the XCD's contiguous range: GROUP_M m-tiles share one sweep over n, so the
  // ~64 workgroups resident on an XCD touch 8 A panels + 8 B panels (<= 4 MiB L2) instead of streaming
  // the whole B matrix once per pair of m-tiles (PMC: FETCH_SIZE 7.5x the operand bytes before).
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
  if (tile_id >= ((M + BM - 1) / BM) * tiles_n) return;                          // uniform per block

  const T* A = static_cast<const T*>(p.A);
  const T* B = static_cast<const T*>(p.B);

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // PF register stages: tiles kt .. kt+PF-1 are in flight from HBM/L2 while tile kt is multiplied
  u32x4 ra[PF][CHA], rb[PF][CHB];
  // Pipeline discipline (keeps hipcc's s_waitcnt vmcnt COUNTED instead of vmcnt(0)):
  //  * every stage issues its global loads unconditionally, in straight-line code, from a clamped
  //    in-range address (no divergent branch, no uniform guard, plain global_load);
  //  * the loaded registers have no consumer until SSTORE, where out-of-range chunks are replaced
  //    by zeros (select at the point where the data is needed anyway);
  //  * steps past the last K tile run on all-zero tiles (they add 0 to the accumulators).
  const int Mc = M - 1, Nc = N - 1;
#define GLOAD(KT, S)                                                                              \
  {                                                                                               \
    _Pragma("unroll") for (int i = 0; i < CHA; ++i) {                                             \
      const int c = tid + i * NT, row = c / CPR, ch = c % CPR;                                   \
      const int gm = m0 + row, gk = (KT) * BK + ch * VEC;                                         \
      ra[S][i] = *reinterpret_cast<const u32x4*>(A + (long)(gm < M ? gm : Mc) * p.lda + (gk < K ? gk : 0)); \
    }                                                                                             \
    _Pragma("unroll") for (int i = 0; i < CHB; ++i) {                                             \
      const int c = tid + i * NT, row = c / CPR, ch = c % CPR;                                   \
      const int gn = n0 + row, gk = (KT) * BK + ch * VEC;                                         \
      rb[S][i] = *reinterpret_cast<const u32x4*>(B + (long)(gn < N ? gn : Nc) * p.ldb + (gk < K ? gk : 0)); \
    }                                                                                             \
  }
#define SSTORE(S, BUF, KT)                                                                        \
  {                                                                                               \
    _Pragma("unroll") for (int i = 0; i < CHA; ++i) {                                             \
      const int c = tid + i * NT, row = c / CPR, ch = c % CPR;                                   \
      const bool ok = (m0 + row) < M && ((KT) * BK + ch * VEC) < K;                               \
      M_::store_chunk(As[BUF], row, ch, ok ? ra[S][i] : zero4);                                   \
    }                                                                                             \
    _Pragma("unroll") for (int i = 0; i < CHB; ++i) {                                             \
      const int c = tid + i * NT, row = c / CPR, ch = c % CPR;                                   \
      const bool ok = (n0 + row) < N && ((KT) * BK + ch * VEC) < K;                               \
      M_::store_chunk(Bs[BUF], row, ch, ok ? rb[S][i] : zero4);                                   \
    }                                                                                             \
  }

  const int nk = (K + BK - 1) / BK;
  const int nk_pad = (nk + PF - 1) / PF * PF;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  // one pipeline step with a LITERAL stage index S (register stages must be statically indexed)
#define STEP(S)                                                                                   \
  {                                                                                               \
    const int kt = kt0 + (S);                                                                     \
    GLOAD(kt + PF, S)                                                                             \
    const T* at = As[(S) & 1] + (wm * WM) * STRIDE;                                               \
    const T* bt = Bs[(S) & 1] + (wn * WN) * STRIDE;                                               \
    _Pragma("unroll") for (int ks = 0; ks < BK; ks += M_::KSTEP) {                                \
      typename M_::frag a[MI], b[NI];                                                             \
      _Pragma("unroll") for (int i = 0; i < MI; ++i) a[i] = M_::load(at, i * 32 + (lane & 31), ks, lane); \
      _Pragma("unroll") for (int j = 0; j < NI; ++j) b[j] = M_::load(bt, j * 32 + (lane & 31), ks, lane); \
      _Pragma("unroll") for (int i = 0; i < MI; ++i)                                              \
        _Pragma("unroll") for (int j = 0; j < NI; ++j) acc[i][j] = M_::mma(b[j], a[i], acc[i][j]); \
    }                                                                                             \
    SSTORE(((S) + 1) % PF, ((S) + 1) & 1, kt + 1)                                                 \
    __syncthreads();                                                                              \
  }
  GLOAD(0, 0)
  GLOAD(1, 1)
  if constexpr (PF == 4) {
    GLOAD(2, 2)
    GLOAD(3, 3)
  }
  SSTORE(0, 0, 0)
  __syncthreads();
  for (int kt0 = 0; kt0 < nk_pad; kt0 += PF) {
    STEP(0)
    STEP(1)
    if constexpr (PF == 4) {
      STEP(2)
      STEP(3)
    }
  }
#undef STEP
#undef GLOAD
#undef SSTORE

  if constexpr (sizeof(T) == 2 && sizeof(OutT) == 2) {
    static_assert(BM * (BN + 8) * 2 + NT * 4 <= (int)sizeof(As), "staged epilogue must fit the A tile buffers");
    if (p.stat_mean) {                                   // block-uniform; the last loop barrier freed As
      staged_store_stats<BM, BN, WM, WN, MI, NI, NT>(acc, p, m0, n0, tm, wm, wn, lane, tid,
                                                     reinterpret_cast<uint16_t*>(&As[0][0]), M, N);
      return;
    }
  }
  gemm_epilogue<OutT, MI, NI>(acc, p, m0 + wm * WM, n0 + wn * WN, lane, M, N);
}

// ------------------------------------------------------------- K-major operands (bf16): TN / NN forms
// Backward GEMMs read activations and weights exactly as the forward pass left them:
//   wgrad  dW[n][k] = sum_t dY[t][n] * X[t][k]   (both operands K-major: the reduction index is the ROW)
//   dgrad  dX[t][k] = sum_n dY[t][n] * W[n][k]   (B operand K-major)
// A K-major tile is staged row-major [k][m] (the global layout, 16-byte chunks along m) with a row stride
// of (BM*2 + 64) bytes, and the MFMA fragments (8 consecutive k for one m per lane) come out of
// ds_read_b64_tr_b16, gfx950's LDS transpose read: lane i of a 16-lane group supplies the address of row
// i>>2, 8-byte piece i&3 of a [4 k][16 m] block and receives column i (probe: tools/probes/tr_read_probe.hip).
// Two reads (k 0-3, 4-7) make one bf16x8 fragment.  The 64-byte row skew puts the 4 rows of a 32-lane access
// on disjoint bank quarters.  No separate transpose kernels, no second copy of the operands in HBM.
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
__device__ __forceinline__ bf16x8 tr_frag(const uint16_t* p0, const uint16_t* p1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)p1);
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// The body takes its workgroup coordinates as arguments: (block, n_blocks) along the tile axis and (split, n_splits)
// along K - the plain kernel passes blockIdx / gridDim, the grouped kernel the position inside one problem's range.
template <typename OutT, int BM, int BN, int WAVES_M, int WAVES_N, bool TA, bool TB, int PF>
__device__ __forceinline__ void gemm_tx_body(const GemmArgs& p, const int block, const int n_blocks, const int split,
                                             const int n_splits) {
  using T = uint16_t;
  constexpr int NT = 64 * WAVES_M * WAVES_N, BK = 64;
  static_assert(PF == 2 || PF == 4, "prefetch depth");
  constexpr int SA = TA ? BM + 32 : 72, SB = TB ? BN + 32 : 72;      // LDS row strides (elements)
  constexpr int ROWS_A = TA ? BK : BM, ROWS_B = TB ? BK : BN;
  constexpr int CPA = TA ? BM / 8 : 8, CPB = TB ? BN / 8 : 8;        // 16-byte chunks per LDS row
  constexpr int CHA = BM * 8 / NT, CHB = BN * 8 / NT;                // chunks per thread per tile
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MI = WM / 32, NI = WN / 32;
  __shared__ __attribute__((aligned(16))) T As[2][ROWS_A * SA];
  __shared__ __attribute__((aligned(16))) T Bs[2][ROWS_B * SB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  int M = p.M;
  if (p.m_dev) { int md = *p.m_dev; M = md < M ? md : M; }
  const int N = p.N;
  // split K (gridDim.y > 1): this workgroup reduces rows [kb, K) of its slice only and adds its partial tile to the
  // fp32 output with float atomics (p.atomic_out)
  int Kall = p.K;
  if (p.k_dev) { const int kd = *p.k_dev; Kall = kd < Kall ? kd : Kall; }      // device-side reduction length (block-uniform)
  const int kper = ((Kall + BK - 1) / BK + n_splits - 1) / n_splits * BK;
  const int kb = split * kper;
  const int K = Kall < kb + kper ? Kall : kb + kper;
  if (kb >= K) return;
  const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  int tile_id;
  {
    const int orig = block, xcd = orig & 7, q = n_blocks >> 3, r = n_blocks & 7;
    tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  if (tile_id >= tiles_m * tiles_n) return;
  const int tm = tile_id / tiles_n, tn = tile_id % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const T* A = static_cast<const T*>(p.A);
  const T* B = static_cast<const T*>(p.B);

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4 ra[PF][CHA], rb[PF][CHB];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  // fused column sums of A (bias gradient): the n-tile-0 workgroup of every m range adds up the chunks it stages
  const bool want_asum = TA && p.asum != nullptr && tn == 0;          // block-uniform
  float csum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) csum[e] = 0.f;
  // chunk c of an operand tile: (LDS row, 16-byte chunk in the row).  Normal operand: row = m, chunk = k/8;
  // K-major operand: row = k, chunk = m/8.  Loads are unconditional from clamped in-range addresses, the
  // zero fill for k >= K happens at the LDS store (same pipeline discipline as gemm_nt_kernel).
#define TX_GLOAD(KT, S)                                                                            \
  {                                                                                                \
    _Pragma("unroll") for (int i = 0; i < CHA; ++i) {                                              \
      const int c = tid + i * NT, row = c / CPA, ch = c % CPA;                                     \
      if constexpr (TA) {                                                                          \
        const int gk = kb + (KT) * BK + row, gm = m0 + ch * 8;                                          \
        ra[S][i] = *reinterpret_cast<const u32x4*>(A + (long)(gk < K ? gk : 0) * p.lda + (gm < M ? gm : 0)); \
      } else {                                                                                     \
        const int gm = m0 + row, gk = kb + (KT) * BK + ch * 8;                                          \
        ra[S][i] = *reinterpret_cast<const u32x4*>(A + (long)(gm < M ? gm : M - 1) * p.lda + (gk < K ? gk : 0)); \
      }                                                                                            \
    }                                                                                              \
    _Pragma("unroll") for (int i = 0; i < CHB; ++i) {                                              \
      const int c = tid + i * NT, row = c / CPB, ch = c % CPB;                                     \
      if constexpr (TB) {                                                                          \
        const int gk = kb + (KT) * BK + row, gn = n0 + ch * 8;                                          \
        rb[S][i] = *reinterpret_cast<const u32x4*>(B + (long)(gk < K ? gk : 0) * p.ldb + (gn < N ? gn : 0)); \
      } else {                                                                                     \
        const int gn = n0 + row, gk = kb + (KT) * BK + ch * 8;                                          \
        rb[S][i] = *reinterpret_cast<const u32x4*>(B + (long)(gn < N ? gn : N - 1) * p.ldb + (gk < K ? gk : 0)); \
      }                                                                                            \
    }                                                                                              \
  }
#define TX_SSTORE(S, BUF, KT)                                                                      \
  {                                                                                                \
    _Pragma("unroll") for (int i = 0; i < CHA; ++i) {                                              \
      const int c = tid + i * NT, row = c / CPA, ch = c % CPA;                                     \
      const bool ok = TA ? (kb + (KT) * BK + row) < K : ((m0 + row) < M && (kb + (KT) * BK + ch * 8) < K);   \
      const u32x4 va = ok ? ra[S][i] : zero4;                                                      \
      *reinterpret_cast<u32x4*>(As[BUF] + row * SA + ch * 8) = va;                                 \
      if (want_asum) {                                                                             \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                            \
          csum[2 * e] += __uint_as_float(va[e] << 16);                                             \
          csum[2 * e + 1] += __uint_as_float(va[e] & 0xffff0000u);                                 \
        }                                                                                          \
      }                                                                                            \
    }                                                                                              \
    _Pragma("unroll") for (int i = 0; i < CHB; ++i) {                                              \
      const int c = tid + i * NT, row = c / CPB, ch = c % CPB;                                     \
      const bool ok = TB ? (kb + (KT) * BK + row) < K : ((n0 + row) < N && (kb + (KT) * BK + ch * 8) < K);   \
      *reinterpret_cast<u32x4*>(Bs[BUF] + row * SB + ch * 8) = ok ? rb[S][i] : zero4;              \
    }                                                                                              \
  }
  // per-lane fragment origins
  //  normal  : row (lane&31) of the wave tile, k chunk 8*(lane>>5)
  //  K-major : tr-read address of this lane: k row 8*(lane>>5) + ((lane&15)>>2), m col 16*((lane>>4)&1) + 4*(lane&3)
  const int a_off = TA ? (8 * (lane >> 5) + ((lane & 15) >> 2)) * SA + wm * WM + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)
                       : (wm * WM + (lane & 31)) * SA + 8 * (lane >> 5);
  const int b_off = TB ? (8 * (lane >> 5) + ((lane & 15) >> 2)) * SB + wn * WN + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)
                       : (wn * WN + (lane & 31)) * SB + 8 * (lane >> 5);
#define TX_STEP(S)                                                                                 \
  {                                                                                                \
    const int kt = kt0 + (S);                                                                      \
    TX_GLOAD(kt + PF, S)                                                                           \
    const T* at = As[(S) & 1] + a_off;                                                             \
    const T* bt = Bs[(S) & 1] + b_off;                                                             \
    _Pragma("unroll") for (int ks = 0; ks < BK; ks += 16) {                                        \
      bf16x8 a[MI], b[NI];                                                                         \
      _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                             \
        if constexpr (TA) a[i] = tr_frag(at + ks * SA + i * 32, at + (ks + 4) * SA + i * 32);      \
        else a[i] = *reinterpret_cast<const bf16x8*>(at + i * 32 * SA + ks);                       \
      }                                                                                            \
      _Pragma("unroll") for (int j = 0; j < NI; ++j) {                                             \
        if constexpr (TB) b[j] = tr_frag(bt + ks * SB + j * 32, bt + (ks + 4) * SB + j * 32);      \
        else b[j] = *reinterpret_cast<const bf16x8*>(bt + j * 32 * SB + ks);                       \
      }                                                                                            \
      _Pragma("unroll") for (int i = 0; i < MI; ++i)                                               \
        _Pragma("unroll") for (int j = 0; j < NI; ++j)                                             \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);     \
    }                                                                                              \
    TX_SSTORE(((S) + 1) % PF, ((S) + 1) & 1, kt + 1)                                               \
    __syncthreads();                                                                               \
  }
  const int nk = (K - kb + BK - 1) / BK;
  const int nk_pad = (nk + PF - 1) / PF * PF;
  TX_GLOAD(0, 0)
  TX_GLOAD(1, 1)
  if constexpr (PF == 4) {
    TX_GLOAD(2, 2)
    TX_GLOAD(3, 3)
  }
  TX_SSTORE(0, 0, 0)
  __syncthreads();
  for (int kt0 = 0; kt0 < nk_pad; kt0 += PF) {
    TX_STEP(0)
    TX_STEP(1)
    if constexpr (PF == 4) {
      TX_STEP(2)
      TX_STEP(3)
    }
  }
#undef TX_STEP
#undef TX_GLOAD
#undef TX_SSTORE
  if constexpr (TA) {
    if (want_asum) {
      // every chunk of thread t covers columns (t % CPA) * 8 .. + 7 of the m range (NT % CPA == 0): fold the
      // NT / CPA threads of a column chunk through LDS (the last loop barrier made the tile buffers free)
      static_assert(NT % CPA == 0 && NT * 9 * 4 <= (int)sizeof(As), "column-sum scratch");
      float* red = reinterpret_cast<float*>(&As[0][0]);
#pragma unroll
      for (int e = 0; e < 8; ++e) red[tid * 9 + e] = csum[e];
      __syncthreads();
      if (tid < BM && m0 + tid < M) {
        const int ch = tid >> 3, e = tid & 7;
        float s = 0.f;
        for (int t = ch; t < NT; t += CPA) s += red[t * 9 + e];
        if (n_splits > 1) unsafeAtomicAdd(p.asum + m0 + tid, p.asum_scale * s);
        else p.asum[m0 + tid] += p.asum_scale * s;
      }
    }
  }
  gemm_epilogue<OutT, MI, NI>(acc, p, m0 + wm * WM, n0 + wn * WN, lane, M, N);
}
template <typename OutT, int BM, int BN, int WAVES_M, int WAVES_N, bool TA, bool TB, int PF>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 2) void gemm_tx_kernel(GemmArgs p) {
  gemm_tx_body<OutT, BM, BN, WAVES_M, WAVES_N, TA, TB, PF>(p, blockIdx.x, gridDim.x, blockIdx.y, gridDim.y);
}

// ------------------------------------------------------------- grouped GEMMs
// The decoder works on M = T*B = 1024 rows: its GEMMs are 256-512 tiles of 64x64 with 16 dependent K steps - one
// workgroup per CU, latency rather than throughput (200-250 TFLOP/s) - and every launch costs ~4.5 us of dispatch on
// top.  Independent products therefore travel together: the ~80 weight gradients of a backward pass (queued until it
// is over, ops.py), the query / output projections of a layer's four context attentions, their input gradients.  One
// launch carries up to GROUP_MAX problems BY VALUE in its kernel arguments; a workgroup finds its problem by walking the
// (8-aligned, so a block's XCD phase is its problem-local one) block prefix sums, which live in scalar registers.
// Several workgroups per CU then hide each other's load latency.
#define GROUP_MAX 24
struct GroupProblem {
  const void* A; const void* B; void* C; float* asum; const float* bias; const int* m_dev; const int* k_dev;
  long lda, ldb, ldc;
  int M, N, K, accumulate, bias_mode, act;
  float alpha, asum_scale;
};
struct GemmGroup {
  GroupProblem pr[GROUP_MAX];
  int start[GROUP_MAX + 1];
  int n;
};
__device__ __forceinline__ int group_find(const GemmGroup& g, int b) {
  int i = 0;
  while (i + 1 < g.n && b >= g.start[i + 1]) ++i;            // block-uniform
  return i;
}
__device__ __forceinline__ GemmArgs group_args(const GroupProblem& q) {
  GemmArgs p;
  p.A = q.A; p.B = q.B; p.C = q.C; p.bias = q.bias; p.aux = nullptr; p.m_dev = q.m_dev; p.k_dev = q.k_dev;
  p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc; p.M = q.M; p.N = q.N; p.K = q.K;
  p.bias_mode = q.bias_mode; p.act = q.act; p.accumulate = q.accumulate; p.alpha = q.alpha;
  p.asum = q.asum; p.asum_scale = q.asum_scale; p.ts = nullptr; p.atomic_out = 0; p.conv_zero = nullptr; p.queue = nullptr;
  p.stat_mean = nullptr; p.stat_m2 = nullptr;
  return p;
}
template <typename OutT, int BM, int BN, bool TA, bool TB, int PF>
__global__ __launch_bounds__(256, 2) void gemm_tx_group_kernel(GemmGroup g) {
  const int i = group_find(g, blockIdx.x);
  const GemmArgs p = group_args(g.pr[i]);
  gemm_tx_body<OutT, BM, BN, 2, 2, TA, TB, PF>(p, blockIdx.x - g.start[i], g.start[i + 1] - g.start[i], 0, 1);
}
// 256x128 tiles, 8 waves (4 x 2): for the long reductions (the article K|V weight gradients: 16384 rows into
// [2048, 1024]) the 128x128 body is bound by operand re-reads through L2; the taller tile halves the re-reads of B
template <typename OutT, bool TA, bool TB>
__global__ __launch_bounds__(512, 1) void gemm_tx_group_wide_kernel(GemmGroup g) {
  const int i = group_find(g, blockIdx.x);
  const GemmArgs p = group_args(g.pr[i]);
  gemm_tx_body<OutT, 256, 128, 4, 2, TA, TB, 2>(p, blockIdx.x - g.start[i], g.start[i + 1] - g.start[i], 0, 1);
}
template <typename OutT, int BM, int BN, int NS = 2>
__global__ __launch_bounds__(256) void gemm_nt_group_kernel(GemmGroup g) {
  const int i = group_find(g, blockIdx.x);
  const GemmArgs p = group_args(g.pr[i]);
  gemm_nt_glds_body<OutT, BM, BN, 2, 2, false, NS>(p, blockIdx.x - g.start[i], g.start[i + 1] - g.start[i]);
}
// LDS stages of the 64x64 direct-to-LDS tile (see gemm_nt_glds_body): 4 = three K tiles in flight (64 KB of LDS, two
// workgroups per CU), 2 = the two-stage form (32 KB, five per CU).  The ring pays where the launch is a short chain of
// round trips - at most two workgroups per CU and a reduction of >= 8 K tiles (tools/bench_conv.py, MI355X, B = 32:
// layer4 conv1 24.5 -> 18.4 us, conv2 53.3 -> 38.3; ResNet-152 5.78 -> 5.50 ms) - and LOSES where many workgroups per
// CU already hide each other's latency (layer1 conv3, K = 64: 43 -> 61 us; layer3 conv3, K = 256: 18.0 -> 24.1).
// TELL_GEMM_RING = 2 / 3 / 4 forces a depth everywhere (A/B aid).
static int ring_env() { return (int)tell_opt(OPT_GEMM_RING); }   // (read per call: tools/bench_conv.py switches it inside one process)
static int small_ring_stages(long tiles64, int K) {
  const int v = ring_env();
  if (v) return v;
  return (tiles64 <= 512 && K >= 512) ? 4 : 2;
}

template <typename OutT, bool TA, bool TB>
static int launch_gemm_tx(const GemmArgs& a_in, hipStream_t stream) {
  GemmArgs a = a_in;
  auto tiles = [&](int bm, int bn) { return (long)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn); };
  const int split_env = (int)tell_opt(OPT_GEMM_SPLITK);   // tuning aid: <= 0 = off
  if constexpr (std::is_same<OutT, float>::value) {
    // Opt-in (TELL_GEMM_SPLITK=n): long reductions into a small fp32 output that is accumulated anyway (weight gradients
    // of the context K/V projections: K = S*B = 16384 rows into [2048, 1024]) split over gridDim.y, partial tiles added
    // with float atomics.  Measured on MI355X it LOSES: 216 us against 139 us for the plain 64x64-tile launch of that
    // shape (6.3 M L2 atomics), so it stays off by default.
    const long t128 = tiles(128, 128);
    if (a.accumulate && !a.m_dev && a.act == 0 && a.bias_mode == 0 && t128 < 256 && a.K >= 2048 && split_env > 0) {
      int splits = (int)((384 + t128 - 1) / t128);
      while (splits > 1 && a.K / splits < 1024) --splits;
      if (split_env > 0) splits = split_env;
      if (splits > 1) {
        a.atomic_out = 1;
        hipLaunchKernelGGL((gemm_tx_kernel<OutT, 128, 128, 2, 2, TA, TB, 2>), dim3((unsigned)t128, (unsigned)splits), dim3(256), 0, stream, a);
        return tell_check_launch("gemm_tx_splitk");
      }
    }
  }
  if (tiles(128, 128) >= 256)
    hipLaunchKernelGGL((gemm_tx_kernel<OutT, 128, 128, 2, 2, TA, TB, 2>), dim3((unsigned)tiles(128, 128)), dim3(256), 0, stream, a);
  else   // few workgroups, latency bound: 4 K tiles in flight
    hipLaunchKernelGGL((gemm_tx_kernel<OutT, 64, 64, 2, 2, TA, TB, 4>), dim3((unsigned)tiles(64, 64)), dim3(256), 0, stream, a);
  return tell_check_launch("gemm_tx");
}

// Which kernel a call runs is decided here and only here: every launch goes through TELL_GEMM_LAUNCH, which records a
// readable label; tell_gemm_nt_plan() runs the same decision with the launch suppressed (bench.py's roofline block
// names kernels by asking, not by mirroring the heuristics).
// Tile counters of the resident 256x256 GEMM launches (gemm_pp2.hip, gemm_q4.hip, gemm_q4e.hip: one counter per XCD): a
// caller-owned, zero-initialised int32 buffer PER DEVICE (tell_gemm_set_tile_queue registers it for the calling thread's
// current device), cut into slots of 8 counters.  A launch leaves its counters zero, so a slot may be reused by any
// launch that cannot overlap it in time.  Launches recorded into a hipGraph keep their slot for the graph's lifetime and
// replay concurrently with whatever the other streams run: they take slots from the FIRST half of the buffer - first
// from the free list (slots a destroyed graph gave back, tell_gemm_tile_queue_release), then fresh ones - and, while the
// capturing thread has a log armed (tell_gemm_tile_queue_log_begin / _end), the slot numbers are recorded so that the
// graph's owner can release them when it drops the graph.  Eager launches walk a ring over the second half (thousands
// of launches deep).  Hand-out is atomic / mutex-protected: the loader, encoder and training threads of one process (and
// the processes of a data-parallel job, each with its own device) launch concurrently - under data parallelism the
// per-XCD counters are the DEFAULT path (training/trainer.py), so this state must not be "last registration wins".
namespace {
constexpr int TQ_MAX_DEV = 64, TQ_WORDS = 8;
struct TileQueueDev {
  int* base = nullptr;
  unsigned slots = 0;                       // 8-counter slots in the buffer; [0, slots / 2) captured, [slots / 2, slots) eager ring
  std::atomic<unsigned> ring{0};
  std::mutex mu;                            // guards fresh / free_list
  unsigned fresh = 0;
  std::vector<unsigned> free_list;
};
TileQueueDev g_tq[TQ_MAX_DEV];
thread_local std::vector<int>* t_tq_log = nullptr;
thread_local std::vector<int> t_tq_log_store;
inline TileQueueDev* tq_current() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return (d >= 0 && d < TQ_MAX_DEV) ? &g_tq[d] : nullptr;
}
}  // namespace
extern "C" int tell_gemm_set_tile_queue(void* counters, int n, hipStream_t) {
  TileQueueDev* q = tq_current();
  TELL_REQUIRE(q != nullptr, "gemm_set_tile_queue: no current device");
  std::lock_guard<std::mutex> lock(q->mu);
  q->base = (counters && n >= 4 * TQ_WORDS) ? static_cast<int*>(counters) : nullptr;
  q->slots = q->base ? (unsigned)n / TQ_WORDS : 0u;
  q->fresh = 0;
  q->free_list.clear();
  q->ring.store(0);
  return TELL_OK;
}
extern "C" int tell_gemm_tile_queue_log_begin(void) {
  t_tq_log_store.clear();
  t_tq_log = &t_tq_log_store;
  return TELL_OK;
}
// -> number of slots the calling thread's captured launches have taken since _log_begin (the log stays open): lets the
// caller size the buffer it hands to _log_end
extern "C" int tell_gemm_tile_queue_log_count(void) { return t_tq_log ? (int)t_tq_log->size() : 0; }
// -> number of slots the calling thread's captured launches took since _log_begin (at most `cap` of them copied to `out`)
extern "C" int tell_gemm_tile_queue_log_end(int* out, int cap) {
  const int n = t_tq_log ? (int)t_tq_log->size() : 0;
  for (int i = 0; i < n && i < cap; ++i) out[i] = (*t_tq_log)[i];
  t_tq_log = nullptr;
  return n;
}
extern "C" int tell_gemm_tile_queue_release(const int* slots, int n) {
  TileQueueDev* q = tq_current();
  if (!q || n <= 0) return TELL_OK;
  std::lock_guard<std::mutex> lock(q->mu);
  for (int i = 0; i < n; ++i)
    if (slots[i] >= 0 && (unsigned)slots[i] < q->slots / 2) q->free_list.push_back((unsigned)slots[i]);
  return TELL_OK;
}
// state of the calling thread's device: out[0] slots, out[1] fresh captured slots handed out, out[2] free-list length
extern "C" int tell_gemm_tile_queue_stats(int* out) {
  TileQueueDev* q = tq_current();
  out[0] = out[1] = out[2] = 0;
  if (!q) return TELL_OK;
  std::lock_guard<std::mutex> lock(q->mu);
  out[0] = (int)q->slots; out[1] = (int)q->fresh; out[2] = (int)q->free_list.size();
  return TELL_OK;
}
// `words` (<= 8) consecutive zeroed counters for one launch, or NULL (-> static tile lists) when no buffer is registered
// for the device / the capture half is used up
int* gemm_tile_queue_slot(int words, hipStream_t stream) {
  TileQueueDev* q = tq_current();
  if (!q || !q->base || words > TQ_WORDS) return nullptr;
  const unsigned half = q->slots / 2;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &st) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (st != hipStreamCaptureStatusNone) {
    unsigned slot;
    {
      std::lock_guard<std::mutex> lock(q->mu);
      if (!q->free_list.empty()) { slot = q->free_list.back(); q->free_list.pop_back(); }
      else if (q->fresh < half) slot = q->fresh++;
      else return nullptr;
    }
    if (t_tq_log) t_tq_log->push_back((int)slot);
    return q->base + (size_t)slot * TQ_WORDS;
  }
  return q->base + (size_t)(half + q->ring.fetch_add(1) % half) * TQ_WORDS;
}
static thread_local char g_gemm_label[96] = "";
static thread_local bool g_gemm_plan = false;
static const char* gemm_label(const char* base, int in_bf16, int out_bf16, int bm, int bn) {
  if (in_bf16 < 0) snprintf(g_gemm_label, sizeof(g_gemm_label), "%s<%s,%d,%d>", base, out_bf16 ? "bf16" : "f32", bm, bn);
  else snprintf(g_gemm_label, sizeof(g_gemm_label), "%s<%s,%s,%d,%d>", base, in_bf16 ? "bf16" : "f32", out_bf16 ? "bf16" : "f32", bm, bn);
  return g_gemm_label;
}
#define TELL_GEMM_LAUNCH(label, kern, grid, block) \
  do { if (g_gemm_plan) (void)(label); else hipLaunchKernelGGL(kern, grid, block, 0, stream, a); } while (0)

template <typename T, typename OutT>
static int launch_gemm(const GemmArgs& a, hipStream_t stream, int* bm_used = nullptr) {
  int bm_dummy;
  if (!bm_used) bm_used = &bm_dummy;
  // Tile choice: the kernel is bound by operand re-reads from L2 (flop/byte of a tile =
  // BM*BN/(BM+BN) per 2-byte element), so take the largest tile that still gives every CU work.
  auto tiles = [&](int bm, int bn) { return (long)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn); };
  if constexpr (sizeof(T) == 2) {
    if (a.K % 64 == 0 && tiles(128, 128) >= 256) {       // direct-to-LDS path
      const int force_env = (int)tell_opt(OPT_GEMM_TILE);   // tuning aid
      const bool no_pp = force_env == 7;                  // 7: the default choice without the ping-pong kernel (A/B)
      const int force = no_pp ? 0 : force_env;
      static const int n_cu = [] { hipDeviceProp_t pr; int d = 0; (void)hipGetDevice(&d); (void)hipGetDeviceProperties(&pr, d); return pr.multiProcessorCount; }();
      // 256x192 (8 waves as 4x2, 64x96 per wave): the QKV projection (N = 3072) quantises to whole rounds with it
      if ((force == 0 || force == 9) && a.N % 192 == 0 && tiles(256, 192) % n_cu == 0 && a.K <= 2048 && !a.accumulate &&
          !a.stat_mean && tiles(256, 256) % n_cu != 0) {
        TELL_GEMM_LAUNCH(gemm_label("gemm_nt_glds_kernel", -1, sizeof(OutT) == 2, 256, 192), (gemm_nt_glds_kernel<OutT, 256, 192, 4, 2>), dim3((unsigned)tiles(256, 192)), dim3(512));
        return g_gemm_plan ? TELL_OK : tell_check_launch("gemm_nt_glds");
      }
      if constexpr (std::is_same<OutT, uint16_t>::value) {
        const bool full = a.M % 256 == 0 && a.N % 256 == 0 && !a.accumulate && a.act != 3 && a.act != 4 && !a.m_dev && !a.stat_mean &&
                          (a.ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(a.C) & 15) == 0 &&
                          (a.bias_mode != 1 || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0);
        // ping-pong 256x256: whole rounds of full tiles (fc1 of RoBERTa: 878 vs 838 TFLOP/s, 4096^3: 1171 vs 1022,
        // 8192^3: 1333 vs 1164); partial rounds lose to the smaller tiles below.  TELL_GEMM_TILE=8 forces it.
        // four waves x 128x128 (gemm_q4.hip): what it needs beyond `full`
        const int q4_env = (int)tell_opt(OPT_GEMM_Q4);   // (per launch: A/B inside one process)
        const bool q4_takes = q4_env && (force == 0 || force == 8) && a.K % 128 == 0 && a.K >= 128 && a.K / 64 < 65536 && a.lda % 8 == 0 &&
                              a.ldb % 8 == 0 && !a.conv_zero && a.act <= 2 && (reinterpret_cast<uintptr_t>(a.A) & 15) == 0 &&
                              (reinterpret_cast<uintptr_t>(a.B) & 15) == 0 && 256L * a.lda * 2 < (1L << 31) && 256L * a.ldb * 2 < (1L << 31);
        // Partial rounds (variable-length batches: B x L = 12288 / 8192 / 4096 rows): the 256x256 ping-pong kernel lost
        // them to 128x128 tiles, q4 does not as long as the rounds it runs are at least 70 % full - at 12288 rows fc2 has
        // 192 tiles = 75 % of one round and still runs in the time of the full 16384-row launch (990 against 830 TFLOP/s
        // effective), qkv is 2.25 rounds in 3.  TELL_Q4_PARTIAL=0: whole rounds only.
        const bool q4_partial_env = tell_opt(OPT_Q4_PARTIAL) != 0;
        const long t256 = tiles(256, 256);
        const bool q4_partial = q4_partial_env && q4_takes && force == 0 && t256 * 10 >= ((t256 + n_cu - 1) / n_cu) * n_cu * 7;
        if (full && !no_pp && ((force == 0 && (t256 % n_cu == 0 || q4_partial)) || (force == 8 && t256 >= n_cu))) {
          // persistent form (tile queue registered, more than one tile per CU): one workgroup per CU pulls tiles
          // MEASURED (MI355X, same box A/B, configs[2]): 1406 / 1420 samples/s without, 1415 / 1404 with; alone qkv 130.2 ->
          // 125.6 us, the other shapes unchanged.  Holding the CU does not buy back the in-step slowdown - the other
          // streams' kernels are work that has to run somewhere - so the form stays opt-in (TELL_GEMM_PERSIST=1).
          // resident form with next-tile prefetch under the epilogue (gemm_pp2.hip) - the default.  MEASURED (MI355X, same
          // box A/B): RoBERTa layer GEMMs alone 442-444 -> 436 us; configs[2] 1500 / 1506 samples/s without, 1524 / 1526 with
          // (in-step launch 148-157 -> 122-139 us).  TELL_GEMM_PP2=0: one workgroup per tile (gemm_nt_pp_kernel), 1: only
          // launches of two rounds or more.
          // four waves x 128x128, hand-placed K loop (gemm_q4.hip) - the default since round 4.  MEASURED (MI355X, interleaved
          // inside one process, M = 16384): RoBERTa layer GEMMs 419-422 us (ping-pong) -> 364-365 us: qkv 968 -> 1124 TFLOP/s,
          // out 922 -> 993, fc1 + GELU 855 -> 1036, fc2 1197 -> 1313 (tools/probes/q4_variants.py).  TELL_GEMM_Q4=0: ping-pong.
          if (q4_takes) {
            // epilogue inside the next tile's K loop (gemm_q4e.hip): per-column bias, K >= 576.  TELL_GEMM_Q4E=0: plain q4
            const int q4e_env = (int)tell_opt(OPT_GEMM_Q4E);
            // MEASURED (static tile lists, M = 16384): qkv (3 tiles per workgroup) 90.2 -> 86.6 us; one tile per workgroup (out,
            // fc2) has no next K loop to hide anything in; with GELU the in-asm drain is slower than hipcc's (fc1 130.5 -> 135.0):
            // taken for act 0 / 1 with at least two tiles per workgroup (TELL_GEMM_Q4E=2: wherever it applies)
            const bool q4e_ok = q4e_env && a.bias_mode == 1 && a.K / 64 >= 13 && 512L * a.ldc < (1L << 31) && tiles(256, 256) % n_cu == 0 &&
                                (q4e_env == 2 || (a.act != 2 && tiles(256, 256) >= 2L * n_cu)) &&
                                !tell_probe(PROBE_Q4_ABL);
            if (g_gemm_plan) { (void)gemm_label(q4e_ok ? "gemm_nt_q4e_kernel" : "gemm_nt_q4_kernel", -1, 1, 256, 256); return TELL_OK; }
            if (q4e_ok) {
              const int rc = launch_gemm_q4e(a, stream, n_cu);
              if (rc <= 0) return rc;
            }
            return launch_gemm_q4(a, stream, n_cu);
          }
          const int pp2_env = (int)tell_opt(OPT_GEMM_PP2);
          if (pp2_env && force == 0 && tiles(256, 256) >= (pp2_env == 2 ? 1 : 2) * (long)n_cu && a.lda % 8 == 0 && a.ldb % 8 == 0 && !a.conv_zero) {
            if (g_gemm_plan) { (void)gemm_label("gemm_nt_pp2_kernel", -1, 1, 256, 256); return TELL_OK; }
            return launch_gemm_pp2(a, stream, n_cu);
          }
          const bool persist_env = tell_opt(OPT_GEMM_PERSIST) == 1;
          int* pq = (persist_env && !g_gemm_plan && tiles(256, 256) > n_cu) ? gemm_tile_queue_slot(1, stream) : nullptr;
          if (pq) {
            GemmArgs ap = a;
            ap.queue = pq;
            hipLaunchKernelGGL((gemm_nt_pp_kernel<OutT>), dim3((unsigned)n_cu), dim3(512), 0, stream, ap);
            return tell_check_launch("gemm_nt_pp");
          }
          TELL_GEMM_LAUNCH(gemm_label("gemm_nt_pp_kernel", -1, sizeof(OutT) == 2, 256, 256), (gemm_nt_pp_kernel<OutT>), dim3((unsigned)tiles(256, 256)), dim3(512));
          return g_gemm_plan ? TELL_OK : tell_check_launch("gemm_nt_pp");
        }
      }
      if (force == 5) {   // 256x256, 8 waves (128x64 per wave), 2-stage
        TELL_GEMM_LAUNCH(gemm_label("gemm_nt_glds_kernel", -1, sizeof(OutT) == 2, 256, 256), (gemm_nt_glds_kernel<OutT, 256, 256, 2, 4>), dim3((unsigned)tiles(256, 256)), dim3(512));
        return g_gemm_plan ? TELL_OK : tell_check_launch("gemm_nt_glds");
      }
      // lockstep 256x256 (8 waves, 1 workgroup/CU) for whole rounds with ragged edges / fp32 output, short K
      *bm_used = 128;
      if (force == 0 && tiles(256, 256) % n_cu == 0 && a.K <= 2048 && !a.accumulate && !a.stat_mean) {
        TELL_GEMM_LAUNCH(gemm_label("gemm_nt_glds_kernel", -1, sizeof(OutT) == 2, 256, 256), (gemm_nt_glds_kernel<OutT, 256, 256, 2, 4>), dim3((unsigned)tiles(256, 256)), dim3(512));
        return g_gemm_plan ? TELL_OK : tell_check_launch("gemm_nt_glds");
      }
      if (force == 2)     // 256x128 (8 waves, 1 workgroup/CU) ties 128x128 (2 workgroups/CU) on MI355X: opt-in only
        TELL_GEMM_LAUNCH(gemm_label("gemm_nt_glds_kernel", -1, sizeof(OutT) == 2, 256, 128), (gemm_nt_glds_kernel<OutT, 256, 128, 4, 2>), dim3((unsigned)tiles(256, 128)), dim3(512));
      else
        TELL_GEMM_LAUNCH(gemm_label("gemm_nt_glds_kernel", -1, sizeof(OutT) == 2, 128, 128), (gemm_nt_glds_kernel<OutT, 128, 128, 2, 2>), dim3((unsigned)tiles(128, 128)), dim3(256));
      return g_gemm_plan ? TELL_OK : tell_check_launch("gemm_nt_glds");
    }
  }
  // Small bf16 GEMMs (fewer than 256 tiles of 128x128: the decoder's M = T*B = 1024-row chain): the direct-to-LDS kernel
  // with 64x64 tiles.  Its 32 KB of LDS (the register-staged 64x64 kernel below takes 36.9 KB) lets a workgroup share a
  // CU with a 256x256 ping-pong workgroup of the RoBERTa stream (128 KB of the CU's 160 KB): the decoder chain then runs
  // UNDER the encoder GEMMs instead of taking turns with them at CU granularity (+0.5 % samples/s at configs[2]).  Not
  // for the decode step's M <= 128 rows: with a handful of workgroups the 4-deep register prefetch of the kernel below
  // wins (11.8 us against 20 us per launch).  TELL_GEMM_SMALL=0 restores the register-staged kernel everywhere (A/B).
  if constexpr (sizeof(T) == 2) {
    // fewer than 256 tiles of 128x128 and a reduction of 1024 or more: gemm_s64.hip (q / out / linear2 of the decoder 7.8 ->
    // 5.7 us, tap logits 7.9 -> 5.7; bit-identical sums).  TELL_GEMM_S64=0: the kernels below (A/B; read per launch)
    const int s64_env = (int)tell_opt(OPT_GEMM_S64);
    if (a.M >= 512 && a.K % 64 == 0 && !a.stat_mean && !a.conv_zero && (s64_env == 2 || (s64_env == 1 && a.K >= 1024)) &&
        gemm_s64_takes(a)) {
      if (g_gemm_plan) { *bm_used = 64; (void)gemm_label("gemm_nt_s64_kernel", -1, sizeof(OutT) == 2, 64, 64); return TELL_OK; }
      const int rc = launch_gemm_s64(a, stream, sizeof(OutT) == 4);
      if (rc <= 0) { *bm_used = 64; return rc; }
    }
    const bool small_glds = tell_opt(OPT_GEMM_SMALL) != 0;
    // (Round 5: an 8-stage ring for the decoder's one-round 1024 x 1024 GEMMs - 7 K tiles in flight, one workgroup per CU -
    //  measured neutral: K = 1024 7.8 -> 8.4 us, K = 4096 23.5 -> 22.4 us, decoder half 6.955 -> 6.953 ms; not kept.)
    // (K = 4096: 35.6 us against 23.7 us for the register-staged kernel below)
    if (small_glds && a.M >= 512 && a.K <= 2048 && a.K % 64 == 0 && !a.stat_mean &&
        (reinterpret_cast<uintptr_t>(a.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.B) & 15) == 0 &&
        (a.lda & 7) == 0 && (a.ldb & 7) == 0) {
      *bm_used = 64;
      const int ring = small_ring_stages(tiles(64, 64), a.K);
      if (ring == 4)
        TELL_GEMM_LAUNCH(gemm_label("gemm_nt_glds_kernel", -1, sizeof(OutT) == 2, 64, 64), (gemm_nt_glds_kernel<OutT, 64, 64, 2, 2, false, 4>), dim3((unsigned)tiles(64, 64)), dim3(256));
      else if (ring == 3)
        TELL_GEMM_LAUNCH(gemm_label("gemm_nt_glds_kernel", -1, sizeof(OutT) == 2, 64, 64), (gemm_nt_glds_kernel<OutT, 64, 64, 2, 2, false, 3>), dim3((unsigned)tiles(64, 64)), dim3(256));
      else
        TELL_GEMM_LAUNCH(gemm_label("gemm_nt_glds_kernel", -1, sizeof(OutT) == 2, 64, 64), (gemm_nt_glds_kernel<OutT, 64, 64, 2, 2>), dim3((unsigned)tiles(64, 64)), dim3(256));
      return g_gemm_plan ? TELL_OK : tell_check_launch("gemm_nt_glds");
    }
  }
  // (a 256x256 register-staged tile measured slower than 256x128 - 236 VGPRs, one workgroup per CU - and
  //  was removed)
  *bm_used = (sizeof(T) == 2 && tiles(256, 128) >= 256) ? 256 : tiles(128, 128) >= 256 ? 128 : 64;
  if (sizeof(T) == 2 && tiles(256, 128) >= 256) {
    TELL_GEMM_LAUNCH(gemm_label("gemm_nt_kernel", sizeof(T) == 2, sizeof(OutT) == 2, 256, 128), (gemm_nt_kernel<T, OutT, 256, 128, 4, 2, 2>), dim3((unsigned)tiles(256, 128)), dim3(512));
  } else if (tiles(128, 128) >= 256) {
    TELL_GEMM_LAUNCH(gemm_label("gemm_nt_kernel", sizeof(T) == 2, sizeof(OutT) == 2, 128, 128), (gemm_nt_kernel<T, OutT, 128, 128, 2, 2, 2>), dim3((unsigned)tiles(128, 128)), dim3(256));
  } else {
    TELL_GEMM_LAUNCH(gemm_label("gemm_nt_kernel", sizeof(T) == 2, sizeof(OutT) == 2, 64, 64), (gemm_nt_kernel<T, OutT, 64, 64, 2, 2, 4>), dim3((unsigned)tiles(64, 64)), dim3(256));
  }
  return g_gemm_plan ? TELL_OK : tell_check_launch("gemm_nt");
}

static thread_local unsigned long long* g_gemm_ts_next = nullptr;
extern "C" int tell_gemm_ts_next(void* ts, hipStream_t) {
  g_gemm_ts_next = static_cast<unsigned long long*>(ts);
  return TELL_OK;
}

extern "C" int tell_gemm_nt(const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                            int M, int N, int K, int in_dtype, int out_dtype, const float* bias,
                            int bias_mode, int act, const void* aux, float alpha, int accumulate,
                            const int* m_dev, hipStream_t stream) {
  TELL_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm_nt: negative dimension");
  if (M == 0 || N == 0) return TELL_OK;
  TELL_REQUIRE(K > 0, "gemm_nt: K must be positive");
  const int vec = in_dtype == TELL_BF16 ? 8 : 4;
  TELL_REQUIRE(in_dtype == TELL_BF16 || in_dtype == TELL_F32, "gemm_nt: bad in_dtype");
  TELL_REQUIRE(out_dtype == TELL_BF16 || out_dtype == TELL_F32, "gemm_nt: bad out_dtype");
  TELL_REQUIRE(K % vec == 0 && lda % vec == 0 && ldb % vec == 0,
               "gemm_nt: K, lda, ldb must be multiples of one 16-byte chunk");
  TELL_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "gemm_nt: A/B must be 16-byte aligned");
  TELL_REQUIRE(bias_mode == 0 || bias != nullptr, "gemm_nt: bias_mode set without bias");
  TELL_REQUIRE((act != 3 && act != 4) || aux != nullptr, "gemm_nt: act=3 / act=4 need aux");
  GemmArgs a;
  a.A = A; a.B = B; a.C = C; a.bias = bias; a.aux = aux; a.m_dev = m_dev;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
  a.bias_mode = bias_mode; a.act = act; a.accumulate = accumulate; a.alpha = alpha;
  a.asum = nullptr; a.asum_scale = 0.f; a.stat_mean = nullptr; a.stat_m2 = nullptr; a.atomic_out = 0; a.ts = nullptr; a.conv_zero = nullptr; a.queue = nullptr;
  if (g_gemm_ts_next && !g_gemm_plan) {
    a.ts = g_gemm_ts_next;
    g_gemm_ts_next = nullptr;
  }
  if (in_dtype == TELL_BF16)
    return out_dtype == TELL_BF16 ? launch_gemm<uint16_t, uint16_t>(a, stream)
                                  : launch_gemm<uint16_t, float>(a, stream);
  return out_dtype == TELL_BF16 ? launch_gemm<float, uint16_t>(a, stream)
                                : launch_gemm<float, float>(a, stream);
}

extern "C" const char* tell_gemm_nt_plan(const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                                         int M, int N, int K, int in_dtype, int out_dtype, const float* bias,
                                         int bias_mode, int act, const void* aux, float alpha, int accumulate,
                                         const int* m_dev, hipStream_t stream) {
  g_gemm_label[0] = 0;
  g_gemm_plan = true;
  (void)tell_gemm_nt(A, lda, B, ldb, C, ldc, M, N, K, in_dtype, out_dtype, bias, bias_mode, act, aux, alpha, accumulate,
                     m_dev, stream);
  g_gemm_plan = false;
  return g_gemm_label;
}

// C[M,N] = act((op(A) . op(B) + bias) * alpha) (+ C), bf16 operands.  trans_a: A is stored [K][M] (K-major,
// row stride lda >= M); otherwise [M][K].  trans_b: B is stored [K][N]; otherwise [N][K] (the tell_gemm_nt form).
extern "C" int tell_gemm_bf16(const void* A, long lda, int trans_a, const void* B, long ldb, int trans_b, void* C,
                              long ldc, int M, int N, int K, int out_dtype, const float* bias, int bias_mode,
                              int act, const void* aux, float alpha, int accumulate, const int* m_dev,
                              float* a_colsum, float a_colsum_scale, hipStream_t stream) {
  TELL_REQUIRE(a_colsum == nullptr || trans_a, "gemm_bf16: a_colsum needs a K-major A");
  if (!trans_a && !trans_b)
    return tell_gemm_nt(A, lda, B, ldb, C, ldc, M, N, K, TELL_BF16, out_dtype, bias, bias_mode, act, aux, alpha,
                        accumulate, m_dev, stream);
  TELL_REQUIRE(M >= 0 && N >= 0 && K > 0, "gemm_bf16: bad dimension");
  if (M == 0 || N == 0) return TELL_OK;
  TELL_REQUIRE(out_dtype == TELL_BF16 || out_dtype == TELL_F32, "gemm_bf16: bad out_dtype");
  TELL_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, "gemm_bf16: lda, ldb must be multiples of 8 elements");
  // a row-major operand is read in 8-element chunks along k: its rows must extend to round_up(K, 8), and the
  // columns K.. of that last chunk must hold zeros (they meet zero-filled rows of the K-major operand)
  const long k8 = ((long)K + 7) / 8 * 8;
  TELL_REQUIRE((trans_a ? lda >= M : lda >= k8) && (trans_b ? ldb >= N : ldb >= k8),
               "gemm_bf16: K-major operands need ld >= extent, row-major ones ld >= round_up(K, 8)");
  TELL_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "gemm_bf16: A/B must be 16-byte aligned");
  TELL_REQUIRE(bias_mode == 0 || bias != nullptr, "gemm_bf16: bias_mode set without bias");
  TELL_REQUIRE(act != 3 || aux != nullptr, "gemm_bf16: act=3 needs aux");
  GemmArgs a;
  a.A = A; a.B = B; a.C = C; a.bias = bias; a.aux = aux; a.m_dev = m_dev;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
  a.bias_mode = bias_mode; a.act = act; a.accumulate = accumulate; a.alpha = alpha;
  a.asum = a_colsum; a.asum_scale = a_colsum_scale; a.stat_mean = nullptr; a.stat_m2 = nullptr; a.atomic_out = 0; a.ts = nullptr; a.conv_zero = nullptr; a.queue = nullptr;
  if (trans_a && trans_b)
    return out_dtype == TELL_BF16 ? launch_gemm_tx<uint16_t, true, true>(a, stream) : launch_gemm_tx<float, true, true>(a, stream);
  if (trans_b)
    return out_dtype == TELL_BF16 ? launch_gemm_tx<uint16_t, false, true>(a, stream) : launch_gemm_tx<float, false, true>(a, stream);
  return out_dtype == TELL_BF16 ? launch_gemm_tx<uint16_t, true, false>(a, stream) : launch_gemm_tx<float, true, false>(a, stream);
}

// n independent bf16 products in a few launches.  Problems are bucketed by form (NT / B K-major / both K-major), output
// type and tile shape; each bucket runs as one launch per GROUP_MAX problems.  A problem the grouped kernels cannot take
// (NT form with K % 64 != 0 or unaligned operands) is launched on its own through the ordinary dispatcher.
template <typename Kern>
static int launch_group(Kern kern, int bm, int bn, const tell_gemm_problem* pr, const int* ids, int n, hipStream_t stream,
                        int threads = 256) {
  for (int base = 0; base < n; base += GROUP_MAX) {
    GemmGroup g;
    g.n = n - base < GROUP_MAX ? n - base : GROUP_MAX;
    g.start[0] = 0;
    for (int i = 0; i < g.n; ++i) {
      const tell_gemm_problem& q = pr[ids[base + i]];
      GroupProblem& t = g.pr[i];
      t.A = q.A; t.B = q.B; t.C = q.C; t.asum = q.asum; t.bias = q.bias; t.lda = q.lda; t.ldb = q.ldb; t.ldc = q.ldc;
      t.M = q.M; t.N = q.N; t.K = q.K; t.accumulate = q.accumulate; t.bias_mode = q.bias_mode; t.act = q.act;
      t.alpha = q.alpha; t.asum_scale = q.asum_scale;
      t.m_dev = q.trans_a ? nullptr : q.lim_dev; t.k_dev = q.trans_a ? q.lim_dev : nullptr;
      const long tiles = (long)((q.M + bm - 1) / bm) * ((q.N + bn - 1) / bn);
      g.start[i + 1] = g.start[i] + (int)((tiles + 7) / 8 * 8);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)g.start[g.n]), dim3(threads), 0, stream, g);
  }
  return tell_check_launch("gemm_grouped");
}
extern "C" int tell_gemm_bf16(const void* A, long lda, int trans_a, const void* B, long ldb, int trans_b, void* C,
                              long ldc, int M, int N, int K, int out_dtype, const float* bias, int bias_mode, int act,
                              const void* aux, float alpha, int accumulate, const int* m_dev, float* a_colsum,
                              float a_colsum_scale, hipStream_t stream);
extern "C" int tell_gemm_grouped(int n, const tell_gemm_problem* pr, hipStream_t stream) {
  if (n <= 0) return TELL_OK;
  const int tile_env = (int)tell_opt(OPT_GROUP_TILE);   // A/B aid: 64 / 128
  enum { FORMS = 3, BUCKETS = FORMS * 2 * 2 + 1, WIDE = FORMS * 2 * 2 };   // form x (bf16, f32) x (64, 128) + long fp32 TN
  int* ids = (int*)alloca(sizeof(int) * BUCKETS * n);
  int* form_of = (int*)alloca(sizeof(int) * n);
  int cnt[BUCKETS] = {0};
  long big_tiles[FORMS][2] = {{0, 0}, {0, 0}, {0, 0}};
  for (int i = 0; i < n; ++i) {
    const tell_gemm_problem& q = pr[i];
    TELL_REQUIRE(q.M > 0 && q.N > 0 && q.K > 0, "gemm_grouped: bad dimension");
    TELL_REQUIRE(q.out_dtype == TELL_F32 || q.out_dtype == TELL_BF16, "gemm_grouped: bad output dtype");
    TELL_REQUIRE(!(q.trans_a && !q.trans_b), "gemm_grouped: A K-major with B row-major is not a form of the step");
    TELL_REQUIRE(q.act == 0 || q.act == 1, "gemm_grouped: act must be 0 (none) or 1 (relu)");
    TELL_REQUIRE(q.asum == nullptr || q.trans_a, "gemm_grouped: fused column sums need a K-major A");
    // a count-limited REDUCTION (K-major A with lim_dev) that finds *lim_dev == 0 returns without touching C: right when
    // the product is added to C, an uninitialised output when it is meant to replace it
    TELL_REQUIRE(!(q.trans_a && q.lim_dev) || q.accumulate, "gemm_grouped: a count-limited K-major product must accumulate");
    const bool aligned = q.lda % 8 == 0 && q.ldb % 8 == 0 && (((uintptr_t)q.A | (uintptr_t)q.B) & 15) == 0;
    int form = q.trans_a ? 2 : q.trans_b ? 1 : 0;
    if (form > 0) TELL_REQUIRE(aligned, "gemm_grouped: K-major operands need 16-byte aligned rows");
    if (form == 0 && !(aligned && q.K % 64 == 0)) form = -1;                         // on its own
    form_of[i] = form;
    if (form >= 0 && q.M >= 128 && q.N >= 128)
      big_tiles[form][q.out_dtype == TELL_F32] += (long)((q.M + 127) / 128) * ((q.N + 127) / 128);
  }
  int rc = TELL_OK;
  for (int i = 0; i < n && !rc; ++i) {
    const tell_gemm_problem& q = pr[i];
    if (form_of[i] < 0) {
      rc = tell_gemm_bf16(q.A, q.lda, 0, q.B, q.ldb, 0, q.C, q.ldc, q.M, q.N, q.K, q.out_dtype, q.bias, q.bias_mode, q.act,
                          nullptr, q.alpha, q.accumulate, nullptr, nullptr, 0.f, stream);
      continue;
    }
    const int f32 = q.out_dtype == TELL_F32;
    // 128x128 tiles halve the LDS and L2 traffic per flop; they need enough tiles in flight to fill the chip twice over
    bool big = q.M >= 128 && q.N >= 128 && big_tiles[form_of[i]][f32] >= 512;
    if (tile_env == 64) big = false;
    if (tile_env == 128) big = q.M >= 128 && q.N >= 128;
    int b = (form_of[i] * 2 + f32) * 2 + (big ? 1 : 0);
    const bool wide_env = tell_opt(OPT_GROUP_WIDE) != 0;   // A/B aid
    // (round 5: 256x128 tiles for the K = 1024 / 2048 weight gradients too - same box, decoder half 6.74 / 6.75 / 6.74 ms
    //  for a threshold of 4096 / 1024 / 2048: the four 128x128 launches, 0.70 ms, become 0.70 ms of wide launches)
    if (wide_env && form_of[i] == 2 && f32 && q.K >= 4096 && q.M >= 256 && q.N >= 128) b = WIDE;
    ids[b * n + cnt[b]++] = i;
  }
  // inside a bucket: longest reductions first - a launch lasts until its last workgroup is done, and a 16384-row
  // reduction dispatched behind twenty 1024-row problems would start when the others are already finishing
  const bool sort_env = tell_opt(OPT_GROUP_SORT) != 0;   // A/B aid
  for (int b = 0; b < BUCKETS && sort_env; ++b) {
    int* v = ids + b * n;
    for (int i = 1; i < cnt[b]; ++i) {                 // insertion sort (stable, n <= ~100)
      const int x = v[i];
      int j = i - 1;
      while (j >= 0 && pr[v[j]].K < pr[x].K) { v[j + 1] = v[j]; --j; }
      v[j + 1] = x;
    }
  }
#define GROUP_RUN(B, KERN, BM, BN) if (cnt[B] && !rc) { rc = launch_group(KERN, BM, BN, pr, ids + (B) * n, cnt[B], stream); cnt[B] = 0; }
  {
    long t0 = 0, t2 = 0;
    int k0 = 1 << 30, k2 = 1 << 30;
    for (int i = 0; i < cnt[0]; ++i) { const tell_gemm_problem& q = pr[ids[i]]; t0 += (long)((q.M + 63) / 64) * ((q.N + 63) / 64); k0 = q.K < k0 ? q.K : k0; }
    for (int i = 0; i < cnt[2]; ++i) { const tell_gemm_problem& q = pr[ids[2 * n + i]]; t2 += (long)((q.M + 63) / 64) * ((q.N + 63) / 64); k2 = q.K < k2 ? q.K : k2; }
    if (cnt[0] && small_ring_stages(t0, k0) >= 3) { GROUP_RUN(0, (gemm_nt_group_kernel<uint16_t, 64, 64, 4>), 64, 64) }
    if (cnt[2] && small_ring_stages(t2, k2) >= 3) { GROUP_RUN(2, (gemm_nt_group_kernel<float, 64, 64, 4>), 64, 64) }
  }
  GROUP_RUN(0, (gemm_nt_group_kernel<uint16_t, 64, 64>), 64, 64)
  GROUP_RUN(1, (gemm_nt_group_kernel<uint16_t, 128, 128>), 128, 128)
  GROUP_RUN(2, (gemm_nt_group_kernel<float, 64, 64>), 64, 64)
  GROUP_RUN(3, (gemm_nt_group_kernel<float, 128, 128>), 128, 128)
  GROUP_RUN(4, (gemm_tx_group_kernel<uint16_t, 64, 64, false, true, 4>), 64, 64)
  GROUP_RUN(5, (gemm_tx_group_kernel<uint16_t, 128, 128, false, true, 2>), 128, 128)
  GROUP_RUN(6, (gemm_tx_group_kernel<float, 64, 64, false, true, 4>), 64, 64)
  GROUP_RUN(7, (gemm_tx_group_kernel<float, 128, 128, false, true, 2>), 128, 128)
  GROUP_RUN(8, (gemm_tx_group_kernel<uint16_t, 64, 64, true, true, 4>), 64, 64)
  GROUP_RUN(9, (gemm_tx_group_kernel<uint16_t, 128, 128, true, true, 2>), 128, 128)
  GROUP_RUN(10, (gemm_tx_group_kernel<float, 64, 64, true, true, 4>), 64, 64)
  GROUP_RUN(11, (gemm_tx_group_kernel<float, 128, 128, true, true, 2>), 128, 128)
  if (cnt[WIDE] && !rc) rc = launch_group((gemm_tx_group_wide_kernel<float, true, true>), 256, 128, pr, ids + WIDE * n, cnt[WIDE], stream, 512);
#undef GROUP_RUN
  return rc;
}

// ------------------------------------------------------------- split-K for skinny GEMMs (decode step)
// A generation step multiplies M = B x beam <= 128 rows by [N, 4096] weights (context_fc, fc2): 16-32 output tiles,
// each a chain of 64 dependent K steps - 23 us of latency on 6 % of the CUs.  The host splits K into slices that run
// as independent problems of ONE grouped launch (tell_gemm_grouped, fp32 partial tiles); this kernel folds the slices
// and applies the epilogue the fused kernel would have applied: out = act((sum_s partial_s + bias) * alpha).
template <typename OutT, int ACT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int splits, long split_stride,
                                                            int M, int N, const float* __restrict__ bias, float alpha,
                                                            OutT* __restrict__ out, long ldc, const int* __restrict__ m_dev) {
  if (m_dev) { const int md = *m_dev; M = md < M ? md : M; }
  const int nq = N >> 2;                                     // quads per row (N % 4 == 0)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)M * nq; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i / nq), n = (int)(i % nq) * 4;
    f32x4_t v = *reinterpret_cast<const f32x4_t*>(partial + (long)m * N + n);
    for (int s = 1; s < splits; ++s) v += *reinterpret_cast<const f32x4_t*>(partial + s * split_stride + (long)m * N + n);
    if (bias) v += *reinterpret_cast<const f32x4_t*>(bias + n);
    v *= alpha;
    epi_act4<ACT>(v);
    if constexpr (sizeof(OutT) == 2) {
      u32x2 w = {pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3])};
      *reinterpret_cast<u32x2*>(out + (long)m * ldc + n) = w;
    } else {
      *reinterpret_cast<f32x4_t*>(out + (long)m * ldc + n) = v;
    }
  }
}
// partial: [splits][M][N] fp32 (slice s at partial + s * split_stride); bias: fp32 [N] or NULL; act 0 / 1 (relu) / 2 (gelu)
extern "C" int tell_splitk_reduce2(const float* partial, int splits, long split_stride, int M, int N, const float* bias,
                                   int act, float alpha, void* out, long ldc, int out_dtype, const int* m_dev,
                                   hipStream_t stream) {
  if (M <= 0 || N <= 0) return TELL_OK;
  TELL_REQUIRE(splits >= 1 && N % 4 == 0 && ldc % 4 == 0 && act >= 0 && act <= 2, "splitk_reduce: bad arguments");
  TELL_REQUIRE((((uintptr_t)partial | (uintptr_t)out | (uintptr_t)bias) & 15) == 0 && split_stride % 4 == 0,
               "splitk_reduce: buffers must be 16-byte aligned");
  long g = ((long)M * (N / 4) + 255) / 256;
  if (g > 2048) g = 2048;
#define SKR(OutT, ACT) hipLaunchKernelGGL((splitk_reduce_kernel<OutT, ACT>), dim3((unsigned)g), dim3(256), 0, stream, partial, splits, split_stride, M, N, bias, alpha, (OutT*)out, ldc, m_dev)
  if (out_dtype == TELL_BF16) { if (act == 0) SKR(uint16_t, 0); else if (act == 1) SKR(uint16_t, 1); else SKR(uint16_t, 2); }
  else { if (act == 0) SKR(float, 0); else if (act == 1) SKR(float, 1); else SKR(float, 2); }
#undef SKR
  return tell_check_launch("splitk_reduce");
}
extern "C" int tell_splitk_reduce(const float* partial, int splits, long split_stride, int M, int N, const float* bias,
                                  int act, float alpha, void* out, long ldc, int out_dtype, hipStream_t stream) {
  return tell_splitk_reduce2(partial, splits, split_stride, M, N, bias, act, alpha, out, ldc, out_dtype, nullptr, stream);
}

int tell_bn_finish_launch(const float* pmean, const float* pm2, long M, int C, int n_chunks, int rows_per_chunk,
                          float eps, float momentum, float* mean, float* invstd, float* running_mean,
                          float* running_var, hipStream_t stream);   // conv.hip

// y[M,N] = A[M,K] . B[N,K]^T (bf16, stored), and the train-mode BatchNorm statistics of y's columns in the same
// pass: mean[N], invstd[N] (+ running stats update).  workspace: 2 * ceil(M/64) * N floats.
extern "C" int tell_gemm_bn_stats(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N,
                                  int K, float eps, float momentum, float* mean, float* invstd, float* running_mean,
                                  float* running_var, float* workspace, hipStream_t stream) {
  TELL_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_bn_stats: bad dimension");
  TELL_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && N % 8 == 0 && ldc % 8 == 0,
               "gemm_bn_stats: K, N and the row strides must be multiples of 8 elements");
  TELL_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && ((uintptr_t)C & 15) == 0,
               "gemm_bn_stats: A/B/C must be 16-byte aligned");
  GemmArgs a;
  a.A = A; a.B = B; a.C = C; a.bias = nullptr; a.aux = nullptr; a.m_dev = nullptr;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
  a.bias_mode = 0; a.act = 0; a.accumulate = 0; a.alpha = 1.f; a.asum = nullptr; a.asum_scale = 0.f; a.atomic_out = 0; a.ts = nullptr; a.conv_zero = nullptr; a.queue = nullptr;
  const long max_tiles = ((long)M + 63) / 64;
  a.stat_mean = workspace;
  a.stat_m2 = workspace + max_tiles * N;
  int bm = 64;
  int rc = launch_gemm<uint16_t, uint16_t>(a, stream, &bm);
  if (rc) return rc;
  return tell_bn_finish_launch(a.stat_mean, a.stat_m2, M, N, (M + bm - 1) / bm, bm, eps, momentum, mean, invstd,
                               running_mean, running_var, stream);
}


// Convolution (1x1 or 3x3, stride 1 or 2, NHWC, Cin a power-of-two multiple of 64) as an IMPLICIT GEMM on the
// direct-to-LDS kernel - the im2col matrix never exists: every lane's DMA source is the shifted input pixel, the
// padding ring reads a zero page - with the statistics of the train-mode BatchNorm that follows
// (resnet.py:92-108 via torchvision's Bottleneck; callback_apex_trainer.py:259 keeps the frozen trunk in train mode):
// per-m-tile column mean / M2 from the stored bf16 tile (GEMM epilogue), merged by one small finish launch.
// mean == NULL: plain convolution (eval mode, running statistics).
//   X [B,H,W,Cin] bf16, Wt [Cout, KH*KW*Cin] bf16 ((kh,kw,c) order), Y [B*OH*OW, Cout] bf16 (raw, BN not applied)
//   workspace: 2 * ceil(M / 64) * Cout floats; zero_page: >= 16 zero bytes, 16-byte aligned
// Measured alternatives for the statistics (tools/bench_conv.py, B = 32, layer3 conv1 9.5 us alone, 15 us this way):
//   * merging the partials INSIDE the GEMM launch by the last workgroup to arrive (agent-scope ticket): 22-26 us for
//     that shape, 85 us for layer1 - one workgroup's dependent loads become the critical path;
//   * column sums / sums of squares from the accumulator registers (lane shuffles) added to [2, Cout] with float
//     atomics: 31-49 us, 320-620 us for layer1 - thousands of same-address atomics serialise in L2.
// the convolution launch itself; stats_ws != NULL: the epilogue also leaves the per-row-chunk statistics there
// ([chunks][Cout] means, then [chunks][Cout] M2 at stats_ws + ceil(M / 64) * Cout); *bm_out = rows per chunk
static int conv_launch(const void* X, const void* Wt, void* Y, int B, int H, int W, int Cin, int KH, int KW, int stride,
                       int pad, int OH, int OW, int Cout, float* stats_ws, const void* zero_page, int* bm_out,
                       hipStream_t stream, const float* bias = nullptr, int act = 0, const void* aux = nullptr) {
  const long Ml = (long)B * OH * OW;
  TELL_REQUIRE(Ml > 0 && Ml < (1L << 31) && Cout > 0, "conv_bn_stats: bad dimension");
  int cshift = 0;
  // the ResNet stem (resnet.py:92-96: 7x7, stride 2, padding 3) over NHWC4 input - X: [B, H, W, 4] (3 channels + zero),
  // Wt: [Cout, 256] = 8 kernel rows (the 8th zero) x 8 window columns (the first zero) x 4 channels (the 4th zero)
  const bool stem = Cin == 4 && KH == 7 && KW == 7 && stride == 2 && pad == 3;
  if (stem) {
    TELL_REQUIRE(W % 2 == 0, "conv_bn_stats: the 7x7 stem gather needs an even image width");
    cshift = -1;
  } else {
    while ((64 << cshift) < Cin) ++cshift;
    TELL_REQUIRE((64 << cshift) == Cin, "conv_bn_stats: Cin must be 64 * 2^n (or the 4-channel 7x7 stem)");
    TELL_REQUIRE(KH == KW && (KH == 1 || KH == 3) && (stride == 1 || stride == 2), "conv_bn_stats: 1x1 / 3x3, stride 1 / 2");
  }
  TELL_REQUIRE(Cout % 8 == 0, "conv_bn_stats: Cout must be a multiple of 8");
  TELL_REQUIRE(((uintptr_t)X & 15) == 0 && ((uintptr_t)Wt & 15) == 0 && ((uintptr_t)Y & 15) == 0 &&
               ((uintptr_t)zero_page & 15) == 0 && zero_page != nullptr, "conv_bn_stats: 16-byte alignment");
  const int M = (int)Ml, N = Cout, K = stem ? 256 : KH * KW * Cin;
  GemmArgs a;
  a.A = X; a.B = Wt; a.C = Y; a.bias = bias; a.aux = aux; a.m_dev = nullptr;
  a.lda = Cin; a.ldb = K; a.ldc = N; a.M = M; a.N = N; a.K = K;
  a.bias_mode = bias ? 1 : 0; a.act = act; a.accumulate = 0; a.alpha = 1.f; a.asum = nullptr; a.asum_scale = 0.f; a.atomic_out = 0;
  a.ts = nullptr; a.queue = nullptr;
  a.conv_zero = (KH == 1 && stride == 1) ? nullptr : zero_page;          // 1x1 / stride 1: A is the activation matrix
  a.conv_H = H; a.conv_W = W; a.conv_OH = OH; a.conv_OW = OW; a.conv_KW = KW; a.conv_stride = stride; a.conv_pad = pad;
  a.conv_cshift = cshift;
  a.stat_mean = a.stat_m2 = nullptr;
  if (stats_ws) {
    a.stat_mean = stats_ws;
    a.stat_m2 = stats_ws + (((long)M + 63) / 64) * N;
  }
  auto tiles = [&](int bm, int bn) { return (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
  const int force = (int)tell_opt(OPT_CONV_TILE);                  // tuning / test aid: 1 / 2 / 3 forces a tile shape
  // 64x64 tiles (4 waves, 32 KB of LDS, 5 workgroups per CU) win on every bottleneck shape of the trunk at B = 32
  // (tools/bench_conv.py: layer3 conv1 15.1 us against 20.4 / 22.3 with 128x64 / 128x128 - these GEMMs are only
  // 50-800 tiles of 128x128, the small tile fills the chip and hides the short K loops behind its neighbours); 128x64
  // ties once there are thousands of tiles
  int pick = (tiles(64, 64) >= 6000 && N % 64 == 0) ? 2 : 3;
  // (Round 5: 128x64 tiles for layer2 at B = 32 - 196 row chunks, which a two-round form of the fused BatchNorm finish +
  //  apply launch could merge per workgroup, instead of 392 chunks and two BatchNorm launches - measured: the trunk got
  //  SLOWER, 4.63 -> 4.80 ms; the 200 KB of chunk statistics per workgroup cost more than the launch they save.)
  // (Round 5 measured DEEPER rings where one round of workgroups covers the output - 64x64 with 8 stages, 128x64 with 6,
  //  128x128 with 4, one workgroup per CU - on every bottleneck shape, profiles/r05_conv_shapes.txt: none wins; layer3
  //  conv2 25.0 (2 stages) / 25.2 (4) / 38.3 us (8).  These launches are not chains of DMA round trips any more: at 32
  //  flop per staged byte a 64x64 tile is bound by what a CU's LDS-DMA path delivers from L2, ~40-45 GB/s per CU, and
  //  several small workgroups per CU pull more than one deep one.  ResNet-152: 4.71 -> 5.08 ms with the deep choice.)
  int ns = 0;                                                      // forced ring depth (TELL_CONV_TILE + TELL_GEMM_RING)
  if (force) { pick = force; ns = ring_env(); }
#define CONV_LAUNCH(BM_, BN_, CV, NS_)                                                                              \
  hipLaunchKernelGGL((gemm_nt_glds_kernel<uint16_t, BM_, BN_, 2, 2, CV, NS_>), dim3((unsigned)tiles(BM_, BN_)), dim3(256), 0, stream, a)
#define CONV_LAUNCH2(BM_, BN_, NS_) do { if (cv) CONV_LAUNCH(BM_, BN_, true, NS_); else CONV_LAUNCH(BM_, BN_, false, NS_); } while (0)
  const bool cv = a.conv_zero != nullptr;
  const bool ring = ns ? ns >= 3 : small_ring_stages(tiles(64, 64), K) >= 3;
  // 64x64 tiles, K >= 1024: the one-wave-per-SIMD K loop of gemm_s64.hip (layer3 conv2 25.0 -> 14.4 us, layer4 conv2 37.5 ->
  // 17.8; shorter reductions stay on the 2-stage body, measured there).  TELL_GEMM_S64=0: the general body (A/B; read per launch)
  const int s64_env = (int)tell_opt(OPT_GEMM_S64);
  if (pick == 3 && !stem && (s64_env == 2 || (s64_env == 1 && K >= 1024))) {
    const int rc = launch_gemm_s64(a, stream, 0);
    if (rc <= 0) { *bm_out = 64; return rc; }
  }
  if (pick == 1) CONV_LAUNCH2(128, 128, 2);
  else if (pick == 2) CONV_LAUNCH2(128, 64, 2);
  else if (ring) CONV_LAUNCH2(64, 64, 4);
  else CONV_LAUNCH2(64, 64, 2);
#undef CONV_LAUNCH2
#undef CONV_LAUNCH
  *bm_out = pick == 3 ? 64 : 128;
  return tell_check_launch("conv_bn_stats");
}
extern "C" int tell_conv_bn_stats(const void* X, const void* Wt, void* Y, int B, int H, int W, int Cin, int KH, int KW,
                                  int stride, int pad, int OH, int OW, int Cout, float eps, float momentum, float* mean,
                                  float* invstd, float* running_mean, float* running_var, float* workspace,
                                  const void* zero_page, hipStream_t stream) {
  if (mean) TELL_REQUIRE(invstd && workspace, "conv_bn_stats: statistics need invstd and workspace");
  int bm = 64;
  int rc = conv_launch(X, Wt, Y, B, H, W, Cin, KH, KW, stride, pad, OH, OW, Cout, mean ? workspace : nullptr, zero_page, &bm,
                       stream);
  if (rc || !mean) return rc;
  const long M = (long)B * OH * OW;
  return tell_bn_finish_launch(workspace, workspace + ((M + 63) / 64) * Cout, M, Cout, (int)((M + bm - 1) / bm), bm, eps,
                               momentum, mean, invstd, running_mean, running_var, stream);
}
// Inference form (model.eval(): running statistics): the BatchNorm is FOLDED into the convolution - w' = w * gamma /
// sqrt(var + eps) per output channel, bias = beta - mean * that (the host builds both once per state of the weights) - so
// conv -> bn (-> + residual) (-> relu) is ONE launch: y = act(conv(x, w') + bias [+ residual]).  relu with a residual:
// relu(... + residual), the Bottleneck's order (resnet.py via torchvision: out += identity; relu).
extern "C" int tell_conv_bias_act(const void* X, const void* Wt, void* Y, int B, int H, int W, int Cin, int KH, int KW,
                                  int stride, int pad, int OH, int OW, int Cout, const float* bias, const void* residual,
                                  int relu, const void* zero_page, hipStream_t stream) {
  TELL_REQUIRE(!residual || relu, "conv_bias_act: a residual without the relu is not a form of the trunk");
  int bm = 64;
  return conv_launch(X, Wt, Y, B, H, W, Cin, KH, KW, stride, pad, OH, OW, Cout, nullptr, zero_page, &bm, stream, bias,
                     residual ? 4 : (relu ? 1 : 0), residual);
}
int tell_bn_finish_apply_launch(const float* pmean, const float* pm2, long M, int C, int n_chunks, int rows_per_chunk,
                                float eps, float momentum, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, const void* residual, void* y, int relu, float* scratch,
                                hipStream_t stream);   // conv.hip
extern "C" int tell_bn_apply(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                             const void* residual, void* y, long M, int C, int relu, int dtype, hipStream_t stream);
// conv -> train-mode BatchNorm (-> + residual) (-> ReLU), y in place: the implicit-GEMM convolution above with the
// statistics in its epilogue, then ONE launch that combines the row chunks and normalises (conv.hip
// bn_finish_apply_kernel) when the combine is cheap enough to repeat per workgroup (<= 128 chunks), else the combine and
// the elementwise pass as two launches.  workspace: 2 * ceil(M / 64) * Cout + 2 * Cout floats.
extern "C" int tell_conv_bn_act(const void* X, const void* Wt, void* Y, int B, int H, int W, int Cin, int KH, int KW,
                                int stride, int pad, int OH, int OW, int Cout, float eps, float momentum, const float* gamma,
                                const float* beta, float* running_mean, float* running_var, const void* residual, int relu,
                                float* workspace, const void* zero_page, hipStream_t stream) {
  TELL_REQUIRE(workspace && gamma && beta, "conv_bn_act: workspace, gamma and beta are required");
  int bm = 64;
  int rc = conv_launch(X, Wt, Y, B, H, W, Cin, KH, KW, stride, pad, OH, OW, Cout, workspace, zero_page, &bm, stream);
  if (rc) return rc;
  const long M = (long)B * OH * OW;
  const long chunks64 = (M + 63) / 64;
  const float* pmean = workspace;
  const float* pm2 = workspace + chunks64 * Cout;
  const int n_chunks = (int)((M + bm - 1) / bm);
  const bool fuse = tell_opt(OPT_BN_FUSE) != 0;                 // A/B aid
  if (fuse) {
    // (opt-in: ResNet alone 4.97 -> 4.93 ms, the training step unchanged - 1443 / 1437 against 1434 / 1432 samples/s)
    const bool comb = tell_opt(OPT_BN_COMBINE) == 1;
    rc = tell_bn_finish_apply_launch(pmean, pm2, M, Cout, n_chunks, bm, eps, momentum, gamma, beta, running_mean, running_var,
                                     residual, Y, relu, comb ? workspace + 2 * chunks64 * Cout + 2 * Cout : nullptr, stream);
    if (rc <= 0) return rc;
  }
  float* mean = workspace + 2 * chunks64 * Cout;
  float* invstd = mean + Cout;
  rc = tell_bn_finish_launch(pmean, pm2, M, Cout, n_chunks, bm, eps, momentum, mean, invstd, running_mean, running_var, stream);
  if (rc) return rc;
  return tell_bn_apply(Y, mean, invstd, gamma, beta, residual, Y, M, Cout, relu, TELL_BF16, stream);
}
