// gemm_nt_pp2_kernel: the 256x256 ping-pong GEMM of gemm.hip (same LDS image, same 4-phase main loop, same wave
// tiles - read its header first) as RESIDENT workgroups that carry the operand stream across tile boundaries.
//
// What it is for: at K = 1024 (RoBERTa's qkv / out / fc1 projections, the article's K|V projection) an output tile is
// only 16 K tiles long, and a workgroup of gemm_nt_pp_kernel spends ~6 us of its ~42 us waiting for its first operand
// tiles (every CU of the chip starts a tile at the same moment: an HBM / L2 burst) with nothing else on the CU, because
// the kernel's 128 KB of LDS and 2 x 247 registers per SIMD leave room for nobody.  Here one workgroup per CU walks
// tiles b, b + grid, b + 2 grid, ... and the first 1.5 K tiles of the NEXT output tile are put in flight right after
// the last MFMA phase of the current one - they land while the epilogue runs.
//
// For that the epilogue may not use the operand stages as its staging area (gemm.hip's does): it goes through the
// 32 KB of LDS the stages leave free (2 x 64 KB + 32 KB = the CU's 160 KB), one 32-row block of every wave per pass
// (64 rows x 512 bytes), four passes.
//
// Counters: loads (LDS-DMA) and stores share vmcnt, and loads and stores retire out of order with respect to each
// other, so after an epilogue the only safe wait for the prefetched tiles is vmcnt(0) - the start of the next tile
// waits for the last store acknowledgement (~1 us) where a fresh workgroup waited for its first operand round trip.
#include "gemm_common.h"
#include "options.h"
#include <type_traits>

typedef __attribute__((address_space(3))) void* pp2_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* pp2_glb_ptr_t;

namespace {
constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF = 128 * 128, TILE = 4 * HALF, SPARE = 32768;

// Block i (0..3) of every wave's accumulators through the spare area: wave (wr, wc) owns rows
// (i >> 1) * 128 + wr * 64 + (i & 1) * 32 + (lane & 31) and columns j * 128 + wc * 32 + 8 g + 4 (lane >> 5) + e.
template <int ACT, int ABL>
__device__ __forceinline__ void pp2_store(f32x16 (&acc)[4][2], const GemmArgs& p, int m0, int n0, int wr, int wc,
                                          int lane, int tid, uint16_t* cs) {
  const int lh = lane >> 5;
  f32x4_t b4[2][4];
  float bm[4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) b4[j][g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) bm[i] = 0.f;
  if (p.bias_mode == 1) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        b4[j][g] = *reinterpret_cast<const f32x4_t*>(p.bias + n0 + j * 128 + wc * 32 + 8 * g + 4 * lh);
  } else if (p.bias_mode == 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) bm[i] = p.bias[m0 + (i >> 1) * 128 + wr * 64 + (i & 1) * 32 + (lane & 31)];
  }
  uint16_t* C = static_cast<uint16_t*>(p.C);
  const uint16_t* R = static_cast<const uint16_t*>(p.res);
  uint32_t dsalt = 0;
  if constexpr (ACT == 5) dsalt = tell_step_salt(p.drop_salt, p.drop_step);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wr * 32 + (lane & 31);                 // row inside the pass: 64 rows x 512 bytes
    u32x4 rres[4];                                         // act 5: this thread's residual pieces of the pass, fetched
    if constexpr (ACT == 5) {                              // before the arithmetic so they land underneath it
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int c = tid + it * 512, row2 = c >> 5, ch = c & 31;
        const int grow = m0 + (i >> 1) * 128 + (row2 >> 5) * 64 + (i & 1) * 32 + (row2 & 31);
        rres[it] = *reinterpret_cast<const u32x4*>(R + (long)grow * p.ld_res + n0 + ch * 8);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = j * 128 + wc * 32 + 8 * g + 4 * lh;
        f32x4_t v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (acc[i][j][4 * g + e] + b4[j][g][e] + bm[i]) * p.alpha;
        if constexpr (ACT == 5) {                          // dropout of the aligned quad (m, n .. n + 3)
          if (p.drop_thr) {
            const int gm = m0 + (i >> 1) * 128 + wr * 64 + (i & 1) * 32 + (lane & 31);
            const uint64_t quad = ((uint64_t)gm * (uint64_t)p.N + (uint64_t)(n0 + col)) >> 2;
            bool k0, k1, k2, k3;
            tell_keep4_bits(tell_quad_x(p.drop_seed, quad), tell_quad_y(dsalt, quad), p.drop_thr, k0, k1, k2, k3);
            v[0] = k0 ? v[0] * p.drop_inv_keep : 0.f; v[1] = k1 ? v[1] * p.drop_inv_keep : 0.f;
            v[2] = k2 ? v[2] * p.drop_inv_keep : 0.f; v[3] = k3 ? v[3] * p.drop_inv_keep : 0.f;
          }
        } else {
          epi_act4<ACT>(v);
        }
        const int ch = (col >> 3) ^ (row & 15);
        u32x2 w = {pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3])};
        *reinterpret_cast<u32x2*>(cs + row * BN + ch * 8 + (col & 7)) = w;
      }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int c = tid + it * 512, row2 = c >> 5, ch = c & 31;
      u32x4 o = *reinterpret_cast<const u32x4*>(cs + row2 * BN + ((ch ^ (row2 & 15)) << 3));
      const int grow = m0 + (i >> 1) * 128 + (row2 >> 5) * 64 + (i & 1) * 32 + (row2 & 31);
      if constexpr (ACT == 5) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = __uint_as_float(o[e] << 16) + __uint_as_float(rres[it][e] << 16);
          const float hi = __uint_as_float(o[e] & 0xffff0000u) + __uint_as_float(rres[it][e] & 0xffff0000u);
          o[e] = pack2_bf16(lo, hi);
        }
      }
      if (ABL != 2 || o[0] == 0x12345678u) *reinterpret_cast<u32x4*>(C + (long)grow * p.ldc + n0 + ch * 8) = o;
    }
    if (i < 3) __syncthreads();
  }
}

// RES: the dropout + residual epilogue (act 5) as its own instantiation - compiled into the plain kernel it cost the main
// loop 14 registers (240 -> 254) and 2-3 % of its speed
template <int ABL, bool RES>
__global__ __launch_bounds__(512) void gemm_nt_pp2_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * TILE + SPARE];
  gemm_ts_enter(p);
  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int K = p.K;
  const int tiles_n = p.N / BN, tiles_m = p.M / BM;
  const int n_tiles = tiles_m * tiles_n;
  const int nk = K / BK;
  const uint16_t* A = static_cast<const uint16_t*>(p.A);
  const uint16_t* B = static_cast<const uint16_t*>(p.B);
  const long a_half = 128 * p.lda, b_half = 128 * p.ldb;

  auto tile_origin = [&](int vb, int& m0, int& n0) __attribute__((always_inline)) {
    const int nwg = n_tiles, xcd = vb & 7, q = nwg >> 3, r = nwg & 7;
    const int tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * tiles_n;
    const int g = tile_id / per_group, first_m = g * GROUP_M;
    const int gm = tiles_m - first_m < GROUP_M ? tiles_m - first_m : GROUP_M;
    const int in_g = tile_id - g * per_group;
    m0 = (first_m + in_g % gm) * BM;
    n0 = (in_g / gm) * BN;
  };
  // first 1.5 K tiles of the output tile at (m0, n0): A0 B0 B1 A1 of K tile 0, A0 B0 of K tile 1 (the order and the
  // count the main loop's vmcnt arithmetic expects).  Pointers are temporaries: nothing lane-derived survives an epilogue.
  auto prefetch = [&](int m0, int n0) __attribute__((always_inline)) {
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const uint16_t* as[2];
    const uint16_t* bs[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int s = (wave * 2 + jj) * 64 + lane, pr = s >> 4, l16 = (s & 15) ^ (pr & 15);
      const int row = 2 * pr + (l16 >> 3);
      as[jj] = A + (long)(m0 + row) * p.lda + (l16 & 7) * 8;
      bs[jj] = B + (long)(n0 + row) * p.ldb + (l16 & 7) * 8;
    }
    unsigned char* d = smem + wave * 2048;
    // issue ORDER matters: the counted waits of the main loop retire half-tiles oldest first (A0 B0 B1 A1, then A0 B0)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
      __builtin_amdgcn_global_load_lds((pp2_glb_ptr_t)as[jj], (pp2_lds_ptr_t)(d + jj * 1024 + 0 * HALF), 16, 0, 0);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
      __builtin_amdgcn_global_load_lds((pp2_glb_ptr_t)bs[jj], (pp2_lds_ptr_t)(d + jj * 1024 + 2 * HALF), 16, 0, 0);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
      __builtin_amdgcn_global_load_lds((pp2_glb_ptr_t)(bs[jj] + b_half), (pp2_lds_ptr_t)(d + jj * 1024 + 3 * HALF), 16, 0, 0);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
      __builtin_amdgcn_global_load_lds((pp2_glb_ptr_t)(as[jj] + a_half), (pp2_lds_ptr_t)(d + jj * 1024 + 1 * HALF), 16, 0, 0);
    if (nk >= 2) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        __builtin_amdgcn_global_load_lds((pp2_glb_ptr_t)(as[jj] + BK), (pp2_lds_ptr_t)(d + jj * 1024 + TILE + 0 * HALF), 16, 0, 0);
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        __builtin_amdgcn_global_load_lds((pp2_glb_ptr_t)(bs[jj] + BK), (pp2_lds_ptr_t)(d + jj * 1024 + TILE + 2 * HALF), 16, 0, 0);
    }
  };

  // Tile order.  Static (p.queue == NULL): tiles b, b + grid, ...  Dynamic: one counter per XCD (workgroup b lives on XCD
  // b & 7 and keeps taking tiles k * 8 + (b & 7), so the XCD-local tile order above is preserved); a workgroup that
  // gets its CU late - inside the training step the other two streams' workgroups hold CUs when a launch begins -
  // simply takes fewer tiles instead of finishing its static share late.  The fetch for the NEXT tile is issued right
  // after the top-of-tile barrier and consumed after the main loop, so its round trip is never waited for; the fetch
  // that can only be the last one of the launch (value per_x + workgroups per XCD - 1) zeroes the counter again.
  const bool dyn = p.queue != nullptr;
  const int xcd_id = blockIdx.x & 7, per_x = n_tiles >> 3, wg_x = (int)gridDim.x >> 3;
  int* const qx = dyn ? p.queue + xcd_id : nullptr;
  volatile int* sq = reinterpret_cast<volatile int*>(smem + 2 * TILE);     // first word of the (idle) spare area
  int vb = blockIdx.x;
  if (dyn) {
    if (tid0 == 0) {
      const int k = atomicAdd(qx, 1);
      if (k == per_x + wg_x - 1) atomicExch(qx, 0);
      *sq = k;
    }
    __syncthreads();
    const int k = *sq;
    __syncthreads();
    vb = k < per_x ? k * 8 + xcd_id : n_tiles;
  }
  if (vb >= n_tiles) return;
  int m0, n0;
  tile_origin(vb, m0, n0);
  prefetch(m0, n0);
  bool first_tile = true;
  for (;;) {
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, lh = lane >> 5;
    const uint16_t* asrc[2];
    const uint16_t* bsrc[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int s = (wave * 2 + jj) * 64 + lane, pr = s >> 4, l16 = (s & 15) ^ (pr & 15);
      const int row = 2 * pr + (l16 >> 3);
      asrc[jj] = A + (long)(m0 + row) * p.lda + (l16 & 7) * 8;
      bsrc[jj] = B + (long)(n0 + row) * p.ldb + (l16 & 7) * 8;
    }
    auto stage = [&](int tt, auto kind_c) __attribute__((always_inline)) {
      constexpr int KIND = decltype(kind_c)::value;        // 0 = A0, 1 = B0, 2 = B1, 3 = A1
      constexpr int H = KIND == 0 ? 0 : KIND == 3 ? 1 : KIND == 1 ? 2 : 3;      // LDS order A0 A1 B0 B1
      unsigned char* dst = smem + (tt & 1) * TILE + H * HALF + wave * 2048;
      const long off = (long)tt * BK + (KIND == 3 ? a_half : KIND == 2 ? b_half : 0);
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        __builtin_amdgcn_global_load_lds((pp2_glb_ptr_t)(((KIND == 0 || KIND == 3) ? asrc[jj] : bsrc[jj]) + off),
                                         (pp2_lds_ptr_t)(dst + jj * 1024), 16, 0, 0);
    };
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

    // K tile 0 (and the stores of the previous epilogue, which share the counter) must have retired; A0 B0 of K tile 1
    // may still fly on the first tile of the launch
    if (first_tile && nk >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    first_tile = false;
    __builtin_amdgcn_s_barrier();
    int knext = 0;
    if (dyn && tid0 == 0) knext = atomicAdd(qx, 1);        // (older than every load of this tile: retired by the counted waits)
    if (wr == 1) __builtin_amdgcn_s_barrier();            // group 1 runs one barrier behind
    for (int t = 0; t < nk; ++t) {
      read_a(t, 0); read_b(t, 0);
      if (t + 1 < nk) stage(t + 1, K2{});
      __builtin_amdgcn_sched_barrier(0);
      mma(0, 0);
      read_b(t, 1);
      if (t + 1 < nk) stage(t + 1, K3{});
      __builtin_amdgcn_sched_barrier(0);
      mma(0, 1);
      read_a(t, 1);
      if (t + 2 < nk) stage(t + 2, K0{});
      __builtin_amdgcn_sched_barrier(0);
      mma(1, 1);
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
    // every wave has passed its last phase: the stages are free - the next output tile's operands start streaming in
    int nvb = vb + (int)gridDim.x;
    if (dyn) {
      if (tid0 == 0) {
        if (knext == per_x + wg_x - 1) atomicExch(qx, 0);
        *sq = knext;
      }
      __syncthreads();
      const int k = *sq;
      nvb = k < per_x ? k * 8 + xcd_id : n_tiles;
      __syncthreads();                                     // (the epilogue stages through the word's area)
    }
    int m1 = 0, n1 = 0;
    if (nvb < n_tiles) {
      tile_origin(nvb, m1, n1);
      prefetch(m1, n1);
    }
    uint16_t* cs = reinterpret_cast<uint16_t*>(smem + 2 * TILE);
    if constexpr (RES) {
      pp2_store<5, ABL>(acc, p, m0, n0, wr, wc, lane, tid, cs);
    } else if (ABL != 1 || acc[0][0][0] == 12345.678f) {
      switch (p.act) {                                     // block-uniform
        case 1: pp2_store<1, ABL>(acc, p, m0, n0, wr, wc, lane, tid, cs); break;
        case 2: pp2_store<2, ABL>(acc, p, m0, n0, wr, wc, lane, tid, cs); break;
        default: pp2_store<0, ABL>(acc, p, m0, n0, wr, wc, lane, tid, cs); break;
      }
    }
    if (nvb >= n_tiles) break;
    vb = nvb; m0 = m1; n0 = n1;                            // (the spare area is next written a whole main loop of barriers later)
  }
  gemm_ts_exit(p);
}
}  // namespace

int launch_gemm_pp2(const GemmArgs& a, hipStream_t stream, int n_cu) {
  const int n_tiles = (a.M / BM) * (a.N / BN);
  // TELL_PP2_GRID: resident workgroups per launch (default: one per CU) - a tuning aid for how many CUs the step's other
  // two streams are left with while a GEMM runs
  const int grid_env = (int)tell_opt(OPT_PP2_GRID);
  const int cap = grid_env > 0 && grid_env < n_cu ? grid_env : n_cu;
  const unsigned grid = (unsigned)(n_tiles < cap ? n_tiles : cap);
  // timing probes (wrong results; probe build only): 1 no epilogue, 2 no global stores
  // per-XCD tile counters (default on; TELL_PP2_DYNAMIC=0: static tile lists).  MEASURED (same box A/B, configs[2]): 1444 / 1450
  // samples/s static, 1456 / 1468 dynamic; the launch inside the step 160-164 -> 142-146 us (100 alone either way)
  const int dyn_env = (int)tell_opt(OPT_PP2_DYNAMIC);
  GemmArgs ad = a;
  ad.queue = nullptr;
  if (dyn_env && n_tiles > (int)grid && grid % 8 == 0 && n_tiles % 8 == 0) ad.queue = gemm_tile_queue_slot(8, stream);
  const GemmArgs& a2 = ad;
  if (a.act == 5) hipLaunchKernelGGL((gemm_nt_pp2_kernel<0, true>), dim3(grid), dim3(512), 0, stream, a2);
#ifdef TELL_PROBES
  else if (tell_probe(PROBE_PP2_ABL) == 1) hipLaunchKernelGGL((gemm_nt_pp2_kernel<1, false>), dim3(grid), dim3(512), 0, stream, a2);
  else if (tell_probe(PROBE_PP2_ABL) == 2) hipLaunchKernelGGL((gemm_nt_pp2_kernel<2, false>), dim3(grid), dim3(512), 0, stream, a2);
#endif
  else hipLaunchKernelGGL((gemm_nt_pp2_kernel<0, false>), dim3(grid), dim3(512), 0, stream, a2);
  return tell_check_launch("gemm_nt_pp2");
}

// out[M,N] = res[M,N] + dropout_p(A[M,K] . B[N,K]^T + bias[n])   (bf16; the transformer sub-layer residual in the GEMM
// epilogue).  MEASURED (MI355X, M = 16384): out-proj 38.1 -> 44.3 us and the LayerNorm behind it 22.0 -> 12.6 (reads one
// tensor), fc2 119.7 -> 121.5 / 16.2 -> 12.6: 3 + 2 us per layer better alone - and 1 % WORSE inside the training step
// (1420-1425 against 1436 samples/s, same box): the 6 us move from a bandwidth-bound kernel that shares the chip with
// the other two streams into the epilogue of a kernel that holds every CU exclusively.  The host mirror therefore uses
// it only on request (TELL_GEMM_RESIDUAL=1).  -> TELL_OK, or 1 when the shape is not one this kernel takes (whole 256x256 tiles, at least one per CU,
// 16-byte aligned rows): the caller then runs tell_gemm_nt and lets tell_layernorm_fwd add the residual.
extern "C" int tell_gemm_nt_dropout_residual(const void* A, long lda, const void* B, long ldb, const float* bias,
                                             const void* res, long ld_res, void* C, long ldc, int M, int N, int K, float p,
                                             uint32_t seed, uint32_t salt, hipStream_t stream) {
  TELL_REQUIRE(p >= 0.f && p < 1.f, "gemm_nt_dropout_residual: p must be in [0,1)");
  static const int n_cu = [] { hipDeviceProp_t pr; int d = 0; (void)hipGetDevice(&d); (void)hipGetDeviceProperties(&pr, d); return pr.multiProcessorCount; }();
  const long tiles = (long)(M / BM) * (N / BN);
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (M <= 0 || M % BM || N % BN || K % BK || K < BK || tiles < n_cu || tiles % n_cu || (lda & 7) || (ldb & 7) || (ldc & 7) ||
      (ld_res & 7) || !al16(A) || !al16(B) || !al16(C) || !al16(res) || (bias && !al16(bias)) || !res)
    return 1;
  GemmArgs a{};
  a.A = A; a.B = B; a.C = C; a.bias = bias; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
  a.bias_mode = bias ? 1 : 0; a.act = 5; a.alpha = 1.f;
  a.res = res; a.ld_res = ld_res; a.drop_thr = p > 0.f ? tell_drop_threshold(p) : 0u; a.drop_inv_keep = 1.f / (1.f - p);
  a.drop_seed = seed; a.drop_salt = salt; a.drop_step = g_tell_rng_step;
  // round 4: the four-wave kernel (gemm_q4.hip) carries the same epilogue form (one rounding: bf16(res + dropout(.)) from
  // fp32, where this file's staged epilogue rounds the dropped product first); TELL_GEMM_Q4=0 keeps the ping-pong kernel
  const int q4_env = (int)tell_opt(OPT_GEMM_Q4);
  if (q4_env && K % 128 == 0 && K >= 128 && K / 64 < 65536 && 512L * lda < (1L << 31) && 512L * ldb < (1L << 31)) return launch_gemm_q4(a, stream, n_cu);
  return launch_gemm_pp2(a, stream, n_cu);
}
