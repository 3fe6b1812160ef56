// gemm_nt_duo_kernel: C[M,N] = epilogue(A[M,K] . B[N,K]^T), bf16 in / bf16 out, full 256x128 tiles - built for TWO
// co-resident workgroups per CU.
//
// Why: the 256x256 ping-pong kernel (gemm.hip) owns a whole CU (128 KB of LDS, 8 waves x 228 registers), so the first
// operand round trip of a tile and its store-issue-bound epilogue (9-13 us per tile, measured) run with the matrix
// cores idle.  At K = 1024 (RoBERTa's qkv / out / fc1 projections: 16 K tiles per output tile) that is 40 % of a
// tile's life: 45 us per round of tiles against 26 us per 1024 k inside the K = 4096 GEMM.  Here a workgroup is 4 waves
// (2 x 2, wave tile 128 x 64 = 128 accumulator registers, the same fragment economy as the ping-pong kernel), its K
// tile is 32 wide and lives in a 3-slot ring of 24 KB (A 256 rows + B 128 rows, 64 bytes per row), so two workgroups
// fit a CU (2 x 72 KB LDS, 2 waves per SIMD at <= 256 registers).  They are independent: while one is in its prologue
// or epilogue the other's MFMAs have the SIMDs to themselves, and inside the main loops the two waves of a SIMD
// alternate between {barrier, DMA issue, fragment reads} and {16 MFMAs} without any hand-made phase offset.
//
// LDS image of a slot (lane-linear for global_load_lds_dwordx4, swizzle on the SOURCE address): row r (64 bytes =
// four 16-byte k-chunks) at byte r*64, chunk c stored at position c ^ ((r >> 2) & 3).  A ds_read_b128 service group
// (16 lanes = 16 rows of one k-chunk, rows {0-3, 12-15, 20-27} or {4-11, 16-19, 28-31}) then touches 16 distinct
// 16-byte pieces of the 256-byte bank row.
// Ring: tile kt+2 is issued after the barrier of iteration kt into the slot tile kt-1 was read from (every wave has
// consumed its kt-1 fragments before it arrives at that barrier: WAR), a wave waits for its own pieces of tile kt with
// a counted vmcnt(6) before the barrier (RAW); loads are never drained inside the loop.
#include "gemm_common.h"

typedef __attribute__((address_space(3))) void* duo_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* duo_glb_ptr_t;

namespace {
constexpr int BM = 256, BN = 128, BK = 32, NS = 3;
constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, SLOT = A_BYTES + B_BYTES;

// bf16 tile through the freed ring (256 rows x 256 bytes, 16-byte chunk index XORed with the row): every output row
// leaves as whole 128-byte lines.
template <int ACT>
__device__ __forceinline__ void duo_store(f32x16 (&acc)[4][2], const GemmArgs& p, int m0, int n0, int wm, int wn,
                                          int lane, int tid, uint16_t* cs) {
  const int lh = lane >> 5;
  // the bias pieces of this lane's 8 column quads and 4 rows are fetched up front as ONE batch of loads: fetched where
  // they are used (inside the block loops, under the block-uniform mode test) every block waits for its own round trip
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
        b4[j][g] = *reinterpret_cast<const f32x4_t*>(p.bias + n0 + wn * 64 + j * 32 + 8 * g + 4 * lh);
  } else if (p.bias_mode == 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) bm[i] = p.bias[m0 + wm * 128 + i * 32 + (lane & 31)];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wm * 128 + i * 32 + (lane & 31);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = wn * 64 + j * 32 + 8 * g + 4 * lh;
        f32x4_t v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (acc[i][j][4 * g + e] + b4[j][g][e] + bm[i]) * p.alpha;
        epi_act4<ACT>(v);
        const int ch = (col >> 3) ^ (row & 15);
        u32x2 w = {pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3])};
        *reinterpret_cast<u32x2*>(cs + row * BN + ch * 8 + (col & 7)) = w;
      }
  }
  __syncthreads();
  uint16_t* C = static_cast<uint16_t*>(p.C);
#pragma unroll
  for (int it = 0; it < BM * 16 / 256; ++it) {
    const int c = tid + it * 256, row = c >> 4, ch = c & 15;
    const u32x4 o = *reinterpret_cast<const u32x4*>(cs + row * BN + ((ch ^ (row & 15)) << 3));
    *reinterpret_cast<u32x4*>(C + (long)(m0 + row) * p.ldc + n0 + ch * 8) = o;
  }
}

template <int ABL, bool REG>
__global__ __launch_bounds__(256, 2) void gemm_nt_duo_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * SLOT];
  gemm_ts_enter(p);
  const int tid = threadIdx.x, lane = tid & 63, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int K = p.K;
  const int tiles_n = p.N / BN, tiles_m = p.M / BM;
  // the second resident of every CU starts late (p.stagger x 64*127 clocks), so the pair never meets in its epilogues
  if (p.stagger > 0 && blockIdx.x >= 256 && blockIdx.x < 512)
    for (int s = 0; s < p.stagger; ++s) __builtin_amdgcn_s_sleep(127);
  int tile_id;
  {
    const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
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

  // DMA sources: piece q (1 KiB = 16 rows x 4 chunks) of an operand, lane -> row q*16 + (lane >> 2), stored chunk
  // position lane & 3, which holds source chunk (lane & 3) ^ ((row >> 2) & 3) = (lane & 3) ^ ((lane >> 4) & 3)
  const int crow = lane >> 2, csrc = (lane & 3) ^ ((lane >> 4) & 3);
  const uint16_t* asrc[4];
  const uint16_t* bsrc[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) asrc[j] = A + (long)(m0 + (wave * 4 + j) * 16 + crow) * p.lda + csrc * 8;
#pragma unroll
  for (int j = 0; j < 2; ++j) bsrc[j] = B + (long)(n0 + (wave * 2 + j) * 16 + crow) * p.ldb + csrc * 8;
  auto issue = [&](int kt, int slot) __attribute__((always_inline)) {
    unsigned char* sa = smem + slot * SLOT + wave * 4096;
    unsigned char* sb = smem + slot * SLOT + A_BYTES + wave * 2048;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((duo_glb_ptr_t)(asrc[j] + kt * BK), (duo_lds_ptr_t)(sa + j * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((duo_glb_ptr_t)(bsrc[j] + kt * BK), (duo_lds_ptr_t)(sb + j * 1024), 16, 0, 0);
  };

  // fragment offsets inside a slot: k-substep ks, lane half lh -> chunk ks*2 + lh
  const int x = (lane >> 2) & 3;
  int a_off[2], b_off[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_off[ks] = (wm * 128 + (lane & 31)) * 64 + (((ks * 2 + lh) ^ x) << 4);
    b_off[ks] = A_BYTES + (wn * 64 + (lane & 31)) * 64 + (((ks * 2 + lh) ^ x) << 4);
  }

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = K / BK;
  if constexpr (REG) {
    // REGISTER-staged ring of TWO slots: G(t) = six global_load_dwordx4 of tile t into registers, W(t) = six ds_write_b128
    // into slot t & 1 (same image as the DMA form), R(t) = twelve fragment reads.  Iteration kt:
    //   lgkmcnt(0) [my W(kt) is in LDS] | barrier | R(kt) | 8 MFMAs | W(kt+1) (waits for G(kt+1), one iteration old) |
    //   G(kt+2) | 8 MFMAs
    // The barrier also says every wave has finished R(kt-1), whose slot W(kt+1) overwrites.
    u32x4 ga[4], gb[2];
    auto gload = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 4; ++j) ga[j] = *reinterpret_cast<const u32x4*>(asrc[j] + kt * BK);
#pragma unroll
      for (int j = 0; j < 2; ++j) gb[j] = *reinterpret_cast<const u32x4*>(bsrc[j] + kt * BK);
    };
    auto lwrite = [&](int slot) __attribute__((always_inline)) {
      unsigned char* sa = smem + slot * SLOT + wave * 4096 + lane * 16;
      unsigned char* sb = smem + slot * SLOT + A_BYTES + wave * 2048 + lane * 16;
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(sa + j * 1024) = ga[j];
#pragma unroll
      for (int j = 0; j < 2; ++j) *reinterpret_cast<u32x4*>(sb + j * 1024) = gb[j];
    };
    gload(0);
    lwrite(0);
    if (nk > 1) gload(1);
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* t = smem + (kt & 1) * SLOT;
      bf16x8 a[2][4], b[2][2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[ks][i] = *reinterpret_cast<const bf16x8*>(t + a_off[ks] + i * 2048);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[ks][j] = *reinterpret_cast<const bf16x8*>(t + b_off[ks] + j * 2048);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0][j], a[0][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nk) {
        if constexpr (ABL != 2) lwrite((kt + 1) & 1);
      }
      if (kt + 2 < nk) {
        if constexpr (ABL == 0) gload(kt + 2);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[1][j], a[1][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
  issue(0, 0);
  if (nk > 1) issue(1, 1);
  int st = 0, nst = 2;
  for (int kt = 0; kt < nk; ++kt) {
    if constexpr (ABL == 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 2 < nk) { if constexpr (ABL == 0) issue(kt + 2, nst); else if constexpr (ABL == 1) issue((kt + 2) & 1, nst); }
    const unsigned char* t = smem + st * SLOT;
    bf16x8 a[2][4], b[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a[ks][i] = *reinterpret_cast<const bf16x8*>(t + a_off[ks] + i * 2048);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[ks][j] = *reinterpret_cast<const bf16x8*>(t + b_off[ks] + j * 2048);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ks][j], a[ks][i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    st = st + 1 == NS ? 0 : st + 1;
    nst = nst + 1 == NS ? 0 : nst + 1;
  }
  }
  __syncthreads();                                       // every wave is done with the ring: it becomes the staging area
  uint16_t* cs = reinterpret_cast<uint16_t*>(smem);
  switch (p.act) {                                       // block-uniform
    case 1: duo_store<1>(acc, p, m0, n0, wm, wn, lane, tid, cs); break;
    case 2: duo_store<2>(acc, p, m0, n0, wm, wn, lane, tid, cs); break;
    default: duo_store<0>(acc, p, m0, n0, wm, wn, lane, tid, cs); break;
  }
  gemm_ts_exit(p);
}
}  // namespace

int launch_gemm_duo(const GemmArgs& a, hipStream_t stream) {
  static const int stagger = getenv("TELL_DUO_STAGGER") ? atoi(getenv("TELL_DUO_STAGGER")) : 0;
  GemmArgs ap = a;
  ap.stagger = stagger;
  const unsigned grid = (unsigned)(a.M / BM) * (unsigned)(a.N / BN);
  static const int abl = getenv("TELL_DUO_ABL") ? atoi(getenv("TELL_DUO_ABL")) : 0;   // timing probes (wrong results)
  static const int reg = getenv("TELL_DUO_REG") ? atoi(getenv("TELL_DUO_REG")) : 0;
  if (reg) {
    if (abl == 1) hipLaunchKernelGGL((gemm_nt_duo_kernel<1, true>), dim3(grid), dim3(256), 0, stream, ap);
    else if (abl == 2) hipLaunchKernelGGL((gemm_nt_duo_kernel<2, true>), dim3(grid), dim3(256), 0, stream, ap);
    else hipLaunchKernelGGL((gemm_nt_duo_kernel<0, true>), dim3(grid), dim3(256), 0, stream, ap);
  } else {
    if (abl == 1) hipLaunchKernelGGL((gemm_nt_duo_kernel<1, false>), dim3(grid), dim3(256), 0, stream, ap);
    else if (abl == 2) hipLaunchKernelGGL((gemm_nt_duo_kernel<2, false>), dim3(grid), dim3(256), 0, stream, ap);
    else hipLaunchKernelGGL((gemm_nt_duo_kernel<0, false>), dim3(grid), dim3(256), 0, stream, ap);
  }
  return tell_check_launch("gemm_nt_duo");
}
