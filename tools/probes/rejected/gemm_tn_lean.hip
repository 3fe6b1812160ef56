// REJECTED (round 5): kept for the record, not compiled into the library.  Lean K-major fp32 weight-gradient kernel (DMA into
// a group-skewed LDS image, tr reads from 2 launch-constant VGPRs + immediates, bias gradient off the matrix cores, ~140
// instructions per K tile against 258) with one problem per XCD.  Correct (55 full-size / decoder / trainer tests green),
// SLOWER in the decoder half: 6.72 ms with the general grouped kernel, 6.99 ms with this body under the launch-wide
// round-robin, 7.18 ms with one problem per XCD (and 7.93 ms with the 16 384-row article gradients on it too).  These
// launches read operands nobody has touched since the forward pass: what they need is bytes in flight - the general body
// holds two workgroups per CU with two K tiles each in registers (128 KB per CU), this one 104 KB of LDS = one workgroup
// with two tiles (64 KB per CU) - not fewer instructions.  It needed gemm_group.h (the GemmGroup structs of gemm.hip in a
// header) and a bucket in tell_gemm_grouped.
// Weight gradients dW[n][k] = sum_t dY[t][n] X[t][k] (both operands K-major: the reduction index is the ROW, the way the
// forward pass left them), fp32 output, as grouped launches - the lean form of gemm.hip's gemm_tx_group_kernel for the
// problems that are whole 128x128 tiles with a reduction that is a multiple of 64 rows (the ~80 weight gradients of a
// decoder backward pass: 1024 = T x B rows; the context K / V projections: 1568 .. 16384 rows go to the wide kernel).
//
// Why: the general K-major body stages through registers (global_load -> VGPR -> ds_write with bounds selects and the
// fused column sums per chunk), 258 instructions per 64-row K tile for 16 MFMAs per wave; with two 4-wave workgroups per
// CU a SIMD issues from one or two waves, one instruction per ~4 clk each ("one wave per SIMD").  Here:
//   * both operand tiles go global -> LDS by buffer_load_dwordx4 ... lds: one launch-constant 32-bit lane offset per
//     operand + one scalar row offset per instruction; a tile past the end of K is a num_records = 0 descriptor, so every
//     step issues the same 8 DMA instructions per wave and the in-flight count is a constant;
//   * LDS image of a [64 k][128 m] tile: DMA instruction g (0..15) carries rows g, g + 16, g + 32, g + 48 (4 x 256 B,
//     lane-linear as the DMA needs) into a 1088-byte group - the 4 CONSECUTIVE k rows a ds_read_b64_tr_b16 touches are
//     then 1088 bytes apart (disjoint bank quarters, what the register-staged body gets from a 64-byte row skew);
//   * fragment addresses = 2 launch-constant VGPRs per operand + immediates (k-substep s: + 256 s; column block: + 64 i),
//     the K loop unrolled over the 3 stages;
//   * the bias gradient (column sums of dY, fused like in the general body) comes off the matrix cores: one more MFMA per
//     A fragment against a fragment of ones, in the workgroups of column-tile 0 only.
// ~100 instructions per K tile for 16 (18) MFMAs.
#include "common.h"
#include "gemm_common.h"
#include "gemm_epi.h"
#include "gemm_group.h"

namespace {
typedef __attribute__((address_space(3))) void* tn_lds_ptr_t;
typedef __attribute__((ext_vector_type(4))) short tn_s16x4;
typedef __attribute__((address_space(3))) tn_s16x4* tn_lds_s16x4_ptr;
constexpr int TN_GROUP = 1088, TN_OP = 16 * TN_GROUP, TN_STAGE = 2 * TN_OP, TN_NS = 3;

template <int I, int N, typename F>
__device__ __forceinline__ void tn_static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); tn_static_for<I + 1, N>(f); }
}
__device__ __forceinline__ bf16x8 tn_frag(const unsigned char* lo, const unsigned char* hi) {
  const tn_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tn_lds_s16x4_ptr)lo);
  const tn_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tn_lds_s16x4_ptr)hi);
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// C[M, N] (+)= A^T . B:  A = [K rows][M] (lda), B = [K rows][N] (ldb), fp32 C; M % 128 == N % 128 == K % 64 == 0
__device__ __forceinline__ void gemm_tn_lean_body(const GemmArgs& p, const int tile_id, unsigned char* smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int M = p.M, N = p.N, K = p.K;
  const int tiles_n = N >> 7;
  const int tm = tile_id / tiles_n, tn = tile_id % tiles_n;
  const int m0 = tm * 128, n0 = tn * 128;

  // ---- DMA: lane = (row 16 * (lane >> 4) of the instruction's four, 16-byte chunk lane & 15)
  const unsigned a_off = (unsigned)(16 * (lane >> 4)) * (unsigned)(p.lda * 2) + (unsigned)(m0 + (lane & 15) * 8) * 2;
  const unsigned b_off = (unsigned)(16 * (lane >> 4)) * (unsigned)(p.ldb * 2) + (unsigned)(n0 + (lane & 15) * 8) * 2;
  const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t srd_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, (int)0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t nil_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t nil_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, 0, 0x00020000);
  const int nk = K >> 6;
  int nt = 0;                                         // next K tile to issue
  const int row_a = (int)(p.lda * 2), row_b = (int)(p.ldb * 2);       // bytes per k row
  auto issue = [&](int stage) __attribute__((always_inline)) {
    const bool more = nt < nk;
    const __amdgpu_buffer_rsrc_t ra = more ? srd_a : nil_a;
    const __amdgpu_buffer_rsrc_t rb = more ? srd_b : nil_b;
    unsigned char* sa = smem + stage * TN_STAGE + (wave * 4) * TN_GROUP;
    unsigned char* sb = sa + TN_OP;
    const int k0 = nt * 64 + wave * 4;                // first of this wave's four row groups
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (tn_lds_ptr_t)(sa + j * TN_GROUP), 16, (int)a_off, (k0 + j) * row_a, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (tn_lds_ptr_t)(sb + j * TN_GROUP), 16, (int)b_off, (k0 + j) * row_b, 0, 0);
    ++nt;
  };

  // ---- tr-read addresses: k row 8 * (lane >> 5) + ((lane & 15) >> 2) (+ 4), column 16 * ((lane >> 4) & 1) + 4 * (lane & 3)
  const int r0 = 8 * (lane >> 5) + ((lane & 15) >> 2), c0 = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int a_lo = r0 * TN_GROUP + (wm * 64 + c0) * 2, a_hi = a_lo + 4 * TN_GROUP;
  const int b_lo = TN_OP + r0 * TN_GROUP + (wn * 64 + c0) * 2, b_hi = b_lo + 4 * TN_GROUP;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // fused column sums of A (bias gradient): column-tile 0's workgroups, the wn == 0 waves (both wn waves read the same A)
  const bool want_asum = p.asum != nullptr && tn == 0 && wn == 0;       // wave-uniform
  f32x16 asum[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) asum[i][r] = 0.f;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 ones_s = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);

#pragma unroll
  for (int st = 0; st < TN_NS - 1; ++st) issue(st);

  auto step = [&](auto st_tag) __attribute__((always_inline)) {
    constexpr int ST = decltype(st_tag)::value;
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((TN_NS - 2) * 8) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned char* ts = smem + ST * TN_STAGE;
    // fragments of k-substep s + 1 are requested before the MFMAs of substep s issue (the LDS latency sits behind this
    // wave's own matrix work: one wave per SIMD has nobody else to hide it); the next tile's DMA goes out behind the
    // first two substeps' reads
    bf16x8 fa[2][2], fb[2][2];
    auto ldfrag = [&](int s, int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[buf][i] = tn_frag(ts + a_lo + s * 256 + i * 64, ts + a_hi + s * 256 + i * 64);
        fb[buf][i] = tn_frag(ts + b_lo + s * 256 + i * 64, ts + b_hi + s * 256 + i * 64);
      }
    };
    ldfrag(0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s + 1 < 4) ldfrag(s + 1, (s + 1) & 1);
      if (s == 0) issue((ST + TN_NS - 1) % TN_NS);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[s & 1][j], fa[s & 1][i], acc[i][j], 0, 0, 0);
      if (want_asum) {
#pragma unroll
        for (int i = 0; i < 2; ++i) asum[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, fa[s & 1][i], asum[i], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  int kt = 0;
  for (; kt + TN_NS <= nk; kt += TN_NS) tn_static_for<0, TN_NS>([&](auto tag) { step(tag); });
  tn_static_for<0, TN_NS - 1>([&](auto tag) { if (kt < nk) { step(tag); ++kt; } });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (trailing num_records = 0 loads still write their zeros)

  if (want_asum && lane < 32) {
    // C layout of mfma(ones, a): every row holds sum_k a[m][k] for column m = lane & 31: register 0 of lanes 0-31
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float* dst = p.asum + m0 + wm * 64 + i * 32 + lane;
      *dst += p.asum_scale * asum[i][0];
    }
  }
  gemm_epilogue<float, 2, 2>(acc, p, m0 + wm * 64, n0 + wn * 64, lane, M, N);
}

// One problem lives on ONE XCD: hardware places workgroup b on XCD b % 8, so XCD x walks its own list of problems (slot
// b >> 3 of that list), 32 workgroups at a time: 4 tile rows x 8 tile columns of a 1024 x 1024 gradient = 4 stripes of dY +
// 8 of X = 3 MB, inside the XCD's 4 MB L2.  (With the launch-wide round-robin of gemm.hip's grouped kernels every XCD pulls
// one stripe of dY and ALL of X of every problem: 432 MB of L2 fills for 96 MB of operands per launch.)
struct TnGroup {
  GroupProblem pr[GROUP_MAX];        // sorted by XCD
  int slot0[GROUP_MAX + 1];          // first slot of problem i inside its XCD's list (slot0[i + 1] - slot0[i] tiles when same XCD)
  int tiles[GROUP_MAX];
  int xfirst[9];                     // problems [xfirst[x], xfirst[x + 1]) belong to XCD x
};
__global__ __launch_bounds__(256) void gemm_tn_lean_group_kernel(TnGroup g) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[TN_NS * TN_STAGE];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  int i = g.xfirst[xcd];
  const int end = g.xfirst[xcd + 1];
  while (i < end && slot >= g.slot0[i] + g.tiles[i]) ++i;             // block-uniform
  if (i >= end) return;
  const GemmArgs p = group_args(g.pr[i]);
  gemm_tn_lean_body(p, slot - g.slot0[i], smem);
}
}  // namespace

int launch_group_tn_lean(const GemmGroup& g, hipStream_t stream) {
  TnGroup t;
  long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int owner[GROUP_MAX], order[GROUP_MAX], tl[GROUP_MAX];
  for (int i = 0; i < g.n; ++i) { tl[i] = (g.pr[i].M >> 7) * (g.pr[i].N >> 7); order[i] = i; }
  for (int i = 1; i < g.n; ++i) {                                     // largest first, then always the least loaded XCD
    const int x = order[i];
    int j = i - 1;
    while (j >= 0 && tl[order[j]] < tl[x]) { order[j + 1] = order[j]; --j; }
    order[j + 1] = x;
  }
  for (int k = 0; k < g.n; ++k) {
    int best = 0;
    for (int x = 1; x < 8; ++x) if (load[x] < load[best]) best = x;
    owner[order[k]] = best;
    load[best] += tl[order[k]];
  }
  int n = 0;
  long most = 0;
  for (int x = 0; x < 8; ++x) {
    t.xfirst[x] = n;
    int slot = 0;
    for (int k = 0; k < g.n; ++k) {
      const int i = order[k];
      if (owner[i] != x) continue;
      t.pr[n] = g.pr[i]; t.slot0[n] = slot; t.tiles[n] = tl[i];
      slot += tl[i]; ++n;
    }
    most = slot > most ? slot : most;
  }
  t.xfirst[8] = n;
  t.slot0[n] = 0;
  if (most == 0) return TELL_OK;
  hipLaunchKernelGGL(gemm_tn_lean_group_kernel, dim3((unsigned)(most * 8)), dim3(256), 0, stream, t);
  return tell_check_launch("gemm_grouped (tn lean)");
}
