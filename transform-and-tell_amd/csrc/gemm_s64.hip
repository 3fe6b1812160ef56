// 64x64-tile NT GEMM / implicit convolution (bf16, K % 64 == 0) with a K loop written for ONE wave per SIMD.
//
// The ResNet bottleneck convolutions at B = 32 and the decoder's M = 1024 GEMMs are 200-1600 tiles of 64x64: a compute
// unit holds one or two 4-wave workgroups, i.e. one or two waves per SIMD, and a SIMD issues at most one instruction
// of a wave every ~4 clocks.  The general direct-to-LDS body (gemm.hip gemm_nt_glds_body) spends ~190 instructions per
// 64-deep K tile on such a tile - run-time stage arithmetic for every ds_read, a division to decode the convolution tap,
// 64-bit source pointers and an EXEC-masked branch per DMA, a run-time vmcnt switch, four ds_read -> wait -> MFMA round
// trips - for 4 MFMAs: 700-1400 clk per K tile where the matrix cores need 128 and the LDS-DMA path, measured alone
// (tools/probes/stage_rate.hip), delivers the tile's 16 KB in ~210 (78 GB/s per workgroup, 145 GB/s per CU from L2
// against the 40 GB/s these launches were getting).  PMC on layer3's 3x3 convolution: 92 % L2 hits, FETCH 14 MB for
// 231 MB staged - the operands were in L2 all along; the waves were busy issuing.
//
// This kernel keeps the tile, the LDS image (source-side swizzle, conflict-free ds_read_b128), the 4-stage ring and the
// epilogues of that body and rewrites the loop:
//   * buffer_load_dwordx4 ... lds with 32-bit per-lane offsets fixed for the whole launch + ONE scalar offset per K tile
//     (plain GEMM: kt * 128 bytes; convolution: the tap's pixel shift + channel block, kept incrementally - no division);
//     a padding pixel is an out-of-range offset (the buffer unit returns zeros: no zero page, no branch), a tile past the
//     end of K a descriptor with num_records = 0 - every step issues the same 4 DMA instructions, so the in-flight count
//     is a constant (s_waitcnt vmcnt(8), no switch);
//   * the K loop unrolled over the 4 stages: every ds_read address is a launch-constant VGPR + an immediate;
//   * all fragment reads of a K tile issued together right after the barrier, the next tile's DMA behind them, then the
//     MFMAs (the compiler counts lgkmcnt down in front of each);
//   * the wave index in a scalar register (LDS destinations of the DMA = s_mov m0).
// MEASURED (MI355X, B = 32, tools/bench_conv.py / bench_decoder_gemms.py, us per launch, general body -> this kernel):
// layer3 conv2 (K = 2304) 25.0 -> 14.4, layer4 conv2 (4608) 37.5 -> 17.8, layer4 conv1 (2048) 12.9 -> 9.7, layer2 conv2
// (1152) 20.5 -> 17.0, layer3 conv1 (1024) 9.1 -> 8.2; decoder q / out / linear2 (1024^3) 7.8 -> 5.7, tap logits 7.9 ->
// 5.7.  Short reductions lose (K = 128 / 256: 13.5 -> 18.9, 10.4 -> 13.0 - 64 KB of LDS is two workgroups per CU where
// the 2-stage body fits five, and there the launch is prologue + epilogue): the callers take this kernel from K = 1024.
// (A 2-stage instantiation - 32 KB, five workgroups per CU - for K < 1024 ties the general body there: K = 256 10.0 against
// 10.1 us, 128: 13.4 / 13.1, 512: 9.7 / 9.2, 576: 16.3 / 20.3; ResNet-152 4.32 against 4.30 ms with it; not kept.)
// (A variant in which the waves split the K TILE instead of the output tile - four independent accumulators, half the
// fragment reads, partial tiles folded through LDS - measured the same or slower, 14.7 / 17.2 / 10.2 us on the first
// three shapes, and changes the summation order; not kept.)
#include "common.h"
#include "gemm_common.h"
#include "gemm_epi.h"

namespace {
typedef __attribute__((address_space(3))) void* s64_lds_ptr_t;
template <int I, int N, typename F>
__device__ __forceinline__ void static_for_tail(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for_tail<I + 1, N>(f); }
}
constexpr int S_STAGE = 16384, S_HALF = 8192;
constexpr unsigned S_OOB = 0x80000000u;            // >= num_records: the buffer unit returns zeros

// S_NS: stages of the ring (S_NS - 1 K tiles in flight); 4 = 64 KB of LDS, two workgroups per CU
template <typename OutT, bool CONV, int S_NS>
__global__ __launch_bounds__(256) void gemm_nt_s64_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[S_NS * S_STAGE];
  gemm_ts_enter(p);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int M = p.M;
  if (p.m_dev) { int md = *p.m_dev; M = md < M ? md : M; }
  const int N = p.N, K = p.K;
  const int tiles_n = (N + 63) >> 6, tiles_m = (M + 63) >> 6;
  // XCD-aware tile mapping + grouped traversal (gemm.hip gemm_nt_glds_body)
  int tile_id;
  {
    const int nwg = (int)gridDim.x, orig = (int)blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  if (tile_id >= tiles_m * tiles_n) return;
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
  const int m0 = tm * 64, n0 = tn * 64;

  // ---- per-lane DMA offsets (bytes), fixed for the launch
  unsigned a_off[2], b_off[2];
  int cv_ih[2], cv_iw[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int s = (wave * 2 + j) * 64 + lane, pr = s >> 4, l16 = (s & 15) ^ (pr & 15);
    int row = m0 + 2 * pr + (l16 >> 3);
    row = row < M ? row : M - 1;
    if constexpr (CONV) {
      const int ow = row % p.conv_OW, t = row / p.conv_OW, oh = t % p.conv_OH, b = t / p.conv_OH;
      cv_ih[j] = oh * p.conv_stride - p.conv_pad;
      cv_iw[j] = ow * p.conv_stride - p.conv_pad;
      // (may be "negative" as a pixel index: only used after the tap shift has been added, 32-bit wrap-around)
      a_off[j] = (unsigned)((((b * p.conv_H + cv_ih[j]) * p.conv_W + cv_iw[j]) << (6 + p.conv_cshift)) * 2 + (l16 & 7) * 16);
    } else {
      cv_ih[j] = cv_iw[j] = 0;
      a_off[j] = (unsigned)row * (unsigned)(p.lda * 2) + (l16 & 7) * 16;
    }
    int rn = n0 + 2 * pr + (l16 >> 3);
    rn = rn < N ? rn : N - 1;
    b_off[j] = (unsigned)rn * (unsigned)(p.ldb * 2) + (l16 & 7) * 16;
  }
  // ---- fragment read offsets inside a stage: row r, 16-byte k-chunk c: (r>>1)*256 + ((((r&1)<<3)|c) ^ ((r>>1)&15))*16
  int a_fo[4], b_fo[4];                                        // one per 16-deep k-substep
  {
    auto fo = [&](int r, int c) { return (r >> 1) * 256 + (((((r & 1) << 3) | c) ^ ((r >> 1) & 15)) << 4); };
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = ks * 2 + (lane >> 5);
      a_fo[ks] = fo(wm * 32 + (lane & 31), c);
      b_fo[ks] = S_HALF + fo(wn * 32 + (lane & 31), c);
    }
  }

  f32x16 acc[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;

  const int nk = K >> 6;
  // ---- the DMA stream: tile `nt` is the next one to issue (sequential); convolution: its tap and channel block
  int nt = 0;
  int tap_kh = 0, tap_kw = 0, tap_c0 = 0;
  const int cin = CONV ? (64 << p.conv_cshift) : 0;
  auto issue = [&](int stage) __attribute__((always_inline)) {
    const bool more = nt < nk;
    unsigned char* sa = smem + stage * S_STAGE + wave * 2048;
    unsigned char* sb = sa + S_HALF;
    // (one scalar select - the num_records word - instead of two whole descriptors)
    const int nrec = more ? (int)S_OOB : 0;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, nrec, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, nrec, 0x00020000);
    if constexpr (CONV) {
      const int delta = (((tap_kh * p.conv_W + tap_kw) << (6 + p.conv_cshift)) + tap_c0) * 2;      // bytes, uniform
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bool in = (unsigned)(cv_ih[j] + tap_kh) < (unsigned)p.conv_H && (unsigned)(cv_iw[j] + tap_kw) < (unsigned)p.conv_W;
        const unsigned vo = in ? a_off[j] + (unsigned)delta : S_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (s64_lds_ptr_t)(sa + j * 1024), 16, (int)vo, 0, 0, 0);
      }
      tap_c0 += 64;
      if (tap_c0 == cin) { tap_c0 = 0; if (++tap_kw == p.conv_KW) { tap_kw = 0; ++tap_kh; } }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (s64_lds_ptr_t)(sa + j * 1024), 16, (int)a_off[j], nt * 128, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (s64_lds_ptr_t)(sb + j * 1024), 16, (int)b_off[j], nt * 128, 0, 0);
    ++nt;
  };

#pragma unroll
  for (int st = 0; st < S_NS - 1; ++st) issue(st);

  // one K tile, stage ST (compile time: every ds_read is `launch-constant VGPR + immediate`)
  auto step = [&](auto st_tag) __attribute__((always_inline)) {
    constexpr int ST = decltype(st_tag)::value;
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((S_NS - 2) * 4) : "memory");   // this wave's pieces of the tile have landed (S_NS - 2 younger tiles in flight)
    __builtin_amdgcn_s_barrier();                        // everyone's have, and everyone is done with the stage issued into below
    asm volatile("" ::: "memory");
    const unsigned char* ts = smem + ST * S_STAGE;
    bf16x8 fa[4], fb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      fa[ks] = *reinterpret_cast<const bf16x8*>(ts + a_fo[ks]);
      fb[ks] = *reinterpret_cast<const bf16x8*>(ts + b_fo[ks]);
    }
    issue((ST + S_NS - 1) % S_NS);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)                        // (the order of the general body: bit-identical sums)
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks], fa[ks], acc[0][0], 0, 0, 0);
  };
  int kt = 0;
  if constexpr (S_NS == 4) {
    for (; kt + 4 <= nk; kt += 4) {
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{});
    }
  } else {
    for (; kt + 8 <= nk; kt += 8) {
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{});
      step(std::integral_constant<int, 4>{});
      step(std::integral_constant<int, 5>{});
      step(std::integral_constant<int, 6>{});
      step(std::integral_constant<int, 7>{});
    }
  }
  static_for_tail<0, S_NS - 1>([&](auto tag) { if (kt < nk) { step(tag); ++kt; } });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the trailing num_records = 0 loads still write their zeros)
  __syncthreads();                                       // every wave is done with the stages: they become the staging area

  gemm_nt_glds_epilogue<OutT, 64, 64, 2, 2, CONV, 4>(acc, p, smem, m0, n0, tm, wm, wn, lane, tid, M, N);
}
}  // namespace

// Is the problem one this kernel takes?  ONE test for the launch path and for tell_gemm_nt_plan (which used to name this
// kernel for every K % 64 == 0 shape, also where the launch then declined and a glds / register kernel ran).
bool gemm_s64_takes(const GemmArgs& a) {
  const bool conv = a.conv_zero != nullptr;
  if (a.K % 64 || a.K < 64 || a.M <= 0 || a.N <= 0) return false;
  if (conv && a.conv_cshift < 0) return false;                                     // the 7x7 stem gather stays where it is
  // 32-bit byte offsets into either operand; an offset of 2^31 is the "padding pixel" marker
  const long a_bytes = conv ? ((long)a.M / ((long)a.conv_OH * a.conv_OW) + 1) * a.conv_H * a.conv_W * (128L << a.conv_cshift)
                            : (long)a.M * a.lda * 2;
  const long b_bytes = (long)a.N * a.ldb * 2;
  if (a_bytes >= (1L << 31) || b_bytes >= (1L << 31) || (a.lda & 7) || (a.ldb & 7)) return false;
  if ((reinterpret_cast<uintptr_t>(a.A) & 15) || (reinterpret_cast<uintptr_t>(a.B) & 15)) return false;
  return (long)((a.M + 63) / 64) * ((a.N + 63) / 64) <= (1L << 30);
}

// -> TELL_OK after launching, 1 when the problem is not one this kernel takes (the caller keeps its own path)
template <typename OutT>
static int launch_s64(const GemmArgs& a, hipStream_t stream) {
  const bool conv = a.conv_zero != nullptr;
  if (!gemm_s64_takes(a)) return 1;
  const long tiles = (long)((a.M + 63) / 64) * ((a.N + 63) / 64);
  const dim3 grid((unsigned)tiles), block(256);
  // (an 8-stage instantiation - 128 KB, seven K tiles in flight - for launches of at most one workgroup per CU, the decoder's
  //  1024 x 1024 outputs whose weights come from HBM: decoder half 6.655 -> 6.650 ms, 15.1 -> 15.2 us per launch; not kept)
  if (conv) hipLaunchKernelGGL((gemm_nt_s64_kernel<OutT, true, 4>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((gemm_nt_s64_kernel<OutT, false, 4>), grid, block, 0, stream, a);
  return tell_check_launch("gemm_nt_s64");
}
int launch_gemm_s64(const GemmArgs& a, hipStream_t stream, int out_f32) {
  return out_f32 ? launch_s64<float>(a, stream) : launch_s64<uint16_t>(a, stream);
}
