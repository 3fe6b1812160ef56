// gemm_nt_q4_kernel: C[M,N] = act((A[M,K] . B[N,K]^T + bias) * alpha), bf16 in / out, whole 256x256 tiles - the RoBERTa
// projection GEMMs at B x 512 rows and the article K|V projection (fairseq TransformerSentenceEncoderLayer; the call
// site of the reference is tell/models/transformer_faces_objects.py:352-353).
//
// Round 4's answer to "the two LDS consumers of the ping-pong kernel collide" (DESIGN 3): FOUR waves own 128x128 of the
// tile each (16 v_mfma_f32_32x32x16_bf16 accumulators = all 256 AGPRs of a lane, one wave per SIMD), which cuts the
// fragment reads from 192 KB to 128 KB per K tile, and the whole K loop is ONE hand-placed instruction stream
// (gemm_q4_loop.inc, written by tools/gen_q4_loop.py - read its header for the schedule): one LDS read or one
// LDS-DMA instruction per MFMA gap, four barriers per K tile, the DMA stream two K tiles ahead in two LDS buffers and
// running on across output-tile boundaries (resident workgroups as in gemm_pp2.hip: the first two K tiles of the next
// output tile land while the epilogue runs).  Accumulators are PHYSICAL registers a[0:255] (clobbers of the statement):
// the epilogue fetches them eight at a time with v_accvgpr_read just before use - as "=a" operands hipcc copied all 256
// into VGPRs right behind the loop (and spilled).  The epilogue stores straight from registers: with the row mapping of
// the LDS image a lane owns 8 consecutive output columns per 16-byte store, and a store instruction costs its ~70 clk
// per CU whether its lanes cover whole lines or not - no LDS staging, no barriers.
// PMC (profiles/r04_pmc_gemm_lds.txt, RoBERTa's four shapes): SQ_LDS_BANK_CONFLICT 590 k -> 0 per launch, SQ_LDS_IDX_ACTIVE
// 10.9 M -> 6.3 M, matrix pipes busy 54 % -> 64 % of wave cycles.
#include "gemm_common.h"
#include "options.h"
#include "gemm_q4_loop.inc"
#include <utility>

namespace {
constexpr int QBM = 256, QBN = 256, QBK = 64;

template <typename F, int... Is>
__device__ __forceinline__ void q4_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void q4_static_for(F&& f) {
  q4_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// "s" operands of the asm statements must be PROVABLY wave-uniform: tile coordinates that went through the LDS broadcast of
// the tile queue are not (to the compiler), so the operand values themselves are passed through v_readfirstlane
__device__ __forceinline__ unsigned long long q4_uni64(unsigned long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
template <typename T> __device__ __forceinline__ unsigned long long q4_ptr(const T* q) {
  return q4_uni64(reinterpret_cast<unsigned long long>(q));
}

typedef __attribute__((ext_vector_type(8))) float f32x8_t;
// gemm_common.h's exact-erf GELU (Abramowitz-Stegun 7.1.28) on the EIGHT values of a 16-byte store at once: four independent
// v_pk_* chains that the scheduler interleaves - one pair at a time the 14 dependent packed operations (each with its
// wait state) ran back to back with nothing between them
__device__ __forceinline__ f32x8_t q4_gelu8(f32x8_t v) {
  const f32x8_t av = __builtin_elementwise_abs(v);
  const f32x8_t x = av * 0.70710678118654752f;
  f32x8_t q = x * 0.0000430638f + 0.0002765672f;
  q = q * x + 0.0001520143f;
  q = q * x + 0.0092705272f;
  q = q * x + 0.0422820123f;
  q = q * x + 0.0705230784f;
  q = q * x + 1.f;
  q *= q; q *= q; q *= q; q *= q;
  f32x8_t e;
#pragma unroll
  for (int k = 0; k < 8; ++k) e[k] = __builtin_amdgcn_rcpf(q[k]);
  const f32x8_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  return (av * -0.5f) * e + __builtin_elementwise_max(v, zero);
}

// the 8 accumulator values behind one 16-byte store: acc[I][j][2 Q + t], j = 0..3, t = 0, 1 (a[16 (4 I + j) + 2 Q + t])
template <int I, int Q> __device__ __forceinline__ void q4_read8(float (&v)[8]) {
  asm volatile(
      "v_accvgpr_read_b32 %0, a[%c8]\n\tv_accvgpr_read_b32 %1, a[%c9]\n\tv_accvgpr_read_b32 %2, a[%c10]\n\t"
      "v_accvgpr_read_b32 %3, a[%c11]\n\tv_accvgpr_read_b32 %4, a[%c12]\n\tv_accvgpr_read_b32 %5, a[%c13]\n\t"
      "v_accvgpr_read_b32 %6, a[%c14]\n\tv_accvgpr_read_b32 %7, a[%c15]"
      : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7])
      : "i"(16 * (4 * I + 0) + 2 * Q), "i"(16 * (4 * I + 0) + 2 * Q + 1), "i"(16 * (4 * I + 1) + 2 * Q),
        "i"(16 * (4 * I + 1) + 2 * Q + 1), "i"(16 * (4 * I + 2) + 2 * Q), "i"(16 * (4 * I + 2) + 2 * Q + 1),
        "i"(16 * (4 * I + 3) + 2 * Q), "i"(16 * (4 * I + 3) + 2 * Q + 1));
}

// Output mapping (tools/gen_q4_loop.py): lane (r = lane & 31, h = lane >> 5) of wave (w_r, w_c) owns for X fragment i the
// row m = 128 w_r + 8 (r >> 1) + 2 i + (r & 1) and, per (q = e >> 1), the 8 columns 128 w_c + 8 G .. + 7 with
// G = 4 (q >> 1) + 2 (q & 1) + h: column 8 G + 2 j + t holds acc[i][j][2 q + t].
// Arithmetic per value: ONE fused multiply-add  acc * alpha + bias'  (bias' = bias * alpha, per column - BROW false - or
// per row), two values per v_pk_fma_f32; with the 8 v_accvgpr_read and 4 v_cvt_pk_bf16_f32 that is 16 VALU issues per
// 16-byte store (the first version spent 34: separate add / multiply per value, and measured 5 us of arithmetic per tile).
template <int ACT, bool BROW, int ABL>
__device__ __forceinline__ void q4_store(const GemmArgs& p, int m0, int n0, int w_r, int w_c, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const int nb = n0 + 128 * w_c + 8 * h;
  const float alpha = p.alpha;
  const f32x2e_t alpha2 = {alpha, alpha};
  f32x2e_t b2[8][4];                                       // BROW false: bias' of the lane's 64 columns
  float bm[4];                                             // BROW true: bias' of the lane's 4 rows
  if constexpr (!BROW) {
    if (p.bias_mode == 1) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int G8 = 16 * (q & 1) + 32 * (q >> 1);
        const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(p.bias + nb + G8);
        const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(p.bias + nb + G8 + 4);
        b2[q][0] = f32x2e_t{lo[0], lo[1]} * alpha2; b2[q][1] = f32x2e_t{lo[2], lo[3]} * alpha2;
        b2[q][2] = f32x2e_t{hi[0], hi[1]} * alpha2; b2[q][3] = f32x2e_t{hi[2], hi[3]} * alpha2;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) b2[q][k] = f32x2e_t{0.f, 0.f};
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) bm[i] = p.bias[m0 + 128 * w_r + 8 * (r >> 1) + 2 * i + (r & 1)] * alpha;
  }
  uint16_t* C = static_cast<uint16_t*>(p.C);
  // ACT 5: C = res + dropout_p(result) - the transformer sub-layer residual (fairseq: x = residual + dropout(out_proj(.)) /
  // dropout(fc2(.))); the mask is the one tell_layernorm_fwd would draw for (seed, salt): element m N + n, common.h quad hash.
  // A store group is two aligned quads of one row; the residual pieces of fragment row I + 1 are fetched while row I is worked on.
  const uint16_t* R = static_cast<const uint16_t*>(p.res);
  uint32_t dsalt = 0;
  if constexpr (ACT == 5) dsalt = tell_step_salt(p.drop_salt, p.drop_step);
  u32x4 rres[2][8];
  auto fetch_res = [&](auto ic) __attribute__((always_inline)) {
    constexpr int I = decltype(ic)::value;
    const int m = m0 + 128 * w_r + 8 * (r >> 1) + 2 * I + (r & 1);
#pragma unroll
    for (int q = 0; q < 8; ++q)
      rres[I & 1][q] = *reinterpret_cast<const u32x4*>(R + (long)m * p.ld_res + nb + 16 * (q & 1) + 32 * (q >> 1));
  };
  if constexpr (ACT == 5) fetch_res(std::integral_constant<int, 0>{});
  q4_static_for<4>([&](auto ic) __attribute__((always_inline)) {
    constexpr int I = decltype(ic)::value;
    const int m = m0 + 128 * w_r + 8 * (r >> 1) + 2 * I + (r & 1);
    uint16_t* crow = C + (long)m * p.ldc + nb;
    if constexpr (ACT == 5 && I < 3) fetch_res(std::integral_constant<int, I + 1>{});
    q4_static_for<8>([&](auto qc) __attribute__((always_inline)) {
      constexpr int Q = decltype(qc)::value;
      constexpr int G8 = 16 * (Q & 1) + 32 * (Q >> 1);
      float a8[8];
      q4_read8<I, Q>(a8);
      u32x4 o;
      f32x8_t v;                                           // columns G8 + 2 k + t: acc[I][k][2 Q + t]
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f32x2e_t w = {a8[2 * k], a8[2 * k + 1]};
        if constexpr (BROW) w = __builtin_elementwise_fma(w, alpha2, f32x2e_t{bm[I], bm[I]});
        else w = __builtin_elementwise_fma(w, alpha2, b2[Q][k]);
        v[2 * k] = w[0]; v[2 * k + 1] = w[1];
      }
      if constexpr (ACT == 1) v = __builtin_elementwise_max(v, f32x8_t{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f});
      else if constexpr (ACT == 2) v = q4_gelu8(v);
      else if constexpr (ACT == 5) {
        if (p.drop_thr) {                                  // block-uniform
          const uint64_t quad = ((uint64_t)m * (uint64_t)p.N + (uint64_t)(nb + G8)) >> 2;
#pragma unroll
          for (int hq = 0; hq < 2; ++hq) {
            bool k0, k1, k2, k3;
            tell_keep4_bits(tell_quad_x(p.drop_seed, quad + hq), tell_quad_y(dsalt, quad + hq), p.drop_thr, k0, k1, k2, k3);
            v[4 * hq] = k0 ? v[4 * hq] * p.drop_inv_keep : 0.f; v[4 * hq + 1] = k1 ? v[4 * hq + 1] * p.drop_inv_keep : 0.f;
            v[4 * hq + 2] = k2 ? v[4 * hq + 2] * p.drop_inv_keep : 0.f; v[4 * hq + 3] = k3 ? v[4 * hq + 3] * p.drop_inv_keep : 0.f;
          }
        }
        const u32x4 rr = rres[I & 1][Q];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          v[2 * k] += __uint_as_float(rr[k] << 16);
          v[2 * k + 1] += __uint_as_float(rr[k] & 0xffff0000u);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = pack2_bf16(v[2 * k], v[2 * k + 1]);
      if (ABL != 2 || o[0] == 0x12345678u) *reinterpret_cast<u32x4*>(crow + G8) = o;
    });
  });
}
template <int ABL>
__device__ __forceinline__ void q4_epilogue(const GemmArgs& p, int m0, int n0, int w_r, int w_c, int lane) {
  if (p.bias_mode == 2) {                                  // block-uniform
    switch (p.act) {
      case 1: q4_store<1, true, ABL>(p, m0, n0, w_r, w_c, lane); break;
      case 2: q4_store<2, true, ABL>(p, m0, n0, w_r, w_c, lane); break;
      default: q4_store<0, true, ABL>(p, m0, n0, w_r, w_c, lane); break;
    }
  } else {
    switch (p.act) {
      case 1: q4_store<1, false, ABL>(p, m0, n0, w_r, w_c, lane); break;
      case 2: q4_store<2, false, ABL>(p, m0, n0, w_r, w_c, lane); break;
      default: q4_store<0, false, ABL>(p, m0, n0, w_r, w_c, lane); break;
    }
  }
}

#define Q4_RUN_MAIN(TEXT)                                                                                             \
  asm volatile(TEXT                                                                                                   \
               :                                                                                                      \
               : "v"(xrd), "v"(wrd), "v"(xvo), "v"(wvo), "s"(xc), "s"(wc), "s"(xn), "s"(wn), "s"(lda32), "s"(ldb32),  \
                 "s"(nkf), "s"(dstw)                                                                                  \
               : Q4_MAIN_CLOBBERS)

// VAR: schedule variant of the K loop (tools/gen_q4_loop.py VARIANTS); ABL: timing probes with wrong results (1: no
// epilogue at all, 2: epilogue arithmetic without the global stores)
template <int VAR, int ABL, bool RES = false>
__global__ __launch_bounds__(256) void gemm_nt_q4_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * Q4_BUF + 64];
  gemm_ts_enter(p);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w_r = wave >> 1, w_c = wave & 1;
  const int tiles_n = p.N / QBN, tiles_m = p.M / QBM;
  const int n_tiles = tiles_m * tiles_n;
  const int nk = p.K / QBK;
  const uint16_t* A = static_cast<const uint16_t*>(p.A);
  const uint16_t* B = static_cast<const uint16_t*>(p.B);

  auto tile_origin = [&](int vb, int& m0, int& n0) __attribute__((always_inline)) {
    const int nwg = n_tiles, xcd = vb & 7, q = nwg >> 3, r = nwg & 7;
    const int tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * tiles_n;
    const int g = tile_id / per_group, first_m = g * GROUP_M;
    const int gm = tiles_m - first_m < GROUP_M ? tiles_m - first_m : GROUP_M;
    const int in_g = tile_id - g * per_group;
    m0 = (first_m + in_g % gm) * QBM;
    n0 = (in_g / gm) * QBN;
  };

  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int r = lane & 31, h = lane >> 5;
  const unsigned xrd = lds0 + (16 * w_r + (r >> 1)) * Q4_PIECE + (r & 1) * 128 + h * 16;
  // W rows: bits 0 and 1 of the piece index swapped, so that the two half-waves of a lane pair own ADJACENT 16-byte column
  // groups (G = 4 (q >> 1) + 2 (q & 1) + h): a store instruction then writes 32 rows x 32 contiguous bytes instead of
  // 64 separate 16-byte pieces (the reads stay conflict-free: the 16 lanes of a ds_read_b128 group still cover 8 pieces
  // x 2 row parities)
  const int rp = ((r >> 1) & ~3) | (((r >> 1) & 1) << 1) | (((r >> 1) >> 1) & 1);
  const unsigned wrd = lds0 + Q4_OPER + (16 * w_c + rp) * Q4_PIECE + (r & 1) * 128 + h * 16;
  const unsigned xvo = (unsigned)((8 * wave + (lane >> 3)) * (int)p.lda * 2 + (lane & 7) * 16);
  const unsigned wvo = (unsigned)((8 * wave + (lane >> 3)) * (int)p.ldb * 2 + (lane & 7) * 16);
  const unsigned lda32 = (unsigned)p.lda * 64u, ldb32 = (unsigned)p.ldb * 64u;     // bytes per 32 rows
  const unsigned dstw = __builtin_amdgcn_readfirstlane(lds0 + wave * Q4_PIECE);

  // Tile order.  Static (p.queue == NULL): tiles b, b + grid, ...  Dynamic (as gemm_pp2.hip): one counter per XCD, workgroup b
  // lives on XCD b & 7 and takes tiles k * 8 + (b & 7); a workgroup that gets its CU late inside the training step takes
  // fewer tiles.  The K loop needs the NEXT tile when it starts (its last two bodies fetch that tile's operands), so the
  // counter is read one tile further ahead than in gemm_pp2.hip: the fetch issued at the top of tile i names tile i + 2
  // and is consumed behind the epilogue.  A workgroup stops fetching at its first out-of-range value, so a launch makes
  // exactly per_x + wg_x fetches per XCD and the one that draws the last value zeroes the counter again.
  const bool dyn = p.queue != nullptr;
  const int xcd_id = blockIdx.x & 7, per_x = n_tiles >> 3, wg_x = (int)gridDim.x >> 3;
  int* const qx = dyn ? p.queue + xcd_id : nullptr;
  volatile int* sq = reinterpret_cast<volatile int*>(smem + 2 * Q4_BUF);
  auto fetch = [&]() __attribute__((always_inline)) {      // thread 0 only
    const int k = atomicAdd(qx, 1);
    if (k == per_x + wg_x - 1) atomicExch(qx, 0);
    return k;
  };
  auto share = [&](int k) __attribute__((always_inline)) { // thread 0's value to the workgroup
    if (tid == 0) *sq = k;
    __syncthreads();
    const int v = __builtin_amdgcn_readfirstlane(*sq);      // (an LDS read is not provably wave-uniform: "s" operands need this)
    __syncthreads();
    return v;
  };
  int vb = blockIdx.x, nvb = vb + (int)gridDim.x;
  if (dyn) {
    const int k0 = share(tid == 0 ? fetch() : 0);
    vb = k0 < per_x ? k0 * 8 + xcd_id : n_tiles;
  }
  if (vb >= n_tiles) return;
  int m0, n0, tile_no = 0;
  (void)tile_no;
  tile_origin(vb, m0, n0);
  {
    const unsigned long long x0 = q4_ptr(A + (long)m0 * p.lda);
    const unsigned long long w0 = q4_ptr(B + (long)n0 * p.ldb);
    asm volatile(Q4_PROLOGUE_ASM
                 :
                 : "v"(xvo), "v"(wvo), "s"(x0), "s"(w0), "s"(lda32), "s"(ldb32), "s"(dstw)
                 : Q4_PROLOGUE_CLOBBERS);
  }
  if (dyn) {                                               // (its round trip overlaps the first operand round trip)
    const int k1 = share(tid == 0 ? fetch() : 0);
    nvb = k1 < per_x ? k1 * 8 + xcd_id : n_tiles;
  }
  for (;;) {
    const bool has_next = nvb < n_tiles;
    int kq = 0;
    if (dyn && has_next && tid == 0) kq = fetch();         // names the tile after the next one; consumed behind the epilogue
    int m1 = m0, n1 = n0;
    if (has_next) tile_origin(nvb, m1, n1);
    const unsigned long long xc = q4_ptr(A + (long)m0 * p.lda + 2 * QBK);
    const unsigned long long wc = q4_ptr(B + (long)n0 * p.ldb + 2 * QBK);
    const unsigned long long xn = q4_ptr(A + (long)m1 * p.lda);
    const unsigned long long wn = q4_ptr(B + (long)n1 * p.ldb);
    const unsigned nkf = __builtin_amdgcn_readfirstlane((unsigned)nk | (has_next ? 0x10000u : 0u));
    if constexpr (ABL == 3) {                              // instrumented: s_memtime stamps of wave 0 into p.aux
      unsigned long long* dbg = const_cast<unsigned long long*>(static_cast<const unsigned long long*>(p.aux));
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      unsigned long long ta, tb;
      asm volatile(Q4_MAIN_ASM_DBG
                   : "=&s"(ta), "=&s"(tb)
                   : "v"(xrd), "v"(wrd), "v"(xvo), "v"(wvo), "s"(xc), "s"(wc), "s"(xn), "s"(wn), "s"(lda32), "s"(ldb32),
                     "s"(nkf), "s"(dstw)
                   : Q4_MAIN_CLOBBERS);
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
      q4_epilogue<0>(p, m0, n0, w_r, w_c, lane);
      const unsigned long long t2 = __builtin_amdgcn_s_memtime();
      if (dbg && tid == 0 && blockIdx.x < 256 && tile_no < 8) {
        unsigned long long* d = dbg + ((long)blockIdx.x * 8 + tile_no) * 8;
        d[0] = t0; d[1] = ta; d[2] = tb; d[3] = t1; d[4] = t2;
      }
      ++tile_no;
    } else {
      if constexpr (VAR == 1) Q4_RUN_MAIN(Q4_MAIN_ASM_1);
      else if constexpr (VAR == 2) Q4_RUN_MAIN(Q4_MAIN_ASM_2);
      else Q4_RUN_MAIN(Q4_MAIN_ASM_0);
      if constexpr (RES) q4_store<5, false, 0>(p, m0, n0, w_r, w_c, lane);
      else if constexpr (ABL != 1) q4_epilogue<ABL>(p, m0, n0, w_r, w_c, lane);
    }
    if (!has_next) break;
    vb = nvb; m0 = m1; n0 = n1;
    if (dyn) {
      const int k2 = share(kq);
      nvb = k2 < per_x ? k2 * 8 + xcd_id : n_tiles;
    } else {
      nvb = vb + (int)gridDim.x;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (no LDS-DMA may outlive the workgroup)
  gemm_ts_exit(p);
}
}  // namespace

int launch_gemm_q4(const GemmArgs& a_in, hipStream_t stream, int n_cu) {
  const int n_tiles = (a_in.M / QBM) * (a_in.N / QBN);
  const unsigned grid = (unsigned)(n_tiles < n_cu ? n_tiles : n_cu);
  // per-XCD tile counters for launches of more than one round (TELL_Q4_DYNAMIC=0: static tile lists)
  const int dyn_env = (int)tell_opt(OPT_Q4_DYNAMIC);
  GemmArgs ad = a_in;
  ad.queue = nullptr;
  if (dyn_env && n_tiles > (int)grid && grid % 8 == 0 && n_tiles % 8 == 0) ad.queue = gemm_tile_queue_slot(8, stream);
  const GemmArgs& a = ad;
  const int var = (int)tell_opt(OPT_Q4_VAR);
  if (a.act == 5) hipLaunchKernelGGL((gemm_nt_q4_kernel<0, 0, true>), dim3(grid), dim3(256), 0, stream, a);
#ifdef TELL_PROBES      // timing probes (1, 2: wrong results; 3: stamps into aux) - tools/probes/q4_variants.py, probe build only
  else if (tell_probe(PROBE_Q4_ABL) == 3) hipLaunchKernelGGL((gemm_nt_q4_kernel<0, 3>), dim3(grid), dim3(256), 0, stream, a);
  else if (tell_probe(PROBE_Q4_ABL) == 1) hipLaunchKernelGGL((gemm_nt_q4_kernel<0, 1>), dim3(grid), dim3(256), 0, stream, a);
  else if (tell_probe(PROBE_Q4_ABL) == 2) hipLaunchKernelGGL((gemm_nt_q4_kernel<0, 2>), dim3(grid), dim3(256), 0, stream, a);
#endif
  else if (var == 1) hipLaunchKernelGGL((gemm_nt_q4_kernel<1, 0>), dim3(grid), dim3(256), 0, stream, a);
  else if (var == 2) hipLaunchKernelGGL((gemm_nt_q4_kernel<2, 0>), dim3(grid), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((gemm_nt_q4_kernel<0, 0>), dim3(grid), dim3(256), 0, stream, a);
  return tell_check_launch("gemm_nt_q4");
}
