// gemm_nt_q4_kernel: C[M,N] = act((A[M,K] . B[N,K]^T + bias) * alpha), bf16 in / out, whole 256x256 tiles - the RoBERTa
// projection GEMMs at B x 512 rows and the article K|V projection (fairseq TransformerSentenceEncoderLayer; the call
// site of the reference is tell/models/transformer_faces_objects.py:352-353).
//
// Round 4's answer to "the two LDS consumers of the ping-pong kernel collide" (DESIGN 3): FOUR waves own 128x128 of the
// tile each (16 v_mfma_f32_32x32x16_bf16 accumulators = 256 AGPRs per lane, one wave per SIMD), which cuts the fragment
// reads from 192 KB to 128 KB per K tile, and the whole K loop is ONE hand-placed instruction stream
// (gemm_q4_loop.inc, written by tools/gen_q4_loop.py - read its header for the schedule): one LDS read or one
// LDS-DMA instruction per MFMA gap, four barriers per K tile, the DMA stream two K tiles ahead in two LDS buffers and
// running on across output-tile boundaries (resident workgroups as in gemm_pp2.hip: the first two K tiles of the next
// output tile land while the epilogue runs).  The epilogue stores straight from registers: with the row mapping of the
// LDS image a lane owns 8 consecutive output columns per 16-byte store, and a store instruction costs its ~70 clk per CU
// whether its lanes cover whole lines or not - no LDS staging, no barriers.
#include "gemm_common.h"
#include "gemm_q4_loop.inc"

namespace {
constexpr int QBM = 256, QBN = 256, QBK = 64;

template <int ACT>
__device__ __forceinline__ void q4_store(f32x16 (&acc)[4][4], const GemmArgs& p, int m0, int n0, int w_r, int w_c, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const int nb = n0 + 128 * w_c;
  f32x4_t b4[8][2];
  float bm[4];
#pragma unroll
  for (int q = 0; q < 8; ++q) { b4[q][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; b4[q][1] = b4[q][0]; }
#pragma unroll
  for (int i = 0; i < 4; ++i) bm[i] = 0.f;
  if (p.bias_mode == 1) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int G = (q & 1) + 4 * (q >> 1) + 2 * h;
      b4[q][0] = *reinterpret_cast<const f32x4_t*>(p.bias + nb + 8 * G);
      b4[q][1] = *reinterpret_cast<const f32x4_t*>(p.bias + nb + 8 * G + 4);
    }
  } else if (p.bias_mode == 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) bm[i] = p.bias[m0 + 128 * w_r + 8 * (r >> 1) + 2 * i + (r & 1)];
  }
  uint16_t* C = static_cast<uint16_t*>(p.C);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + 128 * w_r + 8 * (r >> 1) + 2 * i + (r & 1);
    uint16_t* crow = C + (long)m * p.ldc + nb;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int G = (q & 1) + 4 * (q >> 1) + 2 * h;
      u32x4 o;
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {                   // columns 8 G + 4 jp .. + 3: (j = 2 jp, 2 jp + 1) x (e & 1)
        f32x4_t v = {acc[i][2 * jp][2 * q], acc[i][2 * jp][2 * q + 1], acc[i][2 * jp + 1][2 * q], acc[i][2 * jp + 1][2 * q + 1]};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (v[e] + b4[q][jp][e] + bm[i]) * p.alpha;
        epi_act4<ACT>(v);
        o[2 * jp] = pack2_bf16(v[0], v[1]);
        o[2 * jp + 1] = pack2_bf16(v[2], v[3]);
      }
      *reinterpret_cast<u32x4*>(crow + 8 * G) = o;
    }
  }
}

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
  const unsigned wrd = lds0 + Q4_OPER + (16 * w_c + (r >> 1)) * Q4_PIECE + (r & 1) * 128 + h * 16;
  const unsigned xvo = (unsigned)((8 * wave + (lane >> 3)) * (int)p.lda * 2 + (lane & 7) * 16);
  const unsigned wvo = (unsigned)((8 * wave + (lane >> 3)) * (int)p.ldb * 2 + (lane & 7) * 16);
  const unsigned lda32 = (unsigned)p.lda * 64u, ldb32 = (unsigned)p.ldb * 64u;     // bytes per 32 rows
  const unsigned dstw = __builtin_amdgcn_readfirstlane(lds0 + wave * Q4_PIECE);

  int vb = blockIdx.x;
  if (vb >= n_tiles) return;
  int m0, n0;
  tile_origin(vb, m0, n0);
  {
    const unsigned long long x0 = reinterpret_cast<unsigned long long>(A + (long)m0 * p.lda);
    const unsigned long long w0 = reinterpret_cast<unsigned long long>(B + (long)n0 * p.ldb);
    asm volatile(Q4_PROLOGUE_ASM
                 :
                 : "v"(xvo), "v"(wvo), "s"(x0), "s"(w0), "s"(lda32), "s"(ldb32), "s"(dstw)
                 : Q4_PROLOGUE_CLOBBERS);
  }
  for (;;) {
    const int nvb = vb + (int)gridDim.x;
    const bool has_next = nvb < n_tiles;
    int m1 = m0, n1 = n0;
    if (has_next) tile_origin(nvb, m1, n1);
    const unsigned long long xc = reinterpret_cast<unsigned long long>(A + (long)m0 * p.lda + 2 * QBK);
    const unsigned long long wc = reinterpret_cast<unsigned long long>(B + (long)n0 * p.ldb + 2 * QBK);
    const unsigned long long xn = reinterpret_cast<unsigned long long>(A + (long)m1 * p.lda);
    const unsigned long long wn = reinterpret_cast<unsigned long long>(B + (long)n1 * p.ldb);
    const unsigned nkf = (unsigned)nk | (has_next ? 0x10000u : 0u);
    f32x16 acc[4][4];
    asm volatile(Q4_MAIN_ASM
                 : "=&a"(acc[0][0]), "=&a"(acc[0][1]), "=&a"(acc[0][2]), "=&a"(acc[0][3]),
                   "=&a"(acc[1][0]), "=&a"(acc[1][1]), "=&a"(acc[1][2]), "=&a"(acc[1][3]),
                   "=&a"(acc[2][0]), "=&a"(acc[2][1]), "=&a"(acc[2][2]), "=&a"(acc[2][3]),
                   "=&a"(acc[3][0]), "=&a"(acc[3][1]), "=&a"(acc[3][2]), "=&a"(acc[3][3])
                 : "v"(xrd), "v"(wrd), "v"(xvo), "v"(wvo), "s"(xc), "s"(wc), "s"(xn), "s"(wn), "s"(lda32), "s"(ldb32),
                   "s"(nkf), "s"(dstw)
                 : Q4_MAIN_CLOBBERS);
    switch (p.act) {                                       // block-uniform
      case 1: q4_store<1>(acc, p, m0, n0, w_r, w_c, lane); break;
      case 2: q4_store<2>(acc, p, m0, n0, w_r, w_c, lane); break;
      default: q4_store<0>(acc, p, m0, n0, w_r, w_c, lane); break;
    }
    if (!has_next) break;
    vb = nvb; m0 = m1; n0 = n1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (no LDS-DMA may outlive the workgroup)
  gemm_ts_exit(p);
}
}  // namespace

int launch_gemm_q4(const GemmArgs& a, hipStream_t stream, int n_cu) {
  const int n_tiles = (a.M / QBM) * (a.N / QBN);
  const unsigned grid = (unsigned)(n_tiles < n_cu ? n_tiles : n_cu);
  hipLaunchKernelGGL(gemm_nt_q4_kernel, dim3(grid), dim3(256), 0, stream, a);
  return tell_check_launch("gemm_nt_q4");
}
