// gemm_nt_q4e_kernel: gemm_nt_q4_kernel (csrc/gemm_q4.hip - read its header first: four waves of 128x128, hand-placed K loop)
// with the EPILOGUE INSIDE THE NEXT TILE'S K LOOP: C[M,N] = act((A . B^T + bias[n]) * alpha), bf16, whole 256x256 tiles,
// per-column bias, K >= 576 - RoBERTa's four projection GEMMs and the article K|V projection
// (tell/models/transformer_faces_objects.py:352-353; multi_head.py:488-526).
//
// Why: gemm_nt_q4_kernel's K loop runs at the MFMA floor (2080 clk per K tile, s_memtime), but each output tile then spends
// 10.6 k clk (16.5 k with GELU) in an epilogue during which the matrix cores idle and all 256 CUs store at once - a quarter
// of a K = 1024 tile's life.  Here (tools/gen_q4e_loop.py) the statement of tile T starts by DRAINING tile T-1's
// accumulators into bf16 pairs (the unavoidable VALU part, ~2.5 k clk); 8 of a lane's 32 sixteen-byte stores leave at once,
// the other 24 wait in 96 VGPRs and leave three per K tile from the first eight bodies of tile T's K loop.  The bias of a tile
// is fetched by LDS-DMA while that tile is computed and read from LDS by the drain.  The workgroup's last tile is drained
// by a final statement.  Everything between two statements that must survive lives in physical registers a[0:255]
// (accumulators) and v[32:127] (pending stores): the C++ around the statements only computes addresses.
#include "gemm_common.h"
#include "options.h"
#include "gemm_q4_loop.inc"
#include "gemm_q4e_loop.inc"

namespace {
constexpr int EBM = 256, EBN = 256, EBK = 64;
constexpr int E_SLOT = 2 * Q4_BUF;                          // two 1 KB bias slots, then the tile-queue word

__device__ __forceinline__ unsigned long long q4e_uni64(unsigned long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
template <typename T> __device__ __forceinline__ unsigned long long q4e_ptr(const T* q) {
  return q4e_uni64(reinterpret_cast<unsigned long long>(q));
}

#define Q4E_RUN_MAIN(TEXT)                                                                                              \
  asm volatile(TEXT                                                                                                     \
               :                                                                                                        \
               : "v"(xrd), "v"(wrd), "v"(xvo), "v"(wvo), "v"(coff), "v"(brd), "v"(bvo), "s"(xc), "s"(wc), "s"(xn),      \
                 "s"(wn), "s"(lda32), "s"(ldb32), "s"(nkf), "s"(dstw), "s"(cprev), "s"(bcur), "s"(ldc2), "s"(alpha_bits), \
                 "s"(flags), "s"(bdst)                                                                                  \
               : Q4E_CLOBBERS)
#define Q4E_RUN_FLUSH(TEXT)                                                                                             \
  asm volatile(TEXT : : "v"(coff), "v"(brd), "s"(cprev), "s"(ldc2), "s"(alpha_bits) : Q4E_CLOBBERS)

// PROBE >= 0: ACT 0 with the store schedule PROBES[PROBE] of tools/gen_q4e_loop.py and s_memtime stamps of wave 0 into p.aux
// (tools/probes/q4_variants.py; TELL_Q4E_VAR)
template <int ACT, int PROBE = -1>
__global__ __launch_bounds__(256) void gemm_nt_q4e_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[E_SLOT + 2048 + 64];
  gemm_ts_enter(p);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w_r = wave >> 1, w_c = wave & 1;
  const int tiles_n = p.N / EBN, tiles_m = p.M / EBM;
  const int n_tiles = tiles_m * tiles_n;
  const int nk = p.K / EBK;
  const uint16_t* A = static_cast<const uint16_t*>(p.A);
  const uint16_t* B = static_cast<const uint16_t*>(p.B);
  uint16_t* C = static_cast<uint16_t*>(p.C);

  auto tile_origin = [&](int vb, int& m0, int& n0) __attribute__((always_inline)) {
    const int nwg = n_tiles, xcd = vb & 7, q = nwg >> 3, r = nwg & 7;
    const int tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * tiles_n;
    const int g = tile_id / per_group, first_m = g * GROUP_M;
    const int gm = tiles_m - first_m < GROUP_M ? tiles_m - first_m : GROUP_M;
    const int in_g = tile_id - g * per_group;
    m0 = (first_m + in_g % gm) * EBM;
    n0 = (in_g / gm) * EBN;
  };

  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int r = lane & 31, h = lane >> 5;
  const unsigned xrd = lds0 + (16 * w_r + (r >> 1)) * Q4_PIECE + (r & 1) * 128 + h * 16;
  const int rp = ((r >> 1) & ~3) | (((r >> 1) & 1) << 1) | (((r >> 1) >> 1) & 1);      // (gemm_q4.hip: adjacent column groups per lane pair)
  const unsigned wrd = lds0 + Q4_OPER + (16 * w_c + rp) * Q4_PIECE + (r & 1) * 128 + h * 16;
  const unsigned xvo = (unsigned)((8 * wave + (lane >> 3)) * (int)p.lda * 2 + (lane & 7) * 16);
  const unsigned wvo = (unsigned)((8 * wave + (lane >> 3)) * (int)p.ldb * 2 + (lane & 7) * 16);
  const unsigned lda32 = (unsigned)p.lda * 64u, ldb32 = (unsigned)p.ldb * 64u;
  const unsigned dstw = __builtin_amdgcn_readfirstlane(lds0 + wave * Q4_PIECE);
  // store offset of the lane inside an output tile (row of X fragment 0; fragment i is 2 i rows further down), bytes
  const unsigned coff = (unsigned)(((128 * w_r + 8 * (r >> 1) + (r & 1)) * (int)p.ldc + 128 * w_c + 8 * h) * 2);
  const unsigned ldc2 = (unsigned)p.ldc * 4u;                // bytes per two output rows
  const unsigned bvo = (unsigned)tid * 4u;
  const unsigned alpha_bits = __float_as_uint(p.alpha);

  // tile order: as gemm_nt_q4_kernel (static lists, or one counter per XCD read one tile ahead)
  const bool dyn = p.queue != nullptr;
  const int xcd_id = blockIdx.x & 7, per_x = n_tiles >> 3, wg_x = (int)gridDim.x >> 3;
  int* const qx = dyn ? p.queue + xcd_id : nullptr;
  volatile int* sq = reinterpret_cast<volatile int*>(smem + E_SLOT + 2048);
  auto fetch = [&]() __attribute__((always_inline)) {
    const int k = atomicAdd(qx, 1);
    if (k == per_x + wg_x - 1) atomicExch(qx, 0);
    return k;
  };
  auto share = [&](int k) __attribute__((always_inline)) {
    if (tid == 0) *sq = k;
    __syncthreads();
    const int v = __builtin_amdgcn_readfirstlane(*sq);
    __syncthreads();
    return v;
  };
  int vb = blockIdx.x, nvb = vb + (int)gridDim.x;
  if (dyn) {
    const int k0 = share(tid == 0 ? fetch() : 0);
    vb = k0 < per_x ? k0 * 8 + xcd_id : n_tiles;
  }
  if (vb >= n_tiles) return;
  int m0, n0, mp = 0, np = 0, tile_no = 0;
  tile_origin(vb, m0, n0);
  {
    const unsigned long long x0 = q4e_ptr(A + (long)m0 * p.lda);
    const unsigned long long w0 = q4e_ptr(B + (long)n0 * p.ldb);
    asm volatile(Q4_PROLOGUE_ASM
                 :
                 : "v"(xvo), "v"(wvo), "s"(x0), "s"(w0), "s"(lda32), "s"(ldb32), "s"(dstw)
                 : Q4_PROLOGUE_CLOBBERS);
  }
  if (dyn) {
    const int k1 = share(tid == 0 ? fetch() : 0);
    nvb = k1 < per_x ? k1 * 8 + xcd_id : n_tiles;
  }
  for (;;) {
    const bool has_next = nvb < n_tiles;
    int kq = 0;
    if (dyn && has_next && tid == 0) kq = fetch();
    int m1 = m0, n1 = n0;
    if (has_next) tile_origin(nvb, m1, n1);
    const unsigned long long xc = q4e_ptr(A + (long)m0 * p.lda + 2 * EBK);
    const unsigned long long wc = q4e_ptr(B + (long)n0 * p.ldb + 2 * EBK);
    const unsigned long long xn = q4e_ptr(A + (long)m1 * p.lda);
    const unsigned long long wn = q4e_ptr(B + (long)n1 * p.ldb);
    const unsigned nkf = __builtin_amdgcn_readfirstlane((unsigned)nk | (has_next ? 0x10000u : 0u));
    const unsigned long long cprev = q4e_ptr(C + (long)mp * p.ldc + np);
    const unsigned long long bcur = q4e_ptr(p.bias + n0);
    const unsigned flags = __builtin_amdgcn_readfirstlane(tile_no > 0 ? 0x100u : 0u);
    const int slot = tile_no & 1;
    const unsigned brd = lds0 + E_SLOT + (slot ^ 1) * 1024 + (128 * w_c + 8 * h) * 4;     // the previous tile's bias
    const unsigned bdst = __builtin_amdgcn_readfirstlane(lds0 + E_SLOT + slot * 1024 + wave * 256);
    if constexpr (PROBE >= 0) {
      unsigned long long* dbg = const_cast<unsigned long long*>(static_cast<const unsigned long long*>(p.aux));
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      unsigned long long ta, td, tb;
#define Q4E_RUN_PROBE(TEXT)                                                                                             \
  asm volatile(TEXT                                                                                                     \
               : "=&s"(ta), "=&s"(td), "=&s"(tb)                                                                        \
               : "v"(xrd), "v"(wrd), "v"(xvo), "v"(wvo), "v"(coff), "v"(brd), "v"(bvo), "s"(xc), "s"(wc), "s"(xn),      \
                 "s"(wn), "s"(lda32), "s"(ldb32), "s"(nkf), "s"(dstw), "s"(cprev), "s"(bcur), "s"(ldc2), "s"(alpha_bits), \
                 "s"(flags), "s"(bdst)                                                                                  \
               : Q4E_CLOBBERS)
      Q4E_RUN_PROBE(Q4E_PROBE_ASM_V0);
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
      if (dbg && tid == 0 && blockIdx.x < 256 && tile_no < 8) {
        unsigned long long* d = dbg + ((long)blockIdx.x * 8 + tile_no) * 8;
        d[0] = t0; d[1] = ta; d[2] = td; d[3] = tb; d[4] = t1;
      }
    } else if constexpr (ACT == 1) Q4E_RUN_MAIN(Q4E_MAIN_ASM_ACT1);
    else if constexpr (ACT == 2) Q4E_RUN_MAIN(Q4E_MAIN_ASM_ACT2);
    else Q4E_RUN_MAIN(Q4E_MAIN_ASM_ACT0);
    mp = m0; np = n0; ++tile_no;
    if (!has_next) break;
    vb = nvb; m0 = m1; n0 = n1;
    if (dyn) {
      const int k2 = share(kq);
      nvb = k2 < per_x ? k2 * 8 + xcd_id : n_tiles;
    } else {
      nvb = vb + (int)gridDim.x;
    }
  }
  {
    const unsigned long long cprev = q4e_ptr(C + (long)mp * p.ldc + np);
    const unsigned brd = lds0 + E_SLOT + ((tile_no - 1) & 1) * 1024 + (128 * w_c + 8 * h) * 4;
    if constexpr (ACT == 1) Q4E_RUN_FLUSH(Q4E_FLUSH_ASM_ACT1);
    else if constexpr (ACT == 2) Q4E_RUN_FLUSH(Q4E_FLUSH_ASM_ACT2);
    else Q4E_RUN_FLUSH(Q4E_FLUSH_ASM_ACT0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  gemm_ts_exit(p);
}
}  // namespace

// -> TELL_OK after launching, or 1 when the call is not one this kernel takes (the caller then runs gemm_nt_q4_kernel)
int launch_gemm_q4e(const GemmArgs& a_in, hipStream_t stream, int n_cu) {
  if (a_in.bias_mode != 1 || a_in.act > 2 || a_in.K / EBK < Q4E_NSB + 1 || a_in.K / EBK < 13 || (long)a_in.ldc * 2 * 256 >= (1L << 31)) return 1;
  const int n_tiles = (a_in.M / EBM) * (a_in.N / EBN);
  const unsigned grid = (unsigned)(n_tiles < n_cu ? n_tiles : n_cu);
  const int dyn_env = (int)tell_opt(OPT_Q4_DYNAMIC);
  GemmArgs a = a_in;
  a.queue = nullptr;
  if (dyn_env && n_tiles > (int)grid && grid % 8 == 0 && n_tiles % 8 == 0) a.queue = gemm_tile_queue_slot(8, stream);
#ifdef TELL_PROBES
  if (tell_probe(PROBE_Q4E_VAR) >= 0 && a.act == 0 && a.K / EBK >= 13) {
    hipLaunchKernelGGL((gemm_nt_q4e_kernel<0, 0>), dim3(grid), dim3(256), 0, stream, a);      // the stamped statement
  } else
#endif
  if (a.act == 2) hipLaunchKernelGGL((gemm_nt_q4e_kernel<2>), dim3(grid), dim3(256), 0, stream, a);
  else if (a.act == 1) hipLaunchKernelGGL((gemm_nt_q4e_kernel<1>), dim3(grid), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((gemm_nt_q4e_kernel<0>), dim3(grid), dim3(256), 0, stream, a);
  return tell_check_launch("gemm_nt_q4e");
}
