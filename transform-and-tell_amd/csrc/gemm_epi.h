// Epilogues of the MFMA GEMM kernels (32x32-MFMA accumulator layout): activation / bias / residual stores, the staged
// (through LDS) bf16 / fp32 tile stores of the direct-to-LDS kernels and the BatchNorm-statistics epilogue of the implicit
// convolutions.  Shared by gemm.hip and gemm_s64.hip.
#pragma once
#include "common.h"
#include "gemm_common.h"
#include <type_traits>

template <typename T> struct Mma;
template <> struct Mma<uint16_t> {
  static constexpr int BK = 64, KSTEP = 16, STRIDE = 72;  // elements
  using frag = bf16x8;
  __device__ static __forceinline__ frag load(const uint16_t* tile, int row, int k0, int lane) {
    return *reinterpret_cast<const frag*>(tile + row * STRIDE + k0 + ((lane >> 5) << 3));
  }
  __device__ static __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  __device__ static __forceinline__ void store_chunk(uint16_t* tile, int row, int ch, const u32x4& v) {
    *reinterpret_cast<u32x4*>(tile + row * STRIDE + ch * 8) = v;
  }
};
template <> struct Mma<float> {
  static constexpr int BK = 32, KSTEP = 2, STRIDE = 33;
  using frag = float;
  __device__ static __forceinline__ frag load(const float* tile, int row, int k0, int lane) {
    return tile[row * STRIDE + k0 + (lane >> 5)];
  }
  __device__ static __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
  __device__ static __forceinline__ void store_chunk(float* tile, int row, int ch, const u32x4& v) {
    float* p = tile + row * STRIDE + ch * 4;
    p[0] = __uint_as_float(v[0]); p[1] = __uint_as_float(v[1]);
    p[2] = __uint_as_float(v[2]); p[3] = __uint_as_float(v[3]);
  }
};

// compile-time loop: the register stage index must be a constant or the stages land in scratch
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}


// ------------------------------------------------------------- epilogue (shared by all GEMM kernels)
// Every kernel issues its MFMAs with the operands swapped (B fragment first), so the accumulator tile is
// C^T: lane l owns ONE output row and 4 consecutive output columns per register quad,
//   acc[i][j][4g+e] = C[mw + i*32 + (l&31)][nw + j*32 + 8g + 4*(l>>5) + e].
// That makes the stores 8-byte (bf16) / 16-byte (fp32) vectors instead of 2-byte column scatters, and the
// bf16 conversion a packed v_cvt_pk_bf16_f32.  The activation is a template parameter (one block-uniform
// switch per tile instead of branches per element).


template <typename OutT, int MI, int NI, int ACT>
__device__ __forceinline__ void gemm_epilogue_act(f32x16 (&acc)[MI][NI], const GemmArgs& p, int mw, int nw, int lane,
                                                  int M, int N) {
  OutT* C = static_cast<OutT*>(p.C);
  const OutT* aux = static_cast<const OutT*>(p.aux);
  constexpr uintptr_t AL = 4 * sizeof(OutT) - 1;
  const bool vec_ok = (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & AL) == 0 &&
                      ((ACT != 3 && ACT != 4) || (reinterpret_cast<uintptr_t>(aux) & AL) == 0);
  f32x4_t bn[NI][4];                                               // per-column bias of this lane's 4-column groups
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = nw + j * 32 + 8 * g + 4 * (lane >> 5);
#pragma unroll
      for (int e = 0; e < 4; ++e) bn[j][g][e] = (p.bias_mode == 1 && n + e < N) ? p.bias[n + e] : 0.f;
    }
  // Everything an output row needs from memory - its bias, the previous contents of C (accumulate), the mask / residual
  // pieces (act 3 / 4) - is fetched as ONE batch per 32-row block before any of it is used.  Fetched inside the column
  // loop (under the block-uniform accumulate test) every 4-column group waited for its own round trip: 8-16 dependent
  // HBM / L2 latencies per row block, which for the 16-K-step weight-gradient and residual-accumulating input-gradient
  // GEMMs of the decoder was longer than their main loop.
  float bmr[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = mw + i * 32 + (lane & 31);
    bmr[i] = (p.bias_mode == 2 && m < M) ? p.bias[m] : 0.f;
  }
  bool rmw = p.accumulate != 0;
  if constexpr (std::is_same<OutT, float>::value) rmw = rmw && !p.atomic_out;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = mw + i * 32 + (lane & 31);
    if (m >= M) continue;
    const float bm = bmr[i];
    f32x4_t prev[NI][4], ax[NI][4];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        prev[j][g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        ax[j][g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    if (vec_ok && rmw) {
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = nw + j * 32 + 8 * g + 4 * (lane >> 5);
          if (n + 3 < N) prev[j][g] = Vec4<OutT>::ld(C + (long)m * p.ldc + n);
        }
    }
    if constexpr (ACT == 3 || ACT == 4) {
      if (vec_ok) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n = nw + j * 32 + 8 * g + 4 * (lane >> 5);
            if (n + 3 < N) ax[j][g] = Vec4<OutT>::ld(aux + (long)m * p.ldc + n);
          }
      }
    }
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nw + j * 32 + 8 * g + 4 * (lane >> 5);
        if (n >= N) continue;
        f32x4_t v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (acc[i][j][4 * g + e] + bn[j][g][e] + bm) * p.alpha;
        epi_act4<(ACT == 3 || ACT == 4) ? 0 : ACT>(v);
        OutT* dst = C + (long)m * p.ldc + n;
        if (vec_ok && n + 3 < N) {
          if constexpr (ACT == 3) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ax[j][g][e] > 0.f ? v[e] : 0.f;
          }
          if constexpr (ACT == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] + ax[j][g][e], 0.f);
          }
          if constexpr (std::is_same<OutT, float>::value) {
            if (p.atomic_out) {                                     // block-uniform
#pragma unroll
              for (int e = 0; e < 4; ++e) unsafeAtomicAdd(reinterpret_cast<float*>(dst) + e, v[e]);
              continue;
            }
          }
          if (p.accumulate) v += prev[j][g];
          Vec4<OutT>::st(dst, v);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < N) {
              float x = v[e];
              if constexpr (ACT == 3) x = Elem<OutT>::ld(aux + (long)m * p.ldc + n + e) > 0.f ? x : 0.f;
              if constexpr (ACT == 4) x = fmaxf(x + Elem<OutT>::ld(aux + (long)m * p.ldc + n + e), 0.f);
              if constexpr (std::is_same<OutT, float>::value) {
                if (p.atomic_out) { unsafeAtomicAdd(reinterpret_cast<float*>(dst) + e, x); continue; }
              }
              if (p.accumulate) x += Elem<OutT>::ld(dst + e);
              Elem<OutT>::st(dst + e, x);
            }
        }
      }
  }
}
template <typename OutT, int MI, int NI>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[MI][NI], const GemmArgs& p, int mw, int nw, int lane,
                                              int M, int N) {
  switch (p.act) {                                                 // block-uniform
    case 1: gemm_epilogue_act<OutT, MI, NI, 1>(acc, p, mw, nw, lane, M, N); break;
    case 2: gemm_epilogue_act<OutT, MI, NI, 2>(acc, p, mw, nw, lane, M, N); break;
    case 3: gemm_epilogue_act<OutT, MI, NI, 3>(acc, p, mw, nw, lane, M, N); break;
    case 4: gemm_epilogue_act<OutT, MI, NI, 4>(acc, p, mw, nw, lane, M, N); break;
    default: gemm_epilogue_act<OutT, MI, NI, 0>(acc, p, mw, nw, lane, M, N); break;
  }
}

// Full interior bf16 tile of a direct-to-LDS kernel: staged through free LDS (`cs`, BM rows of CS elements;
// CS == BN means unpadded with the 16-byte chunk index XOR-swizzled by the row), so every output row
// leaves as whole 128-byte lines.  Needs a block barrier BEFORE (cs no longer read as a tile) by the caller.
// LAY 0: a wave owns one contiguous WM x WN block.  LAY 1 (ping-pong kernel): a wave owns 64 rows in each half of
// the tile's rows and 32 columns in each half of its columns (block i: half i>>1, 32-row group i&1; block j: half j).
template <int BM, int BN, int WM, int WN, int MI, int NI, int CS, int NT, int ACT, int LAY = 0>
__device__ __forceinline__ void glds_store_tile_act(f32x16 (&acc)[MI][NI], const GemmArgs& p, int m0, int n0, int wm,
                                                    int wn, int lane, int tid, uint16_t* cs) {
  constexpr bool SWZ = CS == BN;
  constexpr int CPRW = BN / 8;                                     // 16-byte chunks per tile row
  // Bias pieces of this lane's column quads / rows: ONE batch of loads in front of the block loops.  Loaded where they
  // are used, under the block-uniform mode test, every (i, j, g) block waited for its own round trip (s_waitcnt vmcnt(0)
  // x MI*NI*4 in the ISA: 32 dependent L2 round trips in the epilogue of a 128x64 wave tile).
  f32x4_t b4[NI][4];
  float bmr[MI];
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) b4[j][g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < MI; ++i) bmr[i] = 0.f;
  if (p.bias_mode == 1) {
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        b4[j][g] = *reinterpret_cast<const f32x4_t*>(p.bias + n0 + (LAY ? j * (BN / 2) + wn * 32 : wn * WN + j * 32) + 8 * g + 4 * (lane >> 5));
  } else if (p.bias_mode == 2) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
      bmr[i] = p.bias[m0 + (LAY ? (i >> 1) * (BM / 2) + wm * 64 + (i & 1) * 32 + (lane & 31) : wm * WM + i * 32 + (lane & 31))];
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int row = LAY ? (i >> 1) * (BM / 2) + wm * 64 + (i & 1) * 32 + (lane & 31) : wm * WM + i * 32 + (lane & 31);
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = (LAY ? j * (BN / 2) + wn * 32 : wn * WN + j * 32) + 8 * g + 4 * (lane >> 5);
        f32x4_t v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (acc[i][j][4 * g + e] + b4[j][g][e] + bmr[i]) * p.alpha;
        epi_act4<(ACT == 4 || ACT == 3) ? 0 : ACT>(v);
        int ch = col >> 3;
        if constexpr (SWZ) ch ^= row & (CPRW - 1) & 15;
        u32x2 w = {pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3])};
        *reinterpret_cast<u32x2*>(cs + row * CS + ch * 8 + (col & 7)) = w;
      }
  }
  __syncthreads();
  uint16_t* C = static_cast<uint16_t*>(p.C);
#pragma unroll
  for (int i = 0; i < BM * CPRW / NT; ++i) {
    const int c = tid + i * NT, row = c / CPRW;
    int ch = c % CPRW;
    const int sch = SWZ ? ch ^ (row & (CPRW - 1) & 15) : ch;
    u32x4 o = *reinterpret_cast<const u32x4*>(cs + row * CS + sch * 8);
    if constexpr (ACT == 3) {                          // relu backward: keep where the forward output (aux) was positive
      const u32x4 r = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(p.aux) + (long)(m0 + row) * p.ldc + n0 + ch * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {                    // bf16 > 0: sign clear and not zero
        const uint32_t lo = ((r[e] & 0x8000u) == 0 && (r[e] & 0x7fffu) != 0) ? 0x0000ffffu : 0u;
        const uint32_t hi = ((r[e] & 0x80000000u) == 0 && (r[e] & 0x7fff0000u) != 0) ? 0xffff0000u : 0u;
        o[e] &= lo | hi;
      }
    }
    if constexpr (ACT == 4) {                          // relu(tile + residual): the residual arrives as whole 16-byte pieces too
      const u32x4 r = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(p.aux) + (long)(m0 + row) * p.ldc + n0 + ch * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = fmaxf(__uint_as_float(o[e] << 16) + __uint_as_float(r[e] << 16), 0.f);
        const float hi = fmaxf(__uint_as_float(o[e] & 0xffff0000u) + __uint_as_float(r[e] & 0xffff0000u), 0.f);
        o[e] = pack2_bf16(lo, hi);
      }
    }
    *reinterpret_cast<u32x4*>(C + (long)(m0 + row) * p.ldc + n0 + ch * 8) = o;
  }
}
template <int BM, int BN, int WM, int WN, int MI, int NI, int CS, int NT, int LAY = 0>
__device__ __forceinline__ void glds_store_tile(f32x16 (&acc)[MI][NI], const GemmArgs& p, int m0, int n0, int wm,
                                                int wn, int lane, int tid, uint16_t* cs) {
  switch (p.act) {
    case 1: glds_store_tile_act<BM, BN, WM, WN, MI, NI, CS, NT, 1, LAY>(acc, p, m0, n0, wm, wn, lane, tid, cs); break;
    case 2: glds_store_tile_act<BM, BN, WM, WN, MI, NI, CS, NT, 2, LAY>(acc, p, m0, n0, wm, wn, lane, tid, cs); break;
    case 3: glds_store_tile_act<BM, BN, WM, WN, MI, NI, CS, NT, 3, LAY>(acc, p, m0, n0, wm, wn, lane, tid, cs); break;
    case 4: glds_store_tile_act<BM, BN, WM, WN, MI, NI, CS, NT, 4, LAY>(acc, p, m0, n0, wm, wn, lane, tid, cs); break;
    default: glds_store_tile_act<BM, BN, WM, WN, MI, NI, CS, NT, 0, LAY>(acc, p, m0, n0, wm, wn, lane, tid, cs); break;
  }
}
// fp32 outputs (adaptive-softmax logits: [rows, 30265] from K = 64 ... 1024): the tile leaves through LDS as whole
// 16-byte row pieces as well.  Straight from the accumulators a wave-store touches 32 rows x 32 bytes - the 124 MB of
// tail logits took 141 us (0.9 TB/s).  LDS: BM rows of BN floats, 16-byte chunks XORed with the row (32 lanes of a
// store hit 32 rows at the same column).
template <int BM, int BN, int WM, int WN, int MI, int NI, int NT, int ACT>
__device__ __forceinline__ void glds_store_tile_f32_act(f32x16 (&acc)[MI][NI], const GemmArgs& p, int m0, int n0, int wm,
                                                        int wn, int lane, int tid, float* cs) {
  constexpr int CPRW = BN / 4;                                     // 16-byte chunks per tile row
  static_assert((CPRW & (CPRW - 1)) == 0, "power-of-two chunks per row");
  f32x4_t b4[NI][4];                                               // one batch of bias loads (see the bf16 form above)
  float bmr[MI];
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) b4[j][g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < MI; ++i) bmr[i] = 0.f;
  if (p.bias_mode == 1) {
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        b4[j][g] = *reinterpret_cast<const f32x4_t*>(p.bias + n0 + wn * WN + j * 32 + 8 * g + 4 * (lane >> 5));
  } else if (p.bias_mode == 2) {
#pragma unroll
    for (int i = 0; i < MI; ++i) bmr[i] = p.bias[m0 + wm * WM + i * 32 + (lane & 31)];
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int row = wm * WM + i * 32 + (lane & 31);
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = wn * WN + j * 32 + 8 * g + 4 * (lane >> 5);
        f32x4_t v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (acc[i][j][4 * g + e] + b4[j][g][e] + bmr[i]) * p.alpha;
        epi_act4<ACT>(v);
        const int ch = (col >> 2) ^ (row & (CPRW - 1));
        *reinterpret_cast<f32x4_t*>(cs + row * BN + ch * 4) = v;
      }
  }
  __syncthreads();
  float* C = static_cast<float*>(p.C);
#pragma unroll
  for (int i = 0; i < BM * CPRW / NT; ++i) {
    const int c = tid + i * NT, row = c / CPRW, ch = c % CPRW;
    *reinterpret_cast<f32x4_t*>(C + (long)(m0 + row) * p.ldc + n0 + ch * 4) =
        *reinterpret_cast<const f32x4_t*>(cs + row * BN + ((ch ^ (row & (CPRW - 1))) << 2));
  }
}
template <int BM, int BN, int WM, int WN, int MI, int NI, int NT>
__device__ __forceinline__ void glds_store_tile_f32(f32x16 (&acc)[MI][NI], const GemmArgs& p, int m0, int n0, int wm,
                                                    int wn, int lane, int tid, float* cs) {
  switch (p.act) {
    case 1: glds_store_tile_f32_act<BM, BN, WM, WN, MI, NI, NT, 1>(acc, p, m0, n0, wm, wn, lane, tid, cs); break;
    case 2: glds_store_tile_f32_act<BM, BN, WM, WN, MI, NI, NT, 2>(acc, p, m0, n0, wm, wn, lane, tid, cs); break;
    default: glds_store_tile_f32_act<BM, BN, WM, WN, MI, NI, NT, 0>(acc, p, m0, n0, wm, wn, lane, tid, cs); break;
  }
}
__device__ __forceinline__ bool glds_fast_tile_f32(const GemmArgs& p, int m0, int n0, int BM, int BN, int M, int N) {
  return !p.accumulate && !p.atomic_out && p.act != 3 && p.act != 4 && m0 + BM <= M && n0 + BN <= N && (p.ldc & 3) == 0 &&
         (reinterpret_cast<uintptr_t>(p.C) & 15) == 0 &&
         (p.bias_mode != 1 || ((reinterpret_cast<uintptr_t>(p.bias) & 15) == 0 && (n0 & 3) == 0));
}
__device__ __forceinline__ bool glds_fast_tile(const GemmArgs& p, int m0, int n0, int BM, int BN, int M, int N) {
  return !p.accumulate && m0 + BM <= M && n0 + BN <= N && (p.ldc & 7) == 0 &&
         (reinterpret_cast<uintptr_t>(p.C) & 15) == 0 && ((p.act != 4 && p.act != 3) || (reinterpret_cast<uintptr_t>(p.aux) & 15) == 0) &&
         (p.bias_mode != 1 || ((reinterpret_cast<uintptr_t>(p.bias) & 15) == 0 && (n0 & 3) == 0));
}

// Epilogue of a convolution GEMM that feeds a train-mode BatchNorm: the bf16 tile is staged in LDS (cs: BM rows
// of BN+8 elements, then NT floats of scratch), leaves as 16-byte row chunks, and every column's mean / M2 over
// the tile's valid rows is taken from the staged (already rounded) values - the statistics pass over the
// activation in HBM disappears.  Handles ragged last tiles (rows >= M, columns >= N are skipped).
template <int BM, int BN, int WM, int WN, int MI, int NI, int NT>
__device__ __forceinline__ void staged_store_stats(f32x16 (&acc)[MI][NI], const GemmArgs& p, int m0, int n0, int tm,
                                                   int wm, int wn, int lane, int tid, uint16_t* cs, int M, int N) {
  constexpr int CS = BN + 8, CPRW = BN / 8, G = NT / BN;
  static_assert(NT % BN == 0, "row groups");
  float* red = reinterpret_cast<float*>(cs + BM * CS);
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int row = wm * WM + i * 32 + (lane & 31);
    const float bm = (p.bias_mode == 2 && m0 + row < M) ? p.bias[m0 + row] : 0.f;
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = wn * WN + j * 32 + 8 * g + 4 * (lane >> 5);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float b = (p.bias_mode == 1 && n0 + col + e < N) ? p.bias[n0 + col + e] : 0.f;
          v[e] = (acc[i][j][4 * g + e] + b + bm) * p.alpha;
        }
        u32x2 w = {pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3])};
        *reinterpret_cast<u32x2*>(cs + row * CS + col) = w;
      }
  }
  __syncthreads();
  uint16_t* C = static_cast<uint16_t*>(p.C);
#pragma unroll
  for (int i = 0; i < BM * CPRW / NT; ++i) {
    const int c = tid + i * NT, row = c / CPRW, ch = c % CPRW;
    if (m0 + row < M && n0 + ch * 8 < N)
      *reinterpret_cast<u32x4*>(C + (long)(m0 + row) * p.ldc + n0 + ch * 8) =
          *reinterpret_cast<const u32x4*>(cs + row * CS + ch * 8);
  }
  const int c = tid % BN, rg = tid / BN;
  const int rows = M - m0 < BM ? M - m0 : BM;
  float s = 0.f;
  for (int r = rg; r < rows; r += G) s += __uint_as_float((unsigned)cs[r * CS + c] << 16);
  red[rg * BN + c] = s;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int g = 0; g < G; ++g) mean += red[g * BN + c];
  mean /= (float)rows;
  __syncthreads();
  float q = 0.f;
  for (int r = rg; r < rows; r += G) {
    const float d = __uint_as_float((unsigned)cs[r * CS + c] << 16) - mean;
    q += d * d;
  }
  red[rg * BN + c] = q;
  __syncthreads();
  if (rg == 0 && n0 + c < N) {
    float m2 = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) m2 += red[g * BN + c];
    p.stat_mean[(long)tm * N + n0 + c] = mean;
    p.stat_m2[(long)tm * N + n0 + c] = m2;
  }
}

// The epilogue of the direct-to-LDS kernels (gemm_nt_glds_body, gemm_s64.hip): acc in the 32x32-MFMA layout of wave
// (wm, wn), smem = the workgroup's stage buffers (free after the K loop; the caller has synchronised).
template <typename OutT, int BM, int BN, int WAVES_M, int WAVES_N, bool CONV, int NS>
__device__ __forceinline__ void gemm_nt_glds_epilogue(f32x16 (&acc)[BM / WAVES_M / 32][BN / WAVES_N / 32], const GemmArgs& p,
                                                      unsigned char* smem, int m0, int n0, int tm, int wm, int wn, int lane,
                                                      int tid, int M, int N) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MI = WM / 32, NI = WN / 32;
  // ---- epilogue.  Full interior bf16 tiles go through LDS (the tile buffers are free now).
  if constexpr (sizeof(OutT) == 2 && BM * (BN + 8) * 2 + 64 * NW * 4 <= 2 * STAGE && (64 * NW) % BN == 0) {
    if (p.stat_mean) {                                   // block-uniform: conv + BatchNorm statistics
      staged_store_stats<BM, BN, WM, WN, MI, NI, 64 * NW>(acc, p, m0, n0, tm, wm, wn, lane, tid,
                                                          reinterpret_cast<uint16_t*>(smem), M, N);
      return;
    }
  }
  if constexpr (sizeof(OutT) == 2) {
    constexpr int CS = (BM * (BN + 8) * 2 <= 2 * STAGE) ? BN + 8 : BN;   // padded row (elements) when it fits
    static_assert(BM * CS * 2 <= 2 * STAGE, "output tile must fit the freed tile buffers");
    if (glds_fast_tile(p, m0, n0, BM, BN, M, N)) {       // block-uniform
      glds_store_tile<BM, BN, WM, WN, MI, NI, CS, 64 * NW>(acc, p, m0, n0, wm, wn, lane, tid,
                                                           reinterpret_cast<uint16_t*>(smem));
      gemm_ts_exit(p);
      return;
    }
  }
  if constexpr (sizeof(OutT) == 4 && BM * BN * 4 <= 2 * STAGE && !CONV) {
    if (!p.stat_mean && glds_fast_tile_f32(p, m0, n0, BM, BN, M, N)) {       // block-uniform
      __syncthreads();                                                       // every wave is done reading the stages
      glds_store_tile_f32<BM, BN, WM, WN, MI, NI, 64 * NW>(acc, p, m0, n0, wm, wn, lane, tid,
                                                           reinterpret_cast<float*>(smem));
      gemm_ts_exit(p);
      return;
    }
  }
  gemm_epilogue<OutT, MI, NI>(acc, p, m0 + wm * WM, n0 + wn * WN, lane, M, N);
  gemm_ts_exit(p);
}
