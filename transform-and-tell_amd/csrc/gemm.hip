// NT GEMM on the CDNA4 matrix cores:  C[M,N] = epilogue(A[M,K] . B[N,K]^T)
//
// Both operands are row-major with the contraction index contiguous, which is
// what every linear layer of the decoder / RoBERTa / ResNet(1x1, im2col) needs
// (x.W^T).  Backward products (dY.W, dY^T.X) are brought into the same form by
// the transpose kernel in elementwise.hip.
//
//   bf16 : v_mfma_f32_32x32x16_bf16  (lane l supplies row l&31, k-chunk 8*(l>>5)..+8)
//   f32  : v_mfma_f32_32x32x2_f32    (exact-f32 parity mode; lane l: row l&31, k = l>>5)
// C/D layout of both:  col = l&31 (B row = n),  row = (r&3) + 8*(r>>2) + 4*(l>>5) (A row = m).
//
// Tiling: workgroup = 256 threads = 4 waves in 2x2; tile BMxBN in {128x128, 64x64};
// K-step = 128 bytes of K per row (64 bf16 / 32 f32); register-prefetched,
// double-buffered LDS; LDS rows padded (bf16: 144 B stride -> conflict-free
// ds_read_b128 per 16-lane service group; f32: 33-dword stride).
#include "common.h"
#include <type_traits>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // 16-byte staging chunk (native vector: stays in VGPRs)

struct GemmArgs {
  const void* A; const void* B; void* C;
  const float* bias;      // fp32, length N (mode 1) or M (mode 2)
  const void* aux;        // relu-mask source (same layout/dtype as C) for act==3
  const int* m_dev;       // optional device-side effective M (rows >= *m_dev are skipped)
  long lda, ldb, ldc;
  int M, N, K;
  int bias_mode;          // 0 none, 1 per column n, 2 per row m
  int act;                // 0 none, 1 relu, 2 gelu(erf), 3 multiply by (aux > 0)
  int accumulate;         // C = C + result  (beta = 1)
  float alpha;            // result = act((acc + bias) * alpha)
};

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

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

// ------------------------------------------------------------- epilogue (shared by both GEMM kernels)
// acc[i][j][r] holds C[mw + i*32 + (r&3) + 8*(r>>2) + 4*(lane>>5)][nw + j*32 + (lane&31)]
template <typename OutT, int MI, int NI>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[MI][NI], const GemmArgs& p, int mw, int nw, int lane,
                                              int M, int N) {
  OutT* C = static_cast<OutT*>(p.C);
  const OutT* aux = static_cast<const OutT*>(p.aux);
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = nw + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M && n < N) {
          float v = acc[i][j][r];
          if (p.bias_mode == 1) v += p.bias[n];
          else if (p.bias_mode == 2) v += p.bias[m];
          v *= p.alpha;
          if (p.act == 1) v = fmaxf(v, 0.f);
          else if (p.act == 2) v = gelu_erf(v);
          else if (p.act == 3) v = Elem<OutT>::ld(aux + (long)m * p.ldc + n) > 0.f ? v : 0.f;
          OutT* dst = C + (long)m * p.ldc + n;
          if (p.accumulate) v += Elem<OutT>::ld(dst);
          Elem<OutT>::st(dst, v);
        }
      }
    }
  }
}

// ------------------------------------------------------------- direct-to-LDS kernel (bf16, K % 64 == 0)
// global_load_lds_dwordx4: every lane's 16 bytes go straight from L2/HBM into LDS (no VGPR staging, no
// ds_write), the next K tile streams into the other LDS buffer while the MFMAs run on the current one.
// The LDS image of a tile must be lane-linear (wave-uniform base + lane*16), so the bank-conflict
// swizzle is applied on the SOURCE side: slot s (16 B) of the image holds row 2p + (l>>3), chunk l&7 with
// p = s>>4 and l = (s&15) ^ (p&15); fragment reads apply the same involution.  A 16-lane service group
// of ds_read_b128 then touches 16 distinct 16-byte slots of the 256-byte bank row (conflict-free).
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <typename OutT, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_nt_glds_kernel(GemmArgs p) {
  constexpr int NW = WAVES_M * WAVES_N, BK = 64;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int IA = BM / 8 / NW, IB = BN / 8 / NW;   // wave-instructions (1 KiB each) per wave per tile
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MI = WM / 32, NI = WN / 32;
  static_assert(IA >= 1 && IB >= 1, "tile too small for the wave count");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  int M = p.M;
  if (p.m_dev) { int md = *p.m_dev; M = md < M ? md : M; }
  const int N = p.N, K = p.K;
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
  if (tile_id >= ((M + BM - 1) / BM) * tiles_n) return;

  const uint16_t* A = static_cast<const uint16_t*>(p.A);
  const uint16_t* B = static_cast<const uint16_t*>(p.B);
  // per-lane source pointers (row clamped: rows past M/N only feed outputs that are never stored)
  const uint16_t* asrc[IA];
  const uint16_t* bsrc[IB];
#pragma unroll
  for (int j = 0; j < IA; ++j) {
    const int s = (wave * IA + j) * 64 + lane, pr = s >> 4, l16 = (s & 15) ^ (pr & 15);
    int row = m0 + 2 * pr + (l16 >> 3);
    row = row < M ? row : M - 1;
    asrc[j] = A + (long)row * p.lda + (l16 & 7) * 8;
  }
#pragma unroll
  for (int j = 0; j < IB; ++j) {
    const int s = (wave * IB + j) * 64 + lane, pr = s >> 4, l16 = (s & 15) ^ (pr & 15);
    int row = n0 + 2 * pr + (l16 >> 3);
    row = row < N ? row : N - 1;
    bsrc[j] = B + (long)row * p.ldb + (l16 & 7) * 8;
  }
  auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
    unsigned char* sa = smem + stage * STAGE + (wave * IA) * 1024;
    unsigned char* sb = smem + stage * STAGE + A_BYTES + (wave * IB) * 1024;
#pragma unroll
    for (int j = 0; j < IA; ++j)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(asrc[j] + kt * BK), (lds_ptr_t)(sa + j * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < IB; ++j)
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
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int st = kt & 1;
    if (kt + 1 < nk) issue(kt + 1, st ^ 1);          // streams in under the MFMAs below
    const unsigned char* ta = smem + st * STAGE;
    const unsigned char* tb = ta + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = ks * 2 + (lane >> 5);
      bf16x8 a[MI], b[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        a[i] = *reinterpret_cast<const bf16x8*>(ta + a_base[i] + (((a_hi[i] | c) ^ a_x[i]) << 4));
#pragma unroll
      for (int j = 0; j < NI; ++j)
        b[j] = *reinterpret_cast<const bf16x8*>(tb + b_base[j] + (((b_hi[j] | c) ^ b_x[j]) << 4));
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own DMA pieces of tile kt+1 have landed
    __syncthreads();                                     // everyone's have, and buffer `st` is free again
  }
  // ---- epilogue.  Full interior bf16 tiles go through LDS (the tile buffers are free now): the MFMA
  // C layout gives each lane ONE column of 16 rows, so direct stores are 2-byte scatters; staged, every
  // lane writes 16 contiguous bytes and a row of the tile leaves as whole 128-byte lines.
  if constexpr (sizeof(OutT) == 2) {
    constexpr int CS = (BM * (BN + 8) * 2 <= 2 * STAGE) ? BN + 8 : BN;   // padded row (elements) when it fits
    static_assert(BM * CS * 2 <= 2 * STAGE, "output tile must fit the freed tile buffers");
    const bool fast = !p.accumulate && p.act != 3 && m0 + BM <= M && n0 + BN <= N && (p.ldc & 7) == 0 &&
                      (reinterpret_cast<uintptr_t>(p.C) & 15) == 0;
    if (fast) {                                          // block-uniform
      uint16_t* Cs = reinterpret_cast<uint16_t*>(smem);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int col = wn * WN + j * 32 + (lane & 31);
          const float bn_ = p.bias_mode == 1 ? p.bias[n0 + col] : 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            float v = acc[i][j][r] + bn_;
            if (p.bias_mode == 2) v += p.bias[m0 + row];
            v *= p.alpha;
            if (p.act == 1) v = fmaxf(v, 0.f);
            else if (p.act == 2) v = gelu_erf(v);
            Cs[row * CS + col] = f2bf(v);
          }
        }
      __syncthreads();
      constexpr int CPRW = BN / 8, NT = 64 * NW;         // 16-byte chunks per tile row
      uint16_t* C = static_cast<uint16_t*>(p.C);
#pragma unroll
      for (int i = 0; i < BM * CPRW / NT; ++i) {
        const int c = tid + i * NT, row = c / CPRW, ch = c % CPRW;
        *reinterpret_cast<u32x4*>(C + (long)(m0 + row) * p.ldc + n0 + ch * 8) =
            *reinterpret_cast<const u32x4*>(Cs + row * CS + ch * 8);
      }
      return;
    }
  }
  gemm_epilogue<OutT, MI, NI>(acc, p, m0 + wm * WM, n0 + wn * WN, lane, M, N);
}

// ------------------------------------------------------------- direct-to-LDS, 3-stage ring, counted vmcnt
// Same tile image / swizzle as gemm_nt_glds_kernel, but the K loop keeps TWO tiles in flight: tile kt+2 is
// issued while tile kt is multiplied, and the wave only waits until its pieces of tile kt have landed
// (s_waitcnt vmcnt(<pieces per tile>), never 0 in steady state).  One raw s_barrier per K step
// (__syncthreads would emit vmcnt(0) and drain the LDS-DMA queue).
template <typename OutT, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_nt_glds3_kernel(GemmArgs p) {
  constexpr int NW = WAVES_M * WAVES_N, BK = 64, NSTAGE = 3;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int IA = BM / 8 / NW, IB = BN / 8 / NW, PIECES = IA + IB;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MI = WM / 32, NI = WN / 32;
  static_assert(NSTAGE * STAGE <= 160 * 1024, "LDS ring must fit one CU");
  static_assert(PIECES == 6 || PIECES == 8 || PIECES == 4, "vmcnt literals below");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NSTAGE * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  int M = p.M;
  if (p.m_dev) { int md = *p.m_dev; M = md < M ? md : M; }
  const int N = p.N, K = p.K;
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
  if (tile_id >= ((M + BM - 1) / BM) * tiles_n) return;

  const uint16_t* A = static_cast<const uint16_t*>(p.A);
  const uint16_t* B = static_cast<const uint16_t*>(p.B);
  const uint16_t* asrc[IA];
  const uint16_t* bsrc[IB];
#pragma unroll
  for (int j = 0; j < IA; ++j) {
    const int s = (wave * IA + j) * 64 + lane, pr = s >> 4, l16 = (s & 15) ^ (pr & 15);
    int row = m0 + 2 * pr + (l16 >> 3);
    row = row < M ? row : M - 1;
    asrc[j] = A + (long)row * p.lda + (l16 & 7) * 8;
  }
#pragma unroll
  for (int j = 0; j < IB; ++j) {
    const int s = (wave * IB + j) * 64 + lane, pr = s >> 4, l16 = (s & 15) ^ (pr & 15);
    int row = n0 + 2 * pr + (l16 >> 3);
    row = row < N ? row : N - 1;
    bsrc[j] = B + (long)row * p.ldb + (l16 & 7) * 8;
  }
  auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
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
  if (nk > 1) issue(1, 1);
  int st = 0;                                           // stage of tile kt
  for (int kt = 0; kt < nk; ++kt) {
    // wait until this wave's pieces of tile kt landed; the (newer) pieces of tile kt+1 stay in flight
    if (kt + 1 < nk) {
      if constexpr (PIECES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if constexpr (PIECES == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                       // all pieces of tile kt landed; stage of tile kt-1 is free
    asm volatile("" ::: "memory");
    if (kt + 2 < nk) issue(kt + 2, st == 0 ? 2 : st - 1);   // (kt+2) % 3
    const unsigned char* ta = smem + st * STAGE;
    const unsigned char* tb = ta + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = ks * 2 + (lane >> 5);
      bf16x8 a[MI], b[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        a[i] = *reinterpret_cast<const bf16x8*>(ta + a_base[i] + (((a_hi[i] | c) ^ a_x[i]) << 4));
#pragma unroll
      for (int j = 0; j < NI; ++j)
        b[j] = *reinterpret_cast<const bf16x8*>(tb + b_base[j] + (((b_hi[j] | c) ^ b_x[j]) << 4));
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    st = st == 2 ? 0 : st + 1;
  }
  __syncthreads();                                      // every wave is done reading before the epilogue reuses LDS
  if constexpr (sizeof(OutT) == 2) {
    constexpr int CS = BN + 8;
    static_assert(BM * CS * 2 <= NSTAGE * STAGE, "output tile must fit the freed tile buffers");
    const bool fast = !p.accumulate && p.act != 3 && m0 + BM <= M && n0 + BN <= N && (p.ldc & 7) == 0 &&
                      (reinterpret_cast<uintptr_t>(p.C) & 15) == 0;
    if (fast) {
      uint16_t* Cs = reinterpret_cast<uint16_t*>(smem);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int col = wn * WN + j * 32 + (lane & 31);
          const float bn_ = p.bias_mode == 1 ? p.bias[n0 + col] : 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            float v = acc[i][j][r] + bn_;
            if (p.bias_mode == 2) v += p.bias[m0 + row];
            v *= p.alpha;
            if (p.act == 1) v = fmaxf(v, 0.f);
            else if (p.act == 2) v = gelu_erf(v);
            Cs[row * CS + col] = f2bf(v);
          }
        }
      __syncthreads();
      constexpr int CPRW = BN / 8, NT = 64 * NW;
      uint16_t* C = static_cast<uint16_t*>(p.C);
#pragma unroll
      for (int i = 0; i < BM * CPRW / NT; ++i) {
        const int c = tid + i * NT, row = c / CPRW, ch = c % CPRW;
        *reinterpret_cast<u32x4*>(C + (long)(m0 + row) * p.ldc + n0 + ch * 8) =
            *reinterpret_cast<const u32x4*>(Cs + row * CS + ch * 8);
      }
      return;
    }
  }
  gemm_epilogue<OutT, MI, NI>(acc, p, m0 + wm * WM, n0 + wn * WN, lane, M, N);
}

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
        _Pragma("unroll") for (int j = 0; j < NI; ++j) acc[i][j] = M_::mma(a[i], b[j], acc[i][j]); \
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

  gemm_epilogue<OutT, MI, NI>(acc, p, m0 + wm * WM, n0 + wn * WN, lane, M, N);
}

template <typename T, typename OutT>
static int launch_gemm(const GemmArgs& a, hipStream_t stream) {
  // Tile choice: the kernel is bound by operand re-reads from L2 (flop/byte of a tile =
  // BM*BN/(BM+BN) per 2-byte element), so take the largest tile that still gives every CU work.
  auto tiles = [&](int bm, int bn) { return (long)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn); };
  if constexpr (sizeof(T) == 2) {
    if (a.K % 64 == 0 && tiles(128, 128) >= 256) {       // direct-to-LDS path
      static const int force = getenv("TELL_GEMM_TILE") ? atoi(getenv("TELL_GEMM_TILE")) : 0;   // tuning aid
      if (force == 5) {   // 256x256, 8 waves (128x64 per wave), 2-stage
        hipLaunchKernelGGL((gemm_nt_glds_kernel<OutT, 256, 256, 2, 4>), dim3((unsigned)tiles(256, 256)), dim3(512), 0, stream, a);
        return tell_check_launch("gemm_nt_glds");
      }
      if (force == 3) {   // 3-stage ring, counted vmcnt, 256x128, 8 waves
        hipLaunchKernelGGL((gemm_nt_glds3_kernel<OutT, 256, 128, 4, 2>), dim3((unsigned)tiles(256, 128)), dim3(512), 0, stream, a);
        return tell_check_launch("gemm_nt_glds3");
      }
      if (force == 4) {   // 3-stage ring, 128x128, 4 waves
        hipLaunchKernelGGL((gemm_nt_glds3_kernel<OutT, 128, 128, 2, 2>), dim3((unsigned)tiles(128, 128)), dim3(256), 0, stream, a);
        return tell_check_launch("gemm_nt_glds3");
      }
      if (force == 2)     // 256x128 (8 waves, 1 workgroup/CU) ties 128x128 (2 workgroups/CU) on MI355X: opt-in only
        hipLaunchKernelGGL((gemm_nt_glds_kernel<OutT, 256, 128, 4, 2>), dim3((unsigned)tiles(256, 128)), dim3(512), 0, stream, a);
      else
        hipLaunchKernelGGL((gemm_nt_glds_kernel<OutT, 128, 128, 2, 2>), dim3((unsigned)tiles(128, 128)), dim3(256), 0, stream, a);
      return tell_check_launch("gemm_nt_glds");
    }
  }
  // (256x256 with 8 waves measured SLOWER than 256x128 on MI355X - 394 vs 552 TFLOP/s on the RoBERTa
  //  shapes: 236 VGPRs, one workgroup per CU - so it is compiled but not selected)
  if (sizeof(T) == 2 && tiles(256, 256) >= (1L << 40)) {
    hipLaunchKernelGGL((gemm_nt_kernel<T, OutT, 256, 256, 2, 4, 2>), dim3((unsigned)tiles(256, 256)), dim3(512), 0, stream, a);
  } else if (sizeof(T) == 2 && tiles(256, 128) >= 256) {
    hipLaunchKernelGGL((gemm_nt_kernel<T, OutT, 256, 128, 4, 2, 2>), dim3((unsigned)tiles(256, 128)), dim3(512), 0, stream, a);
  } else if (tiles(128, 128) >= 256) {
    hipLaunchKernelGGL((gemm_nt_kernel<T, OutT, 128, 128, 2, 2, 2>), dim3((unsigned)tiles(128, 128)), dim3(256), 0, stream, a);
  } else {
    hipLaunchKernelGGL((gemm_nt_kernel<T, OutT, 64, 64, 2, 2, 4>), dim3((unsigned)tiles(64, 64)), dim3(256), 0, stream, a);
  }
  return tell_check_launch("gemm_nt");
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
  TELL_REQUIRE(act != 3 || aux != nullptr, "gemm_nt: act=3 needs aux");
  GemmArgs a;
  a.A = A; a.B = B; a.C = C; a.bias = bias; a.aux = aux; a.m_dev = m_dev;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
  a.bias_mode = bias_mode; a.act = act; a.accumulate = accumulate; a.alpha = alpha;
  if (in_dtype == TELL_BF16)
    return out_dtype == TELL_BF16 ? launch_gemm<uint16_t, uint16_t>(a, stream)
                                  : launch_gemm<uint16_t, float>(a, stream);
  return out_dtype == TELL_BF16 ? launch_gemm<float, uint16_t>(a, stream)
                                : launch_gemm<float, float>(a, stream);
}
