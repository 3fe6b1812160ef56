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

template <typename T, typename OutT, int BM, int BN, int PF>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs p) {
  using M_ = Mma<T>;
  constexpr int BK = M_::BK, STRIDE = M_::STRIDE, VEC = Elem<T>::VEC;
  constexpr int CPR = BK / VEC;                 // 16-byte chunks per tile row (= 8)
  constexpr int CHA = BM * CPR / 256, CHB = BN * CPR / 256;
  constexpr int WM = BM / 2, WN = BN / 2;       // per-wave tile
  constexpr int MI = WM / 32, NI = WN / 32;
  static_assert(PF == 2 || PF == 4, "prefetch depth 2 or 4 (even: LDS double-buffer parity)");

  __shared__ __attribute__((aligned(16))) T As[2][BM * STRIDE];
  __shared__ __attribute__((aligned(16))) T Bs[2][BN * STRIDE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
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
  const int m0 = (tile_id / tiles_n) * BM, n0 = (tile_id % tiles_n) * BN;
  if (m0 >= M) return;                          // uniform per block

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
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
#define GLOAD(KT, S)                                                                              \
  {                                                                                               \
    _Pragma("unroll") for (int i = 0; i < CHA; ++i) {                                             \
      int c = tid + i * 256, row = c / CPR, ch = c % CPR;                                         \
      int gm = m0 + row, gk = (KT) * BK + ch * VEC;                                               \
      ra[S][i] = (gm < M && gk < K) ? *reinterpret_cast<const u32x4*>(A + (long)gm * p.lda + gk) : zero4; \
    }                                                                                             \
    _Pragma("unroll") for (int i = 0; i < CHB; ++i) {                                             \
      int c = tid + i * 256, row = c / CPR, ch = c % CPR;                                         \
      int gn = n0 + row, gk = (KT) * BK + ch * VEC;                                               \
      rb[S][i] = (gn < N && gk < K) ? *reinterpret_cast<const u32x4*>(B + (long)gn * p.ldb + gk) : zero4; \
    }                                                                                             \
  }
#define SSTORE(S, BUF)                                                                            \
  {                                                                                               \
    _Pragma("unroll") for (int i = 0; i < CHA; ++i) {                                             \
      int c = tid + i * 256;                                                                      \
      M_::store_chunk(As[BUF], c / CPR, c % CPR, ra[S][i]);                                       \
    }                                                                                             \
    _Pragma("unroll") for (int i = 0; i < CHB; ++i) {                                             \
      int c = tid + i * 256;                                                                      \
      M_::store_chunk(Bs[BUF], c / CPR, c % CPR, rb[S][i]);                                       \
    }                                                                                             \
  }

  const int nk = (K + BK - 1) / BK;
  // one pipeline step with a LITERAL stage index S (register stages must be statically indexed)
#define STEP(S)                                                                                   \
  {                                                                                               \
    const int kt = kt0 + (S);                                                                     \
    if (kt < nk) {                                                                                \
      if (kt + PF < nk) GLOAD(kt + PF, S)                                                         \
      const T* at = As[(S) & 1] + (wm * WM) * STRIDE;                                             \
      const T* bt = Bs[(S) & 1] + (wn * WN) * STRIDE;                                             \
      _Pragma("unroll") for (int ks = 0; ks < BK; ks += M_::KSTEP) {                              \
        typename M_::frag a[MI], b[NI];                                                           \
        _Pragma("unroll") for (int i = 0; i < MI; ++i) a[i] = M_::load(at, i * 32 + (lane & 31), ks, lane); \
        _Pragma("unroll") for (int j = 0; j < NI; ++j) b[j] = M_::load(bt, j * 32 + (lane & 31), ks, lane); \
        _Pragma("unroll") for (int i = 0; i < MI; ++i)                                            \
          _Pragma("unroll") for (int j = 0; j < NI; ++j) acc[i][j] = M_::mma(a[i], b[j], acc[i][j]); \
      }                                                                                           \
      if (kt + 1 < nk) SSTORE(((S) + 1) % PF, ((S) + 1) & 1)                                      \
      __syncthreads();                                                                            \
    }                                                                                             \
  }
  GLOAD(0, 0)
  if (1 < nk) GLOAD(1, 1)
  if constexpr (PF == 4) {
    if (2 < nk) GLOAD(2, 2)
    if (3 < nk) GLOAD(3, 3)
  }
  SSTORE(0, 0)
  __syncthreads();
  for (int kt0 = 0; kt0 < nk; kt0 += PF) {
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

  // ------------------------------------------------------------- epilogue
  OutT* C = static_cast<OutT*>(p.C);
  const OutT* aux = static_cast<const OutT*>(p.aux);
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wn * WN + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
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

template <typename T, typename OutT>
static int launch_gemm(const GemmArgs& a, hipStream_t stream) {
  // large tile only when it still fills the 256 CUs; the small tile gets a deeper prefetch
  // (its per-tile MFMA time is too short to cover HBM latency with 2 tiles in flight)
  long tiles128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
  if (tiles128 >= 256) {
    dim3 grid((unsigned)tiles128);
    hipLaunchKernelGGL((gemm_nt_kernel<T, OutT, 128, 128, 2>), grid, dim3(256), 0, stream, a);
  } else {
    long tiles64 = (long)((a.M + 63) / 64) * ((a.N + 63) / 64);
    dim3 grid((unsigned)tiles64);
    hipLaunchKernelGGL((gemm_nt_kernel<T, OutT, 64, 64, 4>), grid, dim3(256), 0, stream, a);
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
