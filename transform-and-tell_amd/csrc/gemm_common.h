// Shared by the GEMM translation units (gemm.hip, gemm_pp2.hip, gemm_q4.hip, gemm_q4e.hip): argument block, epilogue arithmetic, execution-span stamps.
#pragma once
#include "common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // 16-byte staging chunk (native vector: stays in VGPRs)

struct GemmArgs {
  const void* A; const void* B; void* C;
  const float* bias;      // fp32, length N (mode 1) or M (mode 2)
  const void* aux;        // act 3: relu-mask source, act 4: residual added before the relu (same layout/dtype as C)
  const int* m_dev;       // optional device-side effective M (rows >= *m_dev are skipped)
  const int* k_dev = nullptr;   // optional device-side effective K of the K-major forms (rows >= *k_dev of both operands are not read; only the grouped launches set it)
  long lda, ldb, ldc;
  int M, N, K;
  int bias_mode;          // 0 none, 1 per column n, 2 per row m
  int act;                // 0 none, 1 relu, 2 gelu(erf), 3 multiply by (aux > 0), 4 relu(result + aux) (residual block)
  int accumulate;         // C = C + result  (beta = 1)
  float alpha;            // result = act((acc + bias) * alpha)
  float* asum;            // K-major A only: asum[m] += asum_scale * sum_k A[k][m]  (bias gradient of a wgrad GEMM)
  float asum_scale;
  unsigned long long* ts; // measurement aid: ts[0] = min over workgroups of the wall clock at entry, ts[1] = max at exit
  int atomic_out;         // fp32 output shared by several workgroups (split K): accumulate with hardware float atomics
  int* queue;             // persistent ping-pong kernel: the launch's tile counter (zero between launches), or NULL
  // pp2 kernel, act 5: C = aux2[m,n] + dropout(result): the transformer sub-layer residual (fairseq: x = residual +
  // dropout(out_proj(.)) / dropout(fc2(.))) in the epilogue - the mask is the one tell_layernorm_fwd would draw for the
  // same (seed, salt): element index m * N + n, csrc/common.h quad hash
  const void* res; long ld_res; uint32_t drop_thr; float drop_inv_keep; uint32_t drop_seed, drop_salt;
  const uint32_t* drop_step;
  // implicit convolution (direct-to-LDS kernel): A is not a matrix but the NHWC activation [B,H,W,Cin]; row m is output
  // pixel (b, oh, ow), K = KH*KW*Cin in (kh, kw, c) order - each 64-wide K tile lies inside one tap (Cin % 64 == 0), and
  // every lane's DMA source is the shifted input pixel (a 128-byte zero page for the padding ring): no im2col matrix
  const void* conv_zero;  // != NULL selects the mode
  int conv_H, conv_W, conv_OH, conv_OW, conv_KW, conv_stride, conv_pad, conv_cshift;   // Cin = 64 << conv_cshift
  float* stat_mean;       // bf16 NT kernels: per (m-tile, column) mean / M2 of the STORED (bf16-rounded) outputs over
  float* stat_m2;         //   the tile's valid rows, [tiles_m][N] each - the first stage of train-mode BatchNorm
};

// In-kernel execution span (bench.py roofline for launches replayed from hipGraphs, where neither HIP events nor an
// external profiler can bracket a kernel): the first workgroup to arrive stores the device wall clock into ts[0], every
// workgroup folds its exit time into ts[1] (max) - the interval rocprofv3 reports as the kernel's duration.
// tell_gemm_ts_next arms it for the next tell_gemm_nt launch only.
__device__ __forceinline__ void gemm_ts_enter(const GemmArgs& p);
__device__ __forceinline__ void gemm_ts_exit(const GemmArgs& p);

// exact-erf GELU without exp: Abramowitz-Stegun 7.1.28,  erfc(x) = 1 / (1 + a1 x + ... + a6 x^6)^16  for x >= 0
// (|error| <= 3e-7), and  gelu(v) = v Phi(v) = max(v, 0) - |v| erfc(|v| / sqrt 2) / 2.  One v_rcp_f32 and fourteen
// multiply-adds per element (the compiler packs pairs into v_pk_fma_f32 / v_pk_mul_f32) instead of rcp + exp + two
// more multiplies for the 7.1.26 form: the epilogue of the fc1 GEMM is pure VALU work with nothing to overlap it
// (one workgroup per CU), so every instruction there is wall time.  A large argument overflows the power to +inf and
// v_rcp_f32 returns 0, which is the right limit.
typedef __attribute__((ext_vector_type(2))) float f32x2e_t;
__device__ __forceinline__ f32x2e_t gelu_erf2(f32x2e_t v) {          // two elements per instruction (packed fp32 VALU)
  const f32x2e_t av = {fabsf(v[0]), fabsf(v[1])};
  const f32x2e_t x = av * 0.70710678118654752f;
  f32x2e_t q = x * 0.0000430638f + 0.0002765672f;
  q = q * x + 0.0001520143f;
  q = q * x + 0.0092705272f;
  q = q * x + 0.0422820123f;
  q = q * x + 0.0705230784f;
  q = q * x + 1.f;
  q *= q; q *= q; q *= q; q *= q;                                   // ^16
  const f32x2e_t erfc_x = {__builtin_amdgcn_rcpf(q[0]), __builtin_amdgcn_rcpf(q[1])};
  const f32x2e_t relu = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)};
  return (av * -0.5f) * erfc_x + relu;
}
template <int ACT> __device__ __forceinline__ float epi_act(float v) {
  if constexpr (ACT == 1) return fmaxf(v, 0.f);
  else if constexpr (ACT == 2) { const f32x2e_t r = gelu_erf2(f32x2e_t{v, v}); return r[0]; }
  else return v;
}
// four consecutive outputs at once (what every epilogue holds per register quad)
template <int ACT> __device__ __forceinline__ void epi_act4(__attribute__((ext_vector_type(4))) float& v) {
  if constexpr (ACT == 2) {
    const f32x2e_t lo = gelu_erf2(f32x2e_t{v[0], v[1]}), hi = gelu_erf2(f32x2e_t{v[2], v[3]});
    v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = epi_act<ACT>(v[e]);
  }
}
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2e_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {      // v_cvt_pk_bf16_f32 (RNE)
  f32x2e_t f = {lo, hi};
  bf16x2e_t h = __builtin_convertvector(f, bf16x2e_t);
  return *reinterpret_cast<unsigned*>(&h);
}
template <typename OutT> struct Vec4;
template <> struct Vec4<float> {
  __device__ static __forceinline__ f32x4_t ld(const float* p) { return *reinterpret_cast<const f32x4_t*>(p); }
  __device__ static __forceinline__ void st(float* p, f32x4_t v) { *reinterpret_cast<f32x4_t*>(p) = v; }
};
template <> struct Vec4<uint16_t> {
  __device__ static __forceinline__ f32x4_t ld(const uint16_t* p) {
    const u32x2 w = *reinterpret_cast<const u32x2*>(p);
    f32x4_t v = {__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xffff0000u),
                 __uint_as_float(w[1] << 16), __uint_as_float(w[1] & 0xffff0000u)};
    return v;
  }
  __device__ static __forceinline__ void st(uint16_t* p, f32x4_t v) {
    u32x2 w = {pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3])};
    *reinterpret_cast<u32x2*>(p) = w;
  }
};

// ts[0] = entry time of the launch's first workgroup, ts[1] = latest exit, ts[2] = arrivals so far.  The workgroup
// whose arrival ticket is a multiple of the grid size opens a new launch (stores its entry time, clears the exit word):
// no reset launch in front of every sampled GEMM (22 five-microsecond launches per step on the critical stream before).
__device__ __forceinline__ void gemm_ts_enter(const GemmArgs& p) {
  if (p.ts && threadIdx.x == 0) {
    const unsigned long long now = wall_clock64();
    const unsigned long long n = atomicAdd(p.ts + 2, 1ull);
    if (n % gridDim.x == 0) { p.ts[1] = 0ull; p.ts[0] = now; }
  }
}
__device__ __forceinline__ void gemm_ts_exit(const GemmArgs& p) {
  if (p.ts && threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's stores have left
    atomicMax(p.ts + 1, (unsigned long long)wall_clock64());
  }
}

// gemm_pp2.hip: the 256x256 ping-pong kernel as resident workgroups that prefetch the next output tile under the epilogue
int launch_gemm_pp2(const GemmArgs& a, hipStream_t stream, int n_cu);
// gemm_q4.hip: 256x256 tiles, four waves of 128x128, hand-placed K loop (round 4)
int launch_gemm_q4(const GemmArgs& a, hipStream_t stream, int n_cu);
// gemm_q4e.hip: the same K loop with the previous tile's epilogue inside it (-> 1: not a call it takes)
int launch_gemm_q4e(const GemmArgs& a, hipStream_t stream, int n_cu);
// gemm_s64.hip: 64x64 tiles with the one-wave-per-SIMD K loop (plain NT GEMM or implicit convolution; -> 1: not a call it takes)
int launch_gemm_s64(const GemmArgs& a, hipStream_t stream, int out_f32);
bool gemm_s64_takes(const GemmArgs& a);          // the launch path's own eligibility test (also asked by tell_gemm_nt_plan)
int* gemm_tile_queue_slot(int words, hipStream_t stream);    // gemm.hip: `words` zeroed tile counters for one launch, or NULL
