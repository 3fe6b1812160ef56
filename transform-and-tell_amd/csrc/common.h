// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of libtell_hip.so.
// Wave = 64 lanes everywhere; no CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TELL_OK 0
#define TELL_ERR_ARG (-1)
#define TELL_ERR_LAUNCH (-2)

// dtype codes of the C ABI (include/tell_hip.h)
#define TELL_F32 0
#define TELL_BF16 1

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

extern "C" void tell_set_error(const char* msg);

static inline int tell_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    tell_set_error(hipGetErrorString(e));
    (void)what;
    return TELL_ERR_LAUNCH;
  }
  return TELL_OK;
}

#define TELL_REQUIRE(cond, msg)  \
  do {                           \
    if (!(cond)) {               \
      tell_set_error(msg);       \
      return TELL_ERR_ARG;       \
    }                            \
  } while (0)

// ---------------------------------------------------------------- conversions
__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {  // round-to-nearest-even, NaN preserved
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// nn.GLU on one element, a * sigmoid(g) - every kernel that applies it (glu_fwd_kernel, the generation step's linear1
// epilogue, the fused conv-block core) uses this one expression, so they agree bit for bit
// (v_rcp_f32 instead of an IEEE division: 1 ulp of fp32 on a value that is stored as bf16 or enters a bf16 GEMM; the
// division was 10 of the ~25 instructions per element and the fused conv-block core applies GLU 8x redundantly)
__device__ __forceinline__ float tell_glu(float a, float g) { return a * __builtin_amdgcn_rcpf(1.f + __expf(-g)); }

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int VEC = 4;  // elements per 16-byte chunk
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<uint16_t> {  // bf16 storage
  static constexpr int VEC = 8;
  __device__ static __forceinline__ float ld(const uint16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void st(uint16_t* p, float v) { *p = f2bf(v); }
};

// unpack a 16-byte chunk into floats / pack floats into a 16-byte chunk
__device__ __forceinline__ void unpack16(const uint4& c, float (&v)[4], const float*) {
  v[0] = __uint_as_float(c.x); v[1] = __uint_as_float(c.y);
  v[2] = __uint_as_float(c.z); v[3] = __uint_as_float(c.w);
}
__device__ __forceinline__ void unpack16(const uint4& c, float (&v)[8], const uint16_t*) {
  v[0] = __uint_as_float(c.x << 16); v[1] = __uint_as_float(c.x & 0xffff0000u);
  v[2] = __uint_as_float(c.y << 16); v[3] = __uint_as_float(c.y & 0xffff0000u);
  v[4] = __uint_as_float(c.z << 16); v[5] = __uint_as_float(c.z & 0xffff0000u);
  v[6] = __uint_as_float(c.w << 16); v[7] = __uint_as_float(c.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack16(const float (&v)[4], const float*) {
  return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]),
                    __float_as_uint(v[3]));
}
__device__ __forceinline__ uint4 pack16(const float (&v)[8], const uint16_t*) {
  return make_uint4((uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16),
                    (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16),
                    (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16),
                    (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16));
}

// ---------------------------------------------------------------- wave reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------- replayable dropout (hipGraph capture)
// A captured kernel's seed/salt arguments are frozen.  While `g_tell_rng_step` points at a device counter
// (tell_set_rng_step_ptr), every launcher passes that pointer along and the kernel adds counter * odd constant to
// its salt, so a graph replays with fresh masks once the owner bumps the counter; NULL (the default, eager mode)
// leaves the salt untouched.
extern const uint32_t* g_tell_rng_step;
extern const uint32_t* g_tell_pos_step;      // position offset counter of a captured decode step (api.hip)
extern uint32_t* g_tell_pos_next;            // ... and the word that holds the NEXT step's offset (in-graph bookkeeping), or null
__device__ __forceinline__ uint32_t tell_step_salt(uint32_t salt, const uint32_t* step) {
  return step ? salt + *step * 0x632BE5ABu : salt;
}

// ---------------------------------------------------------------- counter-based RNG for dropout
// Stateless.  One hash serves an aligned QUAD of element indices (4i .. 4i+3): a shared 32-bit mix of
// (seed, salt, i) - one full-rate-unfriendly 32-bit multiply - followed by two cheap finalisers, each a 24-bit
// multiply (v_mul_u32_u24, full rate) of one 24-bit window of the mix and a xor-shift.  Finaliser a carries the
// 16-bit fields of elements 0 (low half) and 1 (high half), finaliser b those of elements 2 and 3; element idx is
// kept iff its field is >= floor(p * 2^16).  Kernels whose lanes own consecutive elements (attention probabilities,
// vectorised rows) pay one mix per four decisions; the RoBERTa self-attention kernel, where the hash is the larger
// half of the VALU work, is what this shape is for (56 issue cycles per two decisions against 96 for a full
// two-multiply hash per pair).  The same functions are restated in numpy (tell_amd/rng.py) so tests can rebuild
// masks; tests/test_abi_and_host.py checks keep rate, joint quad patterns and lag correlations.
__device__ __host__ __forceinline__ uint32_t tell_quad_x(uint32_t seed, uint64_t quad) {
  return (uint32_t)quad * 0x9E3779B1u + seed * 0x85EBCA6Bu;
}
__device__ __host__ __forceinline__ uint32_t tell_quad_y(uint32_t salt, uint64_t quad) {
  return (uint32_t)(quad >> 32) * 0x85EBCA77u + salt * 0xC2B2AE3Du + 0x27D4EB2Fu;
}
__device__ __host__ __forceinline__ uint32_t tell_quad_mix(uint32_t x, uint32_t y) {
  x ^= y; x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15;
  return x;
}
__device__ __host__ __forceinline__ uint32_t tell_quad_a(uint32_t h) {       // elements 0, 1
  const uint32_t a = (h & 0xffffffu) * 0xD1B54Bu;
  return a ^ (a >> 15);
}
__device__ __host__ __forceinline__ uint32_t tell_quad_b(uint32_t h) {       // elements 2, 3
  const uint32_t b = (h >> 8) * 0xA54FF5u;
  return b ^ (b >> 15);
}
// the 16-bit field element idx is judged by
__device__ __host__ __forceinline__ uint32_t tell_keep_field(uint32_t seed, uint32_t salt, uint64_t idx) {
  const uint64_t quad = idx >> 2;
  const uint32_t h = tell_quad_mix(tell_quad_x(seed, quad), tell_quad_y(salt, quad));
  const uint32_t w = (idx & 2) ? tell_quad_b(h) : tell_quad_a(h);
  return (idx & 1) ? (w >> 16) : (w & 0xffffu);
}
__device__ __host__ __forceinline__ uint32_t tell_drop_threshold(float p) {   // 16-bit: 0 = no dropout
  double t = (double)p * 65536.0;
  if (t < 0) t = 0;
  if (t > 65535.0) t = 65535.0;
  return (uint32_t)t;
}
// returns the multiplicative keep factor: 0 or 1/(1-p)
__device__ __forceinline__ float tell_keep(uint32_t seed, uint32_t salt, uint64_t idx, uint32_t thr,
                                           float inv_keep) {
  return tell_keep_field(seed, salt, idx) >= thr ? inv_keep : 0.f;
}
// both decisions of the aligned pair starting at the EVEN index idx_even
__device__ __forceinline__ void tell_keep2(uint32_t seed, uint32_t salt, uint64_t idx_even, uint32_t thr,
                                           float inv_keep, float& k0, float& k1) {
  const uint64_t quad = idx_even >> 2;
  const uint32_t h = tell_quad_mix(tell_quad_x(seed, quad), tell_quad_y(salt, quad));
  const uint32_t w = (idx_even & 2) ? tell_quad_b(h) : tell_quad_a(h);
  k0 = (w & 0xffffu) >= thr ? inv_keep : 0.f;
  k1 = (w >> 16) >= thr ? inv_keep : 0.f;
}
// Consecutive quads of one row: the index-dependent part of the hash input is linear in the quad index, so a kernel
// that walks quad0 + d keeps ONE running value x and adds a multiple of TELL_QUAD_STRIDE per quad instead of
// redoing the index multiply.  Valid while the low word of the quad index does not wrap (tell_keep_row_ok).
struct TellKeepRow { uint32_t x0, y; };
#define TELL_QUAD_STRIDE 0x9E3779B1u
__device__ __forceinline__ bool tell_keep_row_ok(uint64_t quad0, uint32_t span) {
  return (uint32_t)quad0 <= 0xFFFFFFFFu - span;
}
__device__ __forceinline__ TellKeepRow tell_keep_row(uint32_t seed, uint32_t salt, uint64_t quad0) {
  TellKeepRow r;
  r.x0 = tell_quad_x(seed, quad0);
  r.y = tell_quad_y(salt, quad0);
  return r;
}
// the four decisions of the quad whose running value is x (= row.x0 + d * TELL_QUAD_STRIDE)
__device__ __forceinline__ void tell_keep4_bits(uint32_t x, uint32_t y, uint32_t thr, bool& k0, bool& k1, bool& k2,
                                                bool& k3) {
  const uint32_t h = tell_quad_mix(x, y);
  const uint32_t a = tell_quad_a(h), b = tell_quad_b(h);
  k0 = (a & 0xffffu) >= thr;
  k1 = (a >> 16) >= thr;
  k2 = (b & 0xffffu) >= thr;
  k3 = (b >> 16) >= thr;
}
