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
__device__ __forceinline__ uint32_t tell_step_salt(uint32_t salt, const uint32_t* step) {
  return step ? salt + *step * 0x632BE5ABu : salt;
}

// ---------------------------------------------------------------- counter-based RNG for dropout
// Stateless.  One 32-bit hash serves an aligned PAIR of element indices (2i, 2i+1): element idx keeps iff
// the 16-bit half (idx & 1) of hash(seed, salt, idx >> 1) is >= floor(p * 2^16).  Kernels whose lanes own
// consecutive elements (attention probabilities, vectorised rows) so pay one hash per two decisions.  The
// same functions are restated in numpy (tell_amd/rng.py) so tests can rebuild masks.
__device__ __host__ __forceinline__ uint32_t tell_hash32(uint32_t seed, uint32_t salt, uint64_t idx) {
  uint32_t x = (uint32_t)idx * 0x9E3779B1u + seed;
  const uint32_t y = (uint32_t)(idx >> 32) * 0x85EBCA77u + salt * 0xC2B2AE3Du + 0x27D4EB2Fu;
  x ^= y; x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
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
  const uint32_t h = tell_hash32(seed, salt, idx >> 1);
  const uint32_t bits = (idx & 1) ? (h >> 16) : (h & 0xffffu);
  return bits >= thr ? inv_keep : 0.f;
}
// Consecutive pairs of one row: the index-dependent part of the hash input is linear in the pair index, so a
// kernel that walks pair_idx0 + d for small compile-time d pays one add per pair instead of two multiplies.
// Valid while the low word of the pair index does not wrap (caller checks tell_keep_row_ok).
struct TellKeepRow { uint32_t x0, y; };
__device__ __forceinline__ bool tell_keep_row_ok(uint64_t pair_idx0, uint32_t span) {
  return (uint32_t)pair_idx0 <= 0xFFFFFFFFu - span;
}
__device__ __forceinline__ TellKeepRow tell_keep_row(uint32_t seed, uint32_t salt, uint64_t pair_idx0) {
  TellKeepRow r;
  r.x0 = (uint32_t)pair_idx0 * 0x9E3779B1u + seed;
  r.y = (uint32_t)(pair_idx0 >> 32) * 0x85EBCA77u + salt * 0xC2B2AE3Du + 0x27D4EB2Fu;
  return r;
}
__device__ __forceinline__ void tell_keep2_row(const TellKeepRow& r, uint32_t d, uint32_t thr, float inv_keep,
                                               float& k0, float& k1) {      // == tell_keep2 at pair index pair_idx0 + d
  uint32_t x = r.x0 + d * 0x9E3779B1u;
  x ^= r.y; x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  k0 = (x & 0xffffu) >= thr ? inv_keep : 0.f;
  k1 = (x >> 16) >= thr ? inv_keep : 0.f;
}
// The same two decisions as booleans, with the caller maintaining x = r.x0 + d * 0x9E3779B1 itself (a kernel that
// walks a fixed pattern of d keeps ONE running value and adds one of two constants per pair instead of holding a
// constant per pair in SGPRs).  keep <=> tell_keep2_row's factor != 0.
#define TELL_PAIR_STRIDE 0x9E3779B1u
__device__ __forceinline__ void tell_keep2_bits(uint32_t x, uint32_t y, uint32_t thr, bool& k0, bool& k1) {
  x ^= y; x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  k0 = (x & 0xffffu) >= thr;
  k1 = (x >> 16) >= thr;
}
// both decisions of the aligned pair starting at the EVEN index idx_even
__device__ __forceinline__ void tell_keep2(uint32_t seed, uint32_t salt, uint64_t idx_even, uint32_t thr,
                                           float inv_keep, float& k0, float& k1) {
  const uint32_t h = tell_hash32(seed, salt, idx_even >> 1);
  k0 = (h & 0xffffu) >= thr ? inv_keep : 0.f;
  k1 = (h >> 16) >= thr ? inv_keep : 0.f;
}
