// Batched small launches of the training step.  The decoder's backward pass ends dozens of tiny reductions whose
// results only the optimizer reads - LayerNorm gamma / beta gradients (column sums of per-row-block partials), the
// bias_k / bias_v gradients of the context attentions (column sums of per-batch rows) - and starts from a dozen
// per-step weight transposes.  Each was a ~5 us launch on the critical stream; here each KIND is one launch whose
// jobs travel by value in the kernel arguments (as tell_gemm_grouped does for GEMMs).
#include "common.h"
#include "../../include/tell_hip.h"

#define MULTI_MAX 48
struct ColsumJobs {
  const float* src[MULTI_MAX];      // [rows][width] fp32, row stride ld
  float* dst0[MULTI_MAX];           // columns [0, w0)      accumulate: dst0[c] += sum_r src[r][c]
  float* dst1[MULTI_MAX];           // columns [w0, width)  -> dst1[c - w0]
  long ld[MULTI_MAX];
  int rows[MULTI_MAX], width[MULTI_MAX], w0[MULTI_MAX];
  int start[MULTI_MAX + 1];         // first block of job i (64 columns per block)
  int n;
};
// 64 columns x 16 row groups per block (1024 threads), four independent loads in flight per thread, fixed-order fold:
// deterministic (the layout of ln_bwd_finish_kernel)
__global__ __launch_bounds__(1024) void colsum_multi_kernel(ColsumJobs j) {
  __shared__ float sm[16][64];
  int i = 0;
  while (i + 1 < j.n && (int)blockIdx.x >= j.start[i + 1]) ++i;            // block-uniform
  const int cx = threadIdx.x & 63, by = threadIdx.x >> 6;
  const int c = ((int)blockIdx.x - j.start[i]) * 64 + cx;
  const float* __restrict__ src = j.src[i];
  const int rows = j.rows[i], width = j.width[i];
  const long ld = j.ld[i];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < width) {
    int r = by;
    for (; r + 48 < rows; r += 64) {
      s0 += src[(long)r * ld + c];
      s1 += src[(long)(r + 16) * ld + c];
      s2 += src[(long)(r + 32) * ld + c];
      s3 += src[(long)(r + 48) * ld + c];
    }
    for (; r < rows; r += 16) s0 += src[(long)r * ld + c];
  }
  sm[by][cx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (by == 0 && c < width) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += sm[k][cx];
    float* dst = c < j.w0[i] ? j.dst0[i] + c : j.dst1[i] + (c - j.w0[i]);
    *dst += s;
  }
}
// n jobs (host arrays of length n): dst0[i][c] += sum_r src[i][r][c] for c < w0[i], dst1[i][c - w0[i]] += ... beyond.
extern "C" int tell_colsum_multi(int n, const void* const* src, const long* ld, const int* rows, const int* width,
                                 const int* w0, void* const* dst0, void* const* dst1, hipStream_t stream) {
  for (int base = 0; base < n; base += MULTI_MAX) {
    ColsumJobs j;
    j.n = n - base < MULTI_MAX ? n - base : MULTI_MAX;
    j.start[0] = 0;
    for (int i = 0; i < j.n; ++i) {
      const int k = base + i;
      TELL_REQUIRE(rows[k] > 0 && width[k] > 0 && w0[k] >= 0 && w0[k] <= width[k] && ld[k] >= width[k],
                   "colsum_multi: bad job geometry");
      TELL_REQUIRE(src[k] && (w0[k] == 0 || dst0[k]) && (w0[k] == width[k] || dst1[k]), "colsum_multi: null pointer");
      j.src[i] = static_cast<const float*>(src[k]);
      j.dst0[i] = static_cast<float*>(dst0[k]);
      j.dst1[i] = static_cast<float*>(dst1[k]);
      j.ld[i] = ld[k]; j.rows[i] = rows[k]; j.width[i] = width[k]; j.w0[i] = w0[k];
      j.start[i + 1] = j.start[i] + (width[k] + 63) / 64;
    }
    hipLaunchKernelGGL(colsum_multi_kernel, dim3(j.start[j.n]), dim3(1024), 0, stream, j);
  }
  return tell_check_launch("colsum_multi");
}

// ------------------------------------------------------------------ n bf16 transposes in one launch
struct TransposeJobs {
  const uint16_t* src[MULTI_MAX];
  uint16_t* dst[MULTI_MAX];
  long ld_src[MULTI_MAX], ld_dst[MULTI_MAX];
  int rows[MULTI_MAX], cols[MULTI_MAX], tiles_x[MULTI_MAX];
  int start[MULTI_MAX + 1];
  int n;
};
// dst[c][r] = src[r][c], 64 x 64 tiles through LDS: 16-byte reads along the source rows, 16-byte writes along the
// destination rows (the one-element-per-thread transpose_kernel of elementwise.hip moves 2 bytes per access)
__global__ __launch_bounds__(256) void transpose_multi_kernel(TransposeJobs j) {
  __shared__ uint16_t tile[64][72];                                          // +8: 16-byte rows stay aligned, column reads spread
  int i = 0;
  while (i + 1 < j.n && (int)blockIdx.x >= j.start[i + 1]) ++i;
  const int t = (int)blockIdx.x - j.start[i];
  const int r0 = (t / j.tiles_x[i]) * 64, c0 = (t % j.tiles_x[i]) * 64;
  const int rows = j.rows[i], cols = j.cols[i];
  const uint16_t* __restrict__ src = j.src[i];
  uint16_t* __restrict__ dst = j.dst[i];
  const long ls = j.ld_src[i], ldd = j.ld_dst[i];
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 2; ++k) {                                              // 64 rows x 8 chunks of 8 elements
    const int q = tid + 256 * k, r = q >> 3, ch = q & 7;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (r0 + r < rows && c0 + ch * 8 + 7 < cols) v = *reinterpret_cast<const uint4*>(src + (long)(r0 + r) * ls + c0 + ch * 8);
    else if (r0 + r < rows) {
      uint16_t e[8];
#pragma unroll
      for (int x = 0; x < 8; ++x) e[x] = (c0 + ch * 8 + x < cols) ? src[(long)(r0 + r) * ls + c0 + ch * 8 + x] : (uint16_t)0;
      v = make_uint4(e[0] | ((uint32_t)e[1] << 16), e[2] | ((uint32_t)e[3] << 16), e[4] | ((uint32_t)e[5] << 16), e[6] | ((uint32_t)e[7] << 16));
    }
    *reinterpret_cast<uint4*>(&tile[r][ch * 8]) = v;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {                                              // output row c (64), 8 chunks of 8 source rows
    const int q = tid + 256 * k, c = q >> 3, ch = q & 7;
    if (c0 + c >= cols) continue;
    uint16_t e[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) e[x] = tile[ch * 8 + x][c];
    uint16_t* d = dst + (long)(c0 + c) * ldd + r0 + ch * 8;
    if (r0 + ch * 8 + 7 < rows)
      *reinterpret_cast<uint4*>(d) = make_uint4(e[0] | ((uint32_t)e[1] << 16), e[2] | ((uint32_t)e[3] << 16),
                                                e[4] | ((uint32_t)e[5] << 16), e[6] | ((uint32_t)e[7] << 16));
    else
      for (int x = 0; x < 8; ++x)
        if (r0 + ch * 8 + x < rows) d[x] = e[x];
  }
}
// dst[i][c][r] = src[i][r][c] (bf16); rows of src and dst must start on 16 bytes (ld % 8 == 0, aligned bases)
extern "C" int tell_transpose_multi(int n, const void* const* src, const long* ld_src, void* const* dst, const long* ld_dst,
                                    const int* rows, const int* cols, hipStream_t stream) {
  for (int base = 0; base < n; base += MULTI_MAX) {
    TransposeJobs j;
    j.n = n - base < MULTI_MAX ? n - base : MULTI_MAX;
    j.start[0] = 0;
    for (int i = 0; i < j.n; ++i) {
      const int k = base + i;
      TELL_REQUIRE(rows[k] > 0 && cols[k] > 0 && ld_src[k] % 8 == 0 && ld_dst[k] % 8 == 0 &&
                   (((uintptr_t)src[k] | (uintptr_t)dst[k]) & 15) == 0, "transpose_multi: 16-byte aligned rows");
      j.src[i] = static_cast<const uint16_t*>(src[k]);
      j.dst[i] = static_cast<uint16_t*>(dst[k]);
      j.ld_src[i] = ld_src[k]; j.ld_dst[i] = ld_dst[k]; j.rows[i] = rows[k]; j.cols[i] = cols[k];
      j.tiles_x[i] = (cols[k] + 63) / 64;
      j.start[i + 1] = j.start[i] + j.tiles_x[i] * ((rows[k] + 63) / 64);
    }
    hipLaunchKernelGGL(transpose_multi_kernel, dim3(j.start[j.n]), dim3(256), 0, stream, j);
  }
  return tell_check_launch("transpose_multi");
}

// ------------------------------------------------------------------ out = add + dropout(x)
// The gradient of a block input that is both the residual and (through the input dropout) the branch input:
// d = d_residual + mask * d_branch / (1 - p) - one pass instead of a dropout launch and an add (decoder_faces_objects.py
// :256-266 backward).  Same hash, same indexing as tell_dropout (element index in the contiguous tensor).
template <typename T>
__global__ __launch_bounds__(256) void dropout_add_kernel(const T* __restrict__ x, const T* __restrict__ add, T* __restrict__ out,
                                                          long n, uint32_t thr, float inv_keep, uint32_t seed, uint32_t salt,
                                                          const uint32_t* __restrict__ step) {
  constexpr int VEC = Elem<T>::VEC;
  const uint32_t salt_eff = tell_step_salt(salt, step);
  const long nv = n / VEC;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    float v[VEC], a[VEC];
    unpack16(*reinterpret_cast<const uint4*>(x + i * VEC), v, (const T*)nullptr);
    unpack16(*reinterpret_cast<const uint4*>(add + i * VEC), a, (const T*)nullptr);
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = a[k] + v[k] * tell_keep(seed, salt_eff, (uint64_t)(i * VEC + k), thr, inv_keep);
    *reinterpret_cast<uint4*>(out + i * VEC) = pack16(v, (const T*)nullptr);
  }
}
extern "C" int tell_dropout_add(const void* x, const void* add, void* out, long n, float p, uint32_t seed, uint32_t salt,
                                int dtype, hipStream_t stream) {
  if (n <= 0) return TELL_OK;
  TELL_REQUIRE(p >= 0.f && p < 1.f, "dropout_add: p must be in [0,1)");
  const int vec = dtype == TELL_BF16 ? 8 : 4;
  TELL_REQUIRE(n % vec == 0 && (((uintptr_t)x | (uintptr_t)add | (uintptr_t)out) & 15) == 0,
               "dropout_add: 16-byte aligned buffers of whole 16-byte chunks");
  const uint32_t thr = p > 0.f ? tell_drop_threshold(p) : 0u;
  const float ik = 1.f / (1.f - p);
  long g = (n / vec + 255) / 256;
  if (g > 2048) g = 2048;
  if (dtype == TELL_BF16) hipLaunchKernelGGL((dropout_add_kernel<uint16_t>), dim3((unsigned)g), dim3(256), 0, stream, (const uint16_t*)x, (const uint16_t*)add, (uint16_t*)out, n, thr, ik, seed, salt, g_tell_rng_step);
  else hipLaunchKernelGGL((dropout_add_kernel<float>), dim3((unsigned)g), dim3(256), 0, stream, (const float*)x, (const float*)add, (float*)out, n, thr, ik, seed, salt, g_tell_rng_step);
  return tell_check_launch("dropout_add");
}
