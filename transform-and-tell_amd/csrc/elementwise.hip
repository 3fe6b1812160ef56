// HBM-bound helpers: cast / transpose (with optional per-row scale = weight-norm
// materialisation), weight-norm backward, GLU, dropout, column sums, row
// gather / scatter, NaN-row masking, RoBERTa layer mix.
// All loops are grid-stride over 16-byte chunks where the layout allows.
#include "common.h"
#include "gemm_common.h"    // gelu_erf2, pack2_bf16, u32x4

static inline int grid_for(long work, int per_block) {
  long g = (work + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > 256 * 16) g = 256 * 16;
  return (int)g;
}

// ---------------------------------------------------------------- cast
template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) Elem<D>::st(dst + i, Elem<S>::ld(src + i));
}

extern "C" int tell_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long n,
                         hipStream_t stream) {
  if (n <= 0) return TELL_OK;
  int g = grid_for(n, 256 * 4);
  if (src_dtype == TELL_F32 && dst_dtype == TELL_BF16)
    hipLaunchKernelGGL((cast_kernel<float, uint16_t>), dim3(g), dim3(256), 0, stream, (const float*)src, (uint16_t*)dst, n);
  else if (src_dtype == TELL_BF16 && dst_dtype == TELL_F32)
    hipLaunchKernelGGL((cast_kernel<uint16_t, float>), dim3(g), dim3(256), 0, stream, (const uint16_t*)src, (float*)dst, n);
  else if (src_dtype == TELL_F32 && dst_dtype == TELL_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), dim3(g), dim3(256), 0, stream, (const float*)src, (float*)dst, n);
  else if (src_dtype == TELL_BF16 && dst_dtype == TELL_BF16)
    hipLaunchKernelGGL((cast_kernel<uint16_t, uint16_t>), dim3(g), dim3(256), 0, stream, (const uint16_t*)src, (uint16_t*)dst, n);
  else { tell_set_error("cast: bad dtype"); return TELL_ERR_ARG; }
  return tell_check_launch("cast");
}

// dst[i] = (D)(src[i] * *scale): data-parallel gradient exchange - the fp32 gradient goes on the wire (bf16 or fp32,
// dst may alias src for fp32) weighted by this rank's share of the global token count (training/trainer.py)
template <typename D>
__global__ __launch_bounds__(256) void scale_cast_kernel(const float* __restrict__ src, D* __restrict__ dst, long n,
                                                         const float* __restrict__ scale) {
  const float sc = scale ? *scale : 1.f;
  const long nv = n / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    Elem<D>::st(dst + 4 * i, v.x * sc); Elem<D>::st(dst + 4 * i + 1, v.y * sc);
    Elem<D>::st(dst + 4 * i + 2, v.z * sc); Elem<D>::st(dst + 4 * i + 3, v.w * sc);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) Elem<D>::st(dst + nv * 4 + threadIdx.x, src[nv * 4 + threadIdx.x] * sc);
}
extern "C" int tell_scale_cast(const float* src, void* dst, int dst_dtype, long n, const float* scale_dev,
                               hipStream_t stream) {
  if (n <= 0) return TELL_OK;
  TELL_REQUIRE(((uintptr_t)src & 15) == 0, "scale_cast: src must be 16-byte aligned");
  int g = grid_for(n / 4 + 1, 256 * 2);
  if (dst_dtype == TELL_BF16) hipLaunchKernelGGL((scale_cast_kernel<uint16_t>), dim3(g), dim3(256), 0, stream, src, (uint16_t*)dst, n, scale_dev);
  else hipLaunchKernelGGL((scale_cast_kernel<float>), dim3(g), dim3(256), 0, stream, src, (float*)dst, n, scale_dev);
  return tell_check_launch("scale_cast");
}

// ---------------------------------------------------------------- transpose (+cast, +row scale)
// dst_t[c][r] = src[r][c] * row_scale[r];  optional dst_plain[r][c] = same value.
template <typename S, typename D>
__global__ __launch_bounds__(256) void transpose_kernel(const S* __restrict__ src, long ld_src,
                                                        D* __restrict__ dst_t, long ld_t,
                                                        D* __restrict__ dst_plain, long ld_p,
                                                        const float* __restrict__ row_scale,
                                                        int rows, int cols) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
  for (int i = ty; i < 64; i += 4) {
    int r = r0 + i, c = c0 + tx;
    float v = 0.f;
    if (r < rows && c < cols) {
      v = Elem<S>::ld(src + (long)r * ld_src + c);
      if (row_scale) v *= row_scale[r];
      if (dst_plain) Elem<D>::st(dst_plain + (long)r * ld_p + c, v);
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  if (dst_t) {
    for (int i = ty; i < 64; i += 4) {
      int c = c0 + i, r = r0 + tx;
      if (r < rows && c < cols) Elem<D>::st(dst_t + (long)c * ld_t + r, tile[tx][i]);
    }
  }
}

extern "C" int tell_transpose(const void* src, long ld_src, int src_dtype, void* dst_t, long ld_t,
                              void* dst_plain, long ld_p, int dst_dtype, const float* row_scale,
                              int rows, int cols, hipStream_t stream) {
  if (rows <= 0 || cols <= 0) return TELL_OK;
  dim3 grid((cols + 63) / 64, (rows + 63) / 64);
#define TL(S, D) hipLaunchKernelGGL((transpose_kernel<S, D>), grid, dim3(256), 0, stream, (const S*)src, ld_src, \
                                    (D*)dst_t, ld_t, (D*)dst_plain, ld_p, row_scale, rows, cols)
  if (src_dtype == TELL_F32 && dst_dtype == TELL_F32) TL(float, float);
  else if (src_dtype == TELL_F32 && dst_dtype == TELL_BF16) TL(float, uint16_t);
  else if (src_dtype == TELL_BF16 && dst_dtype == TELL_BF16) TL(uint16_t, uint16_t);
  else if (src_dtype == TELL_BF16 && dst_dtype == TELL_F32) TL(uint16_t, float);
  else { tell_set_error("transpose: bad dtype"); return TELL_ERR_ARG; }
#undef TL
  return tell_check_launch("transpose");
}

// ---------------------------------------------------------------- weight norm
// scale[r] = g[r] / ||v[r,:]||, norms[r] = ||v[r,:]||   (one wave per row)
__global__ __launch_bounds__(256) void wn_rowscale_kernel(const float* __restrict__ g,
                                                          const float* __restrict__ v, int rows,
                                                          int cols, float* __restrict__ scale,
                                                          float* __restrict__ norms) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* vr = v + (long)row * cols;
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) { float x = vr[c]; s += x * x; }
  s = wave_sum(s);
  if (lane == 0) {
    float nrm = sqrtf(s);
    norms[row] = nrm;
    scale[row] = g[row] / nrm;
  }
}

extern "C" int tell_wn_rowscale(const float* g, const float* v, int rows, int cols, float* scale,
                                float* norms, hipStream_t stream) {
  if (rows <= 0) return TELL_OK;
  hipLaunchKernelGGL(wn_rowscale_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, g, v, rows, cols, scale, norms);
  return tell_check_launch("wn_rowscale");
}

// Working weight of a GehringLinear in ONE pass: w[r,:] = (g[r] / ||v[r,:]||) * v[r,:] in the compute dtype, plus
// norms[r] = ||v[r,:]|| for the backward.  One wave per row; the second sweep over the row hits L2.
template <typename OutT>
__device__ __forceinline__ void wn_weight_row(const float* __restrict__ g, const float* __restrict__ v, int row,
                                              int cols, OutT* __restrict__ w, float* __restrict__ norms, int lane) {
  const float* vr = v + (long)row * cols;
  OutT* wr = w + (long)row * cols;
  float s = 0.f;
  typedef __attribute__((ext_vector_type(4))) float f4v;
  if ((cols & 255) == 0 && cols <= 4096) {
    // the row lives in registers (<= 16 float4 per lane): ONE pass over v with every load in flight before the first use
    // (round 5: the two-pass form read the row, reduced, and read it again - 3.4 TB/s on the 372 MB of the decoder's 20
    // GehringLinears)
    const int nv = cols >> 8;
    f4v x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < nv) x[j] = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(vr) + lane + 64 * j);
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < nv) s += x[j][0] * x[j][0] + x[j][1] * x[j][1] + x[j][2] * x[j][2] + x[j][3] * x[j][3];
    s = wave_sum(s);
    const float nrm = sqrtf(s), sc = g[row] / nrm;
    if (lane == 0) norms[row] = nrm;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < nv) {
        const int c = lane + 64 * j;
        if constexpr (sizeof(OutT) == 2) {
          uint2 o;
          o.x = (uint32_t)f2bf(x[j][0] * sc) | ((uint32_t)f2bf(x[j][1] * sc) << 16);
          o.y = (uint32_t)f2bf(x[j][2] * sc) | ((uint32_t)f2bf(x[j][3] * sc) << 16);
          reinterpret_cast<uint2*>(wr)[c] = o;
        } else {
          reinterpret_cast<float4*>(wr)[c] = make_float4(x[j][0] * sc, x[j][1] * sc, x[j][2] * sc, x[j][3] * sc);
        }
      }
    return;
  }
  if ((cols & 3) == 0) {
    const float4* v4 = reinterpret_cast<const float4*>(vr);
    for (int c = lane; c < cols / 4; c += 64) { const float4 x = v4[c]; s += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w; }
  } else {
    for (int c = lane; c < cols; c += 64) { const float x = vr[c]; s += x * x; }
  }
  s = wave_sum(s);
  const float nrm = sqrtf(s), sc = g[row] / nrm;
  if (lane == 0) norms[row] = nrm;
  if ((cols & 3) == 0) {
    const float4* v4 = reinterpret_cast<const float4*>(vr);
    for (int c = lane; c < cols / 4; c += 64) {
      const float4 x = v4[c];
      if constexpr (sizeof(OutT) == 2) {
        uint2 o;
        o.x = (uint32_t)f2bf(x.x * sc) | ((uint32_t)f2bf(x.y * sc) << 16);
        o.y = (uint32_t)f2bf(x.z * sc) | ((uint32_t)f2bf(x.w * sc) << 16);
        reinterpret_cast<uint2*>(wr)[c] = o;
      } else {
        reinterpret_cast<float4*>(wr)[c] = make_float4(x.x * sc, x.y * sc, x.z * sc, x.w * sc);
      }
    }
  } else {
    for (int c = lane; c < cols; c += 64) Elem<OutT>::st(wr + c, vr[c] * sc);
  }
}
template <typename OutT>
__global__ __launch_bounds__(256) void wn_weight_kernel(const float* __restrict__ g, const float* __restrict__ v,
                                                        int rows, int cols, OutT* __restrict__ w,
                                                        float* __restrict__ norms) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row < rows) wn_weight_row<OutT>(g, v, row, cols, w, norms, threadIdx.x & 63);
}
extern "C" int tell_wn_weight(const float* g, const float* v, int rows, int cols, void* w, int out_dtype,
                              float* norms, hipStream_t stream) {
  if (rows <= 0) return TELL_OK;
  TELL_REQUIRE(((uintptr_t)v & 15) == 0 && ((uintptr_t)w & 15) == 0, "wn_weight: buffers must be 16-byte aligned");
  if (out_dtype == TELL_BF16)
    hipLaunchKernelGGL((wn_weight_kernel<uint16_t>), dim3((rows + 3) / 4), dim3(256), 0, stream, g, v, rows, cols, (uint16_t*)w, norms);
  else
    hipLaunchKernelGGL((wn_weight_kernel<float>), dim3((rows + 3) / 4), dim3(256), 0, stream, g, v, rows, cols, (float*)w, norms);
  return tell_check_launch("wn_weight");
}

// dg[r] += <dW[r], v[r]> / ||v||;  dv[r] += (g/||v||) * (dW[r] - v[r] * <dW[r],v[r]> / ||v||^2)
__device__ __forceinline__ void wn_backward_row(const float* __restrict__ dW, const float* __restrict__ g,
                                                const float* __restrict__ v, const float* __restrict__ norms,
                                                int row, int cols, float* __restrict__ dg, float* __restrict__ dv,
                                                int lane, const bool store = false) {
  const float* vr = v + (long)row * cols;
  const float* dr = dW + (long)row * cols;
  float* o = dv + (long)row * cols;
  const bool vec = (cols & 3) == 0 && (((uintptr_t)dW | (uintptr_t)v | (uintptr_t)dv) & 15) == 0;
  float dot = 0.f;
  if (vec && (cols & 255) == 0 && cols <= 4096) {
    // dW and v rows in registers (<= 2 x 16 float4 per lane): one pass, every load in flight before the dot product
    // (round 5; the two-pass form read both rows twice).  dW is dead after this launch: non-temporal.
    typedef __attribute__((ext_vector_type(4))) float f4v;
    const int nv = cols >> 8;
    f4v d[16], x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < nv) {
        d[j] = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(dr) + lane + 64 * j);
        x[j] = reinterpret_cast<const f4v*>(vr)[lane + 64 * j];
      }
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < nv) dot += d[j][0] * x[j][0] + d[j][1] * x[j][1] + d[j][2] * x[j][2] + d[j][3] * x[j][3];
    dot = wave_sum(dot);
    const float nrm = norms[row], gs = g[row] / nrm, k = dot / (nrm * nrm);
    if (lane == 0) dg[row] = (store ? 0.f : dg[row]) + dot / nrm;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < nv) {
        const int c = lane + 64 * j;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!store) a = reinterpret_cast<float4*>(o)[c];
        a.x += gs * (d[j][0] - x[j][0] * k); a.y += gs * (d[j][1] - x[j][1] * k);
        a.z += gs * (d[j][2] - x[j][2] * k); a.w += gs * (d[j][3] - x[j][3] * k);
        reinterpret_cast<float4*>(o)[c] = a;
      }
    return;
  }
  if (vec) {
    for (int c = lane; c < cols / 4; c += 64) {
      const float4 d = reinterpret_cast<const float4*>(dr)[c], x = reinterpret_cast<const float4*>(vr)[c];
      dot += d.x * x.x + d.y * x.y + d.z * x.z + d.w * x.w;
    }
  } else {
    for (int c = lane; c < cols; c += 64) dot += dr[c] * vr[c];
  }
  dot = wave_sum(dot);
  const float nrm = norms[row], gs = g[row] / nrm, k = dot / (nrm * nrm);
  // accumulate into the (zeroed) grad buffers - or, `store` (wave-uniform): this launch is the only writer of dg / dv in
  // the step and the optimizer left them alone (tell_bertadam_step2 keep_grad): no read of the old value
  if (lane == 0) dg[row] = (store ? 0.f : dg[row]) + dot / nrm;
  if (vec) {
    for (int c = lane; c < cols / 4; c += 64) {
      const float4 d = reinterpret_cast<const float4*>(dr)[c], x = reinterpret_cast<const float4*>(vr)[c];
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!store) a = reinterpret_cast<float4*>(o)[c];
      a.x += gs * (d.x - x.x * k); a.y += gs * (d.y - x.y * k); a.z += gs * (d.z - x.z * k); a.w += gs * (d.w - x.w * k);
      reinterpret_cast<float4*>(o)[c] = a;
    }
  } else {
    for (int c = lane; c < cols; c += 64) o[c] = (store ? 0.f : o[c]) + gs * (dr[c] - vr[c] * k);
  }
}
__global__ __launch_bounds__(256) void wn_backward_kernel(const float* __restrict__ dW,
                                                          const float* __restrict__ g,
                                                          const float* __restrict__ v,
                                                          const float* __restrict__ norms, int rows,
                                                          int cols, float* __restrict__ dg,
                                                          float* __restrict__ dv) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row < rows) wn_backward_row(dW, g, v, norms, row, cols, dg, dv, threadIdx.x & 63);
}

extern "C" int tell_wn_backward(const float* dW, const float* g, const float* v, const float* norms,
                                int rows, int cols, float* dg, float* dv, hipStream_t stream) {
  if (rows <= 0) return TELL_OK;
  hipLaunchKernelGGL(wn_backward_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, dW, g, v, norms, rows, cols, dg, dv);
  return tell_check_launch("wn_backward");
}

// The same two row kernels over MANY GehringLinears in one launch (the decoder has 20: at 10-15 us a launch for 1-4 M
// elements each they are latency, not bandwidth).  The per-tensor pointers travel BY VALUE in the kernel arguments
// (no device table to keep alive, and a captured graph bakes them in); a wave finds its tensor by walking the row
// prefix sums, which sit in scalar registers.
#define WN_MULTI_MAX 32
struct WNMulti {
  const float* a[WN_MULTI_MAX];        // forward: unused          backward: dW
  const float* g[WN_MULTI_MAX];
  const float* v[WN_MULTI_MAX];
  void* w[WN_MULTI_MAX];               // forward: working weight  backward: dv
  float* norms[WN_MULTI_MAX];
  float* dg[WN_MULTI_MAX];
  int cols[WN_MULTI_MAX];
  int row_start[WN_MULTI_MAX + 1];
  int n;
  unsigned store;                      // backward: bit i = tensor i's dg / dv are stored, not accumulated
};
template <typename OutT, bool BWD>
__global__ __launch_bounds__(256) void wn_multi_kernel(WNMulti m) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= m.row_start[m.n]) return;
  int i = 0;
  while (i + 1 < m.n && row >= m.row_start[i + 1]) ++i;       // wave-uniform
  const int r = row - m.row_start[i];
  if constexpr (BWD) wn_backward_row(m.a[i], m.g[i], m.v[i], m.norms[i], r, m.cols[i], m.dg[i], static_cast<float*>(m.w[i]), lane, (m.store >> i) & 1u);
  else wn_weight_row<OutT>(m.g[i], m.v[i], r, m.cols[i], static_cast<OutT*>(m.w[i]), m.norms[i], lane);
}
template <bool BWD>
static int wn_multi_launch(int n, const void* const* a, const void* const* g, const void* const* v, void* const* w,
                           void* const* norms, void* const* dg, const int* rows, const int* cols, int out_dtype,
                           hipStream_t stream, const int* store = nullptr) {
  static_assert(WN_MULTI_MAX <= 32, "WNMulti::store is one bit per tensor");
  for (int base = 0; base < n; base += WN_MULTI_MAX) {
    WNMulti m;
    m.n = n - base < WN_MULTI_MAX ? n - base : WN_MULTI_MAX;
    m.row_start[0] = 0;
    m.store = 0u;
    for (int i = 0; i < m.n; ++i) {
      const int j = base + i;
      TELL_REQUIRE(rows[j] > 0 && cols[j] > 0, "wn_multi: empty tensor");
      TELL_REQUIRE(BWD || (((uintptr_t)v[j] | (uintptr_t)w[j]) & 15) == 0, "wn_weight_multi: buffers must be 16-byte aligned");
      m.a[i] = BWD ? static_cast<const float*>(a[j]) : nullptr;
      m.g[i] = static_cast<const float*>(g[j]);
      m.v[i] = static_cast<const float*>(v[j]);
      m.w[i] = w[j];
      m.norms[i] = static_cast<float*>(norms[j]);
      m.dg[i] = BWD ? static_cast<float*>(dg[j]) : nullptr;
      m.cols[i] = cols[j];
      m.row_start[i + 1] = m.row_start[i] + rows[j];
      if (BWD && store && store[j]) m.store |= 1u << i;
    }
    const dim3 grid((m.row_start[m.n] + 3) / 4);
    if (BWD) hipLaunchKernelGGL((wn_multi_kernel<float, true>), grid, dim3(256), 0, stream, m);
    else if (out_dtype == TELL_BF16) hipLaunchKernelGGL((wn_multi_kernel<uint16_t, false>), grid, dim3(256), 0, stream, m);
    else hipLaunchKernelGGL((wn_multi_kernel<float, false>), grid, dim3(256), 0, stream, m);
  }
  return tell_check_launch(BWD ? "wn_backward_multi" : "wn_weight_multi");
}
// host arrays of n device pointers / sizes
extern "C" int tell_wn_weight_multi(int n, const void* const* g, const void* const* v, void* const* w,
                                    void* const* norms, const int* rows, const int* cols, int out_dtype,
                                    hipStream_t stream) {
  if (n <= 0) return TELL_OK;
  return wn_multi_launch<false>(n, nullptr, g, v, w, norms, nullptr, rows, cols, out_dtype, stream);
}
extern "C" int tell_wn_backward_multi(int n, const void* const* dW, const void* const* g, const void* const* v,
                                      const void* const* norms, const int* rows, const int* cols, void* const* dg,
                                      void* const* dv, hipStream_t stream) {
  if (n <= 0) return TELL_OK;
  return wn_multi_launch<true>(n, dW, g, v, dv, const_cast<void* const*>(norms), dg, rows, cols, TELL_F32, stream);
}
// store: host int[n] or NULL - tensor j's dg / dv are WRITTEN (beta = 0) instead of accumulated: for gradients that have
// this launch as their only writer in a step and that the optimizer does not zero (tell_bertadam_step2 keep_grad)
extern "C" int tell_wn_backward_multi2(int n, const void* const* dW, const void* const* g, const void* const* v,
                                       const void* const* norms, const int* rows, const int* cols, void* const* dg,
                                       void* const* dv, const int* store, hipStream_t stream) {
  if (n <= 0) return TELL_OK;
  return wn_multi_launch<true>(n, dW, g, v, dv, const_cast<void* const*>(norms), dg, rows, cols, TELL_F32, stream, store);
}

// ---------------------------------------------------------------- GLU  (h = [a | gate], y = a * sigmoid(gate))
template <typename T>
__global__ void glu_fwd_kernel(const T* __restrict__ h, T* __restrict__ y, long rows, int C) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = rows * C, stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    long r = i / C; int c = (int)(i % C);
    float a = Elem<T>::ld(h + r * 2 * C + c), g = Elem<T>::ld(h + r * 2 * C + C + c);
    Elem<T>::st(y + i, tell_glu(a, g));
  }
}
template <typename T>
__global__ void glu_bwd_kernel(const T* __restrict__ h, const T* __restrict__ dy, T* __restrict__ dh,
                               long rows, int C) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = rows * C, stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    long r = i / C; int c = (int)(i % C);
    float a = Elem<T>::ld(h + r * 2 * C + c), g = Elem<T>::ld(h + r * 2 * C + C + c);
    float s = 1.f / (1.f + __expf(-g)), d = Elem<T>::ld(dy + i);
    Elem<T>::st(dh + r * 2 * C + c, d * s);
    Elem<T>::st(dh + r * 2 * C + C + c, d * a * s * (1.f - s));
  }
}
extern "C" int tell_glu_fwd(const void* h, void* y, long rows, int C, int dtype, hipStream_t stream) {
  if (rows <= 0) return TELL_OK;
  int g = grid_for(rows * C, 256 * 2);
  if (dtype == TELL_BF16) hipLaunchKernelGGL((glu_fwd_kernel<uint16_t>), dim3(g), dim3(256), 0, stream, (const uint16_t*)h, (uint16_t*)y, rows, C);
  else hipLaunchKernelGGL((glu_fwd_kernel<float>), dim3(g), dim3(256), 0, stream, (const float*)h, (float*)y, rows, C);
  return tell_check_launch("glu_fwd");
}
extern "C" int tell_glu_bwd(const void* h, const void* dy, void* dh, long rows, int C, int dtype,
                            hipStream_t stream) {
  if (rows <= 0) return TELL_OK;
  int g = grid_for(rows * C, 256 * 2);
  if (dtype == TELL_BF16) hipLaunchKernelGGL((glu_bwd_kernel<uint16_t>), dim3(g), dim3(256), 0, stream, (const uint16_t*)h, (const uint16_t*)dy, (uint16_t*)dh, rows, C);
  else hipLaunchKernelGGL((glu_bwd_kernel<float>), dim3(g), dim3(256), 0, stream, (const float*)h, (const float*)dy, (float*)dh, rows, C);
  return tell_check_launch("glu_bwd");
}

// ---------------------------------------------------------------- dropout  y = x * keep(idx)
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, long n, uint32_t thr,
                               float inv_keep, uint32_t seed, uint32_t salt,
                               const uint32_t* __restrict__ step) {
  salt = tell_step_salt(salt, step);
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride)
    Elem<T>::st(y + i, Elem<T>::ld(x + i) * tell_keep(seed, salt, (uint64_t)i, thr, inv_keep));
}
extern "C" int tell_dropout(const void* x, void* y, long n, float p, uint32_t seed, uint32_t salt,
                            int dtype, hipStream_t stream) {
  if (n <= 0) return TELL_OK;
  TELL_REQUIRE(p >= 0.f && p < 1.f, "dropout: p must be in [0,1)");
  int g = grid_for(n, 256 * 4);
  uint32_t thr = tell_drop_threshold(p);
  float ik = 1.f / (1.f - p);
  if (dtype == TELL_BF16) hipLaunchKernelGGL((dropout_kernel<uint16_t>), dim3(g), dim3(256), 0, stream, (const uint16_t*)x, (uint16_t*)y, n, thr, ik, seed, salt, g_tell_rng_step);
  else hipLaunchKernelGGL((dropout_kernel<float>), dim3(g), dim3(256), 0, stream, (const float*)x, (float*)y, n, thr, ik, seed, salt, g_tell_rng_step);
  return tell_check_launch("dropout");
}

// ---------------------------------------------------------------- exact-erf GELU as its own launch (bf16, in place or not)
// y = gelu(x), 16 bytes per thread and trip.  Same formula as the GEMM epilogue's act 2 (gemm_common.h gelu_erf2); used
// where the activation is kept OUT of the dominant GEMM's epilogue: that kernel owns every CU it runs on, so arithmetic in
// its epilogue is paid with idle matrix cores, while this launch is bandwidth-bound and shares the chip.
__global__ __launch_bounds__(256) void gelu_vec_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, long nv) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(x + i * 8);
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x2e_t r = gelu_erf2(f32x2e_t{__uint_as_float(v[q] << 16), __uint_as_float(v[q] & 0xffff0000u)});
      o[q] = pack2_bf16(r[0], r[1]);
    }
    *reinterpret_cast<u32x4*>(y + i * 8) = o;
  }
}
extern "C" int tell_gelu(const void* x, void* y, long n, int dtype, hipStream_t stream) {
  if (n <= 0) return TELL_OK;
  TELL_REQUIRE(dtype == TELL_BF16 && n % 8 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0,
               "gelu: bf16, a multiple of 8 elements, 16-byte aligned buffers");
  const long nv = n / 8;
  const int g = (int)((nv + 255) / 256 > 16384 ? 16384 : (nv + 255) / 256);
  hipLaunchKernelGGL(gelu_vec_kernel, dim3(g), dim3(256), 0, stream, (const uint16_t*)x, (uint16_t*)y, nv);
  return tell_check_launch("gelu");
}

// ---------------------------------------------------------------- column sums (bias grads): out[c] (+)= scale * sum_r x[r][c]
// stage 1: grid (C/64, n_chunks); 256 threads = 64 columns x 4 row lanes; stage 2 sums the chunk partials
// in a fixed order (deterministic, no atomics).
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, long ld, int rows, int C,
                                                             int rows_per_chunk, float* __restrict__ partial,
                                                             const int* __restrict__ m_dev, float* __restrict__ direct,
                                                             int accumulate, float scale) {
  __shared__ float part[4][65];
  if (m_dev) { int md = *m_dev; rows = md < rows ? md : rows; }
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  const int r0 = blockIdx.y * rows_per_chunk;
  const int r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
  float s = 0.f;
  if (c < C)
    for (int r = r0 + ry; r < r1; r += 4) s += Elem<T>::ld(x + (long)r * ld + c);
  part[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && c < C) {
    const float t = part[0][cx] + part[1][cx] + part[2][cx] + part[3][cx];
    if (direct) direct[c] = (accumulate ? direct[c] : 0.f) + scale * t;      // single chunk: no second pass
    else partial[(long)blockIdx.y * C + c] = t;
  }
}
__global__ void colsum_finish_kernel(const float* __restrict__ partial, int n_chunks, int C, float* __restrict__ out,
                                     int accumulate, float scale) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float t = 0.f;
  for (int k = 0; k < n_chunks; ++k) t += partial[(long)k * C + c];
  t *= scale;
  out[c] = accumulate ? out[c] + t : t;
}
extern "C" int tell_colsum_chunks(int rows) { int n = (rows + 127) / 128; return n < 1 ? 1 : (n > 128 ? 128 : n); }
// workspace: tell_colsum_chunks(rows) * C floats
extern "C" int tell_colsum(const void* x, long ld, int rows, int C, int dtype, float* out,
                           int accumulate, const int* m_dev, float scale, float* workspace, hipStream_t stream) {
  if (C <= 0) return TELL_OK;
  const int nch = tell_colsum_chunks(rows);
  const int rpc = (rows + nch - 1) / nch;
  dim3 grid((C + 63) / 64, nch);
  float* direct = nch == 1 ? out : nullptr;
  if (dtype == TELL_BF16) hipLaunchKernelGGL((colsum_partial_kernel<uint16_t>), grid, dim3(256), 0, stream, (const uint16_t*)x, ld, rows, C, rpc, workspace, m_dev, direct, accumulate, scale);
  else hipLaunchKernelGGL((colsum_partial_kernel<float>), grid, dim3(256), 0, stream, (const float*)x, ld, rows, C, rpc, workspace, m_dev, direct, accumulate, scale);
  if (!direct) hipLaunchKernelGGL(colsum_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, workspace, nch, C, out, accumulate, scale);
  return tell_check_launch("colsum");
}

// ---------------------------------------------------------------- row gather / scatter-add (compacted cluster rows)
template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ src, long ld_src, const int* __restrict__ idx,
                                   const int* __restrict__ count_dev, int cap, T* __restrict__ dst,
                                   long ld_dst, int C) {
  int n = count_dev ? *count_dev : cap;
  if (n > cap) n = cap;
  for (int r = blockIdx.x; r < n; r += gridDim.x) {
    const T* s = src + (long)idx[r] * ld_src;
    T* d = dst + (long)r * ld_dst;
    for (int c = threadIdx.x; c < C; c += blockDim.x) d[c] = s[c];
  }
}
// dst[idx[r]] += src[r] * scale   (indices unique -> no atomics)
template <typename T>
__global__ void scatter_add_rows_kernel(const T* __restrict__ src, long ld_src, const int* __restrict__ idx,
                                        const int* __restrict__ count_dev, int cap, T* __restrict__ dst,
                                        long ld_dst, int C) {
  int n = count_dev ? *count_dev : cap;
  if (n > cap) n = cap;
  for (int r = blockIdx.x; r < n; r += gridDim.x) {
    const T* s = src + (long)r * ld_src;
    T* d = dst + (long)idx[r] * ld_dst;
    for (int c = threadIdx.x; c < C; c += blockDim.x)
      Elem<T>::st(d + c, Elem<T>::ld(d + c) + Elem<T>::ld(s + c));
  }
}
extern "C" int tell_gather_rows(const void* src, long ld_src, const int* idx, const int* count_dev,
                                int cap, void* dst, long ld_dst, int C, int dtype, hipStream_t stream) {
  if (cap <= 0) return TELL_OK;
  int g = cap < 2048 ? cap : 2048;
  if (dtype == TELL_BF16) hipLaunchKernelGGL((gather_rows_kernel<uint16_t>), dim3(g), dim3(256), 0, stream, (const uint16_t*)src, ld_src, idx, count_dev, cap, (uint16_t*)dst, ld_dst, C);
  else hipLaunchKernelGGL((gather_rows_kernel<float>), dim3(g), dim3(256), 0, stream, (const float*)src, ld_src, idx, count_dev, cap, (float*)dst, ld_dst, C);
  return tell_check_launch("gather_rows");
}
extern "C" int tell_scatter_add_rows(const void* src, long ld_src, const int* idx, const int* count_dev,
                                     int cap, void* dst, long ld_dst, int C, int dtype,
                                     hipStream_t stream) {
  if (cap <= 0) return TELL_OK;
  int g = cap < 2048 ? cap : 2048;
  if (dtype == TELL_BF16) hipLaunchKernelGGL((scatter_add_rows_kernel<uint16_t>), dim3(g), dim3(256), 0, stream, (const uint16_t*)src, ld_src, idx, count_dev, cap, (uint16_t*)dst, ld_dst, C);
  else hipLaunchKernelGGL((scatter_add_rows_kernel<float>), dim3(g), dim3(256), 0, stream, (const float*)src, ld_src, idx, count_dev, cap, (float*)dst, ld_dst, C);
  return tell_check_launch("scatter_add_rows");
}

// ---------------------------------------------------------------- NaN-padded rows -> mask + zeros (+cast)
// transformer_faces_objects.py:373-379: mask[r] = any(isnan(x[r,:])); x[r,:] = 0 where masked.
template <typename D>
__global__ __launch_bounds__(256) void nan_rows_kernel(const float* __restrict__ x, int rows, int C,
                                                       D* __restrict__ y, uint8_t* __restrict__ mask) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long)row * C;
  int bad = 0;
  for (int c = lane; c < C; c += 64) bad |= (xr[c] != xr[c]);
  bad = __any(bad);
  if (lane == 0) mask[row] = bad ? 1 : 0;
  D* yr = y + (long)row * C;
  for (int c = lane; c < C; c += 64) Elem<D>::st(yr + c, bad ? 0.f : xr[c]);
}
extern "C" int tell_nan_rows(const float* x, int rows, int C, void* y, int out_dtype, uint8_t* mask,
                             hipStream_t stream) {
  if (rows <= 0) return TELL_OK;
  dim3 grid((rows + 3) / 4);
  if (out_dtype == TELL_BF16) hipLaunchKernelGGL((nan_rows_kernel<uint16_t>), grid, dim3(256), 0, stream, x, rows, C, (uint16_t*)y, mask);
  else hipLaunchKernelGGL((nan_rows_kernel<float>), grid, dim3(256), 0, stream, x, rows, C, (float*)y, mask);
  return tell_check_launch("nan_rows");
}

// ---------------------------------------------------------------- RoBERTa layer mix (transformer_faces_objects.py:355-364)
// out[i] = sum_l softmax(w)[l] * H[l][i];   H is a contiguous stack [L][n]
template <typename T>
__global__ __launch_bounds__(256) void mix_fwd_kernel(const T* __restrict__ H, const float* __restrict__ w,
                                                      int L, long n, T* __restrict__ out) {
  __shared__ float sw[64];
  if (threadIdx.x < 64) {
    float v = threadIdx.x < L ? w[threadIdx.x] : -INFINITY;
    float m = wave_max(v);
    float e = threadIdx.x < L ? __expf(v - m) : 0.f;
    float s = wave_sum(e);
    sw[threadIdx.x] = e / s;
  }
  __syncthreads();
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float acc = 0.f;
    for (int l = 0; l < L; ++l) acc += sw[l] * Elem<T>::ld(H + (long)l * n + i);
    Elem<T>::st(out + i, acc);
  }
}
// partial[b][l] = sum over this block's elements of dOut[i] * H[l][i]
template <typename T>
__global__ __launch_bounds__(256) void mix_bwd_kernel(const T* __restrict__ H, const T* __restrict__ dOut,
                                                      int L, long n, float* __restrict__ partial) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int l = 0; l < L; ++l) {
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
      acc += Elem<T>::ld(dOut + i) * Elem<T>::ld(H + (long)l * n + i);
    acc = wave_sum(acc);
    if (lane == 0) red[wave][l] = acc;
  }
  __syncthreads();
  if (threadIdx.x < L)
    partial[(long)blockIdx.x * L + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
// 16-byte-chunk variants (L <= 32): every thread owns one chunk position and walks the L layers, so the stack is
// streamed exactly once with 16 B per lane (HBM-bound: L*n elements read, n written / L*n + n read).
template <typename T>
__global__ __launch_bounds__(256) void mix_fwd_vec_kernel(const T* __restrict__ H, const float* __restrict__ w,
                                                          int L, long n, T* __restrict__ out) {
  constexpr int VEC = Elem<T>::VEC;
  __shared__ float sw[64];
  if (threadIdx.x < 64) {
    float v = threadIdx.x < L ? w[threadIdx.x] : -INFINITY;
    float m = wave_max(v);
    float e = threadIdx.x < L ? __expf(v - m) : 0.f;
    float s = wave_sum(e);
    sw[threadIdx.x] = e / s;
  }
  __syncthreads();
  const long nv = n / VEC;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    // every layer's 16 bytes requested before the first use (round 5: five at a time - `#pragma unroll 5` - ran the
    // 840 MB stack at 3.9 TB/s where the backward kernel, whose loop is unrolled whole, reads the same stack at 4.8);
    // the stack is read once per step: non-temporal
    typedef __attribute__((ext_vector_type(4))) unsigned int u4;
    u4 hv[32];
#pragma unroll
    for (int l = 0; l < 32; ++l)
      if (l < L) hv[l] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(H + (long)l * n + i * VEC));
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
#pragma unroll
    for (int l = 0; l < 32; ++l)
      if (l < L) {
        float v[VEC];
        unpack16(make_uint4(hv[l][0], hv[l][1], hv[l][2], hv[l][3]), v, (const T*)nullptr);
        const float wl = sw[l];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] += wl * v[k];
      }
    *reinterpret_cast<uint4*>(out + i * VEC) = pack16(acc, (const T*)nullptr);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void mix_bwd_vec_kernel(const T* __restrict__ H, const T* __restrict__ dOut,
                                                          int L, long n, float* __restrict__ partial) {
  constexpr int VEC = Elem<T>::VEC;
  __shared__ float red[4][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[32];
#pragma unroll
  for (int l = 0; l < 32; ++l) acc[l] = 0.f;
  const long nv = n / VEC;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    float d[VEC];
    unpack16(*reinterpret_cast<const uint4*>(dOut + i * VEC), d, (const T*)nullptr);
#pragma unroll
    for (int l = 0; l < 32; ++l) {
      if (l < L) {
        float v[VEC];
        unpack16(*reinterpret_cast<const uint4*>(H + (long)l * n + i * VEC), v, (const T*)nullptr);
        float a = acc[l];
#pragma unroll
        for (int k = 0; k < VEC; ++k) a += d[k] * v[k];
        acc[l] = a;
      }
    }
  }
#pragma unroll
  for (int l = 0; l < 32; ++l) {
    const float a = wave_sum(acc[l]);
    if (lane == 0) red[wave][l] = a;
  }
  __syncthreads();
  if (threadIdx.x < L)
    partial[(long)blockIdx.x * L + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
extern "C" int tell_mix_fwd(const void* H, const float* w, int L, long n, void* out, int dtype,
                            hipStream_t stream) {
  TELL_REQUIRE(L >= 1 && L <= 64, "mix_fwd: L must be in [1,64]");
  if (L <= 32 && n % 8 == 0 && (((uintptr_t)H | (uintptr_t)out) & 15) == 0 && dtype == TELL_BF16) {
    hipLaunchKernelGGL((mix_fwd_vec_kernel<uint16_t>), dim3(grid_for(n / 8, 256)), dim3(256), 0, stream, (const uint16_t*)H, w, L, n, (uint16_t*)out);
    return tell_check_launch("mix_fwd_vec");
  }
  int g = grid_for(n, 256 * 4);
  if (dtype == TELL_BF16) hipLaunchKernelGGL((mix_fwd_kernel<uint16_t>), dim3(g), dim3(256), 0, stream, (const uint16_t*)H, w, L, n, (uint16_t*)out);
  else hipLaunchKernelGGL((mix_fwd_kernel<float>), dim3(g), dim3(256), 0, stream, (const float*)H, w, L, n, (float*)out);
  return tell_check_launch("mix_fwd");
}
// partial must hold n_blocks*L floats; returns n_blocks through *n_blocks_out
extern "C" int tell_mix_bwd(const void* H, const void* dOut, int L, long n, float* partial,
                            int n_blocks, int dtype, hipStream_t stream) {
  TELL_REQUIRE(L >= 1 && L <= 64, "mix_bwd: L must be in [1,64]");
  TELL_REQUIRE(n_blocks >= 1, "mix_bwd: n_blocks");
  if (L <= 32 && n % 8 == 0 && (((uintptr_t)H | (uintptr_t)dOut) & 15) == 0 && dtype == TELL_BF16) {
    hipLaunchKernelGGL((mix_bwd_vec_kernel<uint16_t>), dim3(n_blocks), dim3(256), 0, stream, (const uint16_t*)H, (const uint16_t*)dOut, L, n, partial);
    return tell_check_launch("mix_bwd_vec");
  }
  if (dtype == TELL_BF16) hipLaunchKernelGGL((mix_bwd_kernel<uint16_t>), dim3(n_blocks), dim3(256), 0, stream, (const uint16_t*)H, (const uint16_t*)dOut, L, n, partial);
  else hipLaunchKernelGGL((mix_bwd_kernel<float>), dim3(n_blocks), dim3(256), 0, stream, (const float*)H, (const float*)dOut, L, n, partial);
  return tell_check_launch("mix_bwd");
}

// gw[l] += sm[l] * (d[l] - sum_j sm[j] d[j]),  sm = softmax(w), d[l] = sum over the n_blocks partial rows of tell_mix_bwd:
// the gradient of the 25 mixing logits (transformer_faces_objects.py:355-364 backward), one workgroup.
__global__ __launch_bounds__(1024) void mix_wgrad_kernel(const float* __restrict__ partial, int n_blocks, int L,
                                                         const float* __restrict__ w, float* __restrict__ gw) {
  __shared__ float part[16][64];
  __shared__ float d[64], sm[64];
  const int t = threadIdx.x, l = t & 63, g = t >> 6;              // 16 row groups x 64 columns (coalesced over l)
  float s = 0.f;
  if (l < L)
    for (int b = g; b < n_blocks; b += 16) s += partial[(long)b * L + l];
  part[g][l] = s;
  __syncthreads();
  if (t < 64) {
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) v += part[q][t];
    d[t] = v;
  }
  float mx = -INFINITY;
  for (int j = 0; j < L; ++j) mx = fmaxf(mx, w[j]);
  float den = 0.f;
  for (int j = 0; j < L; ++j) den += __expf(w[j] - mx);
  if (t < L) sm[t] = __expf(w[t] - mx) / den;
  __syncthreads();
  float dot = 0.f;
  for (int j = 0; j < L; ++j) dot += sm[j] * d[j];
  if (t < L) gw[t] += sm[t] * (d[t] - dot);
}
extern "C" int tell_mix_wgrad(const float* partial, int n_blocks, int L, const float* w, float* gw, hipStream_t stream) {
  TELL_REQUIRE(L >= 1 && L <= 64 && n_blocks >= 1, "mix_wgrad: L must be in [1,64]");
  hipLaunchKernelGGL(mix_wgrad_kernel, dim3(1), dim3(1024), 0, stream, partial, n_blocks, L, w, gw);
  return tell_check_launch("mix_wgrad");
}

// out = x / (ln 2 * n): the summed cross entropy in nats -> bits per target token (transformer_faces_objects.py:85-88),
// and - same arithmetic - the gradient of that sum from the gradient of the loss.
__global__ void loss_bits_kernel(const float* __restrict__ x, const int* __restrict__ n, float* __restrict__ out) {
  out[0] = x[0] / 0.69314718055994531f / (float)n[0];
}
extern "C" int tell_loss_bits(const float* x, const int* n_valid, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(loss_bits_kernel, dim3(1), dim3(1), 0, stream, x, n_valid, out);
  return tell_check_launch("loss_bits");
}

// ---------------------------------------------------------------- y += alpha * x (fp32 or T), used for grad accumulation of tied weights
template <typename T>
__global__ void axpy_kernel(const T* __restrict__ x, T* __restrict__ y, long n, float alpha) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) Elem<T>::st(y + i, Elem<T>::ld(y + i) + alpha * Elem<T>::ld(x + i));
}
extern "C" int tell_axpy(const void* x, void* y, long n, float alpha, int dtype, hipStream_t stream) {
  if (n <= 0) return TELL_OK;
  int g = grid_for(n, 256 * 4);
  if (dtype == TELL_BF16) hipLaunchKernelGGL((axpy_kernel<uint16_t>), dim3(g), dim3(256), 0, stream, (const uint16_t*)x, (uint16_t*)y, n, alpha);
  else hipLaunchKernelGGL((axpy_kernel<float>), dim3(g), dim3(256), 0, stream, (const float*)x, (float*)y, n, alpha);
  return tell_check_launch("axpy");
}

// ---------------------------------------------------------------- out = x0 + x1 + ... (up to 8 dense tensors): gradient fan-in
// of an activation that feeds several branches (the 4 context attentions + their residuals read the same X,
// decoder_faces_objects.py:271-352) in ONE pass instead of n-1 pairwise adds
struct SumNArgs { const void* x[8]; };
template <typename T>
__global__ __launch_bounds__(256) void sum_n_kernel(SumNArgs a, int n_in, T* __restrict__ out, long n) {
  constexpr int VEC = Elem<T>::VEC;
  const long nv = n / VEC;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < n_in) {
        float v[VEC];
        unpack16(*reinterpret_cast<const uint4*>(static_cast<const T*>(a.x[j]) + i * VEC), v, (const T*)nullptr);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] += v[k];
      }
    }
    *reinterpret_cast<uint4*>(out + i * VEC) = pack16(acc, (const T*)nullptr);
  }
}
extern "C" int tell_sum_n(const void* x0, const void* x1, const void* x2, const void* x3, const void* x4,
                          const void* x5, const void* x6, const void* x7, int n_in, void* out, long n, int dtype,
                          hipStream_t stream) {
  if (n <= 0) return TELL_OK;
  const int vec = dtype == TELL_BF16 ? 8 : 4;
  TELL_REQUIRE(n_in >= 1 && n_in <= 8 && n % vec == 0, "sum_n: 1..8 inputs, n a multiple of one 16-byte chunk");
  SumNArgs a = {{x0, x1, x2, x3, x4, x5, x6, x7}};
  for (int j = 0; j < n_in; ++j) TELL_REQUIRE(a.x[j] && ((uintptr_t)a.x[j] & 15) == 0, "sum_n: inputs must be 16-byte aligned");
  int g = grid_for(n / vec, 256);
  if (dtype == TELL_BF16) hipLaunchKernelGGL((sum_n_kernel<uint16_t>), dim3(g), dim3(256), 0, stream, a, n_in, (uint16_t*)out, n);
  else hipLaunchKernelGGL((sum_n_kernel<float>), dim3(g), dim3(256), 0, stream, a, n_in, (float*)out, n);
  return tell_check_launch("sum_n");
}

// ---------------------------------------------------------------- relu backward: dx = dy * (y > 0)
template <typename T>
__global__ void relu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) Elem<T>::st(dx + i, Elem<T>::ld(y + i) > 0.f ? Elem<T>::ld(dy + i) : 0.f);
}
extern "C" int tell_relu_bwd(const void* dy, const void* y, void* dx, long n, int dtype, hipStream_t stream) {
  if (n <= 0) return TELL_OK;
  int g = grid_for(n, 256 * 4);
  if (dtype == TELL_BF16) hipLaunchKernelGGL((relu_bwd_kernel<uint16_t>), dim3(g), dim3(256), 0, stream, (const uint16_t*)dy, (const uint16_t*)y, (uint16_t*)dx, n);
  else hipLaunchKernelGGL((relu_bwd_kernel<float>), dim3(g), dim3(256), 0, stream, (const float*)dy, (const float*)y, (float*)dx, n);
  return tell_check_launch("relu_bwd");
}
