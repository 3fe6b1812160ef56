// ResNet-152 trunk helpers (tell/models/resnet.py:92-108), NHWC activations so that
// every convolution is an NT GEMM on the matrix cores: 1x1/s1 convs read the
// activation matrix [B*H*W, Cin] directly; 3x3, 7x7 and strided 1x1 convs go
// through im2col rows [B*OH*OW, KH*KW*Cin] (K padded to a 16-byte multiple) against
// weights stored [Cout, KH, KW, Cin].  BatchNorm runs with BATCH statistics in
// training (callback_apex_trainer.py:259 puts the frozen trunk in train mode).
#include "common.h"
#include "options.h"
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;   // native 16-byte vector (stays in VGPRs)

template <typename S, typename D>
__global__ void nchw_to_nhwc_kernel(const S* __restrict__ x, D* __restrict__ y, int B, int C, int H, int W) {
  const long n = (long)B * C * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long r = i / C;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int b = (int)(r / H);
    Elem<D>::st(y + i, Elem<S>::ld(x + (((long)b * C + c) * H + h) * W + w));
  }
}
extern "C" int tell_nchw_to_nhwc(const float* x, void* y, int B, int C, int H, int W, int out_dtype,
                                 hipStream_t stream) {
  long n = (long)B * C * H * W;
  if (n <= 0) return TELL_OK;
  int g = (int)((n + 1023) / 1024 > 4096 ? 4096 : (n + 1023) / 1024);
  if (out_dtype == TELL_BF16) hipLaunchKernelGGL((nchw_to_nhwc_kernel<float, uint16_t>), dim3(g), dim3(256), 0, stream, x, (uint16_t*)y, B, C, H, W);
  else hipLaunchKernelGGL((nchw_to_nhwc_kernel<float, float>), dim3(g), dim3(256), 0, stream, x, (float*)y, B, C, H, W);
  return tell_check_launch("nchw_to_nhwc");
}

// fp32 NCHW image (C <= 4) -> bf16 NHWC4 pixels (8 bytes each, missing channels zero): the input layout of the implicit
// 7x7 stem gather (gemm.hip).  One thread per pixel: plane reads coalesce across the wave, one 8-byte store.
__global__ __launch_bounds__(256) void nchw_to_nhwc4_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, int B, int C,
                                                            long HW) {
  const long n = (long)B * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW, hw = i - b * HW;
    const float* src = x + b * C * HW + hw;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < C) v[c] = src[(long)c * HW];
    uint2 o;
    o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    reinterpret_cast<uint2*>(y)[i] = o;
  }
}
extern "C" int tell_nchw_to_nhwc4(const float* x, void* y, int B, int C, int H, int W, hipStream_t stream) {
  const long n = (long)B * H * W;
  if (n <= 0) return TELL_OK;
  TELL_REQUIRE(C >= 1 && C <= 4 && ((uintptr_t)y & 7) == 0, "nchw_to_nhwc4: 1..4 channels, 8-byte aligned output");
  const int g = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
  hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(g), dim3(256), 0, stream, x, (uint16_t*)y, B, C, (long)H * W);
  return tell_check_launch("nchw_to_nhwc4");
}

// col[m][(kh*KW + kw)*Cin + c] = x[b, oh*s - pad + kh, ow*s - pad + kw, c]  (0 outside / K padding)
template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const T* __restrict__ x, T* __restrict__ col, int B, int H,
                                                     int W, int Cin, int KH, int KW, int stride, int pad,
                                                     int OH, int OW, int Kp) {
  const long M = (long)B * OH * OW;
  const int Kreal = KH * KW * Cin;
  const long n = M * Kp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % Kp);
    const long m = i / Kp;
    T v = (T)0;
    if (k < Kreal) {
      const int c = k % Cin, kk = k / Cin, kw = kk % KW, kh = kk / KW;
      const int ow = (int)(m % OW);
      const long r = m / OW;
      const int oh = (int)(r % OH), b = (int)(r / OH);
      const int ih = oh * stride - pad + kh, iw = ow * stride - pad + kw;
      if (ih >= 0 && ih < H && iw >= 0 && iw < W) v = x[(((long)b * H + ih) * W + iw) * Cin + c];
    }
    col[i] = v;
  }
}
// The 3-channel 7x7 stem (bf16, Kp % 8 == 0): one thread per 16-byte OUTPUT chunk (8 consecutive k of one row).  The
// element-per-thread kernel above spends its time in five integer divisions per 2-byte store (195 us for the 154 MB
// matrix of a 32-image batch: 0.8 TB/s); here the k -> (kh, kw, c) decomposition is a table in LDS built once per
// workgroup (input offset and tap coordinates per k), the row decomposition is done once per chunk, and the matrix
// leaves as whole 16-byte pieces.  The gathers are 2-byte loads from a 9.6 MB input that lives in L2.
__global__ __launch_bounds__(256) void im2col_chunk_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ col,
                                                           int B, int H, int W, int Cin, int KH, int KW, int stride,
                                                           int pad, int OH, int OW, int Kp) {
  extern __shared__ int tab[];                     // [Kp]: (offset << 8) | (kh << 4) | kw ; -1 for the K padding
  const int Kreal = KH * KW * Cin;
  for (int k = threadIdx.x; k < Kp; k += blockDim.x) {
    int v = -1;
    if (k < Kreal) {
      const int c = k % Cin, kk = k / Cin, kw = kk % KW, kh = kk / KW;
      v = (((kh * W + kw) * Cin + c) << 8) | (kh << 4) | kw;
    }
    tab[k] = v;
  }
  __syncthreads();
  const int cpr = Kp >> 3;                         // chunks per row
  const long n = (long)B * OH * OW * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cpr);
    const long m = i / cpr;
    const int ow = (int)(m % OW);
    const long r = m / OW;
    const int oh = (int)(r % OH), b = (int)(r / OH);
    const int ih0 = oh * stride - pad, iw0 = ow * stride - pad;
    const uint16_t* base = x + (((long)b * H + ih0) * W + iw0) * Cin;
    uint16_t e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = tab[ch * 8 + j];
      const int kh = (t >> 4) & 15, kw = t & 15;
      const bool in = t >= 0 && (unsigned)(ih0 + kh) < (unsigned)H && (unsigned)(iw0 + kw) < (unsigned)W;
      e[j] = in ? base[t >> 8] : (uint16_t)0;
    }
    u32x4_t o = {(uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16),
                 (uint32_t)e[4] | ((uint32_t)e[5] << 16), (uint32_t)e[6] | ((uint32_t)e[7] << 16)};
    *reinterpret_cast<u32x4_t*>(col + m * Kp + ch * 8) = o;
  }
}
// 16-byte-chunk variant for Cin % VEC == 0 (every conv except the 3-channel stem)
template <typename T>
__global__ __launch_bounds__(256) void im2col_vec_kernel(const T* __restrict__ x, T* __restrict__ col, int B, int H,
                                                         int W, int Cin, int KH, int KW, int stride, int pad,
                                                         int OH, int OW) {
  constexpr int VEC = Elem<T>::VEC;
  const int cpk = Cin / VEC;                       // chunks per (kh,kw) tap
  const long M = (long)B * OH * OW;
  const long n = M * KH * KW * cpk;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cpk);
    long r = i / cpk;
    const int kw = (int)(r % KW); r /= KW;
    const int kh = (int)(r % KH); r /= KH;
    const long m = r;
    const int ow = (int)(m % OW);
    const long r2 = m / OW;
    const int oh = (int)(r2 % OH), b = (int)(r2 / OH);
    const int ih = oh * stride - pad + kh, iw = ow * stride - pad + kw;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (ih >= 0 && ih < H && iw >= 0 && iw < W)
      v = *reinterpret_cast<const uint4*>(x + (((long)b * H + ih) * W + iw) * Cin + cc * VEC);
    *reinterpret_cast<uint4*>(col + i * VEC) = v;
  }
}

// Same gather, but every element first goes through the producer's BatchNorm (+ReLU): the normalised activation
// of a conv -> BN -> ReLU -> 3x3 conv chain is never written to HBM.  Padding taps stay exact zeros (the padding
// belongs to the post-activation tensor).
template <typename T>
__global__ __launch_bounds__(256) void im2col_bn_vec_kernel(const T* __restrict__ x, T* __restrict__ col, int B, int H,
                                                            int W, int Cin, int KH, int KW, int stride, int pad,
                                                            int OH, int OW, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int relu) {
  constexpr int VEC = Elem<T>::VEC;
  const int cpk = Cin / VEC;
  const long M = (long)B * OH * OW;
  const long n = M * KH * KW * cpk;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cpk);
    long r = i / cpk;
    const int kw = (int)(r % KW); r /= KW;
    const int kh = (int)(r % KH); r /= KH;
    const long m = r;
    const int ow = (int)(m % OW);
    const long r2 = m / OW;
    const int oh = (int)(r2 % OH), b = (int)(r2 / OH);
    const int ih = oh * stride - pad + kh, iw = ow * stride - pad + kw;
    uint4 out = make_uint4(0, 0, 0, 0);
    if (ih >= 0 && ih < H && iw >= 0 && iw < W) {
      float v[VEC];
      unpack16(*reinterpret_cast<const uint4*>(x + (((long)b * H + ih) * W + iw) * Cin + cc * VEC), v, (const T*)nullptr);
      const int c0 = cc * VEC;
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const float o = (v[k] - mean[c0 + k]) * invstd[c0 + k] * gamma[c0 + k] + beta[c0 + k];
        v[k] = relu ? fmaxf(o, 0.f) : o;
      }
      out = pack16(v, (const T*)nullptr);
    }
    *reinterpret_cast<uint4*>(col + i * VEC) = out;
  }
}
// requires Cin % (16 bytes / element) == 0 and an unpadded K (every 3x3 conv of the trunk)
extern "C" int tell_im2col_bn(const void* x, void* col, int B, int H, int W, int Cin, int KH, int KW, int stride,
                              int pad, int OH, int OW, const float* mean, const float* invstd, const float* gamma,
                              const float* beta, int relu, int dtype, hipStream_t stream) {
  const int vec = dtype == TELL_BF16 ? 8 : 4;
  long n = (long)B * OH * OW * KH * KW * Cin;
  if (n <= 0) return TELL_OK;
  TELL_REQUIRE(Cin % vec == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)col & 15) == 0,
               "im2col_bn: Cin must be a multiple of one 16-byte chunk and the buffers 16-byte aligned");
  long nv = n / vec;
  int gv = (int)((nv + 255) / 256 > 16384 ? 16384 : (nv + 255) / 256);
  if (dtype == TELL_BF16) hipLaunchKernelGGL((im2col_bn_vec_kernel<uint16_t>), dim3(gv), dim3(256), 0, stream, (const uint16_t*)x, (uint16_t*)col, B, H, W, Cin, KH, KW, stride, pad, OH, OW, mean, invstd, gamma, beta, relu);
  else hipLaunchKernelGGL((im2col_bn_vec_kernel<float>), dim3(gv), dim3(256), 0, stream, (const float*)x, (float*)col, B, H, W, Cin, KH, KW, stride, pad, OH, OW, mean, invstd, gamma, beta, relu);
  return tell_check_launch("im2col_bn");
}

extern "C" int tell_im2col(const void* x, void* col, int B, int H, int W, int Cin, int KH, int KW, int stride,
                           int pad, int OH, int OW, int Kp, int dtype, hipStream_t stream) {
  long n = (long)B * OH * OW * Kp;
  if (n <= 0) return TELL_OK;
  TELL_REQUIRE(Kp >= KH * KW * Cin, "im2col: padded K smaller than KH*KW*Cin");
  const int vec = dtype == TELL_BF16 ? 8 : 4;
  if (Cin % vec == 0 && Kp == KH * KW * Cin && ((uintptr_t)x & 15) == 0 && ((uintptr_t)col & 15) == 0) {
    long nv = n / vec;
    int gv = (int)((nv + 255) / 256 > 16384 ? 16384 : (nv + 255) / 256);
    if (dtype == TELL_BF16) hipLaunchKernelGGL((im2col_vec_kernel<uint16_t>), dim3(gv), dim3(256), 0, stream, (const uint16_t*)x, (uint16_t*)col, B, H, W, Cin, KH, KW, stride, pad, OH, OW);
    else hipLaunchKernelGGL((im2col_vec_kernel<float>), dim3(gv), dim3(256), 0, stream, (const float*)x, (float*)col, B, H, W, Cin, KH, KW, stride, pad, OH, OW);
    return tell_check_launch("im2col_vec");
  }
  if (dtype == TELL_BF16 && Kp % 8 == 0 && KH < 16 && KW < 16 && Kp <= 4096 && (long)KH * W * Cin < (1 << 22) &&
      ((uintptr_t)col & 15) == 0) {
    const long nc = n / 8;
    const int gc = (int)((nc + 255) / 256 > 16384 ? 16384 : (nc + 255) / 256);
    hipLaunchKernelGGL(im2col_chunk_kernel, dim3(gc), dim3(256), Kp * sizeof(int), stream, (const uint16_t*)x, (uint16_t*)col,
                       B, H, W, Cin, KH, KW, stride, pad, OH, OW, Kp);
    return tell_check_launch("im2col_chunk");
  }
  int g = (int)((n + 1023) / 1024 > 8192 ? 8192 : (n + 1023) / 1024);
  if (dtype == TELL_BF16) hipLaunchKernelGGL((im2col_kernel<uint16_t>), dim3(g), dim3(256), 0, stream, (const uint16_t*)x, (uint16_t*)col, B, H, W, Cin, KH, KW, stride, pad, OH, OW, Kp);
  else hipLaunchKernelGGL((im2col_kernel<float>), dim3(g), dim3(256), 0, stream, (const float*)x, (float*)col, B, H, W, Cin, KH, KW, stride, pad, OH, OW, Kp);
  return tell_check_launch("im2col");
}

// ---------------------------------------------------------------- BatchNorm (batch statistics)
// stage 1: per (row chunk, 32-column group): count, mean, M2 of the chunk (fp32, two-pass inside the chunk)
template <typename T>
__global__ __launch_bounds__(256) void bn_partial_kernel(const T* __restrict__ x, long M, int C, int BN_ROWS,
                                                         float* __restrict__ pmean, float* __restrict__ pm2) {
  __shared__ float red[8][33];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  const long r0 = (long)blockIdx.y * BN_ROWS;
  const long r1 = r0 + BN_ROWS < M ? r0 + BN_ROWS : M;
  const float cnt = (float)(r1 - r0);
  float s = 0.f;
  if (c < C) for (long r = r0 + ry; r < r1; r += 8) s += Elem<T>::ld(x + r * C + c);
  red[ry][cx] = s;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) mean += red[k][cx];
  mean /= cnt;
  __syncthreads();
  float q = 0.f;
  if (c < C) for (long r = r0 + ry; r < r1; r += 8) { float d = Elem<T>::ld(x + r * C + c) - mean; q += d * d; }
  red[ry][cx] = q;
  __syncthreads();
  if (ry == 0 && c < C) {
    float m2 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) m2 += red[k][cx];
    pmean[(long)blockIdx.y * C + c] = mean;
    pm2[(long)blockIdx.y * C + c] = m2;
  }
}
// stage 1 (vectorised): each thread owns one 16-byte column chunk and walks rows; statistics are
// accumulated SHIFTED by the chunk's first row (sum(x-s), sum((x-s)^2)) so the single pass stays
// accurate in fp32; row lanes are reduced through LDS.
template <typename T>
__global__ __launch_bounds__(256) void bn_partial_vec_kernel(const T* __restrict__ x, long M, int C, int BN_ROWS,
                                                             float* __restrict__ pmean, float* __restrict__ pm2) {
  constexpr int VEC = Elem<T>::VEC;
  __shared__ float red_s[256 * VEC], red_q[256 * VEC];
  const int cpr = C / VEC;
  const int CB = cpr < 256 ? cpr : 256, RL = 256 / CB;
  const int tid = threadIdx.x, cl = tid % CB, rl = tid / CB;
  const int cc = blockIdx.x * CB + cl;
  const long r0 = (long)blockIdx.y * BN_ROWS;
  const long r1 = r0 + BN_ROWS < M ? r0 + BN_ROWS : M;
  float sh[VEC], s[VEC], q[VEC];
  unpack16(*reinterpret_cast<const uint4*>(x + r0 * C + (long)cc * VEC), sh, (const T*)nullptr);
#pragma unroll
  for (int k = 0; k < VEC; ++k) { s[k] = 0.f; q[k] = 0.f; }
  for (long r = r0 + rl; r < r1; r += RL) {
    float v[VEC];
    unpack16(*reinterpret_cast<const uint4*>(x + r * C + (long)cc * VEC), v, (const T*)nullptr);
#pragma unroll
    for (int k = 0; k < VEC; ++k) { const float d = v[k] - sh[k]; s[k] += d; q[k] += d * d; }
  }
#pragma unroll
  for (int k = 0; k < VEC; ++k) { red_s[(rl * CB + cl) * VEC + k] = s[k]; red_q[(rl * CB + cl) * VEC + k] = q[k]; }
  __syncthreads();
  if (rl == 0) {
    const float n = (float)(r1 - r0);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      float ts = 0.f, tq = 0.f;
      for (int j = 0; j < RL; ++j) { ts += red_s[(j * CB + cl) * VEC + k]; tq += red_q[(j * CB + cl) * VEC + k]; }
      pmean[(long)blockIdx.y * C + cc * VEC + k] = sh[k] + ts / n;
      pm2[(long)blockIdx.y * C + cc * VEC + k] = tq - ts * ts / n;
    }
  }
}
// stage 2: Chan combine over the chunks, one wave per channel (lanes stride the chunks, then butterfly)
__global__ __launch_bounds__(256) void bn_finish_kernel(const float* __restrict__ pmean, const float* __restrict__ pm2,
                                                        long M, int C, int n_chunks, int BN_ROWS, float eps,
                                                        float momentum, float* __restrict__ mean,
                                                        float* __restrict__ invstd, float* __restrict__ running_mean,
                                                        float* __restrict__ running_var) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  float n = 0.f, mu = 0.f, m2 = 0.f;
  for (int k = lane; k < n_chunks; k += 64) {
    const long r0 = (long)k * BN_ROWS;
    const float nb = (float)((r0 + BN_ROWS < M ? r0 + BN_ROWS : M) - r0);
    const float mb = pmean[(long)k * C + c], qb = pm2[(long)k * C + c];
    const float nt = n + nb, delta = mb - mu;
    mu = (n * mu + nb * mb) / nt;
    m2 = m2 + qb + delta * delta * n * nb / nt;
    n = nt;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float nb = __shfl_xor(n, o, 64), mb = __shfl_xor(mu, o, 64), qb = __shfl_xor(m2, o, 64);
    const float nt = n + nb;
    if (nt > 0.f) {
      const float delta = mb - mu;
      mu = (n * mu + nb * mb) / nt;           // symmetric form: both partners compute the same value
      m2 = m2 + qb + delta * delta * n * nb / nt;
    }
    n = nt;
  }
  if (lane == 0) {
    const float var = m2 / n;
    mean[c] = mu;
    invstd[c] = rsqrtf(var + eps);
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (n > 1.f ? m2 / (n - 1.f) : var);
    }
  }
}
int tell_bn_finish_launch(const float* pmean, const float* pm2, long M, int C, int n_chunks, int rows_per_chunk,
                          float eps, float momentum, float* mean, float* invstd, float* running_mean,
                          float* running_var, hipStream_t stream) {
  hipLaunchKernelGGL(bn_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, stream, pmean, pm2, M, C, n_chunks,
                     rows_per_chunk, eps, momentum, mean, invstd, running_mean, running_var);
  return tell_check_launch("bn_finish");
}
static inline int bn_rows_per_chunk(long M) {
  long r = (M + 255) / 256;                   // ~256 row chunks -> enough workgroups for every layer shape
  if (r < 8) r = 8;
  return (int)((r + 7) / 8 * 8);
}
extern "C" long tell_bn_chunks(long M) { const int r = bn_rows_per_chunk(M); return (M + r - 1) / r; }
// workspace: 2 * tell_bn_chunks(M) * C floats
extern "C" int tell_bn_stats(const void* x, long M, int C, float eps, float momentum, float* mean, float* invstd,
                             float* running_mean, float* running_var, float* workspace, int dtype,
                             hipStream_t stream) {
  if (M <= 0 || C <= 0) return TELL_OK;
  const int rpc = bn_rows_per_chunk(M);
  const long nch = tell_bn_chunks(M);
  float* pmean = workspace;
  float* pm2 = workspace + nch * C;
  const int vec = dtype == TELL_BF16 ? 8 : 4;
  const int cpr = C / vec;
  const bool pow2 = cpr > 0 && (cpr & (cpr - 1)) == 0;
  if (C % vec == 0 && pow2 && ((uintptr_t)x & 15) == 0 && (cpr <= 256 || cpr % 256 == 0)) {
    dim3 grid(cpr <= 256 ? 1 : cpr / 256, (unsigned)nch);
    if (dtype == TELL_BF16) hipLaunchKernelGGL((bn_partial_vec_kernel<uint16_t>), grid, dim3(256), 0, stream, (const uint16_t*)x, M, C, rpc, pmean, pm2);
    else hipLaunchKernelGGL((bn_partial_vec_kernel<float>), grid, dim3(256), 0, stream, (const float*)x, M, C, rpc, pmean, pm2);
  } else {
    dim3 grid((C + 31) / 32, (unsigned)nch);
    if (dtype == TELL_BF16) hipLaunchKernelGGL((bn_partial_kernel<uint16_t>), grid, dim3(256), 0, stream, (const uint16_t*)x, M, C, rpc, pmean, pm2);
    else hipLaunchKernelGGL((bn_partial_kernel<float>), grid, dim3(256), 0, stream, (const float*)x, M, C, rpc, pmean, pm2);
  }
  hipLaunchKernelGGL(bn_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, stream, pmean, pm2, M, C, (int)nch, rpc, eps, momentum, mean, invstd, running_mean, running_var);
  return tell_check_launch("bn_stats");
}

// y = [relu]( (x - mean) * invstd * gamma + beta [+ residual] )
template <typename T>
__global__ void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                const float* __restrict__ invstd, const float* __restrict__ gamma,
                                const float* __restrict__ beta, const T* __restrict__ residual,
                                T* __restrict__ y, long M, int C, int relu) {
  const long n = M * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    float v = (Elem<T>::ld(x + i) - mean[c]) * invstd[c] * gamma[c] + beta[c];
    if (residual) v += Elem<T>::ld(residual + i);
    if (relu) v = fmaxf(v, 0.f);
    Elem<T>::st(y + i, v);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_vec_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const T* __restrict__ residual, T* __restrict__ y, long M,
                                                           int C, int relu) {
  constexpr int VEC = Elem<T>::VEC;
  const int cpr = C / VEC;
  const long n = M * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % cpr) * VEC;
    float v[VEC], r[VEC];
    unpack16(*reinterpret_cast<const uint4*>(x + i * VEC), v, (const T*)nullptr);
    if (residual) unpack16(*reinterpret_cast<const uint4*>(residual + i * VEC), r, (const T*)nullptr);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      float o = (v[k] - mean[c0 + k]) * invstd[c0 + k] * gamma[c0 + k] + beta[c0 + k];
      if (residual) o += r[k];
      v[k] = relu ? fmaxf(o, 0.f) : o;
    }
    *reinterpret_cast<uint4*>(y + i * VEC) = pack16(v, (const T*)nullptr);
  }
}

extern "C" int tell_bn_apply(const void* x, const float* mean, const float* invstd, const float* gamma,
                             const float* beta, const void* residual, void* y, long M, int C, int relu,
                             int dtype, hipStream_t stream) {
  long n = M * C;
  if (n <= 0) return TELL_OK;
  const int vec = dtype == TELL_BF16 ? 8 : 4;
  if (C % vec == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)residual & 15) == 0) {
    long nv = n / vec;
    int gv = (int)((nv + 255) / 256 > 16384 ? 16384 : (nv + 255) / 256);
    if (dtype == TELL_BF16) hipLaunchKernelGGL((bn_apply_vec_kernel<uint16_t>), dim3(gv), dim3(256), 0, stream, (const uint16_t*)x, mean, invstd, gamma, beta, (const uint16_t*)residual, (uint16_t*)y, M, C, relu);
    else hipLaunchKernelGGL((bn_apply_vec_kernel<float>), dim3(gv), dim3(256), 0, stream, (const float*)x, mean, invstd, gamma, beta, (const float*)residual, (float*)y, M, C, relu);
    return tell_check_launch("bn_apply_vec");
  }
  int g = (int)((n + 1023) / 1024 > 8192 ? 8192 : (n + 1023) / 1024);
  if (dtype == TELL_BF16) hipLaunchKernelGGL((bn_apply_kernel<uint16_t>), dim3(g), dim3(256), 0, stream, (const uint16_t*)x, mean, invstd, gamma, beta, (const uint16_t*)residual, (uint16_t*)y, M, C, relu);
  else hipLaunchKernelGGL((bn_apply_kernel<float>), dim3(g), dim3(256), 0, stream, (const float*)x, mean, invstd, gamma, beta, (const float*)residual, (float*)y, M, C, relu);
  return tell_check_launch("bn_apply");
}

// Train-mode BatchNorm behind a convolution whose GEMM epilogue left per-row-chunk statistics (pmean / pm2: [n_chunks][C],
// csrc/gemm.hip staged_store_stats): the Chan combine of the chunks AND the normalisation (+ residual) (+ ReLU) in ONE
// launch.  A workgroup owns 64 channels x a slab of rows; its prologue merges the n_chunks partials of its 64 channels
// (thread = channel x one of 4 chunk strides, then a fixed-order fold through LDS: deterministic), which costs
// n_chunks x 512 bytes of L2 reads per workgroup - used for n_chunks <= 128 (layer3 / layer4 of the trunk at B = 32:
// 25-98 chunks), where that is a fraction of the slab itself; beyond, the combine stays a launch of its own.
// The workgroups of the first row slab also write the running statistics (momentum update, unbiased variance).
// Round 5: the launch is a chain of memory round trips, not bandwidth (6.4 MB in 11 us at layer3's [6272, 256]): chunk
// means -> barrier -> chunk M2 -> barrier -> rows pass 1 -> rows pass 2, each ~1.5 us because everything it reads was
// written by the convolution's workgroups on OTHER XCDs (nothing is in this XCD's L2).  Now ONE trip: the workgroup's
// rows of y (and of the residual) - up to 4 passes of 32 rows - are requested FIRST, then every chunk statistic of the
// thread (<= 32 chunks x 2 values, all in flight together), and the combine is a single shifted pass
//   d_k = mean_k - c,  S = sum n_k d_k,  Q = sum (M2_k + n_k d_k^2),  mean = c + S / n,  M2 = Q - S^2 / n
// with c = the first chunk's mean (each thread loads it itself): exact algebra, and the subtraction only ever sees the
// spread of the chunk means around one of them - no cancellation against the mean itself (the two-pass form it replaces
// needed the global mean before the second pass could start).  Folded over the 4 chunk strides in a fixed order.
// MEASURED (MI355X, same box, ResNet-152 train B = 32): 4.76 -> 4.63 ms; layer3 conv1 + BN + ReLU 19.0 -> 17.4 us, conv2
// 35.0 -> 33.2, layer4 conv1 20.8 -> 18.9 (tools/bench_conv.py).
#define BNFA_PASSES 4
__global__ __launch_bounds__(256) void bn_finish_apply_kernel(const float* __restrict__ pmean, const float* __restrict__ pm2,
                                                              long M, int C, int n_chunks, int rows_per_chunk, float eps,
                                                              float momentum, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ running_mean,
                                                              float* __restrict__ running_var,
                                                              const uint16_t* __restrict__ residual, uint16_t* __restrict__ y,
                                                              int relu, int rows_per_block) {
  __shared__ float ss[4][64], sq[4][64], s_mean[64], s_scale[64], s_beta[64];
  const int tid = threadIdx.x, cl = tid & 63, g = tid >> 6;
  const int c = blockIdx.x * 64 + cl;
  // ---- 1. this thread's rows: 8 channels (16 bytes) x one row per pass
  const int o = tid & 7, rl = tid >> 3;
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  const long col = (long)blockIdx.x * 64 + o * 8;
  u32x4_t yv[BNFA_PASSES], rv[BNFA_PASSES];
#pragma unroll
  for (int ps = 0; ps < BNFA_PASSES; ++ps) {
    const long r = r0 + rl + 32 * ps;
    const long rc = r < r1 ? r : r1 - 1;                       // (clamped: in-bounds, never stored)
    yv[ps] = *reinterpret_cast<const u32x4_t*>(y + rc * C + col);
    if (residual) rv[ps] = *reinterpret_cast<const u32x4_t*>(residual + rc * C + col);
  }
  // ---- 2. chunk statistics of channel c, chunks g, g + 4, ...
  const float n_last = (float)(M - (long)(n_chunks - 1) * rows_per_chunk), n_full = (float)rows_per_chunk;
  const float shift = pmean[c];                               // chunk 0's mean
  float mk[32], qk[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const int k = g + 4 * j, kk = k < n_chunks ? k : n_chunks - 1;
    mk[j] = pmean[(long)kk * C + c];
    qk[j] = pm2[(long)kk * C + c];
  }
  float S = 0.f, Q = 0.f;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const int k = g + 4 * j;
    const float nk = k >= n_chunks ? 0.f : (k == n_chunks - 1 ? n_last : n_full);
    const float d = mk[j] - shift;
    S += nk * d;
    Q += (k < n_chunks ? qk[j] : 0.f) + nk * d * d;
  }
  ss[g][cl] = S; sq[g][cl] = Q;
  __syncthreads();
  if (tid < 64) {
    const float St = (ss[0][cl] + ss[1][cl]) + (ss[2][cl] + ss[3][cl]);
    const float Qt = (sq[0][cl] + sq[1][cl]) + (sq[2][cl] + sq[3][cl]);
    const float n = (float)M;
    const float mu = shift + St / n;
    float m2 = Qt - St * St / n;
    m2 = m2 > 0.f ? m2 : 0.f;
    const float var = m2 / n;
    s_mean[cl] = mu;
    s_scale[cl] = rsqrtf(var + eps) * gamma[c];
    s_beta[cl] = beta[c];
    if (blockIdx.y == 0 && running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (n > 1.f ? m2 / (n - 1.f) : var);
    }
  }
  __syncthreads();
  float mean8[8], scale8[8], beta8[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { mean8[k] = s_mean[o * 8 + k]; scale8[k] = s_scale[o * 8 + k]; beta8[k] = s_beta[o * 8 + k]; }
  // ---- 3. normalise the rows already in registers
#pragma unroll
  for (int ps = 0; ps < BNFA_PASSES; ++ps) {
    const long r = r0 + rl + 32 * ps;
    if (r >= r1) break;
    float v[8], rs[8];
    const uint4 yw = make_uint4(yv[ps][0], yv[ps][1], yv[ps][2], yv[ps][3]);
    unpack16(yw, v, (const uint16_t*)nullptr);
    if (residual) {
      const uint4 rw = make_uint4(rv[ps][0], rv[ps][1], rv[ps][2], rv[ps][3]);
      unpack16(rw, rs, (const uint16_t*)nullptr);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float t = (v[k] - mean8[k]) * scale8[k] + beta8[k];
      if (residual) t += rs[k];
      v[k] = relu ? fmaxf(t, 0.f) : t;
    }
    *reinterpret_cast<uint4*>(y + r * C + col) = pack16(v, (const uint16_t*)nullptr);
  }
}
// More than 128 row chunks (layer1 / layer2 of the trunk: 392-1568 at B = 32): the chunk statistics are first merged into
// <= 128 super-chunks by a launch that is parallel over (64 channels, super-chunk) - the same two-pass Chan merge, S chunks
// each - and the fused finish + apply above then runs on the super-chunks.  (The single-workgroup-per-64-channels finish
// it replaces walks all chunks from one CU: 8 us alone, 29 us inside the training step where it waits behind everything.)
__global__ __launch_bounds__(256) void bn_combine_kernel(const float* __restrict__ pmean, const float* __restrict__ pm2,
                                                         long M, int C, int n_chunks, int rows_per_chunk, int S,
                                                         float* __restrict__ qmean, float* __restrict__ qm2) {
  __shared__ float sa[4][64], sb[4][64];
  const int tid = threadIdx.x, cl = tid & 63, g = tid >> 6;
  const int c = blockIdx.x * 64 + cl, sc = blockIdx.y;
  const int k0 = sc * S, k1 = k0 + S < n_chunks ? k0 + S : n_chunks;
  const float n_last = (float)(M - (long)(n_chunks - 1) * rows_per_chunk), n_full = (float)rows_per_chunk;
  float acc = 0.f, cnt = 0.f;
  for (int k = k0 + g; k < k1; k += 4) {
    const float nk = k == n_chunks - 1 ? n_last : n_full;
    acc += nk * pmean[(long)k * C + c];
    cnt += nk;
  }
  sa[g][cl] = acc; sb[g][cl] = cnt;
  __syncthreads();
  const float ntot = (sb[0][cl] + sb[1][cl]) + (sb[2][cl] + sb[3][cl]);
  const float mu = ((sa[0][cl] + sa[1][cl]) + (sa[2][cl] + sa[3][cl])) / ntot;
  __syncthreads();
  acc = 0.f;
  for (int k = k0 + g; k < k1; k += 4) {
    const float d = pmean[(long)k * C + c] - mu;
    acc += pm2[(long)k * C + c] + (k == n_chunks - 1 ? n_last : n_full) * d * d;
  }
  sa[g][cl] = acc;
  __syncthreads();
  if (tid < 64) {
    qmean[(long)sc * C + c] = mu;
    qm2[(long)sc * C + c] = (sa[0][cl] + sa[1][cl]) + (sa[2][cl] + sa[3][cl]);
  }
}
// -> TELL_OK after launching, or 1 when the shape is not one the fused kernel takes (the caller runs finish + apply)
// scratch (optional): 256 * C floats for the super-chunk statistics of the n_chunks > 128 case
int tell_bn_finish_apply_launch(const float* pmean, const float* pm2, long M, int C, int n_chunks, int rows_per_chunk,
                                float eps, float momentum, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, const void* residual, void* y, int relu, float* scratch,
                                hipStream_t stream) {
  if (C % 64 != 0 || ((uintptr_t)y & 15) != 0 || ((uintptr_t)residual & 15) != 0) return 1;
  const int slabs = C / 64;
  if (n_chunks > 128) {
    if (!scratch) return 1;
    const int S = (n_chunks + 127) / 128, G = (n_chunks + S - 1) / S;
    float* qmean = scratch;
    float* qm2 = scratch + (long)128 * C;
    hipLaunchKernelGGL(bn_combine_kernel, dim3(slabs, G), dim3(256), 0, stream, pmean, pm2, M, C, n_chunks, rows_per_chunk, S,
                       qmean, qm2);
    int rc = tell_check_launch("bn_combine");
    if (rc) return rc;
    pmean = qmean; pm2 = qm2; n_chunks = G; rows_per_chunk *= S;
  }
  // ~768 workgroups of 32 .. 128 rows (1 .. 4 passes of 32, all requested up front) x 64 channels
  const long wgs = tell_opt(OPT_BN_WGS) > 0 ? tell_opt(OPT_BN_WGS) : 768;   // tuning aid
  long per = (M * slabs + wgs - 1) / wgs;
  per = (per + 31) / 32 * 32;
  if (per < 32) per = 32;
  if (per > 32 * BNFA_PASSES) per = 32 * BNFA_PASSES;
  if ((M + per - 1) / per > 65535) return 1;          // (grid.y limit: more than 65535 x 128 rows -> the two-launch path)
  const unsigned gy = (unsigned)((M + per - 1) / per);
  hipLaunchKernelGGL(bn_finish_apply_kernel, dim3(slabs, gy), dim3(256), 0, stream, pmean, pm2, M, C, n_chunks,
                     rows_per_chunk, eps, momentum, gamma, beta, running_mean, running_var, (const uint16_t*)residual,
                     (uint16_t*)y, relu, (int)per);
  return tell_check_launch("bn_finish_apply");
}

// ToTensor + Normalize of the readers (nytimes_faces_ner_matched.py:67-69) on the device: uint8 [B,H,W,3] (decoded
// pixels as the shards store them) -> float32 [B,3,H,W] = (x / 255 - mean[c]) / std[c], the model's `image` input
__global__ __launch_bounds__(256) void image_normalize_kernel(const uint8_t* __restrict__ x, float* __restrict__ y,
                                                              long npix, long hw, float m0, float m1, float m2,
                                                              float s0, float s1, float s2) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const long b = i / hw, r = i - b * hw;
    const uint8_t* p = x + i * 3;
    float* o = y + b * 3 * hw + r;
    o[0] = ((float)p[0] * (1.f / 255.f) - m0) / s0;
    o[hw] = ((float)p[1] * (1.f / 255.f) - m1) / s1;
    o[2 * hw] = ((float)p[2] * (1.f / 255.f) - m2) / s2;
  }
}
extern "C" int tell_image_normalize(const uint8_t* x, float* y, int B, int H, int W, float m0, float m1, float m2,
                                    float s0, float s1, float s2, hipStream_t stream) {
  const long hw = (long)H * W, npix = hw * B;
  if (npix <= 0) return TELL_OK;
  TELL_REQUIRE(s0 > 0.f && s1 > 0.f && s2 > 0.f, "image_normalize: std must be positive");
  int g = (int)((npix + 255) / 256 > 8192 ? 8192 : (npix + 255) / 256);
  hipLaunchKernelGGL(image_normalize_kernel, dim3(g), dim3(256), 0, stream, x, y, npix, hw, m0, m1, m2, s0, s1, s2);
  return tell_check_launch("image_normalize");
}

// 3x3 / stride 2 / pad 1 max pooling, NHWC
template <typename T>
__global__ void maxpool_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int OH,
                               int OW) {
  const long n = (long)B * OH * OW * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long r = i / C;
    const int ow = (int)(r % OW); r /= OW;
    const int oh = (int)(r % OH);
    const int b = (int)(r / OH);
    float m = -INFINITY;
    for (int kh = 0; kh < 3; ++kh)
      for (int kw = 0; kw < 3; ++kw) {
        const int ih = oh * 2 - 1 + kh, iw = ow * 2 - 1 + kw;
        if (ih >= 0 && ih < H && iw >= 0 && iw < W)
          m = fmaxf(m, Elem<T>::ld(x + (((long)b * H + ih) * W + iw) * C + c));
      }
    Elem<T>::st(y + i, m);
  }
}
// bf16, C % 8 == 0: one thread per (output pixel, 8-channel chunk) - nine 16-byte loads, one 16-byte store (the scalar
// kernel above: 62 us for the stem's 51 MB -> 13 MB; this is the ~15 us the traffic costs)
__global__ __launch_bounds__(256) void maxpool_vec_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int B,
                                                          int H, int W, int C, int OH, int OW) {
  const int cpp = C >> 3;
  const long n = (long)B * OH * OW * cpp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cpp);
    long r = i / cpp;
    const int ow = (int)(r % OW); r /= OW;
    const int oh = (int)(r % OH);
    const int b = (int)(r / OH);
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ih = oh * 2 - 1 + kh, iw = ow * 2 - 1 + kw;
        if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
          const u32x4_t v = *reinterpret_cast<const u32x4_t*>(x + (((long)b * H + ih) * W + iw) * C + cc * 8);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            m[2 * q] = fmaxf(m[2 * q], __uint_as_float(v[q] << 16));
            m[2 * q + 1] = fmaxf(m[2 * q + 1], __uint_as_float(v[q] & 0xffff0000u));
          }
        }
      }
    u32x4_t o;                                     // maxima of bf16 values are bf16 values: truncation is exact
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = (__float_as_uint(m[2 * q]) >> 16) | (__float_as_uint(m[2 * q + 1]) & 0xffff0000u);
    *reinterpret_cast<u32x4_t*>(y + i * 8) = o;
  }
}
extern "C" int tell_maxpool3x3s2(const void* x, void* y, int B, int H, int W, int C, int OH, int OW, int dtype,
                                 hipStream_t stream) {
  long n = (long)B * OH * OW * C;
  if (n <= 0) return TELL_OK;
  if (dtype == TELL_BF16 && C % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0) {
    const long nc = n / 8;
    const int gc = (int)((nc + 255) / 256 > 16384 ? 16384 : (nc + 255) / 256);
    hipLaunchKernelGGL(maxpool_vec_kernel, dim3(gc), dim3(256), 0, stream, (const uint16_t*)x, (uint16_t*)y, B, H, W, C, OH, OW);
    return tell_check_launch("maxpool_vec");
  }
  int g = (int)((n + 1023) / 1024 > 8192 ? 8192 : (n + 1023) / 1024);
  if (dtype == TELL_BF16) hipLaunchKernelGGL((maxpool_kernel<uint16_t>), dim3(g), dim3(256), 0, stream, (const uint16_t*)x, (uint16_t*)y, B, H, W, C, OH, OW);
  else hipLaunchKernelGGL((maxpool_kernel<float>), dim3(g), dim3(256), 0, stream, (const float*)x, (float*)y, B, H, W, C, OH, OW);
  return tell_check_launch("maxpool");
}
