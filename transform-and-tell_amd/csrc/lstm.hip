// LSTM-decoder pieces of the GloVe/LSTM baseline (tell/models/decoder_flattened_lstm.py):
//   lstm_cell     the gate non-linearities and state update of nn.LSTMCell (:20-26, called at :160-161); the two
//                 gate GEMMs (x W_ih^T + b_ih, h W_hh^T + b_hh) are tell_gemm_nt calls, their sum is taken here
//   dot_attn      AttentionLayer.forward (:40-63) between the input projection and the output projection:
//                 scores = <source_hids[l,b,:], x[b,:]>, key-padding mask, softmax over l, weighted sum
//   tanh          the tanh around output_proj (:62)
// All of it is latency-bound elementwise / reduction work at B x H = 16 x 1536: one pass, fp32 math, coalesced rows.
#include "common.h"

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// gates: [B, 4H] fp32, ACTIVATED (i, f, g, o in nn.LSTMCell's chunk order) - saved for the backward pass
template <typename T>
__global__ void lstm_cell_fwd_kernel(const T* g1, const T* g2, const T* c_prev, T* h, T* c, float* gates, int B, int H) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * H) return;
  const int b = idx / H, j = idx % H;
  const long base = (long)b * 4 * H + j;
  float pre[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) pre[k] = Elem<T>::ld(g1 + base + (long)k * H) + Elem<T>::ld(g2 + base + (long)k * H);
  const float i = sigmoidf_(pre[0]), f = sigmoidf_(pre[1]), g = tanhf(pre[2]), o = sigmoidf_(pre[3]);
  const float cn = f * Elem<T>::ld(c_prev + idx) + i * g;
  Elem<T>::st(c + idx, cn);
  Elem<T>::st(h + idx, o * tanhf(cn));
  gates[base] = i; gates[base + H] = f; gates[base + 2L * H] = g; gates[base + 3L * H] = o;
}

// dh / dc may be null (no gradient arrived on that output).  c is the NEW cell state as stored by the forward pass.
template <typename T>
__global__ void lstm_cell_bwd_kernel(const T* dh, const T* dc, const float* gates, const T* c, const T* c_prev,
                                     T* dgates, T* dc_prev, int B, int H) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * H) return;
  const int b = idx / H, j = idx % H;
  const long base = (long)b * 4 * H + j;
  const float i = gates[base], f = gates[base + H], g = gates[base + 2L * H], o = gates[base + 3L * H];
  const float tc = tanhf(Elem<T>::ld(c + idx));
  const float gh = dh ? Elem<T>::ld(dh + idx) : 0.f;
  const float dcn = (dc ? Elem<T>::ld(dc + idx) : 0.f) + gh * o * (1.f - tc * tc);
  Elem<T>::st(dgates + base, dcn * g * i * (1.f - i));
  Elem<T>::st(dgates + base + H, dcn * Elem<T>::ld(c_prev + idx) * f * (1.f - f));
  Elem<T>::st(dgates + base + 2L * H, dcn * i * (1.f - g * g));
  Elem<T>::st(dgates + base + 3L * H, gh * tc * o * (1.f - o));
  Elem<T>::st(dc_prev + idx, dcn * f);
}

constexpr int DA_MAXL = 1024;

// one workgroup (256 threads) per batch element; src[l,b,:] at src + l*s_sl + b*s_sb; probs: [L,B] fp32
template <typename T>
__global__ __launch_bounds__(256) void dot_attn_fwd_kernel(const T* src, long s_sl, long s_sb, const T* x,
                                                           const uint8_t* mask, T* ctx, float* probs, int L, int B,
                                                           int D) {
  __shared__ float sc[DA_MAXL];
  __shared__ float red[8];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* xb = x + (long)b * D;
  for (int l = wave; l < L; l += 4) {
    const T* s = src + l * s_sl + b * s_sb;
    float acc = 0.f;
    for (int d = lane; d < D; d += 64) acc += Elem<T>::ld(s + d) * Elem<T>::ld(xb + d);
    acc = wave_sum(acc);
    if (lane == 0) sc[l] = (mask && mask[(long)b * L + l]) ? -INFINITY : acc;
  }
  __syncthreads();
  float m = -INFINITY;
  for (int l = tid; l < L; l += 256) m = fmaxf(m, sc[l]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int l = tid; l < L; l += 256) sum += __expf(sc[l] - m);
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
  for (int l = tid; l < L; l += 256) {
    const float pr = __expf(sc[l] - m) * inv;
    sc[l] = pr;
    probs[(long)l * B + b] = pr;
  }
  __syncthreads();
  for (int d = tid; d < D; d += 256) {
    float acc = 0.f;
    for (int l = 0; l < L; ++l) acc += sc[l] * Elem<T>::ld(src + l * s_sl + b * s_sb + d);
    Elem<T>::st(ctx + (long)b * D + d, acc);
  }
}

// gradient w.r.t. the projected query x and, when dsrc != null (weigh_bert: the article states are a trainable mix of
// the RoBERTa layers), w.r.t. the source states: dsrc[l,b,:] = probs[l,b] dctx[b,:] + dscore[l] x[b,:], contiguous [L,B,D]
template <typename T>
__global__ __launch_bounds__(256) void dot_attn_bwd_kernel(const T* src, long s_sl, long s_sb, const float* probs,
                                                           const T* dctx, const T* x, T* dx, T* dsrc, int L, int B,
                                                           int D) {
  __shared__ float ds[DA_MAXL];
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* gb = dctx + (long)b * D;
  for (int l = wave; l < L; l += 4) {                       // dp[l] = <dctx, src[l]>
    const T* s = src + l * s_sl + b * s_sb;
    float acc = 0.f;
    for (int d = lane; d < D; d += 64) acc += Elem<T>::ld(s + d) * Elem<T>::ld(gb + d);
    acc = wave_sum(acc);
    if (lane == 0) ds[l] = acc;
  }
  __syncthreads();
  float s = 0.f;
  for (int l = tid; l < L; l += 256) s += probs[(long)l * B + b] * ds[l];
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  s = red[0] + red[1] + red[2] + red[3];
  for (int l = tid; l < L; l += 256) ds[l] = probs[(long)l * B + b] * (ds[l] - s);     // softmax backward
  __syncthreads();
  for (int d = tid; d < D; d += 256) {
    float acc = 0.f;
    for (int l = 0; l < L; ++l) acc += ds[l] * Elem<T>::ld(src + l * s_sl + b * s_sb + d);
    Elem<T>::st(dx + (long)b * D + d, acc);
  }
  if (dsrc) {
    const T* xb = x + (long)b * D;
    for (int d = tid; d < D; d += 256) {
      const float g = Elem<T>::ld(gb + d), xv = Elem<T>::ld(xb + d);
      for (int l = 0; l < L; ++l)
        Elem<T>::st(dsrc + ((long)l * B + b) * D + d, probs[(long)l * B + b] * g + ds[l] * xv);
    }
  }
}

template <typename T> __global__ void tanh_fwd_kernel(const T* x, T* y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) Elem<T>::st(y + i, tanhf(Elem<T>::ld(x + i)));
}
template <typename T> __global__ void tanh_bwd_kernel(const T* dy, const T* y, T* dx, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float v = Elem<T>::ld(y + i); Elem<T>::st(dx + i, Elem<T>::ld(dy + i) * (1.f - v * v)); }
}

#define DISPATCH(dtype, ...) do { if ((dtype) == TELL_BF16) { using T = uint16_t; __VA_ARGS__; } else { using T = float; __VA_ARGS__; } } while (0)

extern "C" int tell_lstm_cell_fwd(const void* g1, const void* g2, const void* c_prev, void* h, void* c, float* gates,
                                  int B, int H, int dtype, hipStream_t stream) {
  if (B <= 0 || H <= 0) return TELL_OK;
  TELL_REQUIRE(dtype == TELL_BF16 || dtype == TELL_F32, "lstm_cell_fwd: bad dtype");
  const long n = (long)B * H;
  DISPATCH(dtype, hipLaunchKernelGGL((lstm_cell_fwd_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                                     (const T*)g1, (const T*)g2, (const T*)c_prev, (T*)h, (T*)c, gates, B, H));
  return tell_check_launch("lstm_cell_fwd");
}

extern "C" int tell_lstm_cell_bwd(const void* dh, const void* dc, const float* gates, const void* c, const void* c_prev,
                                  void* dgates, void* dc_prev, int B, int H, int dtype, hipStream_t stream) {
  if (B <= 0 || H <= 0) return TELL_OK;
  TELL_REQUIRE(dtype == TELL_BF16 || dtype == TELL_F32, "lstm_cell_bwd: bad dtype");
  const long n = (long)B * H;
  DISPATCH(dtype, hipLaunchKernelGGL((lstm_cell_bwd_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                                     (const T*)dh, (const T*)dc, gates, (const T*)c, (const T*)c_prev, (T*)dgates,
                                     (T*)dc_prev, B, H));
  return tell_check_launch("lstm_cell_bwd");
}

extern "C" int tell_dot_attn_fwd(const void* src, long src_stride_l, long src_stride_b, const void* x,
                                 const unsigned char* mask, void* ctx, float* probs, int L, int B, int D, int dtype,
                                 hipStream_t stream) {
  if (B <= 0) return TELL_OK;
  TELL_REQUIRE(L >= 1 && L <= DA_MAXL, "dot_attn_fwd: source length must be in 1..1024");
  TELL_REQUIRE(dtype == TELL_BF16 || dtype == TELL_F32, "dot_attn_fwd: bad dtype");
  DISPATCH(dtype, hipLaunchKernelGGL((dot_attn_fwd_kernel<T>), dim3(B), dim3(256), 0, stream, (const T*)src, src_stride_l,
                                     src_stride_b, (const T*)x, mask, (T*)ctx, probs, L, B, D));
  return tell_check_launch("dot_attn_fwd");
}

extern "C" int tell_dot_attn_bwd(const void* src, long src_stride_l, long src_stride_b, const float* probs,
                                 const void* dctx, const void* x, void* dx, void* dsrc, int L, int B, int D, int dtype,
                                 hipStream_t stream) {
  if (B <= 0) return TELL_OK;
  TELL_REQUIRE(L >= 1 && L <= DA_MAXL, "dot_attn_bwd: source length must be in 1..1024");
  TELL_REQUIRE(dtype == TELL_BF16 || dtype == TELL_F32, "dot_attn_bwd: bad dtype");
  DISPATCH(dtype, hipLaunchKernelGGL((dot_attn_bwd_kernel<T>), dim3(B), dim3(256), 0, stream, (const T*)src, src_stride_l,
                                     src_stride_b, probs, (const T*)dctx, (const T*)x, (T*)dx, (T*)dsrc, L, B, D));
  return tell_check_launch("dot_attn_bwd");
}

extern "C" int tell_tanh_fwd(const void* x, void* y, long n, int dtype, hipStream_t stream) {
  if (n <= 0) return TELL_OK;
  DISPATCH(dtype, hipLaunchKernelGGL((tanh_fwd_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                                     (const T*)x, (T*)y, n));
  return tell_check_launch("tanh_fwd");
}

extern "C" int tell_tanh_bwd(const void* dy, const void* y, void* dx, long n, int dtype, hipStream_t stream) {
  if (n <= 0) return TELL_OK;
  DISPATCH(dtype, hipLaunchKernelGGL((tanh_bwd_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                                     (const T*)dy, (const T*)y, (T*)dx, n));
  return tell_check_launch("tanh_bwd");
}
