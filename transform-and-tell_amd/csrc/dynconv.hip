// DynamicConv core (tell/modules/convolutions/dynamic.py:285-336), T x B x C layout:
//   taps  = softmax_K(tap_logits[t,b,h,:])            (over ALL K taps, :302-304)
//   tapsd = DropConnect(taps)                          (:305)
//   y[t,b,h*R+c] = sum_k tapsd[k] * x[t-(K-1)+k, b, h*R+c],   x[<0] = 0  (causal)
// The reference builds a dense [B*H,T,T+K-1] band matrix and bmm's it; here each
// wave owns one (t,b,h) and gathers its K input rows directly (coalesced 64-lane
// rows; HBM-bound: AI = 2K / ((2 + H*K/C) * sizeof) flop/byte).
// Softmax over the taps is done with wavefront shuffles (lane k holds tap k).
#include "common.h"

template <typename T>
__global__ __launch_bounds__(256) void dynconv_fwd_kernel(const T* __restrict__ x,
                                                          const T* __restrict__ logits,
                                                          T* __restrict__ y, float* __restrict__ taps,
                                                          int Tn, int B, int H, int K, int R,
                                                          uint32_t thr, float inv_keep, uint32_t seed, uint32_t salt,
    const uint32_t* __restrict__ step) {
  salt = tell_step_salt(salt, step);
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);   // (t*B + b)*H + h
  if (wid >= (long)Tn * B * H) return;
  const int h = (int)(wid % H);
  const long tb = wid / H;
  const int b = (int)(tb % B), t = (int)(tb / B);
  const int C = H * R;

  float lg = lane < K ? Elem<T>::ld(logits + tb * (long)H * K + (long)h * K + lane) : -INFINITY;
  const float m = wave_max(lg);
  float e = lane < K ? __expf(lg - m) : 0.f;
  const float w = e / wave_sum(e);
  if (lane < K && taps) taps[wid * K + lane] = w;
  float wd = w;
  if (thr && lane < K) wd *= tell_keep(seed, salt, (uint64_t)(wid * K + lane), thr, inv_keep);

  const int k_lo = (K - 1 - t) > 0 ? (K - 1 - t) : 0;   // taps reaching before t=0 see zeros
  for (int c0 = 0; c0 < R; c0 += 64) {           // shuffles stay outside lane-divergent code
    const int c = c0 + lane;
    const bool ok = c < R;
    float acc = 0.f;
    for (int k = k_lo; k < K; ++k) {
      const int ts = t - (K - 1) + k;
      const float wk = __shfl(wd, k, 64);
      if (ok) acc += wk * Elem<T>::ld(x + ((long)ts * B + b) * C + (long)h * R + c);
    }
    if (ok) Elem<T>::st(y + ((long)t * B + b) * C + (long)h * R + c, acc);
  }
}

// dlogits[t,b,h,k] from dy:  dtapd[k] = <dy[t,b,h,:], x[t-(K-1)+k,b,h,:]>,
// then DropConnect and softmax backward.
template <typename T>
__global__ __launch_bounds__(256) void dynconv_bwd_taps_kernel(const T* __restrict__ x,
                                                               const T* __restrict__ dy,
                                                               const float* __restrict__ taps,
                                                               T* __restrict__ dlogits, int Tn, int B,
                                                               int H, int K, int R, uint32_t thr,
                                                               float inv_keep, uint32_t seed, uint32_t salt,
    const uint32_t* __restrict__ step) {
  salt = tell_step_salt(salt, step);
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (long)Tn * B * H) return;
  const int h = (int)(wid % H);
  const long tb = wid / H;
  const int b = (int)(tb % B), t = (int)(tb / B);
  const int C = H * R;
  const T* dyr = dy + tb * (long)C + (long)h * R;
  const int k_lo = (K - 1 - t) > 0 ? (K - 1 - t) : 0;
  float mine = 0.f;                              // lane k ends up holding dtapd[k]
  for (int k = k_lo; k < K; ++k) {
    const int ts = t - (K - 1) + k;
    const T* xr = x + ((long)ts * B + b) * C + (long)h * R;
    float s = 0.f;
    for (int c = lane; c < R; c += 64) s += Elem<T>::ld(dyr + c) * Elem<T>::ld(xr + c);
    s = wave_sum(s);
    if (lane == k) mine = s;
  }
  float w = lane < K ? taps[wid * K + lane] : 0.f;
  float dw = mine;
  if (thr && lane < K) dw *= tell_keep(seed, salt, (uint64_t)(wid * K + lane), thr, inv_keep);
  const float dot = wave_sum(w * dw);
  if (lane < K) Elem<T>::st(dlogits + tb * (long)H * K + (long)h * K + lane, w * (dw - dot));
}

// dx[t',b,h*R+c] = sum_k tapsd[t'+(K-1)-k, b, h, k] * dy[t'+(K-1)-k, b, h*R+c]
template <typename T>
__global__ __launch_bounds__(256) void dynconv_bwd_x_kernel(const T* __restrict__ dy,
                                                            const float* __restrict__ taps,
                                                            T* __restrict__ dx, int accumulate, int Tn,
                                                            int B, int H, int K, int R, uint32_t thr,
                                                            float inv_keep, uint32_t seed, uint32_t salt,
    const uint32_t* __restrict__ step) {
  salt = tell_step_salt(salt, step);
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (long)Tn * B * H) return;
  const int h = (int)(wid % H);
  const long tb = wid / H;
  const int b = (int)(tb % B), tp = (int)(tb / B);
  const int C = H * R;
  for (int c = lane; c < R; c += 64) {
    float acc = 0.f;
    for (int k = K - 1; k >= 0; --k) {
      const int t = tp + (K - 1) - k;
      if (t >= Tn) break;
      const long w_id = ((long)t * B + b) * H + h;
      float wd = taps[w_id * K + k];
      if (thr) wd *= tell_keep(seed, salt, (uint64_t)(w_id * K + k), thr, inv_keep);
      acc += wd * Elem<T>::ld(dy + ((long)t * B + b) * C + (long)h * R + c);
    }
    T* d = dx + ((long)tp * B + b) * C + (long)h * R + c;
    Elem<T>::st(d, accumulate ? Elem<T>::ld(d) + acc : acc);
  }
}

extern "C" int tell_dynconv_fwd(const void* x, const void* logits, void* y, float* taps, int T, int B,
                                int H, int K, int R, float p, uint32_t seed, uint32_t salt, int dtype,
                                hipStream_t stream) {
  if ((long)T * B * H <= 0) return TELL_OK;
  TELL_REQUIRE(K >= 1 && K <= 64, "dynconv: kernel size must be in [1,64] (one lane per tap)");
  TELL_REQUIRE(p >= 0.f && p < 1.f, "dynconv: p must be in [0,1)");
  uint32_t thr = p > 0.f ? tell_drop_threshold(p) : 0u;
  float ik = 1.f / (1.f - p);
  long waves = (long)T * B * H;
  dim3 grid((unsigned)((waves + 3) / 4));
  if (dtype == TELL_BF16)
    hipLaunchKernelGGL((dynconv_fwd_kernel<uint16_t>), grid, dim3(256), 0, stream, (const uint16_t*)x, (const uint16_t*)logits, (uint16_t*)y, taps, T, B, H, K, R, thr, ik, seed, salt, g_tell_rng_step);
  else
    hipLaunchKernelGGL((dynconv_fwd_kernel<float>), grid, dim3(256), 0, stream, (const float*)x, (const float*)logits, (float*)y, taps, T, B, H, K, R, thr, ik, seed, salt, g_tell_rng_step);
  return tell_check_launch("dynconv_fwd");
}

extern "C" int tell_dynconv_bwd(const void* x, const void* dy, const float* taps, void* dx,
                                int dx_accumulate, void* dlogits, int T, int B, int H, int K, int R,
                                float p, uint32_t seed, uint32_t salt, int dtype, hipStream_t stream) {
  if ((long)T * B * H <= 0) return TELL_OK;
  TELL_REQUIRE(K >= 1 && K <= 64, "dynconv: kernel size must be in [1,64] (one lane per tap)");
  TELL_REQUIRE(p >= 0.f && p < 1.f, "dynconv: p must be in [0,1)");
  uint32_t thr = p > 0.f ? tell_drop_threshold(p) : 0u;
  float ik = 1.f / (1.f - p);
  long waves = (long)T * B * H;
  dim3 grid((unsigned)((waves + 3) / 4));
  if (dtype == TELL_BF16) {
    hipLaunchKernelGGL((dynconv_bwd_taps_kernel<uint16_t>), grid, dim3(256), 0, stream, (const uint16_t*)x, (const uint16_t*)dy, taps, (uint16_t*)dlogits, T, B, H, K, R, thr, ik, seed, salt, g_tell_rng_step);
    hipLaunchKernelGGL((dynconv_bwd_x_kernel<uint16_t>), grid, dim3(256), 0, stream, (const uint16_t*)dy, taps, (uint16_t*)dx, dx_accumulate, T, B, H, K, R, thr, ik, seed, salt, g_tell_rng_step);
  } else {
    hipLaunchKernelGGL((dynconv_bwd_taps_kernel<float>), grid, dim3(256), 0, stream, (const float*)x, (const float*)dy, taps, (float*)dlogits, T, B, H, K, R, thr, ik, seed, salt, g_tell_rng_step);
    hipLaunchKernelGGL((dynconv_bwd_x_kernel<float>), grid, dim3(256), 0, stream, (const float*)dy, taps, (float*)dx, dx_accumulate, T, B, H, K, R, thr, ik, seed, salt, g_tell_rng_step);
  }
  return tell_check_launch("dynconv_bwd");
}
