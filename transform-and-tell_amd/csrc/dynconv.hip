// DynamicConv core (tell/modules/convolutions/dynamic.py:285-336), T x B x C layout:
//   taps  = softmax_K(tap_logits[t,b,h,:])            (over ALL K taps, :302-304)
//   tapsd = DropConnect(taps)                          (:305)
//   y[t,b,h*R+c] = sum_k tapsd[k] * x[t-(K-1)+k, b, h*R+c],   x[<0] = 0  (causal)
// The reference builds a dense [B*H,T,T+K-1] band matrix and bmm's it; here each
// wave owns one (t,b,h) and gathers its K input rows directly (coalesced 64-lane
// rows; HBM-bound: AI = 2K / ((2 + H*K/C) * sizeof) flop/byte).
// Softmax over the taps is done with wavefront shuffles (lane k holds tap k).
#include "common.h"
#include <stdlib.h>

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


// ---------------------------------------------------------------- LDS-tiled variants (head width R = 64, T <= 128)
// One workgroup per (b, h): the head's [T, 64] slice of x (post-GLU) is staged ONCE in LDS with 16-byte loads (fp32 in
// LDS), all T outputs come from the staged tile - the K-fold re-read of x through L2 of the wave-per-row kernel above
// is gone.  Tap softmax by wavefront shuffles (lane k = tap k) as before.  Backward is ONE pass: x, dy and the
// DropConnect-ed taps of the head live in LDS, every wave produces the tap-logit gradients of its rows and the dx of
// its rows (the two-kernel version read dy K times and the taps K times from L2).
#define DC_R 64
#define DC_TMAX 128
#define DC_KMAX 32
template <typename T, int STRIDE>
__device__ __forceinline__ void dc_stage(const T* __restrict__ src, float* __restrict__ dst, int Tn, int B, int b, int C,
                                         int h, int tid) {
  constexpr int VEC = Elem<T>::VEC, CPR = DC_R / VEC;          // 16-byte chunks per 64-channel row
  for (int i = tid; i < Tn * CPR; i += 256) {
    const int t = i / CPR, ch = i % CPR;
    float v[VEC];
    unpack16(*reinterpret_cast<const uint4*>(src + ((long)t * B + b) * C + (long)h * DC_R + ch * VEC), v, (const T*)nullptr);
#pragma unroll
    for (int k = 0; k < VEC; ++k) dst[t * STRIDE + ch * VEC + k] = v[k];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dynconv_fwd_lds_kernel(const T* __restrict__ x, const T* __restrict__ logits,
                                                              T* __restrict__ y, float* __restrict__ taps, int Tn, int B,
                                                              int H, int K, uint32_t thr, float inv_keep, uint32_t seed,
                                                              uint32_t salt, const uint32_t* __restrict__ step) {
  __shared__ float xs[DC_TMAX * DC_R];
  __shared__ float lgs[DC_TMAX * DC_KMAX];      // the head's tap logits: one staging round trip, not one per row
  salt = tell_step_salt(salt, step);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x % H, C = H * DC_R;
  dc_stage<T, DC_R>(x, xs, Tn, B, b, C, h, tid);
  // thread -> (row tid / 32 + 8 u, tap tid % 32): four rows' loads in flight per thread, no divisions (K <= 32)
  for (int t0 = 0; t0 < Tn; t0 += 32) {
    float v[4];
    const int k = tid & 31;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u * 8 + (tid >> 5);
      v[u] = (t < Tn && k < K) ? Elem<T>::ld(logits + ((long)t * B + b) * (long)H * K + (long)h * K + k) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u * 8 + (tid >> 5);
      if (t < Tn && k < K) lgs[t * DC_KMAX + k] = v[u];
    }
  }
  __syncthreads();
  for (int t = wave; t < Tn; t += 4) {
    const long tb = (long)t * B + b, wid = tb * H + h;
    const float lg = lane < K ? lgs[t * DC_KMAX + lane] : -INFINITY;
    const float m = wave_max(lg);
    const float e = lane < K ? __expf(lg - m) : 0.f;
    const float w = e / wave_sum(e);
    if (lane < K && taps) taps[wid * K + lane] = w;
    float wd = w;
    if (thr && lane < K) wd *= tell_keep(seed, salt, (uint64_t)(wid * K + lane), thr, inv_keep);
    const int k_lo = (K - 1 - t) > 0 ? (K - 1 - t) : 0;       // taps reaching before t = 0 see zeros
    // (K is a run-time value: without the unroll hint every tap was its own LDS round trip - 60 clk per tap and row with
    //  two waves per SIMD to hide it; eight taps' reads are now in flight together)
    float acc = 0.f;
#pragma unroll 8
    for (int k = k_lo; k < K; ++k)                       // (k is wave-uniform: v_readlane, not an LDS permute)
      acc += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wd), k)) * xs[(t - (K - 1) + k) * DC_R + lane];
    Elem<T>::st(y + tb * C + (long)h * DC_R + lane, acc);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dynconv_bwd_lds_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                              const float* __restrict__ taps, T* __restrict__ dx,
                                                              int dx_accumulate, T* __restrict__ dlogits, int Tn, int B,
                                                              int H, int K, uint32_t thr, float inv_keep, uint32_t seed,
                                                              uint32_t salt, const uint32_t* __restrict__ step) {
  constexpr int XS = DC_R + 1;             // x rows padded by one float: lane k walks row t-(K-1)+k, column c -> K banks
  extern __shared__ float sm[];
  float* xs = sm;                          // [Tn][65]
  float* dys = xs + Tn * XS;               // [Tn][64]
  float* tw = dys + Tn * DC_R;             // [Tn][K]  softmax taps
  float* tk = tw + Tn * K;                 // [Tn][K]  DropConnect keep factors (0 or 1/(1-p))
  salt = tell_step_salt(salt, step);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x % H, C = H * DC_R;
  dc_stage<T, XS>(x, xs, Tn, B, b, C, h, tid);
  dc_stage<T, DC_R>(dy, dys, Tn, B, b, C, h, tid);
  for (int t0 = 0; t0 < Tn; t0 += 32) {                 // as in the forward kernel: (row, tap) from the thread index
    float v[4];
    const int k = tid & 31;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u * 8 + (tid >> 5);
      v[u] = (t < Tn && k < K) ? taps[(((long)t * B + b) * H + h) * K + k] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u * 8 + (tid >> 5);
      if (t < Tn && k < K) {
        const long wid = ((long)t * B + b) * H + h;
        tw[t * K + k] = v[u];
        tk[t * K + k] = thr ? tell_keep(seed, salt, (uint64_t)(wid * K + k), thr, inv_keep) : 1.f;
      }
    }
  }
  __syncthreads();
  for (int t = wave; t < Tn; t += 4) {
    const long tb = (long)t * B + b;
    // ---- tap-logit gradients of row t: lane k owns dtapd[k] = <dy[t,:], x[t-(K-1)+k,:]> (dy broadcast, x rows on
    //      distinct banks): no cross-lane reduction per tap; then DropConnect and the softmax backward by shuffles
    const int row = t - (K - 1) + lane;
    float dw = 0.f;
    if (lane < K && row >= 0) {
      const float* xr = xs + row * XS;
      const float* dr = dys + t * DC_R;
#pragma unroll 8
      for (int c = 0; c < DC_R; ++c) dw += dr[c] * xr[c];
    }
    const float w = lane < K ? tw[t * K + lane] : 0.f;
    if (lane < K) dw *= tk[t * K + lane];
    const float dot = wave_sum(w * dw);
    if (lane < K) Elem<T>::st(dlogits + tb * (long)H * K + (long)h * K + lane, w * (dw - dot));
    // ---- dx of row t: sum_k tapsd[t+(K-1)-k][k] * dy[t+(K-1)-k][:]   (lane = channel)
    float acc = 0.f;
    const int n_src = Tn - t < K ? Tn - t : K;            // rows t .. t + n_src - 1 reach row t through tap K - 1 - j
#pragma unroll 8
    for (int j = 0; j < n_src; ++j) {
      const int tt = t + j, k = K - 1 - j;
      acc += tw[tt * K + k] * tk[tt * K + k] * dys[tt * DC_R + lane];
    }
    T* d = dx + tb * C + (long)h * DC_R + lane;
    Elem<T>::st(d, dx_accumulate ? Elem<T>::ld(d) + acc : acc);
  }
}

static inline bool dc_lds_ok(const void* x, const void* y, int T, int R, int K, int dtype) {
  const int vec = dtype == TELL_BF16 ? 8 : 4;
  static const bool off = getenv("TELL_DYNCONV_LDS") && atoi(getenv("TELL_DYNCONV_LDS")) == 0;     // A/B switch
  return !off && R == DC_R && T <= DC_TMAX && K <= DC_KMAX && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && (DC_R % vec) == 0;
}

extern "C" int tell_dynconv_fwd(const void* x, const void* logits, void* y, float* taps, int T, int B,
                                int H, int K, int R, float p, uint32_t seed, uint32_t salt, int dtype,
                                hipStream_t stream) {
  if ((long)T * B * H <= 0) return TELL_OK;
  TELL_REQUIRE(K >= 1 && K <= 64, "dynconv: kernel size must be in [1,64] (one lane per tap)");
  TELL_REQUIRE(p >= 0.f && p < 1.f, "dynconv: p must be in [0,1)");
  uint32_t thr = p > 0.f ? tell_drop_threshold(p) : 0u;
  float ik = 1.f / (1.f - p);
  if (dc_lds_ok(x, y, T, R, K, dtype) && (H * R) % 8 == 0) {
    if (dtype == TELL_BF16) hipLaunchKernelGGL((dynconv_fwd_lds_kernel<uint16_t>), dim3(B * H), dim3(256), 0, stream, (const uint16_t*)x, (const uint16_t*)logits, (uint16_t*)y, taps, T, B, H, K, thr, ik, seed, salt, g_tell_rng_step);
    else hipLaunchKernelGGL((dynconv_fwd_lds_kernel<float>), dim3(B * H), dim3(256), 0, stream, (const float*)x, (const float*)logits, (float*)y, taps, T, B, H, K, thr, ik, seed, salt, g_tell_rng_step);
    return tell_check_launch("dynconv_fwd_lds");
  }
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
  if (dc_lds_ok(x, dy, T, R, K, dtype) && T <= 64 && (H * R) % 8 == 0 && ((uintptr_t)dx & 15) == 0) {   // <= 48 KB of LDS
    const size_t smem = ((size_t)T * (2 * DC_R + 1) + (size_t)2 * T * K) * sizeof(float);
    if (dtype == TELL_BF16) hipLaunchKernelGGL((dynconv_bwd_lds_kernel<uint16_t>), dim3(B * H), dim3(256), smem, stream, (const uint16_t*)x, (const uint16_t*)dy, taps, (uint16_t*)dx, dx_accumulate, (uint16_t*)dlogits, T, B, H, K, thr, ik, seed, salt, g_tell_rng_step);
    else hipLaunchKernelGGL((dynconv_bwd_lds_kernel<float>), dim3(B * H), dim3(256), smem, stream, (const float*)x, (const float*)dy, taps, (float*)dx, dx_accumulate, (float*)dlogits, T, B, H, K, thr, ik, seed, salt, g_tell_rng_step);
    return tell_check_launch("dynconv_bwd_lds");
  }
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
