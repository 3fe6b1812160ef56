// DynamicConv core (tell/modules/convolutions/dynamic.py:285-336), T x B x C layout:
//   taps  = softmax_K(tap_logits[t,b,h,:])            (over ALL K taps, :302-304)
//   tapsd = DropConnect(taps)                          (:305)
//   y[t,b,h*R+c] = sum_k tapsd[k] * x[t-(K-1)+k, b, h*R+c],   x[<0] = 0  (causal)
// The reference builds a dense [B*H,T,T+K-1] band matrix and bmm's it; here each
// wave owns one (t,b,h) and gathers its K input rows directly (coalesced 64-lane
// rows; HBM-bound: AI = 2K / ((2 + H*K/C) * sizeof) flop/byte).
// Softmax over the taps is done with wavefront shuffles (lane k holds tap k).
#include "common.h"
#include "options.h"
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
  const bool off = tell_opt(OPT_DYNCONV_LDS) == 0;     // A/B switch
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

// ---------------------------------------------------------------- fused conv-block core (round 4)
// decoder_faces_objects.py:259-261 + dynamic.py:285-336 as ONE launch:  gl = GLU(h1);  tap logits = gl . W_tap^T;
// taps = softmax_K;  y = DropConnect(taps) (*) gl.   The unfused path is three launches (tell_glu_fwd, the tap-logit
// GEMM, tell_dynconv_fwd) that write and re-read gl (2 x 2 MB) and the logits (1 MB) and each pay ~5-13 us of launch +
// one memory round trip for ~1 us of work.  Here one workgroup owns (b, a group of DCB_HG heads): it stages the WHOLE
// GLU output of its batch column - [T <= 32, E] bf16, 66 KB of LDS; the tap logits of a head reduce over all E channels -
// from h1 (the 8 workgroups of a batch column repeat that read, 128 KB each, through the L2 of the one XCD they share), computes the [32, HG*K <= 64] tap
// logits with v_mfma_f32_32x32x16_bf16 - the four waves split E, B fragments come straight from W_tap in global memory
// (as the generation step's skinny kernel does), partial tiles meet in LDS - and then runs the softmax / DropConnect /
// K-tap sum of the LDS kernel above on its heads' 64-channel columns of the staged tile.  Written: gl (its own
// channels; backward needs it), y, the softmax taps (fp32, what tell_dynconv_bwd reads).  The logits never leave the
// chip and stay fp32 (the unfused path rounds them to bf16 between the GEMM and the softmax).
// Algorithmic bytes at T = B = 32, E = 1024, K = 31: h1 4.2 MB + W_tap 1 MB read, gl 2.1 + y 2.1 + taps 2.0 MB written.
#define DCB_HG 2
#define DCB_T 32
typedef __attribute__((ext_vector_type(4))) unsigned int dcb_u32x4;
template <int E>
__global__ __launch_bounds__(256) void dynconv_block_fwd_kernel(const uint16_t* __restrict__ h1,
                                                                const uint16_t* __restrict__ wtap,
                                                                uint16_t* __restrict__ gl, uint16_t* __restrict__ y,
                                                                float* __restrict__ taps, int Tn, int B, int H, int K,
                                                                uint32_t thr, float inv_keep, uint32_t seed, uint32_t salt,
                                                                const uint32_t* __restrict__ step, int abl) {
  constexpr int GS = E + 8;                                  // LDS row stride (elements): 16 bytes of skew per row
  constexpr int CPR = E / 8;                                 // 16-byte chunks per row
  constexpr int NP = 64, PS = NP + 1;                        // tap columns of a head group (padded), partial-tile row stride
  extern __shared__ __attribute__((aligned(16))) unsigned char dcb_sm[];
  uint16_t* gs = reinterpret_cast<uint16_t*>(dcb_sm);                        // [32][GS] bf16
  float* part = reinterpret_cast<float*>(dcb_sm + (size_t)DCB_T * GS * 2);   // [4][32][PS]
  salt = tell_step_salt(salt, step);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // workgroup -> (b, head group): the `groups` workgroups of one batch column share its h1 rows, so they sit on ONE XCD
  // (blockIdx & 7) and the column comes out of HBM once, not once per XCD's L2
  const int groups = H / DCB_HG;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int b = (slot / groups) * 8 + xcd, h0 = (slot % groups) * DCB_HG;
  if (b >= B) return;
  const int c_lo = h0 * DC_R, c_hi = c_lo + DCB_HG * DC_R;
  // ---- B fragments of this wave's first tap block: in flight while the GLU tile is staged
  const int kw = wave * (E / 4);                                             // this wave reduces channels [kw, kw + E/4)
  const int n_rows = H * K;
  const int n0 = h0 * K;
  auto wrow = [&](int nb) -> const uint16_t* {
    int n = n0 + nb * 32 + (lane & 31);
    n = n < n_rows ? n : n_rows - 1;                                         // (columns past the group are never read back)
    return wtap + (long)n * E + kw + 8 * (lane >> 5);
  };
  constexpr int KS = E / 4 / 16;                                             // k steps of 16 per wave
  dcb_u32x4 bf0[KS], bf1[KS];
  {
    const uint16_t* w0 = wrow(0);
    const uint16_t* w1 = wrow(1);
#pragma unroll
    for (int s = 0; s < KS; ++s) bf0[s] = *reinterpret_cast<const dcb_u32x4*>(w0 + 16 * s);
#pragma unroll
    for (int s = 0; s < KS; ++s) bf1[s] = *reinterpret_cast<const dcb_u32x4*>(w1 + 16 * s);
  }
  // ---- GLU of the batch column into LDS (rows >= Tn: zeros), own channels to HBM.  Chunk i = tid + 256 j is row
  // 2 j + (tid >> 7), 16-byte piece tid & 127 (E = 1024): 16 pieces per thread, loaded 8 at a time (16 loads in flight -
  // issued one piece per loop trip the 16 round trips through L2 were most of the kernel)
  static_assert(CPR == 128 && DCB_T * CPR == 16 * 256, "thread -> chunk mapping below");
  {
    const int ch = tid & 127, tr = tid >> 7;
    const bool own = ch * 8 >= c_lo && ch * 8 < c_hi;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      uint4 va[8], vg[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int t = 2 * (8 * half + j) + tr;
        const uint16_t* src = h1 + ((long)(t < Tn ? t : 0) * B + b) * (2 * E) + ch * 8;
        va[j] = *reinterpret_cast<const uint4*>(src);
        vg[j] = *reinterpret_cast<const uint4*>(src + E);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int t = 2 * (8 * half + j) + tr;
        float a[8], g[8], r[8];
        unpack16(va[j], a, (const uint16_t*)nullptr);
        unpack16(vg[j], g, (const uint16_t*)nullptr);
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (abl & 1) ? a[e] : tell_glu(a[e], g[e]);           // (glu_fwd_kernel's arithmetic)
        uint4 o = pack16(r, (const uint16_t*)nullptr);
        if (t >= Tn) o = make_uint4(0u, 0u, 0u, 0u);
        else if (own) *reinterpret_cast<uint4*>(gl + ((long)t * B + b) * E + ch * 8) = o;
        *reinterpret_cast<uint4*>(gs + t * GS + ch * 8) = o;
      }
    }
  }
  __syncthreads();
  // ---- tap logits: D[t][n] partial over this wave's channels; lane holds column n = lane & 31, rows acc_row(r)
  if (!(abl & 2)) {
    const uint16_t* arow = gs + (lane & 31) * GS + kw + 8 * (lane >> 5);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(arow + 16 * s);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, bf0[s]), acc0, 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(arow + 16 * s);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, bf1[s]), acc1, 0, 0, 0);
    }
    float* pw = part + wave * (DCB_T * PS);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      pw[t * PS + (lane & 31)] = acc0[r];
      pw[t * PS + 32 + (lane & 31)] = acc1[r];
    }
  }
  __syncthreads();
  // ---- softmax over the taps + DropConnect: four threads per (head, row), eight taps each, folded with quad DPP moves
  // (the stand-alone kernels reduce a row with 12 dependent ds_bpermute shuffles - hidden there by twelve waves per CU,
  // 11 us here at one wave per SIMD).  The dropped, boundary-zeroed taps go to LDS for the tap sums below.
  float* wds = part + 4 * DCB_T * PS;                                         // [2 heads x 32 rows][32 taps]
  {
    const int r = tid >> 2, q = tid & 3, hl_ = r >> 5, t = r & 31;
    const long wid = ((long)(t < Tn ? t : 0) * B + b) * H + h0 + hl_;
    float lg[8], mx = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int k = q + 4 * jj, n = hl_ * K + (k < K ? k : 0);
      const float v = part[t * PS + n] + part[(DCB_T + t) * PS + n] + part[(2 * DCB_T + t) * PS + n] + part[(3 * DCB_T + t) * PS + n];
      lg[jj] = k < K ? v : -INFINITY;
      mx = fmaxf(mx, lg[jj]);
    }
    mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mx), 0xB1, 0xf, 0xf, true)));   // lane ^ 1
    mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mx), 0x4E, 0xf, 0xf, true)));   // lane ^ 2
    float sm = 0.f;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      lg[jj] = (q + 4 * jj) < K ? __expf(lg[jj] - mx) : 0.f;
      sm += lg[jj];
    }
    sm += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sm), 0xB1, 0xf, 0xf, true));
    sm += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sm), 0x4E, 0xf, 0xf, true));
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int k = q + 4 * jj;
      float v = 0.f;
      if (k < K && t < Tn) {
        const float w = lg[jj] / sm;
        taps[wid * K + k] = w;
        v = w;
        if (thr) v *= tell_keep(seed, salt, (uint64_t)(wid * K + k), thr, inv_keep);
        if (k < K - 1 - t) v = 0.f;                                           // taps that reach before t = 0
      }
      wds[r * 32 + k] = v;
    }
  }
  __syncthreads();
  // ---- K-tap sums.  A wave owns one head and 16 consecutive rows, four rows at a time: row t0 + i at tap k reads
  // x[t0 + i - (K-1) + k] - a four-row window that slides by ONE row per tap, so a tap step is one LDS read, four
  // v_readlane and four FMAs with four independent chains in flight.  Taps that would reach before t = 0 were zeroed
  // above instead of skipped; their (clamped) reads hit row 0, which those rows also see through a live tap.
  static_assert(DCB_HG == 2, "wave -> (head, row block) mapping below");
  const int C = E;
  const int hl = wave & 1, h = h0 + hl;
  const uint16_t* xcol = gs + h * DC_R + lane;
  auto ldx = [&](int r) -> float {
    r = r < 0 ? 0 : (r > DCB_T - 1 ? DCB_T - 1 : r);
    return bf2f(xcol[r * GS]);
  };
  if (!(abl & 4))
  for (int t0 = (wave >> 1) * 16; t0 < (wave >> 1) * 16 + 16 && t0 < Tn; t0 += 4) {
    float wd[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wd[i] = wds[(hl * 32 + t0 + i) * 32 + (lane & 31)];       // (lanes 32 .. 63 are never selected)
    const int base = t0 - (K - 1);
    float x0 = ldx(base), x1 = ldx(base + 1), x2 = ldx(base + 2), x3 = ldx(base + 3);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    // taps in blocks of 8 (lanes K .. 63 of wd hold zeros, so running to the next multiple of 8 adds zeros): the block's
    // eight new rows are read together - fetched one per tap step, each read's latency sat in front of the next step
    for (int kb = 0; kb < K; kb += 8) {
      float xn[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) xn[j] = ldx(base + kb + 4 + j);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a0 += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wd[0]), kb + j)) * x0;
        a1 += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wd[1]), kb + j)) * x1;
        a2 += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wd[2]), kb + j)) * x2;
        a3 += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wd[3]), kb + j)) * x3;
        x0 = x1; x1 = x2; x2 = x3; x3 = xn[j];
      }
    }
    const float av[4] = {a0, a1, a2, a3};
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (t0 + i < Tn) y[((long)(t0 + i) * B + b) * C + (long)h * DC_R + lane] = f2bf(av[i]);
  }
}

// h1 [T*B, 2E] bf16 (linear1's output, a | gate), w_tap [H*K, E] bf16 (DynamicConv weight_linear, no bias) ->
// gl [T*B, E], y [T*B, E] bf16, taps [T*B*H, K] fp32.  -> TELL_OK, or 1 when the shape is not one this kernel takes
// (T <= 32, K <= 32, head width 64, E = 1024, even head count): the caller then runs the three separate launches.
extern "C" int tell_dynconv_block_fwd(const void* h1, const void* w_tap, void* gl, void* y, float* taps, int T, int B,
                                      int H, int K, float p, uint32_t seed, uint32_t salt, hipStream_t stream) {
  if ((long)T * B * H <= 0) return TELL_OK;
  TELL_REQUIRE(p >= 0.f && p < 1.f, "dynconv_block: p must be in [0,1)");
  const bool off = tell_opt(OPT_DYNCONV_BLOCK) == 0;      // A/B switch
  const int E = H * DC_R;
  if (off || T > DCB_T || K < 1 || K > DC_KMAX || DCB_HG * K > 64 || E != 1024 || H % DCB_HG || !taps ||
      ((((uintptr_t)h1 | (uintptr_t)w_tap | (uintptr_t)gl | (uintptr_t)y) & 15) != 0))
    return 1;
  const uint32_t thr = p > 0.f ? tell_drop_threshold(p) : 0u;
  const float ik = 1.f / (1.f - p);
  const size_t smem = (size_t)DCB_T * (1024 + 8) * 2 + (size_t)4 * DCB_T * 65 * sizeof(float) + (size_t)DCB_HG * DCB_T * 32 * sizeof(float);
  static const bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dynconv_block_fwd_kernel<1024>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
    return true;
  }();
  (void)attr;
  const int abl = (int)tell_probe(PROBE_DCB_ABL);      // timing probe (wrong results): probe build only
  hipLaunchKernelGGL((dynconv_block_fwd_kernel<1024>), dim3((B + 7) / 8 * 8 * (H / DCB_HG)), dim3(256), smem, stream,
                     (const uint16_t*)h1, (const uint16_t*)w_tap, (uint16_t*)gl, (uint16_t*)y, taps, T, B, H, K, thr, ik, seed,
                     salt, g_tell_rng_step, abl);
  return tell_check_launch("dynconv_block_fwd");
}
