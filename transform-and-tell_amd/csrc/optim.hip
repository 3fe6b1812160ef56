// Multi-tensor BertAdam over flat fp32 buffers (optimizer `bert_adam` of
// expt/nytimes/9_transformer_objects/config.yaml:126-149; update rule of
// pytorch_pretrained_bert.BertAdam - third-party, restated, parity unpinned):
//   per tensor:  g <- g * min(1, max_grad_norm / (||g||_2 + 1e-6))       (per-TENSOR clip)
//   m <- b1 m + (1-b1) g ;  v <- b2 v + (1-b2) g^2      (no bias correction)
//   p <- p - lr_t * ( m / (sqrt(v) + eps) + wd * p )
// Layout: every tensor starts at a multiple of CHUNK elements inside the flat
// buffers (zero padded), so a chunk belongs to exactly one tensor.  HBM-bound:
// 4 reads + 3 writes of fp32 per element.
#include "common.h"
#include "options.h"

#define OPT_CHUNK 1024
// MEASURED (MI355X, same box, round 5): on bare buffers of the decoder's size (tools/probes/stream_probe.hip, 176 M
// parameters, 34 B each) one chunk per iteration 1108 us = 5.66 TB/s, four chunks per iteration 1032 us, four chunks + non-
// temporal loads / stores 1003 us = 6.26 TB/s (a float4 copy on that box: 5.3-5.7 TB/s); decoder half of the step alone
// (tools/decoder_profile.py, TELL_ADAM_VAR = 0..4): 6.93 / 6.82 / 6.75 / 6.80 / 6.73 ms.
#define TELL_ADAM_DEFAULT 4
#include <stdlib.h>

// partial[c] = sum of squares of (grad * grad_scale) over chunk c
// (wire != NULL: the gradient of this step is the bf16 buffer the data-parallel exchange left - read it as it is
//  instead of widening 2 B/param back into the fp32 buffer first)
__device__ __forceinline__ float4 load_grad4(const float* __restrict__ grad, const uint16_t* __restrict__ wire, long o) {
  if (wire) {
    const uint2 w = reinterpret_cast<const uint2*>(wire)[o];
    return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                       __uint_as_float(w.y & 0xffff0000u));
  }
  return reinterpret_cast<const float4*>(grad)[o];
}
// U chunks per block iteration, every load of the iteration in flight before the first reduction (one chunk per iteration
// left a block with a single 16-byte load outstanding per thread between two barriers)
template <int U>
__global__ __launch_bounds__(256) void sqsum_chunks_kernel(const float* __restrict__ grad, const uint16_t* __restrict__ wire,
                                                           long n_chunks, float grad_scale, float* __restrict__ partial) {
  __shared__ float red[U][4];
  for (long c0 = (long)blockIdx.x * U; c0 < n_chunks; c0 += (long)gridDim.x * U) {
    float4 g[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      g[u] = c0 + u < n_chunks ? load_grad4(grad, wire, (c0 + u) * OPT_CHUNK / 4 + threadIdx.x) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float s = (g[u].x * g[u].x + g[u].y * g[u].y + g[u].z * g[u].z + g[u].w * g[u].w) * grad_scale * grad_scale;
      s = wave_sum(s);
      if ((threadIdx.x & 63) == 0) red[u][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x < U && c0 + threadIdx.x < n_chunks)
      partial[c0 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    __syncthreads();
  }
}
// norms[t] = sqrt(sum of the tensor's chunk partials); one 256-thread block per tensor, four loads in flight per thread,
// partial sums folded in a fixed order (deterministic).  (One wave per tensor walked the 50 k chunks of the embedding
// table one dependent load at a time: 145 us.)
__global__ __launch_bounds__(256) void tensor_norms_kernel(const float* __restrict__ partial,
                                                           const long* __restrict__ chunk_begin,
                                                           int n_tensors, float* __restrict__ norms,
                                                           int* __restrict__ skip) {
  __shared__ float red[4];
  const int t = blockIdx.x, lane = threadIdx.x & 63;
  const long c0 = chunk_begin[t], c1 = chunk_begin[t + 1];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  long c = c0 + threadIdx.x;
  for (; c + 768 < c1; c += 1024) {
    s0 += partial[c]; s1 += partial[c + 256]; s2 += partial[c + 512]; s3 += partial[c + 768];
  }
  for (; c < c1; c += 256) s0 += partial[c];
  float s = wave_sum((s0 + s1) + (s2 + s3));
  if (lane == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    s = (red[0] + red[1]) + (red[2] + red[3]);
    norms[t] = sqrtf(s);
    if (skip && !(fabsf(s) <= 3.4e38f)) atomicOr(skip, 2);      // NaN / Inf gradient (apex O2 skips such steps)
  }
}
// skip[0] = 1 if the loss is NaN / Inf (callback_apex_trainer.py:225-227 skips the batch), else 0
__global__ void loss_flag_kernel(const float* __restrict__ loss, int* __restrict__ skip) {
  skip[0] = (fabsf(loss[0]) <= 3.4e38f) ? 0 : 1;
}
extern "C" int tell_loss_flag(const float* loss, int* skip, hipStream_t stream) {
  hipLaunchKernelGGL(loss_flag_kernel, dim3(1), dim3(1), 0, stream, loss, skip);
  return tell_check_launch("loss_flag");
}
// warmup_linear (pytorch_pretrained_bert WarmupLinearSchedule.get_lr_) evaluated on the device from a device step counter
// that only SUCCESSFUL updates advance: a batch skipped for a non-finite loss / gradient never reached optimizer.step()
// in the reference (callback_apex_trainer.py:225-227), so it must not cost a tick of the schedule either - and the host
// cannot know about the skip without a synchronisation.  t_total <= 0: constant learning rate.
__global__ void lr_schedule_kernel(const int* __restrict__ step, float lr, float warmup, float t_total,
                                   float* __restrict__ lr_dev) {
  double f = 1.0;
  if (t_total > 0.f) {
    const double x = (double)step[0] / (double)t_total, w = (double)warmup;
    f = x < w ? x / w : fmax((x - 1.0) / (w - 1.0), 0.0);
  }
  lr_dev[0] = (float)((double)lr * f);
}
// One pass over the flat buffers: BertAdam update of the fp32 masters, the bf16 working copy the next forward
// reads (shadow; no per-tensor cast kernels), and the zeroing of the gradient for the next step.
// U: chunks per block iteration (all 4 U loads of a thread in flight before the first use); NT: the streams nobody reads
// again soon (master, m, v, the zeroed gradient) go out with non-temporal stores and come in with non-temporal loads -
// the bf16 shadow, which the next forward pass reads, keeps the default policy.
template <int U, bool NT>
__global__ __launch_bounds__(256) void bertadam_update_kernel(float* __restrict__ param,
                                                              float* __restrict__ grad,
                                                              float* __restrict__ m, float* __restrict__ v,
                                                              const int* __restrict__ chunk_tensor,
                                                              const float* __restrict__ norms, long n_chunks,
                                                              const float* __restrict__ lr_dev, float b1,
                                                              float b2, float eps, float wd, float max_norm,
                                                              float grad_scale, uint16_t* __restrict__ shadow,
                                                              int zero_grad, int* __restrict__ skip,
                                                              const uint16_t* __restrict__ wire,
                                                              int* __restrict__ step_dev,
                                                              const int* __restrict__ keep_grad) {
  typedef __attribute__((ext_vector_type(4))) float f4;
  const float lr = *lr_dev;
  // keep_grad[t] != 0: tensor t's gradient is rewritten whole (beta = 0 stores) by its single producer in the next
  // backward pass - not zeroed here (4 B / parameter less to write, and the producer does not read it back)
  if (skip && skip[0] != 0) {      // non-finite loss or gradient: leave p, m, v and the shadow alone, only clear the gradient
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(skip + 1, 1);          // running count of skipped steps
    if (zero_grad)
      for (long c = blockIdx.x; c < n_chunks; c += gridDim.x)
        if (!keep_grad || !keep_grad[chunk_tensor[c]])
          reinterpret_cast<float4*>(grad)[c * OPT_CHUNK / 4 + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  if (step_dev && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(step_dev, 1);   // (nobody reads it inside this launch)
  auto ldf = [](const float* base, long o) __attribute__((always_inline)) {
    const f4* q = reinterpret_cast<const f4*>(base) + o;
    return NT ? __builtin_nontemporal_load(q) : *q;
  };
  auto stf = [](float* base, long o, f4 val) __attribute__((always_inline)) {
    f4* q = reinterpret_cast<f4*>(base) + o;
    if (NT) __builtin_nontemporal_store(val, q); else *q = val;
  };
  for (long c0 = (long)blockIdx.x * U; c0 < n_chunks; c0 += (long)gridDim.x * U) {
    f4 g[U], p[U], mm[U], vv[U];
    float coef[U];
    int tensor[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long c = c0 + u < n_chunks ? c0 + u : n_chunks - 1;               // (clamped: the tail repeats the last chunk's loads)
      const long o = c * OPT_CHUNK / 4 + threadIdx.x;
      if (wire) { const float4 w = load_grad4(grad, wire, o); g[u] = f4{w.x, w.y, w.z, w.w}; }
      else g[u] = ldf(grad, o);
      p[u] = ldf(param, o); mm[u] = ldf(m, o); vv[u] = ldf(v, o);
      tensor[u] = chunk_tensor[c];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      coef[u] = grad_scale;
      if (max_norm > 0.f) {
        const float cc = max_norm / (norms[tensor[u]] + 1e-6f);
        if (cc < 1.f) coef[u] *= cc;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (c0 + u >= n_chunks) break;
      const long o = (c0 + u) * OPT_CHUNK / 4 + threadIdx.x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gg = g[u][k] * coef[u];
        mm[u][k] = b1 * mm[u][k] + (1.f - b1) * gg;
        vv[u][k] = b2 * vv[u][k] + (1.f - b2) * gg * gg;
        p[u][k] -= lr * (mm[u][k] / (sqrtf(vv[u][k]) + eps) + wd * p[u][k]);
      }
      stf(param, o, p[u]); stf(m, o, mm[u]); stf(v, o, vv[u]);
      if (shadow) {
        uint2 sw;
        sw.x = (uint32_t)f2bf(p[u][0]) | ((uint32_t)f2bf(p[u][1]) << 16);
        sw.y = (uint32_t)f2bf(p[u][2]) | ((uint32_t)f2bf(p[u][3]) << 16);
        reinterpret_cast<uint2*>(shadow)[o] = sw;
      }
      if (zero_grad && !(keep_grad && keep_grad[tensor[u]])) stf(grad, o, f4{0.f, 0.f, 0.f, 0.f});
    }
  }
}

extern "C" int tell_opt_chunk(void) { return OPT_CHUNK; }

// workspace `partial`: n_chunks floats; `norms`: n_tensors floats
// keep_grad: device int32[n_tensors] or NULL - tensors whose gradient the zeroing leaves alone (see the kernel)
extern "C" int tell_bertadam_step2(float* param, float* grad, float* m, float* v,
                                   const int* chunk_tensor, const long* chunk_begin, long n_chunks,
                                   int n_tensors, float* partial, float* norms, const float* lr_dev,
                                   float b1, float b2, float eps, float wd, float max_norm,
                                   float grad_scale, void* shadow_bf16, int zero_grad, int* skip,
                                   const void* grad_wire_bf16, int* step_dev, float lr_base, float warmup, float t_total,
                                   const int* keep_grad, hipStream_t stream) {
  if (n_chunks <= 0) return TELL_OK;
  if (step_dev)      // device-side schedule: this step's learning rate from the count of updates applied so far
    hipLaunchKernelGGL(lr_schedule_kernel, dim3(1), dim3(1), 0, stream, step_dev, lr_base, warmup, t_total,
                       const_cast<float*>(lr_dev));
  TELL_REQUIRE(((uintptr_t)grad_wire_bf16 & 7) == 0, "bertadam: the bf16 gradient must be 8-byte aligned");
  const uint16_t* wire = static_cast<const uint16_t*>(grad_wire_bf16);
  TELL_REQUIRE(((uintptr_t)param & 15) == 0 && ((uintptr_t)grad & 15) == 0, "bertadam: buffers must be 16-byte aligned");
  // TELL_ADAM_VAR (A/B aid, read once): 0 = one chunk per iteration, default cache policy (the round-4 kernel's shape);
  // 1 = U 2; 2 = U 4; 3 = U 2 + non-temporal; 4 = U 4 + non-temporal.  tools/probes/stream_probe.hip has the same shapes
  // on bare buffers.
  const int var = tell_opt(OPT_ADAM_VAR) >= 0 ? (int)tell_opt(OPT_ADAM_VAR) : TELL_ADAM_DEFAULT;
  const int grid_env = tell_opt(OPT_ADAM_GRID) > 0 ? (int)tell_opt(OPT_ADAM_GRID) : 4096;
  const int U = (var == 1 || var == 3) ? 2 : (var == 2 || var == 4) ? 4 : 1;
  const long iters = (n_chunks + U - 1) / U;
  int g = iters < grid_env ? (int)iters : grid_env;
  if (max_norm > 0.f) {
    const long it4 = (n_chunks + 3) / 4;
    const int gs = it4 < 4096 ? (int)it4 : 4096;
    hipLaunchKernelGGL((sqsum_chunks_kernel<4>), dim3(gs), dim3(256), 0, stream, grad, wire, n_chunks, grad_scale, partial);
    hipLaunchKernelGGL(tensor_norms_kernel, dim3(n_tensors), dim3(256), 0, stream, partial, chunk_begin, n_tensors, norms, skip);
  }
#define TELL_ADAM_LAUNCH(UU, NTT) hipLaunchKernelGGL((bertadam_update_kernel<UU, NTT>), dim3(g), dim3(256), 0, stream, param, grad, m, v, chunk_tensor, norms, n_chunks, lr_dev, b1, b2, eps, wd, max_norm, grad_scale, (uint16_t*)shadow_bf16, zero_grad, skip, wire, step_dev, keep_grad)
  switch (var) {
    case 1: TELL_ADAM_LAUNCH(2, false); break;
    case 2: TELL_ADAM_LAUNCH(4, false); break;
    case 3: TELL_ADAM_LAUNCH(2, true); break;
    case 4: TELL_ADAM_LAUNCH(4, true); break;
    default: TELL_ADAM_LAUNCH(1, false); break;
  }
#undef TELL_ADAM_LAUNCH
  return tell_check_launch("bertadam_step");
}
extern "C" int tell_bertadam_step(float* param, float* grad, float* m, float* v,
                                  const int* chunk_tensor, const long* chunk_begin, long n_chunks,
                                  int n_tensors, float* partial, float* norms, const float* lr_dev,
                                  float b1, float b2, float eps, float wd, float max_norm,
                                  float grad_scale, void* shadow_bf16, int zero_grad, int* skip,
                                  const void* grad_wire_bf16, int* step_dev, float lr_base, float warmup, float t_total,
                                  hipStream_t stream) {
  return tell_bertadam_step2(param, grad, m, v, chunk_tensor, chunk_begin, n_chunks, n_tensors, partial, norms, lr_dev, b1, b2,
                             eps, wd, max_norm, grad_scale, shadow_bf16, zero_grad, skip, grad_wire_bf16, step_dev, lr_base,
                             warmup, t_total, nullptr, stream);
}

// fill n floats with a value (grad zeroing without a memset node per tensor)
__global__ void fill_kernel(float* __restrict__ x, long n, float value) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) x[i] = value;
}
extern "C" int tell_fill_f32(float* x, long n, float value, hipStream_t stream) {
  if (n <= 0) return TELL_OK;
  long g = (n + 1023) / 1024;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(fill_kernel, dim3((int)g), dim3(256), 0, stream, x, n, value);
  return tell_check_launch("fill_f32");
}
