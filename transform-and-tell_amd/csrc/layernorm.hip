// Fused  y = LayerNorm(res + dropout(x)) * gamma + beta   (post-LN blocks of the
// decoder, decoder_faces_objects.py:263-266, and RoBERTa's post-LN encoder).
// One wave per row, fp32 statistics, two-pass variance.  Rows are re-read from
// L1/L2 (a 1024-wide bf16 row is 2 KB), so HBM traffic is one read + one write.
#include "common.h"
#include "options.h"
#include <stdlib.h>
// MEASURED (MI355X, tools/bench_layernorm.py, [16384, 1024] bf16 + residual): one row per wave 20.8 us with dropout / 20.7
// without; two rows per wave, loads up front 19.9 / 18.1; the same with non-temporal loads 22.4 / 21.0.
#define TELL_LN_DEFAULT 1

template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, long ld_x,
                                                     const T* __restrict__ res, long ld_r,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     long ld_y, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int rows, int C,
                                                     float eps, uint32_t thr, float inv_keep,
                                                     uint32_t seed, uint32_t salt,
    const uint32_t* __restrict__ step) {
  salt = tell_step_salt(salt, step);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + (long)row * ld_x;
  const T* rr = res ? res + (long)row * ld_r : nullptr;
  auto z = [&](int c) -> float {
    float v = Elem<T>::ld(xr + c);
    if (thr) v *= tell_keep(seed, salt, (uint64_t)row * C + c, thr, inv_keep);
    if (rr) v += Elem<T>::ld(rr + c);
    return v;
  };
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += z(c);
  const float mu = wave_sum(s) / C;
  float q = 0.f;
  for (int c = lane; c < C; c += 64) { float d = z(c) - mu; q += d * d; }
  const float rs = rsqrtf(wave_sum(q) / C + eps);
  if (lane == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
  T* yr = y + (long)row * ld_y;
  for (int c = lane; c < C; c += 64) Elem<T>::st(yr + c, (z(c) - mu) * rs * gamma[c] + beta[c]);
}

// Fast path: C == 64 * VEC * NCH; every lane keeps its NCH 16-byte chunks in registers:
// one coalesced read of x (+res), statistics from registers, one coalesced write.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void ln_fwd_vec_kernel(const T* __restrict__ x, long ld_x,
                                                         const T* __restrict__ res, long ld_r,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, T* __restrict__ y,
                                                         long ld_y, float* __restrict__ mean,
                                                         float* __restrict__ rstd, int rows, float eps,
                                                         uint32_t thr, float inv_keep, uint32_t seed, uint32_t salt,
    const uint32_t* __restrict__ step) {
  salt = tell_step_salt(salt, step);
  constexpr int VEC = Elem<T>::VEC, C = 64 * VEC * NCH;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[NCH][VEC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c0 = (lane + 64 * i) * VEC;
    unpack16(*reinterpret_cast<const uint4*>(x + (long)row * ld_x + c0), v[i], (const T*)nullptr);
    if (thr) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) v[i][k] *= tell_keep(seed, salt, (uint64_t)row * C + c0 + k, thr, inv_keep);
    }
    if (res) {
      float r[VEC];
      unpack16(*reinterpret_cast<const uint4*>(res + (long)row * ld_r + c0), r, (const T*)nullptr);
#pragma unroll
      for (int k = 0; k < VEC; ++k) v[i][k] += r[k];
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) s += v[i][k];
  }
  const float mu = wave_sum(s) / C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int k = 0; k < VEC; ++k) { const float d = v[i][k] - mu; q += d * d; }
  const float rs = rsqrtf(wave_sum(q) / C + eps);
  if (lane == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c0 = (lane + 64 * i) * VEC;
    float o[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) o[k] = (v[i][k] - mu) * rs * gamma[c0 + k] + beta[c0 + k];
    *reinterpret_cast<uint4*>(y + (long)row * ld_y + c0) = pack16(o, (const T*)nullptr);
  }
}

// RPW rows per wave, every 16-byte load of the wave's rows (x and the residual) requested before the first use; NT: the
// two inputs are dead after this launch in the encoder's chain (x = a GEMM output, res = the previous sub-layer's output)
// and come in with non-temporal loads.  Same arithmetic, same reduction order per row as ln_fwd_vec_kernel: bit-identical.
template <typename T, int NCH, int RPW, bool NT>
__global__ __launch_bounds__(256) void ln_fwd_vec_rows_kernel(const T* __restrict__ x, long ld_x,
                                                              const T* __restrict__ res, long ld_r,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, T* __restrict__ y,
                                                              long ld_y, float* __restrict__ mean,
                                                              float* __restrict__ rstd, int rows, float eps,
                                                              uint32_t thr, float inv_keep, uint32_t seed, uint32_t salt,
    const uint32_t* __restrict__ step) {
  typedef __attribute__((ext_vector_type(4))) unsigned int u4;
  salt = tell_step_salt(salt, step);
  constexpr int VEC = Elem<T>::VEC, C = 64 * VEC * NCH;
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
  if (row0 >= rows) return;
  u4 xv[RPW][NCH], rv[RPW][NCH];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int row = row0 + r < rows ? row0 + r : rows - 1;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c0 = (lane + 64 * i) * VEC;
      const u4* px = reinterpret_cast<const u4*>(x + (long)row * ld_x + c0);
      xv[r][i] = NT ? __builtin_nontemporal_load(px) : *px;
      if (res) {
        const u4* pr = reinterpret_cast<const u4*>(res + (long)row * ld_r + c0);
        rv[r][i] = NT ? __builtin_nontemporal_load(pr) : *pr;
      }
    }
  }
  float gm[NCH][VEC], bt[NCH][VEC];
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int k = 0; k < VEC; ++k) { gm[i][k] = gamma[(lane + 64 * i) * VEC + k]; bt[i][k] = beta[(lane + 64 * i) * VEC + k]; }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int row = row0 + r;
    if (row >= rows) break;
    float v[NCH][VEC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c0 = (lane + 64 * i) * VEC;
      unpack16(make_uint4(xv[r][i][0], xv[r][i][1], xv[r][i][2], xv[r][i][3]), v[i], (const T*)nullptr);
      if (thr) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[i][k] *= tell_keep(seed, salt, (uint64_t)row * C + c0 + k, thr, inv_keep);
      }
      if (res) {
        float q[VEC];
        unpack16(make_uint4(rv[r][i][0], rv[r][i][1], rv[r][i][2], rv[r][i][3]), q, (const T*)nullptr);
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[i][k] += q[k];
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) s += v[i][k];
    }
    const float mu = wave_sum(s) / C;
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int k = 0; k < VEC; ++k) { const float d = v[i][k] - mu; q2 += d * d; }
    const float rs = rsqrtf(wave_sum(q2) / C + eps);
    if (lane == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c0 = (lane + 64 * i) * VEC;
      float o[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) o[k] = (v[i][k] - mu) * rs * gm[i][k] + bt[i][k];
      *reinterpret_cast<uint4*>(y + (long)row * ld_y + c0) = pack16(o, (const T*)nullptr);
    }
  }
}

extern "C" int tell_layernorm_fwd(const void* x, long ld_x, const void* res, long ld_r,
                                  const float* gamma, const float* beta, void* y, long ld_y,
                                  float* mean, float* rstd, int rows, int C, float eps, float p,
                                  uint32_t seed, uint32_t salt, int dtype, hipStream_t stream) {
  if (rows <= 0) return TELL_OK;
  TELL_REQUIRE(p >= 0.f && p < 1.f, "layernorm_fwd: p must be in [0,1)");
  uint32_t thr = p > 0.f ? tell_drop_threshold(p) : 0u;
  float ik = 1.f / (1.f - p);
  dim3 grid((rows + 3) / 4);
  const int vec = dtype == TELL_BF16 ? 8 : 4;
  const bool aligned = ld_x % vec == 0 && ld_y % vec == 0 && (res == nullptr || ld_r % vec == 0) &&
                       ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)res & 15) == 0;
#define LNV(T, NCH) hipLaunchKernelGGL((ln_fwd_vec_kernel<T, NCH>), grid, dim3(256), 0, stream, (const T*)x, ld_x, \
    (const T*)res, ld_r, gamma, beta, (T*)y, ld_y, mean, rstd, rows, eps, thr, ik, seed, salt, g_tell_rng_step)
  // TELL_LN_VAR (A/B aid, read per call): 0 = one row per wave; 1 = two rows per wave, loads up front.
  // bf16, C = 1024, >= 4096 rows (the encoder's [B x 512, 1024] rows).
  {
    const int var = tell_opt(OPT_LN_VAR) >= 0 ? (int)tell_opt(OPT_LN_VAR) : TELL_LN_DEFAULT;
    if (var && aligned && dtype == TELL_BF16 && C == 1024 && rows >= 4096) {
      dim3 g2((rows + 7) / 8);
#define LNR(NTT) hipLaunchKernelGGL((ln_fwd_vec_rows_kernel<uint16_t, 2, 2, NTT>), g2, dim3(256), 0, stream, (const uint16_t*)x, ld_x, \
    (const uint16_t*)res, ld_r, gamma, beta, (uint16_t*)y, ld_y, mean, rstd, rows, eps, thr, ik, seed, salt, g_tell_rng_step)
      LNR(false);          // (the non-temporal form lost: see TELL_LN_DEFAULT)
#undef LNR
      return tell_check_launch("layernorm_fwd_rows");
    }
  }
  if (aligned && C % (64 * vec) == 0 && C / (64 * vec) <= 4 && C / (64 * vec) != 3) {
    const int nch = C / (64 * vec);
    if (dtype == TELL_BF16) { if (nch == 1) LNV(uint16_t, 1); else if (nch == 2) LNV(uint16_t, 2); else LNV(uint16_t, 4); }
    else { if (nch == 1) LNV(float, 1); else if (nch == 2) LNV(float, 2); else LNV(float, 4); }
    return tell_check_launch("layernorm_fwd_vec");
  }
#undef LNV
  if (dtype == TELL_BF16)
    hipLaunchKernelGGL((ln_fwd_kernel<uint16_t>), grid, dim3(256), 0, stream, (const uint16_t*)x, ld_x, (const uint16_t*)res, ld_r, gamma, beta, (uint16_t*)y, ld_y, mean, rstd, rows, C, eps, thr, ik, seed, salt, g_tell_rng_step);
  else
    hipLaunchKernelGGL((ln_fwd_kernel<float>), grid, dim3(256), 0, stream, (const float*)x, ld_x, (const float*)res, ld_r, gamma, beta, (float*)y, ld_y, mean, rstd, rows, C, eps, thr, ik, seed, salt, g_tell_rng_step);
  return tell_check_launch("layernorm_fwd");
}

// Backward.  z = res + dropout(x) is recomputed from the saved inputs.
//   dz = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma
//   dres = dz (optionally accumulated into an existing buffer), dx = dz * keep
// dgamma / dbeta: each block reduces its ROWS_PER_BLOCK rows into partial[block][2][C];
// tell_ln_bwd_finish sums the partials (deterministic, no atomics).
#define LN_BWD_ROWS 4
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, long ld_dy,
                                                     const T* __restrict__ x, long ld_x,
                                                     const T* __restrict__ res, long ld_r,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ mean,
                                                     const float* __restrict__ rstd,
                                                     T* __restrict__ dx, long ld_dx,
                                                     T* __restrict__ dres, long ld_dres, int dres_acc,
                                                     float* __restrict__ partial, int rows, int C,
                                                     uint32_t thr, float inv_keep, uint32_t seed, uint32_t salt,
    const uint32_t* __restrict__ step) {
  salt = tell_step_salt(salt, step);
  extern __shared__ float sm[];                 // [4 waves][2][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* my_g = sm + (long)wave * 2 * C;
  float* my_b = my_g + C;
  for (int c = lane; c < C; c += 64) { my_g[c] = 0.f; my_b[c] = 0.f; }
  const int r_begin = blockIdx.x * LN_BWD_ROWS;
  for (int rr_ = wave; rr_ < LN_BWD_ROWS; rr_ += 4) {
    const int row = r_begin + rr_;
    if (row >= rows) break;
    const T* xr = x + (long)row * ld_x;
    const T* rr = res ? res + (long)row * ld_r : nullptr;
    const T* dyr = dy + (long)row * ld_dy;
    const float mu = mean[row], rs = rstd[row];
    auto xhat = [&](int c, float& keep) -> float {
      float v = Elem<T>::ld(xr + c);
      keep = thr ? tell_keep(seed, salt, (uint64_t)row * C + c, thr, inv_keep) : 1.f;
      v *= keep;
      if (rr) v += Elem<T>::ld(rr + c);
      return (v - mu) * rs;
    };
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < C; c += 64) {
      float k; float xh = xhat(c, k);
      float d = Elem<T>::ld(dyr + c);
      float g = d * gamma[c];
      s1 += g; s2 += g * xh;
      my_g[c] += d * xh;                        // lane-private columns -> no race
      my_b[c] += d;
    }
    s1 = wave_sum(s1) / C; s2 = wave_sum(s2) / C;
    for (int c = lane; c < C; c += 64) {
      float k; float xh = xhat(c, k);
      float g = Elem<T>::ld(dyr + c) * gamma[c];
      float dz = rs * (g - s1 - xh * s2);
      if (dres) {
        T* d = dres + (long)row * ld_dres + c;
        Elem<T>::st(d, dres_acc ? Elem<T>::ld(d) + dz : dz);
      }
      if (dx) Elem<T>::st(dx + (long)row * ld_dx + c, dz * k);
    }
  }
  __syncthreads();
  float* out = partial + (long)blockIdx.x * 2 * C;
  for (int c = threadIdx.x; c < 2 * C; c += 256)
    out[c] = sm[c] + sm[2 * C + c] + sm[4 * C + c] + sm[6 * C + c];
}

// Fast path (C == 64 * VEC * NCH, 16-byte aligned rows): one wave per row, every operand read once with
// 16-byte loads and kept in registers; the row's dgamma/dbeta contributions go through LDS once per block.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void ln_bwd_vec_kernel(const T* __restrict__ dy, long ld_dy,
                                                         const T* __restrict__ x, long ld_x,
                                                         const T* __restrict__ res, long ld_r,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ rstd,
                                                         T* __restrict__ dx, long ld_dx,
                                                         T* __restrict__ dres, long ld_dres, int dres_acc,
                                                         float* __restrict__ partial, int rows,
                                                         uint32_t thr, float inv_keep, uint32_t seed, uint32_t salt,
    const uint32_t* __restrict__ step) {
  salt = tell_step_salt(salt, step);
  constexpr int VEC = Elem<T>::VEC, C = 64 * VEC * NCH;
  __shared__ float sm[4][2][C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * LN_BWD_ROWS + wave;
  const bool live = row < rows;                                      // wave-uniform
  float xh[NCH][VEC], g[NCH][VEC], keep[NCH][VEC];
  float s1 = 0.f, s2 = 0.f, rs = 0.f;
  if (live) {
    const float mu = mean[row];
    rs = rstd[row];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c0 = (lane + 64 * i) * VEC;
      float d[VEC];
      unpack16(*reinterpret_cast<const uint4*>(x + (long)row * ld_x + c0), xh[i], (const T*)nullptr);
      unpack16(*reinterpret_cast<const uint4*>(dy + (long)row * ld_dy + c0), d, (const T*)nullptr);
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        keep[i][k] = thr ? tell_keep(seed, salt, (uint64_t)row * C + c0 + k, thr, inv_keep) : 1.f;
        xh[i][k] *= keep[i][k];
      }
      if (res) {
        float r[VEC];
        unpack16(*reinterpret_cast<const uint4*>(res + (long)row * ld_r + c0), r, (const T*)nullptr);
#pragma unroll
        for (int k = 0; k < VEC; ++k) xh[i][k] += r[k];
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        xh[i][k] = (xh[i][k] - mu) * rs;
        g[i][k] = d[k] * gamma[c0 + k];
        s1 += g[i][k];
        s2 += g[i][k] * xh[i][k];
        sm[wave][0][c0 + k] = d[k] * xh[i][k];
        sm[wave][1][c0 + k] = d[k];
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        sm[wave][0][(lane + 64 * i) * VEC + k] = 0.f;
        sm[wave][1][(lane + 64 * i) * VEC + k] = 0.f;
      }
  }
  if (live) {
    s1 = wave_sum(s1) / C;
    s2 = wave_sum(s2) / C;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c0 = (lane + 64 * i) * VEC;
      float dz[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) dz[k] = rs * (g[i][k] - s1 - xh[i][k] * s2);
      if (dres) {
        T* dp = dres + (long)row * ld_dres + c0;
        float o[VEC];
        if (dres_acc) {
          unpack16(*reinterpret_cast<const uint4*>(dp), o, (const T*)nullptr);
#pragma unroll
          for (int k = 0; k < VEC; ++k) o[k] += dz[k];
        } else {
#pragma unroll
          for (int k = 0; k < VEC; ++k) o[k] = dz[k];
        }
        *reinterpret_cast<uint4*>(dp) = pack16(o, (const T*)nullptr);
      }
      if (dx) {
        float o[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) o[k] = dz[k] * keep[i][k];
        *reinterpret_cast<uint4*>(dx + (long)row * ld_dx + c0) = pack16(o, (const T*)nullptr);
      }
    }
  }
  __syncthreads();
  float* out = partial + (long)blockIdx.x * 2 * C;
  const float* s = &sm[0][0][0];
  for (int c = threadIdx.x; c < 2 * C; c += 256) out[c] = s[c] + s[2 * C + c] + s[4 * C + c] + s[6 * C + c];
}

// 64 columns x 16 partial-groups per block (1024 threads): coalesced reads, every thread keeps 4 independent loads in
// flight, LDS combine.  (4 groups of serial dependent loads cost 20 us per LayerNorm at 256 partial rows.)
__global__ __launch_bounds__(1024) void ln_bwd_finish_kernel(const float* __restrict__ partial, int n_blocks, int C,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             int accumulate) {
  __shared__ float sm[16][64];
  const int cx = threadIdx.x & 63, by = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < 2 * C) {
    const long ld = 2L * C;
    int b = by;
    for (; b + 48 < n_blocks; b += 64) {
      s0 += partial[(long)b * ld + c];
      s1 += partial[(long)(b + 16) * ld + c];
      s2 += partial[(long)(b + 32) * ld + c];
      s3 += partial[(long)(b + 48) * ld + c];
    }
    for (; b < n_blocks; b += 16) s0 += partial[(long)b * ld + c];
  }
  sm[by][cx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (by == 0 && c < 2 * C) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += sm[k][cx];
    float* dst = c < C ? dgamma + c : dbeta + (c - C);
    *dst = accumulate ? *dst + s : s;
  }
}

extern "C" int tell_layernorm_bwd_blocks(int rows) { return (rows + LN_BWD_ROWS - 1) / LN_BWD_ROWS; }

// partial: workspace of tell_layernorm_bwd_blocks(rows) * 2 * C floats
extern "C" int tell_layernorm_bwd(const void* dy, long ld_dy, const void* x, long ld_x, const void* res,
                                  long ld_r, const float* gamma, const float* mean, const float* rstd,
                                  void* dx, long ld_dx, void* dres, long ld_dres, int dres_accumulate,
                                  float* dgamma, float* dbeta, int dparam_accumulate, float* partial,
                                  int rows, int C, float p, uint32_t seed, uint32_t salt, int dtype,
                                  hipStream_t stream) {
  if (rows <= 0) return TELL_OK;
  TELL_REQUIRE(p >= 0.f && p < 1.f, "layernorm_bwd: p must be in [0,1)");
  TELL_REQUIRE((long)C * 8 * sizeof(float) <= 64 * 1024, "layernorm_bwd: C too large for LDS partials");
  uint32_t thr = p > 0.f ? tell_drop_threshold(p) : 0u;
  float ik = 1.f / (1.f - p);
  int nb = tell_layernorm_bwd_blocks(rows);
  size_t smem = (size_t)C * 8 * sizeof(float);
  const int vec = dtype == TELL_BF16 ? 8 : 4;
  auto al = [&](const void* q, long ld) { return q == nullptr || (ld % vec == 0 && ((uintptr_t)q & 15) == 0); };
  const bool aligned = al(dy, ld_dy) && al(x, ld_x) && al(res, ld_r) && al(dx, ld_dx) && al(dres, ld_dres);
  const int nch = C % (64 * vec) == 0 ? C / (64 * vec) : 0;
#define LNB(T, NCH) hipLaunchKernelGGL((ln_bwd_vec_kernel<T, NCH>), dim3(nb), dim3(256), 0, stream, (const T*)dy, ld_dy, \
    (const T*)x, ld_x, (const T*)res, ld_r, gamma, mean, rstd, (T*)dx, ld_dx, (T*)dres, ld_dres, dres_accumulate,      \
    partial, rows, thr, ik, seed, salt, g_tell_rng_step)
  if (aligned && (nch == 1 || nch == 2 || (nch == 4 && dtype == TELL_F32))) {
    if (dtype == TELL_BF16) { if (nch == 1) LNB(uint16_t, 1); else LNB(uint16_t, 2); }
    else { if (nch == 1) LNB(float, 1); else if (nch == 2) LNB(float, 2); else LNB(float, 4); }
  } else if (dtype == TELL_BF16)
    hipLaunchKernelGGL((ln_bwd_kernel<uint16_t>), dim3(nb), dim3(256), smem, stream, (const uint16_t*)dy, ld_dy, (const uint16_t*)x, ld_x, (const uint16_t*)res, ld_r, gamma, mean, rstd, (uint16_t*)dx, ld_dx, (uint16_t*)dres, ld_dres, dres_accumulate, partial, rows, C, thr, ik, seed, salt, g_tell_rng_step);
  else
    hipLaunchKernelGGL((ln_bwd_kernel<float>), dim3(nb), dim3(256), smem, stream, (const float*)dy, ld_dy, (const float*)x, ld_x, (const float*)res, ld_r, gamma, mean, rstd, (float*)dx, ld_dx, (float*)dres, ld_dres, dres_accumulate, partial, rows, C, thr, ik, seed, salt, g_tell_rng_step);
  int rc = tell_check_launch("layernorm_bwd");
  if (rc) return rc;
#undef LNB
  if (!dgamma) return TELL_OK;      // the caller folds `partial` ([blocks][2C]: gamma | beta terms) later (tell_colsum_multi)
  hipLaunchKernelGGL(ln_bwd_finish_kernel, dim3((2 * C + 63) / 64), dim3(1024), 0, stream, partial, nb, C, dgamma, dbeta, dparam_accumulate);
  return tell_check_launch("layernorm_bwd_finish");
}

// ---------------------------------------------------------------- n LayerNorms over one residual in one launch
// cat_i LayerNorm_i(res + dropout(x_i)): the context block of a decoder layer (decoder_faces_objects.py:283-352) ends in
// n = 4 LayerNorms of [rows, C] that share the residual and write adjacent column slices of context_fc's input.  One
// launch each way instead of n (and, backward, n finish launches): forward is the vec kernel with the problem index on
// blockIdx.y; backward walks the n problems of a row inside one wave, so the residual gradient sum_i dz_i is
// accumulated in REGISTERS and written once (the per-problem launches re-read and re-rounded it n-1 times).
#define LN_CAT_MAX 8
struct LNCatArgs {
  const void* x[LN_CAT_MAX];
  const float* gamma[LN_CAT_MAX];
  const float* beta[LN_CAT_MAX];
  void* dx[LN_CAT_MAX];
  float* dgamma[LN_CAT_MAX];
  float* dbeta[LN_CAT_MAX];
  uint32_t salt[LN_CAT_MAX];
  int n;
};
template <typename T, int NCH>
__global__ __launch_bounds__(256) void ln_cat_fwd_kernel(LNCatArgs a, long ld_x, const T* __restrict__ res, long ld_r,
                                                         T* __restrict__ y, long ld_y, float* __restrict__ mean,
                                                         float* __restrict__ rstd, int rows, float eps, uint32_t thr,
                                                         float inv_keep, uint32_t seed, const uint32_t* __restrict__ step) {
  constexpr int VEC = Elem<T>::VEC, C = 64 * VEC * NCH;
  const int i_p = blockIdx.y;
  const uint32_t salt = tell_step_salt(a.salt[i_p], step);
  const T* __restrict__ x = static_cast<const T*>(a.x[i_p]);
  const float* __restrict__ gamma = a.gamma[i_p];
  const float* __restrict__ beta = a.beta[i_p];
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[NCH][VEC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c0 = (lane + 64 * i) * VEC;
    unpack16(*reinterpret_cast<const uint4*>(x + (long)row * ld_x + c0), v[i], (const T*)nullptr);
    if (thr) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) v[i][k] *= tell_keep(seed, salt, (uint64_t)row * C + c0 + k, thr, inv_keep);
    }
    float r[VEC];
    unpack16(*reinterpret_cast<const uint4*>(res + (long)row * ld_r + c0), r, (const T*)nullptr);
#pragma unroll
    for (int k = 0; k < VEC; ++k) { v[i][k] += r[k]; s += v[i][k]; }
  }
  const float mu = wave_sum(s) / C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int k = 0; k < VEC; ++k) { const float d = v[i][k] - mu; q += d * d; }
  const float rs = rsqrtf(wave_sum(q) / C + eps);
  if (lane == 0) { mean[(long)i_p * rows + row] = mu; rstd[(long)i_p * rows + row] = rs; }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c0 = (lane + 64 * i) * VEC;
    float o[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) o[k] = (v[i][k] - mu) * rs * gamma[c0 + k] + beta[c0 + k];
    *reinterpret_cast<uint4*>(y + (long)row * ld_y + (long)i_p * C + c0) = pack16(o, (const T*)nullptr);
  }
}

// partial: [n][blocks][2][C]
template <typename T, int NCH>
__global__ __launch_bounds__(256) void ln_cat_bwd_kernel(LNCatArgs a, const T* __restrict__ dcat, long ld_d, long ld_x,
                                                         const T* __restrict__ res, long ld_r,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         long ld_dx, T* __restrict__ dres, long ld_dres,
                                                         float* __restrict__ partial, int rows, uint32_t thr,
                                                         float inv_keep, uint32_t seed, const uint32_t* __restrict__ step) {
  constexpr int VEC = Elem<T>::VEC, C = 64 * VEC * NCH;
  __shared__ float sm[4][2][C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * LN_BWD_ROWS + wave;
  const bool live = row < rows;                                      // wave-uniform
  float rres[NCH][VEC], dr[NCH][VEC];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if (live) unpack16(*reinterpret_cast<const uint4*>(res + (long)row * ld_r + (lane + 64 * i) * VEC), rres[i], (const T*)nullptr);
#pragma unroll
    for (int k = 0; k < VEC; ++k) dr[i][k] = 0.f;
  }
  for (int i_p = 0; i_p < a.n; ++i_p) {
    const uint32_t salt = tell_step_salt(a.salt[i_p], step);
    const T* __restrict__ x = static_cast<const T*>(a.x[i_p]);
    const float* __restrict__ gamma = a.gamma[i_p];
    T* __restrict__ dx = static_cast<T*>(a.dx[i_p]);
    float xh[NCH][VEC], g[NCH][VEC], keep[NCH][VEC];
    float s1 = 0.f, s2 = 0.f, rs = 0.f;
    if (live) {
      const float mu = mean[(long)i_p * rows + row];
      rs = rstd[(long)i_p * rows + row];
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c0 = (lane + 64 * i) * VEC;
        float d[VEC];
        unpack16(*reinterpret_cast<const uint4*>(x + (long)row * ld_x + c0), xh[i], (const T*)nullptr);
        unpack16(*reinterpret_cast<const uint4*>(dcat + (long)row * ld_d + (long)i_p * C + c0), d, (const T*)nullptr);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          keep[i][k] = thr ? tell_keep(seed, salt, (uint64_t)row * C + c0 + k, thr, inv_keep) : 1.f;
          xh[i][k] = (xh[i][k] * keep[i][k] + rres[i][k] - mu) * rs;
          g[i][k] = d[k] * gamma[c0 + k];
          s1 += g[i][k];
          s2 += g[i][k] * xh[i][k];
          sm[wave][0][c0 + k] = d[k] * xh[i][k];
          sm[wave][1][c0 + k] = d[k];
        }
      }
      s1 = wave_sum(s1) / C;
      s2 = wave_sum(s2) / C;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c0 = (lane + 64 * i) * VEC;
        float o[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const float dz = rs * (g[i][k] - s1 - xh[i][k] * s2);
          dr[i][k] += dz;
          o[k] = dz * keep[i][k];
        }
        if (dx) *reinterpret_cast<uint4*>(dx + (long)row * ld_dx + c0) = pack16(o, (const T*)nullptr);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          sm[wave][0][(lane + 64 * i) * VEC + k] = 0.f;
          sm[wave][1][(lane + 64 * i) * VEC + k] = 0.f;
        }
    }
    __syncthreads();
    float* out = partial + ((long)i_p * gridDim.x + blockIdx.x) * 2 * C;
    const float* s = &sm[0][0][0];
    for (int c = threadIdx.x; c < 2 * C; c += 256) out[c] = s[c] + s[2 * C + c] + s[4 * C + c] + s[6 * C + c];
    __syncthreads();
  }
  if (live && dres) {
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      *reinterpret_cast<uint4*>(dres + (long)row * ld_dres + (lane + 64 * i) * VEC) = pack16(dr[i], (const T*)nullptr);
  }
}
// as ln_bwd_finish_kernel, problem blockIdx.y (always accumulating)
__global__ __launch_bounds__(1024) void ln_cat_bwd_finish_kernel(LNCatArgs a, const float* __restrict__ partial,
                                                                 int n_blocks, int C) {
  __shared__ float sm[16][64];
  const int cx = threadIdx.x & 63, by = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  const float* part = partial + (long)blockIdx.y * n_blocks * 2 * C;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < 2 * C) {
    const long ld = 2L * C;
    int b = by;
    for (; b + 48 < n_blocks; b += 64) {
      s0 += part[(long)b * ld + c];
      s1 += part[(long)(b + 16) * ld + c];
      s2 += part[(long)(b + 32) * ld + c];
      s3 += part[(long)(b + 48) * ld + c];
    }
    for (; b < n_blocks; b += 16) s0 += part[(long)b * ld + c];
  }
  sm[by][cx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (by == 0 && c < 2 * C) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += sm[k][cx];
    float* dst = c < C ? a.dgamma[blockIdx.y] + c : a.dbeta[blockIdx.y] + (c - C);
    *dst += s;
  }
}

static int ln_cat_check(int n, int C, int dtype, const char* who) {
  TELL_REQUIRE(n >= 1 && n <= LN_CAT_MAX, "layernorm_cat: 1..8 LayerNorms per launch");
  TELL_REQUIRE(dtype == TELL_BF16 && (C == 512 || C == 1024), "layernorm_cat: bf16 rows of 512 or 1024 columns");
  (void)who;
  return TELL_OK;
}
// y[:, i*C:(i+1)*C] = LayerNorm_i(res + dropout_p(x_i)); mean / rstd: [n, rows].  Host arrays of n entries: x, gamma,
// beta, salts.  bf16, C = 512 or 1024, 16-byte aligned rows.
extern "C" int tell_layernorm_cat_fwd(int n, const void* const* x, long ld_x, const void* res, long ld_r,
                                      const float* const* gamma, const float* const* beta, void* y, long ld_y,
                                      float* mean, float* rstd, int rows, int C, float eps, float p, uint32_t seed,
                                      const uint32_t* salts, int dtype, hipStream_t stream) {
  if (rows <= 0) return TELL_OK;
  int rc = ln_cat_check(n, C, dtype, "fwd");
  if (rc) return rc;
  TELL_REQUIRE(p >= 0.f && p < 1.f, "layernorm_cat_fwd: p must be in [0,1)");
  TELL_REQUIRE(ld_x % 8 == 0 && ld_r % 8 == 0 && ld_y % 8 == 0 && (((uintptr_t)res | (uintptr_t)y) & 15) == 0,
               "layernorm_cat_fwd: rows must be 16-byte aligned");
  LNCatArgs a;
  a.n = n;
  for (int i = 0; i < n; ++i) {
    TELL_REQUIRE(((uintptr_t)x[i] & 15) == 0, "layernorm_cat_fwd: rows must be 16-byte aligned");
    a.x[i] = x[i]; a.gamma[i] = gamma[i]; a.beta[i] = beta[i]; a.salt[i] = salts[i];
    a.dx[i] = nullptr; a.dgamma[i] = a.dbeta[i] = nullptr;
  }
  const uint32_t thr = p > 0.f ? tell_drop_threshold(p) : 0u;
  const float ik = 1.f / (1.f - p);
  const dim3 grid((rows + 3) / 4, n);
  if (C == 512)
    hipLaunchKernelGGL((ln_cat_fwd_kernel<uint16_t, 1>), grid, dim3(256), 0, stream, a, ld_x, (const uint16_t*)res, ld_r, (uint16_t*)y, ld_y, mean, rstd, rows, eps, thr, ik, seed, g_tell_rng_step);
  else
    hipLaunchKernelGGL((ln_cat_fwd_kernel<uint16_t, 2>), grid, dim3(256), 0, stream, a, ld_x, (const uint16_t*)res, ld_r, (uint16_t*)y, ld_y, mean, rstd, rows, eps, thr, ik, seed, g_tell_rng_step);
  return tell_check_launch("layernorm_cat_fwd");
}
// Backward of the above: dx_i (NULL entries allowed), dres = sum_i dz_i (or NULL), dgamma_i / dbeta_i ACCUMULATED.
// partial: workspace of n * tell_layernorm_bwd_blocks(rows) * 2 * C floats.
extern "C" int tell_layernorm_cat_bwd(int n, const void* dcat, long ld_dcat, const void* const* x, long ld_x,
                                      const void* res, long ld_r, const float* const* gamma, const float* mean,
                                      const float* rstd, void* const* dx, long ld_dx, void* dres, long ld_dres,
                                      float* const* dgamma, float* const* dbeta, float* partial, int rows, int C,
                                      float p, uint32_t seed, const uint32_t* salts, int dtype, hipStream_t stream) {
  if (rows <= 0) return TELL_OK;
  int rc = ln_cat_check(n, C, dtype, "bwd");
  if (rc) return rc;
  TELL_REQUIRE(p >= 0.f && p < 1.f, "layernorm_cat_bwd: p must be in [0,1)");
  TELL_REQUIRE(ld_x % 8 == 0 && ld_r % 8 == 0 && ld_dcat % 8 == 0 && ld_dx % 8 == 0 && ld_dres % 8 == 0 &&
               (((uintptr_t)res | (uintptr_t)dcat | (uintptr_t)dres) & 15) == 0,
               "layernorm_cat_bwd: rows must be 16-byte aligned");
  LNCatArgs a;
  a.n = n;
  for (int i = 0; i < n; ++i) {
    TELL_REQUIRE((((uintptr_t)x[i] | (uintptr_t)dx[i]) & 15) == 0, "layernorm_cat_bwd: rows must be 16-byte aligned");
    a.x[i] = x[i]; a.gamma[i] = gamma[i]; a.beta[i] = nullptr; a.salt[i] = salts[i];
    a.dx[i] = dx[i]; a.dgamma[i] = dgamma ? dgamma[i] : nullptr; a.dbeta[i] = dbeta ? dbeta[i] : nullptr;
  }
  const uint32_t thr = p > 0.f ? tell_drop_threshold(p) : 0u;
  const float ik = 1.f / (1.f - p);
  const int nb = tell_layernorm_bwd_blocks(rows);
  if (C == 512)
    hipLaunchKernelGGL((ln_cat_bwd_kernel<uint16_t, 1>), dim3(nb), dim3(256), 0, stream, a, (const uint16_t*)dcat, ld_dcat, ld_x, (const uint16_t*)res, ld_r, mean, rstd, ld_dx, (uint16_t*)dres, ld_dres, partial, rows, thr, ik, seed, g_tell_rng_step);
  else
    hipLaunchKernelGGL((ln_cat_bwd_kernel<uint16_t, 2>), dim3(nb), dim3(256), 0, stream, a, (const uint16_t*)dcat, ld_dcat, ld_x, (const uint16_t*)res, ld_r, mean, rstd, ld_dx, (uint16_t*)dres, ld_dres, partial, rows, thr, ik, seed, g_tell_rng_step);
  rc = tell_check_launch("layernorm_cat_bwd");
  if (rc) return rc;
  if (!dgamma) return TELL_OK;      // deferred: partial is [n][blocks][2C]
  hipLaunchKernelGGL(ln_cat_bwd_finish_kernel, dim3((2 * C + 63) / 64, n), dim3(1024), 0, stream, a, partial, nb, C);
  return tell_check_launch("layernorm_cat_bwd_finish");
}
