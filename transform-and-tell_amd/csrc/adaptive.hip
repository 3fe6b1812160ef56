// Adaptive input embedding + adaptive softmax helpers with STATIC shapes
// (tell/modules/token_embedders/adaptive.py:61-76, tell/modules/softmax.py:144-222,
//  tell/modules/criteria/adaptive_loss.py:27-73).
//
// The reference uses boolean masks / nonzero() / index_select with host syncs.
// Here a single-workgroup partition kernel builds, on the device, ascending
// compacted row lists per vocabulary band; the band GEMMs then run over a
// fixed-capacity buffer with a device-side row count (gemm_nt's m_dev), so the
// whole step has no data-dependent launch and no host synchronisation.
#include "common.h"
#include "options.h"

#define MAX_BANDS 4

struct PartitionArgs {
  const long* ids;      // [N] int64 token ids
  int N, n_bands;
  int cut[MAX_BANDS];   // upper bounds: band b = [cut[b-1], cut[b])
  int pad_idx;          // ids == pad_idx are not counted in n_valid
  int* band_rows;       // [n_bands][N]  ascending row indices of each band
  int* band_local;      // [n_bands][N]  id - band_lo of those rows
  int* band_count;      // [n_bands]
  int* slot;            // [N]  b*N + position of the row inside its band list
  int* head_target;     // [N]  id (band 0) or cut[0] + b - 1 (softmax.py:158)
  int* n_valid;         // [1]  number of ids != pad_idx (adaptive_loss.py:62-65)
};

__global__ __launch_bounds__(1024) void partition_kernel(PartitionArgs p) {
  __shared__ int wave_tot[16];
  __shared__ int base_s;
  __shared__ int valid_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) valid_s = 0;
  for (int b = 0; b < p.n_bands; ++b) {
    const int lo = b == 0 ? 0 : p.cut[b - 1], hi = p.cut[b];
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int start = 0; start < p.N; start += 1024) {
      const int i = start + tid;
      long id = i < p.N ? p.ids[i] : -1;
      const bool flag = i < p.N && id >= lo && id < hi;
      const unsigned long long bal = __ballot(flag);
      const int wpre = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) wave_tot[wave] = __popcll(bal);
      __syncthreads();
      int off = base_s, tot = 0;
      for (int w = 0; w < 16; ++w) { if (w < wave) off += wave_tot[w]; tot += wave_tot[w]; }
      if (flag) {
        const int j = off + wpre;
        p.band_rows[(long)b * p.N + j] = i;
        p.band_local[(long)b * p.N + j] = (int)(id - lo);
        if (p.slot) p.slot[i] = b * p.N + j;
        if (p.head_target) p.head_target[i] = b == 0 ? (int)id : p.cut[0] + b - 1;
      }
      if (b == 0 && p.n_valid) {
        const unsigned long long vb = __ballot(i < p.N && id != p.pad_idx);
        if (lane == 0) atomicAdd(&valid_s, __popcll(vb));
      }
      __syncthreads();
      if (tid == 0) base_s += tot;
      __syncthreads();
    }
    if (tid == 0) p.band_count[b] = base_s;
  }
  __syncthreads();
  if (tid == 0 && p.n_valid) *p.n_valid = valid_s;
}

extern "C" int tell_adaptive_partition(const long* ids, int N, const int* cutoffs, int n_bands, int pad_idx,
                                       int* band_rows, int* band_local, int* band_count, int* slot,
                                       int* head_target, int* n_valid, hipStream_t stream) {
  TELL_REQUIRE(n_bands >= 1 && n_bands <= MAX_BANDS, "partition: 1..4 bands");
  if (N <= 0) return TELL_OK;
  PartitionArgs p;
  p.ids = ids; p.N = N; p.n_bands = n_bands; p.pad_idx = pad_idx;
  for (int b = 0; b < MAX_BANDS; ++b) p.cut[b] = b < n_bands ? cutoffs[b] : 0;
  p.band_rows = band_rows; p.band_local = band_local; p.band_count = band_count; p.slot = slot;
  p.head_target = head_target; p.n_valid = n_valid;
  hipLaunchKernelGGL(partition_kernel, dim3(1), dim3(1024), 0, stream, p);
  return tell_check_launch("adaptive_partition");
}

// ------------------------------------------------------------------ embedding finalize
// out[row(n)] = scale * band_out[slot[n]] + pos_table[position(n)]
//   position (positional.py:231-268, right padding): pad -> pad_idx, else pad_idx + 1 + t + start_pos
//   row(n): n = b*T + t  ->  t*B + b when `tbc` (decoder layout) else n
template <typename T>
__global__ __launch_bounds__(256) void embed_finalize_kernel(const T* __restrict__ band_out,
                                                             const int* __restrict__ slot,
                                                             const long* __restrict__ ids,
                                                             const float* __restrict__ pos_table,
                                                             int pos_rows, T* __restrict__ out, int Bn,
                                                             int Tn, int E, float scale, int pos_pad,
                                                             int start_pos, int tbc,
                                                             const uint32_t* __restrict__ step) {
  // inside a captured decode step the position offset advances with the graph's device counter (common.h)
  if (step) start_pos += (int)*step;
  const int n = blockIdx.x;
  const int b = n / Tn, t = n % Tn;
  const long id = ids[n];
  int pos = id == pos_pad ? pos_pad : pos_pad + 1 + t + start_pos;
  if (pos >= pos_rows) pos = pos_rows - 1;       // host guarantees the table is large enough
  const T* src = band_out + (long)slot[n] * E;
  const float* pr = pos_table + (long)pos * E;
  T* dst = out + (long)(tbc ? t * Bn + b : n) * E;
  for (int c = threadIdx.x; c < E; c += 256) Elem<T>::st(dst + c, scale * Elem<T>::ld(src + c) + pr[c]);
}
extern "C" int tell_embed_finalize(const void* band_out, const int* slot, const long* ids,
                                   const float* pos_table, int pos_rows, void* out, int B, int T, int E,
                                   float scale, int pos_pad, int start_pos, int tbc, int dtype,
                                   hipStream_t stream) {
  if (B * T <= 0) return TELL_OK;
  if (dtype == TELL_BF16) hipLaunchKernelGGL((embed_finalize_kernel<uint16_t>), dim3(B * T), dim3(256), 0, stream, (const uint16_t*)band_out, slot, ids, pos_table, pos_rows, (uint16_t*)out, B, T, E, scale, pos_pad, start_pos, tbc, g_tell_pos_step);
  else hipLaunchKernelGGL((embed_finalize_kernel<float>), dim3(B * T), dim3(256), 0, stream, (const float*)band_out, slot, ids, pos_table, pos_rows, (float*)out, B, T, E, scale, pos_pad, start_pos, tbc, g_tell_pos_step);
  return tell_check_launch("embed_finalize");
}

// backward of finalize: dband[slot[n]] = scale * dout[row(n)]
template <typename T>
__global__ __launch_bounds__(256) void embed_finalize_bwd_kernel(const T* __restrict__ dout,
                                                                 const int* __restrict__ slot,
                                                                 T* __restrict__ dband, int Bn, int Tn,
                                                                 int E, float scale, int tbc) {
  const int n = blockIdx.x;
  const int b = n / Tn, t = n % Tn;
  const T* src = dout + (long)(tbc ? t * Bn + b : n) * E;
  T* dst = dband + (long)slot[n] * E;
  for (int c = threadIdx.x; c < E; c += 256) Elem<T>::st(dst + c, scale * Elem<T>::ld(src + c));
}
extern "C" int tell_embed_finalize_bwd(const void* dout, const int* slot, void* dband, int B, int T, int E,
                                       float scale, int tbc, int dtype, hipStream_t stream) {
  if (B * T <= 0) return TELL_OK;
  if (dtype == TELL_BF16) hipLaunchKernelGGL((embed_finalize_bwd_kernel<uint16_t>), dim3(B * T), dim3(256), 0, stream, (const uint16_t*)dout, slot, (uint16_t*)dband, B, T, E, scale, tbc);
  else hipLaunchKernelGGL((embed_finalize_bwd_kernel<float>), dim3(B * T), dim3(256), 0, stream, (const float*)dout, slot, (float*)dband, B, T, E, scale, tbc);
  return tell_check_launch("embed_finalize_bwd");
}

// gradient of the band table: demb[local[j]] += drows[j]; local row `padding_idx` receives nothing
// (nn.Embedding(padding_idx), adaptive.py:42).
// Round 6: DETERMINISTIC - a segmented sum by table row instead of fp32 atomics.  (With atomics the order in which the
// duplicates of a token met in a table row depended on workgroup scheduling: two otherwise identical trainers differed by
// 1-3e-6 of the loss after a few steps, and a test tolerance had to follow.)  The workgroup of the FIRST band row that
// names a table row is that row's only writer: it lists the later band rows with the same id in increasing order (a block
// scan over 1024 candidates at a time), adds them up in that order and adds the sum to the gradient row once.
template <typename T>
__global__ __launch_bounds__(256) void embed_table_grad_kernel(const T* __restrict__ drows, long ld,
                                                               const int* __restrict__ local,
                                                               const int* __restrict__ count_dev, int cap,
                                                               float* __restrict__ demb, int dim,
                                                               int padding_idx) {
  __shared__ int s_wave[4];
  __shared__ int s_list[1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int n = *count_dev;
  if (n > cap) n = cap;
  for (int j = blockIdx.x; j < n; j += gridDim.x) {
    const int row = local[j];
    if (row == padding_idx) continue;                          // (uniform)
    int dup = 0;
    for (int i = tid; i < j; i += 256) dup |= (local[i] == row) ? 1 : 0;
    if (__syncthreads_or(dup)) continue;                       // an earlier band row owns this table row (uniform)
    for (int c0 = 0; c0 < dim; c0 += 1024) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int base = j; base < n; base += 1024) {
        // ordered list of the band rows base .. base + 1023 that name `row`: thread t owns candidates 4t .. 4t + 3
        int hit[4], cnt = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = base + tid * 4 + e;
          hit[e] = (i < n && local[i] == row) ? 1 : 0;
          cnt += hit[e];
        }
        int incl = cnt;                                        // inclusive scan over the wave, then over the 4 waves
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int v = __shfl_up(incl, o, 64);
          if (lane >= o) incl += v;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int off = incl - cnt;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        const int total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (hit[e]) s_list[off++] = base + tid * 4 + e;
        __syncthreads();
        for (int m = 0; m < total; ++m) {
          const T* src = drows + (long)s_list[m] * ld + c0;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c = tid + q * 256;
            if (c0 + c < dim) acc[q] += Elem<T>::ld(src + c);
          }
        }
        __syncthreads();                                       // (s_list / s_wave are rewritten by the next chunk)
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = c0 + tid + q * 256;
        if (c < dim) demb[(long)row * dim + c] += acc[q];
      }
    }
  }
}
extern "C" int tell_embed_table_grad(const void* drows, long ld, const int* local, const int* count_dev,
                                     int cap, float* demb, int dim, int padding_idx, int dtype,
                                     hipStream_t stream) {
  if (cap <= 0) return TELL_OK;
  int g = cap < 1024 ? cap : 1024;
  if (dtype == TELL_BF16) hipLaunchKernelGGL((embed_table_grad_kernel<uint16_t>), dim3(g), dim3(256), 0, stream, (const uint16_t*)drows, ld, local, count_dev, cap, demb, dim, padding_idx);
  else hipLaunchKernelGGL((embed_table_grad_kernel<float>), dim3(g), dim3(256), 0, stream, (const float*)drows, ld, local, count_dev, cap, demb, dim, padding_idx);
  return tell_check_launch("embed_table_grad");
}

// ------------------------------------------------------------------ cross entropy over one cluster
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = fmaxf(r, red[w]);
  __syncthreads();
  return r;
}
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += red[w];
  __syncthreads();
  return r;
}

// per row i (< *m_dev): lse[i] = logsumexp(logits[i,:]); loss[i] = lse - logits[i,tgt] unless tgt == ignore
// target of compacted row i is targets[row_idx ? row_idx[i] : i]
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, long ld, int M, int V,
                                                     const int* __restrict__ targets,
                                                     const int* __restrict__ row_idx,
                                                     const int* __restrict__ m_dev, int ignore_index,
                                                     float* __restrict__ lse, float* __restrict__ loss) {
  __shared__ float red[4];
  int Me = M;
  if (m_dev) { int md = *m_dev; Me = md < M ? md : M; }
  const int i = blockIdx.x;
  if (i >= Me) { if (threadIdx.x == 0) { loss[i] = 0.f; lse[i] = 0.f; } return; }
  const float* row = logits + (long)i * ld;
  float mx = -INFINITY;
  for (int j = threadIdx.x; j < V; j += 256) mx = fmaxf(mx, row[j]);
  mx = block_max(mx, red);
  float s = 0.f;
  for (int j = threadIdx.x; j < V; j += 256) s += __expf(row[j] - mx);
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    const float l = mx + __logf(s);
    const int tg = targets[row_idx ? row_idx[i] : i];
    lse[i] = l;
    loss[i] = tg == ignore_index ? 0.f : l - row[tg];
  }
}
// dlogits[i,j] = (softmax - onehot) * g,  g = *gscale (device scalar), 0 for ignored rows
template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, long ld, int M, int V,
                                                     const int* __restrict__ targets,
                                                     const int* __restrict__ row_idx,
                                                     const int* __restrict__ m_dev, int ignore_index,
                                                     const float* __restrict__ lse,
                                                     const float* __restrict__ gscale, T* __restrict__ dlogits,
                                                     long ld_d) {
  int Me = M;
  if (m_dev) { int md = *m_dev; Me = md < M ? md : M; }
  const int i = blockIdx.x;
  if (i >= Me) return;
  const int tg = targets[row_idx ? row_idx[i] : i];
  const float g = tg == ignore_index ? 0.f : *gscale;
  const float l = lse[i];
  const float* row = logits + (long)i * ld;
  T* d = dlogits + (long)i * ld_d;
  for (int j = threadIdx.x; j < V; j += 256)
    Elem<T>::st(d + j, (__expf(row[j] - l) - (j == tg ? 1.f : 0.f)) * g);
}
extern "C" int tell_ce_fwd(const float* logits, long ld, int M, int V, const int* targets, const int* row_idx,
                           const int* m_dev, int ignore_index, float* lse, float* loss, hipStream_t stream) {
  if (M <= 0) return TELL_OK;
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(M), dim3(256), 0, stream, logits, ld, M, V, targets, row_idx, m_dev, ignore_index, lse, loss);
  return tell_check_launch("ce_fwd");
}
extern "C" int tell_ce_bwd(const float* logits, long ld, int M, int V, const int* targets, const int* row_idx,
                           const int* m_dev, int ignore_index, const float* lse, const float* gscale,
                           void* dlogits, long ld_d, int dtype, hipStream_t stream) {
  if (M <= 0) return TELL_OK;
  if (dtype == TELL_BF16) hipLaunchKernelGGL((ce_bwd_kernel<uint16_t>), dim3(M), dim3(256), 0, stream, logits, ld, M, V, targets, row_idx, m_dev, ignore_index, lse, gscale, (uint16_t*)dlogits, ld_d);
  else hipLaunchKernelGGL((ce_bwd_kernel<float>), dim3(M), dim3(256), 0, stream, logits, ld, M, V, targets, row_idx, m_dev, ignore_index, lse, gscale, (float*)dlogits, ld_d);
  return tell_check_launch("ce_bwd");
}

// deterministic sum of n (< = *m_dev if given) floats -> out[0] (+= if accumulate)
__global__ __launch_bounds__(1024) void sum_kernel(const float* __restrict__ x, int n,
                                                   const int* __restrict__ m_dev, float* __restrict__ out,
                                                   int accumulate) {
  __shared__ float red[16];
  if (m_dev) { int md = *m_dev; n = md < n ? md : n; }
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) s += x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    out[0] = accumulate ? out[0] + t : t;
  }
}
extern "C" int tell_sum_f32(const float* x, int n, const int* m_dev, float* out, int accumulate,
                            hipStream_t stream) {
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, stream, x, n, m_dev, out, accumulate);
  return tell_check_launch("sum_f32");
}

// ------------------------------------------------------------------ generation head (softmax.py:193-222 + topk(1))
// One workgroup per row: log-softmax of the head and of every tail, combined
// log-probs  lp(w) = head_lsm[w]  (w < c0)   |   tail_lsm_i[w - cut_i] + head_lsm[c0 + i];
// writes argmax token + its log-prob and, optionally, the full log-prob row.
struct LogProbArgs {
  const float* head; long ld_head; int head_n;    // head_n = c0 + n_tails
  const float* tail[3]; long ld_tail[3]; int tail_n[3];
  int n_tails, c0, rows;
  float* log_probs; long ld_lp;                   // optional [rows, vocab]
  int* token; float* token_lp;
};
// (1024 threads per row, 4 loads in flight per thread: with 256 threads and one load at a time the 50 k logits of a row
//  took 87 us of dependent latency - 8 % of a decode step)
__global__ __launch_bounds__(1024) void logprob_argmax_kernel(LogProbArgs p) {
  __shared__ float red[16];
  __shared__ float best_v[16];
  __shared__ int best_i[16];
  const int i = blockIdx.x;
  const float* hrow = p.head + (long)i * p.ld_head;
  float mx = -INFINITY;
#pragma unroll 4
  for (int j = threadIdx.x; j < p.head_n; j += 1024) mx = fmaxf(mx, hrow[j]);
  mx = block_max(mx, red);
  float s = 0.f;
#pragma unroll 4
  for (int j = threadIdx.x; j < p.head_n; j += 1024) s += __expf(hrow[j] - mx);
  s = block_sum(s, red);
  const float lse_h = mx + __logf(s);
  float bv = -INFINITY; int bi = 0x7fffffff;
  float* lp = p.log_probs ? p.log_probs + (long)i * p.ld_lp : nullptr;
#pragma unroll 4
  for (int j = threadIdx.x; j < p.c0; j += 1024) {
    const float v = hrow[j] - lse_h;
    if (lp) lp[j] = v;
    if (v > bv || (v == bv && j < bi)) { bv = v; bi = j; }
  }
  int base = p.c0;
  for (int c = 0; c < p.n_tails; ++c) {
    const float* trow = p.tail[c] + (long)i * p.ld_tail[c];
    const int n = p.tail_n[c];
    float m2 = -INFINITY;
  #pragma unroll 4
  for (int j = threadIdx.x; j < n; j += 1024) m2 = fmaxf(m2, trow[j]);
    m2 = block_max(m2, red);
    float s2 = 0.f;
  #pragma unroll 4
  for (int j = threadIdx.x; j < n; j += 1024) s2 += __expf(trow[j] - m2);
    s2 = block_sum(s2, red);
    const float off = (hrow[p.c0 + c] - lse_h) - (m2 + __logf(s2));
  #pragma unroll 4
  for (int j = threadIdx.x; j < n; j += 1024) {
      const float v = trow[j] + off;
      if (lp) lp[base + j] = v;
      if (v > bv || (v == bv && base + j < bi)) { bv = v; bi = base + j; }
    }
    base += n;
  }
  // block arg-max (lowest index wins ties)
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { best_v[threadIdx.x >> 6] = bv; best_i[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w)
      if (best_v[w] > bv || (best_v[w] == bv && best_i[w] < bi)) { bv = best_v[w]; bi = best_i[w]; }
    if (p.token) p.token[i] = bi;
    if (p.token_lp) p.token_lp[i] = bv;
  }
}
// The same arg-max with every logit of the row in registers: all loads of a row are requested at once (the three-pass
// form above is a chain of ~36 dependent load groups per row: 33 us of a 600 us decode step), the max / sum-exp /
// arg-max passes then run over registers.  Same comparisons as above (v = logit - lse, lowest index wins a tie).
// Capacity (float4 per thread x 1024 threads): head 2 (8192 logits), tails 4 / 8 / 2 (16384 / 32768 / 8192); rows start
// on 16 bytes.  64 of the 128 registers a 1024-thread workgroup leaves each thread.
__device__ constexpr int LPF_CAP[4] = {2, 4, 8, 2};
__device__ constexpr int LPF_OFF[4] = {0, 2, 6, 14};
// K = 1: arg-max (token, token_lp); K = 4 / 8: every thread keeps its K best while scanning its registers, the block pops
// the global best k times (beam search: tokens / lps [rows, k], best first).
template <int K>
__global__ __launch_bounds__(1024) void logprob_regs_kernel(LogProbArgs p, int k, int* __restrict__ tokens,
                                                            float* __restrict__ lps) {
  __shared__ float red[4][16];
  __shared__ float best_v[16];
  __shared__ int best_i[16];
  __shared__ int win_i;
  const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nseg = 1 + p.n_tails;
  const float* rowp[4]; int n[4];
  rowp[0] = p.head + (long)i * p.ld_head; n[0] = p.head_n;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    rowp[c + 1] = c < p.n_tails ? p.tail[c] + (long)i * p.ld_tail[c] : rowp[0];
    n[c + 1] = c < p.n_tails ? p.tail_n[c] : 0;
  }
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 x[16];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int q = 0; q < LPF_CAP[s]; ++q) {
      const int j = (q * 1024 + tid) * 4;
      f4 v = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      if (j + 3 < n[s]) v = *reinterpret_cast<const f4*>(rowp[s] + j);
      else if (j < n[s]) {
        v.x = rowp[s][j];
        if (j + 1 < n[s]) v.y = rowp[s][j + 1];
        if (j + 2 < n[s]) v.z = rowp[s][j + 2];
      }
      x[LPF_OFF[s] + q] = v;
    }
  float mx[4], sm[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < LPF_CAP[s]; ++q) {
      const f4 v = x[LPF_OFF[s] + q];
      m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    m = wave_max(m);
    if (lane == 0) red[s][wave] = m;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float m = red[s][0];
#pragma unroll
    for (int w = 1; w < 16; ++w) m = fmaxf(m, red[s][w]);
    mx[s] = m;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float t = 0.f;
    if (s < nseg) {
#pragma unroll
      for (int q = 0; q < LPF_CAP[s]; ++q) {
        const f4 v = x[LPF_OFF[s] + q];
        t += (__expf(v.x - mx[s]) + __expf(v.y - mx[s])) + (__expf(v.z - mx[s]) + __expf(v.w - mx[s]));
      }
    }
    t = wave_sum(t);
    if (lane == 0) red[s][wave] = t;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[s][w];
    sm[s] = t;
  }
  const float lse_h = mx[0] + __logf(sm[0]);
  float tv[K]; int ti[K];
#pragma unroll
  for (int q = 0; q < K; ++q) { tv[q] = -INFINITY; ti[q] = 0x7fffffff; }
  auto better = [](float v, int j, float w, int m) { return v > w || (v == w && j < m); };
  auto consider = [&](float v, int j) {
    if (!better(v, j, tv[K - 1], ti[K - 1])) return;
    tv[K - 1] = v; ti[K - 1] = j;
#pragma unroll
    for (int q = K - 1; q > 0; --q)
      if (better(tv[q], ti[q], tv[q - 1], ti[q - 1])) {
        const float fv = tv[q]; tv[q] = tv[q - 1]; tv[q - 1] = fv;
        const int fi = ti[q]; ti[q] = ti[q - 1]; ti[q - 1] = fi;
      }
  };
#pragma unroll
  for (int q = 0; q < LPF_CAP[0]; ++q) {
    const int j = (q * 1024 + tid) * 4;
    if (j < p.c0) consider(x[q].x - lse_h, j);
    if (j + 1 < p.c0) consider(x[q].y - lse_h, j + 1);
    if (j + 2 < p.c0) consider(x[q].z - lse_h, j + 2);
    if (j + 3 < p.c0) consider(x[q].w - lse_h, j + 3);
  }
  int base = p.c0;
#pragma unroll
  for (int s = 1; s < 4; ++s) {
    if (s < nseg) {
      const float off = (rowp[0][p.c0 + s - 1] - lse_h) - (mx[s] + __logf(sm[s]));
#pragma unroll
      for (int q = 0; q < LPF_CAP[s]; ++q) {
        const int j = (q * 1024 + tid) * 4;
        const f4 v = x[LPF_OFF[s] + q];
        if (j < n[s]) consider(v.x + off, base + j);
        if (j + 1 < n[s]) consider(v.y + off, base + j + 1);
        if (j + 2 < n[s]) consider(v.z + off, base + j + 2);
        if (j + 3 < n[s]) consider(v.w + off, base + j + 3);
      }
      base += n[s];
    }
  }
  for (int r = 0; r < k; ++r) {                       // pop the block-wide best k times (lowest index wins ties)
    float bv = tv[0]; int bi = ti[0];
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    __syncthreads();                                   // previous round's win_i / best_* consumed
    if (lane == 0) { best_v[wave] = bv; best_i[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 16; ++w)
        if (better(best_v[w], best_i[w], bv, bi)) { bv = best_v[w]; bi = best_i[w]; }
      if (K == 1) {
        if (p.token) p.token[i] = bi;
        if (p.token_lp) p.token_lp[i] = bv;
      } else {
        tokens[(long)i * k + r] = bi;
        lps[(long)i * k + r] = bv;
      }
      win_i = bi;
    }
    if (K == 1) break;
    __syncthreads();
    if (ti[0] == win_i) {                              // the owner of the winner advances its list
#pragma unroll
      for (int q = 0; q < K - 1; ++q) { tv[q] = tv[q + 1]; ti[q] = ti[q + 1]; }
      tv[K - 1] = -INFINITY; ti[K - 1] = 0x7fffffff;
    }
  }
}

// Top-k variant for beam search: per row the k best (log-prob, token) pairs of the adaptive softmax, sorted best
// first, without materialising [rows, vocab] (a beam of k only ever needs each hypothesis' own k best tokens).
// Every thread keeps its k best in registers while streaming the row; the block then pops the global best k times.
template <int K>
__global__ __launch_bounds__(256) void logprob_topk_kernel(LogProbArgs p, int k, int* __restrict__ tokens,
                                                           float* __restrict__ lps) {
  __shared__ float red[4];
  __shared__ float best_v[4];
  __shared__ int best_i[4];
  __shared__ int win_i;
  const int i = blockIdx.x;
  const float* hrow = p.head + (long)i * p.ld_head;
  float tv[K]; int ti[K];
#pragma unroll
  for (int q = 0; q < K; ++q) { tv[q] = -INFINITY; ti[q] = 0x7fffffff; }
  auto better = [](float v, int j, float w, int m) { return v > w || (v == w && j < m); };
  auto push = [&](float v, int j) {
    if (!better(v, j, tv[K - 1], ti[K - 1])) return;
    tv[K - 1] = v; ti[K - 1] = j;
#pragma unroll
    for (int q = K - 1; q > 0; --q)
      if (better(tv[q], ti[q], tv[q - 1], ti[q - 1])) {
        const float fv = tv[q]; tv[q] = tv[q - 1]; tv[q - 1] = fv;
        const int fi = ti[q]; ti[q] = ti[q - 1]; ti[q - 1] = fi;
      }
  };
  float mx = -INFINITY;
  for (int j = threadIdx.x; j < p.head_n; j += 256) mx = fmaxf(mx, hrow[j]);
  mx = block_max(mx, red);
  float s = 0.f;
  for (int j = threadIdx.x; j < p.head_n; j += 256) s += __expf(hrow[j] - mx);
  s = block_sum(s, red);
  const float lse_h = mx + __logf(s);
  for (int j = threadIdx.x; j < p.c0; j += 256) push(hrow[j] - lse_h, j);
  int base = p.c0;
  for (int c = 0; c < p.n_tails; ++c) {
    const float* trow = p.tail[c] + (long)i * p.ld_tail[c];
    const int n = p.tail_n[c];
    float m2 = -INFINITY;
    for (int j = threadIdx.x; j < n; j += 256) m2 = fmaxf(m2, trow[j]);
    m2 = block_max(m2, red);
    float s2 = 0.f;
    for (int j = threadIdx.x; j < n; j += 256) s2 += __expf(trow[j] - m2);
    s2 = block_sum(s2, red);
    const float off = (hrow[p.c0 + c] - lse_h) - (m2 + __logf(s2));
    for (int j = threadIdx.x; j < n; j += 256) push(trow[j] + off, base + j);
    base += n;
  }
  for (int r = 0; r < k; ++r) {                       // pop the block-wide best k times
    float bv = tv[0]; int bi = ti[0];
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    __syncthreads();                                   // previous round's win_i / best_* consumed
    if ((threadIdx.x & 63) == 0) { best_v[threadIdx.x >> 6] = bv; best_i[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 4; ++w)
        if (better(best_v[w], best_i[w], bv, bi)) { bv = best_v[w]; bi = best_i[w]; }
      tokens[(long)i * k + r] = bi;
      lps[(long)i * k + r] = bv;
      win_i = bi;
    }
    __syncthreads();
    if (ti[0] == win_i) {                              // the owner of the winner advances its list
#pragma unroll
      for (int q = 0; q < K - 1; ++q) { tv[q] = tv[q + 1]; ti[q] = ti[q + 1]; }
      tv[K - 1] = -INFINITY; ti[K - 1] = 0x7fffffff;
    }
  }
}
extern "C" int tell_adaptive_logprob_topk(const float* head, long ld_head, int c0, int n_tails,
                                          const float* tail0, long ld0, int n0, const float* tail1, long ld1,
                                          int n1, const float* tail2, long ld2, int n2, int rows, int k,
                                          int* tokens, float* lps, hipStream_t stream) {
  TELL_REQUIRE(n_tails >= 0 && n_tails <= 3, "logprob_topk: up to 3 tails");
  TELL_REQUIRE(k >= 1 && k <= 8, "logprob_topk: 1 <= k <= 8");
  if (rows <= 0) return TELL_OK;
  LogProbArgs p;
  p.head = head; p.ld_head = ld_head; p.head_n = c0 + n_tails; p.c0 = c0; p.n_tails = n_tails; p.rows = rows;
  p.tail[0] = tail0; p.ld_tail[0] = ld0; p.tail_n[0] = n0;
  p.tail[1] = tail1; p.ld_tail[1] = ld1; p.tail_n[1] = n1;
  p.tail[2] = tail2; p.ld_tail[2] = ld2; p.tail_n[2] = n2;
  p.log_probs = nullptr; p.ld_lp = 0; p.token = nullptr; p.token_lp = nullptr;
  const bool aligned = ld_head % 4 == 0 && ((uintptr_t)head % 16) == 0 &&
                       (n_tails < 1 || (ld0 % 4 == 0 && ((uintptr_t)tail0 % 16) == 0)) &&
                       (n_tails < 2 || (ld1 % 4 == 0 && ((uintptr_t)tail1 % 16) == 0)) &&
                       (n_tails < 3 || (ld2 % 4 == 0 && ((uintptr_t)tail2 % 16) == 0));
  const bool regs_env = tell_opt(OPT_ARGMAX_REGS) != 0;      // A/B aid
  if (regs_env && aligned && p.head_n <= 2 * 4096 && (n_tails < 1 || n0 <= 4 * 4096) && (n_tails < 2 || n1 <= 8 * 4096) &&
      (n_tails < 3 || n2 <= 2 * 4096)) {
    if (k <= 4) hipLaunchKernelGGL((logprob_regs_kernel<4>), dim3(rows), dim3(1024), 0, stream, p, k, tokens, lps);
    else hipLaunchKernelGGL((logprob_regs_kernel<8>), dim3(rows), dim3(1024), 0, stream, p, k, tokens, lps);
    return tell_check_launch("logprob_topk (registers)");
  }
  if (k <= 4) hipLaunchKernelGGL((logprob_topk_kernel<4>), dim3(rows), dim3(256), 0, stream, p, k, tokens, lps);
  else hipLaunchKernelGGL((logprob_topk_kernel<8>), dim3(rows), dim3(256), 0, stream, p, k, tokens, lps);
  return tell_check_launch("logprob_topk");
}

extern "C" int tell_adaptive_logprob_argmax(const float* head, long ld_head, int c0, int n_tails,
                                            const float* tail0, long ld0, int n0, const float* tail1,
                                            long ld1, int n1, const float* tail2, long ld2, int n2,
                                            int rows, float* log_probs, long ld_lp, int* token,
                                            float* token_lp, hipStream_t stream) {
  TELL_REQUIRE(n_tails >= 0 && n_tails <= 3, "logprob_argmax: up to 3 tails");
  if (rows <= 0) return TELL_OK;
  LogProbArgs p;
  p.head = head; p.ld_head = ld_head; p.head_n = c0 + n_tails; p.c0 = c0; p.n_tails = n_tails; p.rows = rows;
  p.tail[0] = tail0; p.ld_tail[0] = ld0; p.tail_n[0] = n0;
  p.tail[1] = tail1; p.ld_tail[1] = ld1; p.tail_n[1] = n1;
  p.tail[2] = tail2; p.ld_tail[2] = ld2; p.tail_n[2] = n2;
  p.log_probs = log_probs; p.ld_lp = ld_lp; p.token = token; p.token_lp = token_lp;
  const bool aligned = ld_head % 4 == 0 && ((uintptr_t)head % 16) == 0 &&
                       (n_tails < 1 || (ld0 % 4 == 0 && ((uintptr_t)tail0 % 16) == 0)) &&
                       (n_tails < 2 || (ld1 % 4 == 0 && ((uintptr_t)tail1 % 16) == 0)) &&
                       (n_tails < 3 || (ld2 % 4 == 0 && ((uintptr_t)tail2 % 16) == 0));
  const bool regs_env = tell_opt(OPT_ARGMAX_REGS) != 0;      // A/B aid
  if (!log_probs && regs_env && aligned && p.head_n <= 2 * 4096 && (n_tails < 1 || n0 <= 4 * 4096) &&
      (n_tails < 2 || n1 <= 8 * 4096) && (n_tails < 3 || n2 <= 2 * 4096)) {
    hipLaunchKernelGGL((logprob_regs_kernel<1>), dim3(rows), dim3(1024), 0, stream, p, 1, (int*)nullptr, (float*)nullptr);
    return tell_check_launch("logprob_argmax (registers)");
  }
  hipLaunchKernelGGL(logprob_argmax_kernel, dim3(rows), dim3(1024), 0, stream, p);
  return tell_check_launch("logprob_argmax");
}

// ------------------------------------------------------------------ greedy generation: one step's bookkeeping
// What the loop of transformer_faces_objects.py:443-494 does per token after the arg-max, for all rows at once: a row
// that has not finished records token and log-prob (divided by the sampling temperature), a row that emits EOS now is
// marked finished and remembers the step; every row's token becomes the next step's input (finished rows keep decoding
// into the void - the batch keeps its shape - and their outputs stay padding).  Sixteen elementwise ATen launches per
// generated token before.
__global__ void greedy_update_kernel(const int* __restrict__ tok, const float* __restrict__ lp,
                                     uint8_t* __restrict__ finished, long* __restrict__ ids, long ld_ids,
                                     float* __restrict__ lps, long ld_lps, long* __restrict__ done_step,
                                     long* __restrict__ cur, int B, int i_host, int eos, float inv_temp, int* counter,
                                     const int* step_dev) {
  const int i = step_dev ? *step_dev + 1 : i_host;             // (in a captured step: the registered counter holds i - 1)
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (counter && b == 0) *counter = i;       // position offset of the NEXT replay of a step graph captured at step 1: (i + 1) - 1
  if (b >= B) return;
  const bool fin = finished[b] != 0;
  const int t = tok[b];
  if (!fin) {
    ids[(long)b * ld_ids + i + 1] = t;
    lps[(long)b * ld_lps + i] = lp[b] * inv_temp;
    if (t == eos) { done_step[b] = i + 1; finished[b] = 1; }
  }
  cur[b] = t;
}
// tok int32 [B], lp fp32 [B], finished uint8 [B], ids int64 [B, ld_ids], lps fp32 [B, ld_lps], done_step int64 [B],
// cur int64 [B] (the next step's input tokens); i = index of the step that produced tok; counter (optional): the device
// int32 a captured decode step reads as its position offset - set to i (saves the per-step fill launch in front of the graph)
extern "C" int tell_greedy_update(const int* tok, const float* lp, uint8_t* finished, long* ids, long ld_ids, float* lps,
                                  long ld_lps, long* done_step, long* cur, int B, int i, int eos, float inv_temp, int* counter,
                                  const int* step_dev, hipStream_t stream) {
  if (B <= 0) return TELL_OK;
  hipLaunchKernelGGL(greedy_update_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, tok, lp, finished, ids, ld_ids, lps,
                     ld_lps, done_step, cur, B, i, eos, inv_temp, counter, step_dev);
  return tell_check_launch("greedy_update");
}
