// RoBERTa-large input embedding (fairseq TransformerSentenceEncoder: token table +
// learned positions, positions = cumsum(non-pad) * non-pad + padding_idx).
#include "common.h"

// one workgroup per sequence; S <= 1024
__global__ __launch_bounds__(1024) void roberta_positions_kernel(const long* __restrict__ ids, int S, int pad,
                                                                 int* __restrict__ pos) {
  __shared__ int wave_tot[16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool ok = tid < S && ids[(long)b * S + tid] != pad;
  const unsigned long long bal = __ballot(ok);
  const int incl = __popcll(bal & ((2ull << lane) - 1ull));   // inclusive prefix inside the wave
  if (lane == 0) wave_tot[wave] = __popcll(bal);
  __syncthreads();
  int off = 0;
  for (int w = 0; w < wave; ++w) off += wave_tot[w];
  if (tid < S) pos[(long)b * S + tid] = ok ? off + incl + pad : pad;
}
// out[n] = word[ids[n]] + posemb[pos[n]]
template <typename T>
__global__ __launch_bounds__(256) void embed2_kernel(const long* __restrict__ ids, const int* __restrict__ pos,
                                                     const T* __restrict__ word, const T* __restrict__ posemb,
                                                     T* __restrict__ out, int E) {
  const long n = blockIdx.x;
  const T* w = word + ids[n] * (long)E;
  const T* q = posemb + (long)pos[n] * E;
  T* o = out + n * (long)E;
  for (int c = threadIdx.x; c < E; c += 256) Elem<T>::st(o + c, Elem<T>::ld(w + c) + Elem<T>::ld(q + c));
}
extern "C" int tell_roberta_embed(const long* ids, int B, int S, int pad, const void* word, const void* posemb,
                                  int* pos_ws, void* out, int E, int dtype, hipStream_t stream) {
  if (B * S <= 0) return TELL_OK;
  TELL_REQUIRE(S <= 1024, "roberta_embed: sequence longer than 1024");
  hipLaunchKernelGGL(roberta_positions_kernel, dim3(B), dim3(1024), 0, stream, ids, S, pad, pos_ws);
  if (dtype == TELL_BF16) hipLaunchKernelGGL((embed2_kernel<uint16_t>), dim3(B * S), dim3(256), 0, stream, ids, pos_ws, (const uint16_t*)word, (const uint16_t*)posemb, (uint16_t*)out, E);
  else hipLaunchKernelGGL((embed2_kernel<float>), dim3(B * S), dim3(256), 0, stream, ids, pos_ws, (const float*)word, (const float*)posemb, (float*)out, E);
  return tell_check_launch("roberta_embed");
}

// zero the rows whose mask byte is set (fairseq: x *= 1 - padding_mask after the embedding LayerNorm)
template <typename T>
__global__ __launch_bounds__(256) void mask_rows_kernel(T* __restrict__ x, const uint8_t* __restrict__ mask,
                                                        long rows, int C) {
  for (long r = blockIdx.x; r < rows; r += gridDim.x)
    if (mask[r])
      for (int c = threadIdx.x; c < C; c += 256) x[r * C + c] = (T)0;
}
extern "C" int tell_mask_rows(void* x, const uint8_t* mask, long rows, int C, int dtype, hipStream_t stream) {
  if (rows <= 0) return TELL_OK;
  int g = rows < 4096 ? (int)rows : 4096;
  if (dtype == TELL_BF16) hipLaunchKernelGGL((mask_rows_kernel<uint16_t>), dim3(g), dim3(256), 0, stream, (uint16_t*)x, mask, rows, C);
  else hipLaunchKernelGGL((mask_rows_kernel<float>), dim3(g), dim3(256), 0, stream, (float*)x, mask, rows, C);
  return tell_check_launch("mask_rows");
}
