/* tell_hip.h - C ABI of libtell_hip.so: the MI355X (gfx950) kernels of the
 * Transform-and-Tell caption hot path.
 *
 * The reference (alasdairtran/transform-and-tell) is pure Python: it has no FFI.
 * Each entry point below replaces the ATen/cuDNN/cuBLAS op *sequence* that the
 * cited reference lines issue; the Python host modules in
 * `transform-and-tell_amd/` (same class names, constructor arguments and
 * state_dict keys as the reference) call these through ctypes.  INTEGRATION.md
 * shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; caller owns all memory; the library
 *     allocates nothing.  Its state, all of it set through this header: a thread-local
 *     error string; four REGISTERED device pointers - two that kernels read while a
 *     hipGraph is recorded / replayed (tell_set_rng_step_ptr, tell_set_pos_step_ptr: the
 *     dropout step and decode position counters), the per-device tile-counter buffer of
 *     the resident GEMM launches (tell_gemm_set_tile_queue) and a thread-local one-shot
 *     hook that arms the next tell_gemm_nt launch with a span stamp (tell_gemm_ts_next);
 *     and the table of named integer OPTIONS below (tell_set_option), which is the only
 *     way to steer a launcher's choice of kernel.  The library reads NO environment
 *     variable;
 *   - every call is asynchronous on `stream` (pass torch's current stream);
 *   - return 0 on success, <0 on error (tell_last_error() explains);
 *   - `dtype`: TELL_F32 = 0 (exact-f32 parity mode, f32 MFMA), TELL_BF16 = 1;
 *     activations / working copies of weights are in `dtype`, statistics, biases,
 *     norm parameters, losses, master weights and gradients of weights are fp32;
 *   - `*_dev` pointers are DEVICE scalars read by the kernel (static shapes, no
 *     host synchronisation: row counts of vocabulary bands, learning rate, ...);
 *   - dropout: keep(idx) = hash(seed, salt, idx) >= p*2^32, scaled by 1/(1-p);
 *     the same (seed, salt) given to the backward call reproduces the mask.
 * All file:line citations are relative to the reference root.
 */
#ifndef TELL_HIP_H
#define TELL_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TELL_F32 0
#define TELL_BF16 1
typedef struct ihipStream_t* tell_stream_t; /* hipStream_t */

/* ---- library ------------------------------------------------------------ */
int tell_abi_version(void);
/* Named run-time options (csrc/options.h lists keys and defaults: "gemm_q4", "q4_dynamic", "gemm_s64", "conv_tile", ...).
 * Each is one process-global integer read by the launchers when a launch is ISSUED (a launch recorded into a hipGraph
 * keeps the choice it was recorded with).  Every option selects between kernels that compute the SAME result (to the
 * rounding the parity tests state); the wrong-result timing ablations of tools/probes/ exist only in the probe build
 * (-DTELL_PROBES, libtell_hip_probes.so; tell_probe_build() = 1) - in the shipped library their keys are unknown.
 * tell_set_option -> 0, or -1 for an unknown key (tell_last_error names it); tell_get_option -> the value, or LONG_MIN
 * for an unknown key; tell_option_key(i) / tell_option_default(i) enumerate the table (NULL / LONG_MIN past its end). */
int tell_set_option(const char* key, long value);
long tell_get_option(const char* key);
const char* tell_option_key(int index);
long tell_option_default(int index);
int tell_probe_build(void);
int tell_device_count(void);
const char* tell_last_error(void);
void tell_set_error(const char* msg);
/* host evaluation of the dropout RNG / threshold used by every kernel (csrc/common.h): the 16-bit field element idx
 * is judged by (kept iff field >= threshold) */
uint32_t tell_keep_field_host(uint32_t seed, uint32_t salt, uint64_t idx);
/* hipGraph capture of kernels with dropout: while `counter` (device uint32) is registered, every RNG kernel adds
 * counter * odd-constant to its salt, so a captured graph draws fresh masks after the owner increments it;
 * NULL (default) = eager behaviour.  The stream argument is ignored (uniform binding signature). */
int tell_set_rng_step_ptr(const void* counter, tell_stream_t stream);
/* the same mechanism for the position offset of the embedder (positional.py:170-173 keeps it in incremental_state):
 * while registered, tell_embed_finalize adds *counter to start_pos - one captured decode step serves every position */
int tell_set_pos_step_ptr(const void* counter, tell_stream_t stream);
/* in-graph bookkeeping of a captured decode step: `next` (device uint32, or NULL) holds the position offset of the NEXT
 * step; tell_embed_gather_step - the step's first kernel - reads it and publishes it in the counter registered above,
 * tell_greedy_update / tell_beam_update - its last - write the following offset back (their `counter` argument) */
int tell_set_pos_next_ptr(void* next, tell_stream_t stream);
uint32_t tell_drop_threshold_host(float p);
/* measurement aid (bench.py roofline): rate of the device wall clock in kHz (100 000 on MI355X) */
int tell_wall_clock_khz(void);
/* measurement aid (bench.py --cu-hog, the single-GPU rehearsal of what RCCL's channel kernels do to a step whose dominant
 * GEMM needs whole compute units - SURVEY 8e): n workgroups that each hold one CU (64 KB of LDS) for `ticks` of the device
 * wall clock (capped at 4 s), or until *stop (device int32, may be NULL) becomes non-zero, and do nothing else */
int tell_cu_hog(int n_workgroups, long ticks, int* stop, tell_stream_t stream);
/* arm the NEXT tell_gemm_nt launch of this thread: its direct-to-LDS / ping-pong kernel records its execution span
 * into ts[0] (first workgroup in) / ts[1] (last workgroup out), device wall-clock ticks; ts is uint64[3], zero before
 * the first use (ts[2] counts workgroup arrivals: every launch of the same grid re-opens the span by itself) */
int tell_gemm_ts_next(void* ts, tell_stream_t stream);
/* register the tile counters of the RESIDENT 256x256 GEMM launches (gemm_nt_pp2 / q4 / q4e kernels) FOR THE CALLING THREAD'S
 * CURRENT DEVICE: `counters` = n zero-initialised int32 on that device, owned by the caller and alive for as long as GEMMs
 * are launched there (NULL / 0 unregisters: resident launches then walk static tile lists).  The buffer is cut into slots
 * of 8 counters (one per XCD); a launch with more tiles than workgroups takes one slot as its work queues and leaves it
 * zero.  Launches recorded into a hipGraph keep their slot while the graph lives and take it from the first half of the
 * buffer (released slots first, then fresh ones; when the half is used up, later captures use static lists); eager
 * launches walk a ring over the second half.  Hand-out is thread-safe.  The host mirror registers 2^20 counters per device. */
int tell_gemm_set_tile_queue(void* counters, int n, tell_stream_t stream);
/* bookkeeping of the slots captured launches keep (host-only calls, no stream): _log_begin arms a thread-local record of the
 * slot numbers the calling thread's captured launches take from now on; _log_end returns how many were taken (copying at
 * most `cap` of them to `out`) and disarms it; _release gives slots back to the current device's free list (call it when
 * the graph that recorded them is destroyed); _stats: out[0] = slots, out[1] = fresh captured slots handed out so far,
 * out[2] = free-list length. */
int tell_gemm_tile_queue_log_begin(void);
int tell_gemm_tile_queue_log_count(void);   /* slots logged so far (the log stays open): sizes the buffer for _log_end */
int tell_gemm_tile_queue_log_end(int* out, int cap);
int tell_gemm_tile_queue_release(const int* slots, int n);
int tell_gemm_tile_queue_stats(int* out);

/* ---- exact-erf GELU as its own launch (bf16; y may be x) ----------------------
 * the activation between fc1 and fc2 of fairseq's TransformerSentenceEncoderLayer (activation_fn = gelu) when it is kept
 * out of the fc1 GEMM's epilogue (act 2 of tell_gemm_nt is the fused form; same formula). */
int tell_gelu(const void* x, void* y, long n, int dtype, tell_stream_t stream);

/* ---- GEMM with the transformer sub-layer residual in its epilogue -----------
 * out[M,N] = res[M,N] + dropout_p(A[M,K] . B[N,K]^T + bias[n]), bf16, mask = the one tell_layernorm_fwd draws for
 * (seed, salt) over an [M,N] input.  Replaces  x = residual + F.dropout(self.out_proj(attn)) / F.dropout(self.fc2(h))
 * of fairseq's TransformerSentenceEncoderLayer as called by roberta.extract_features
 * (transformer_faces_objects.py:352-353); the LayerNorm that follows then reads ONE tensor.
 * Returns 1 (nothing launched) when the shape is not whole rounds of 256x256 tiles: run tell_gemm_nt +
 * tell_layernorm_fwd(res) instead. */
int tell_gemm_nt_dropout_residual(const void* A, long lda, const void* B, long ldb, const float* bias, const void* res,
                                  long ld_res, void* C, long ldc, int M, int N, int K, float p, uint32_t seed,
                                  uint32_t salt, tell_stream_t stream);

/* ---- GEMM (every nn.Linear / F.linear / 1x1 conv on the path) --------------
 * C[M,N] = act((A[M,K] . B[N,K]^T + bias) * alpha) (+ C if accumulate)
 * Replaces F.linear in tell/modules/linear.py:8-33 (GehringLinear),
 * multi_head.py:488-526 (in_proj_q/k/v, out_proj), dynamic.py:300 (weight_linear),
 * softmax.py:24-40,182-189 (head / tail projections), adaptive.py:73 (band
 * projections) and the convolutions of resnet.py:92-108 (NHWC rows).
 * bias_mode 0 none | 1 bias[n] | 2 bias[m];  act 0 none | 1 relu | 2 gelu(erf) |
 * 3 multiply by (aux[m,n] > 0) (relu backward) | 4 relu(result + aux[m,n]) (residual block);  m_dev: optional device row count. */
int tell_gemm_nt(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                 int in_dtype, int out_dtype, const float* bias, int bias_mode, int act, const void* aux,
                 float alpha, int accumulate, const int* m_dev, tell_stream_t stream);

/* The kernel tell_gemm_nt would launch for exactly these arguments, as a readable label ("gemm_nt_pp_kernel<bf16,256,256>",
 * "gemm_nt_glds_kernel<bf16,128,128>", ...); nothing is launched.  Measurement aid (bench.py's roofline block). */
const char* tell_gemm_nt_plan(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                 int in_dtype, int out_dtype, const float* bias, int bias_mode, int act, const void* aux,
                 float alpha, int accumulate, const int* m_dev, tell_stream_t stream);

/* bf16 GEMM with K-major operands (no transposed copies in HBM): trans_a -> A stored [K][M] (lda >= M),
 * trans_b -> B stored [K][N].  The backward forms of the same reference lines: autograd of F.linear gives
 * grad_weight = grad_out^T . x (trans_a = trans_b = 1) and grad_in = grad_out . W (trans_b = 1).
 * a_colsum (trans_a only, may be NULL): a_colsum[m] += a_colsum_scale * sum_k A[k][m] - grad_bias = column sums
 * of grad_out, taken from the tiles the wgrad GEMM stages anyway. */
int tell_gemm_bf16(const void* A, long lda, int trans_a, const void* B, long ldb, int trans_b, void* C, long ldc,
                   int M, int N, int K, int out_dtype, const float* bias, int bias_mode, int act, const void* aux,
                   float alpha, int accumulate, const int* m_dev, float* a_colsum, float a_colsum_scale,
                   tell_stream_t stream);

/* n independent bf16 products in a few launches (the weight gradients of a backward pass queued until its end, the
   query / output projections of a decoder layer's context attentions, their input gradients):
       C_i[M,N] (+)= act((op(A_i) op(B_i)^T + bias_i) * alpha_i)
   trans_a / trans_b as in tell_gemm_bf16: 0/0 the NT form (A [M,K], B [N,K]); 0/1 B stored [K,N] (dX = dY W);
   1/1 A stored [K,M] as well (dW = dY^T X), which may carry asum_i[m] += asum_scale_i * sum_k A_i[k][m] (the bias
   gradient).  bias: fp32 per column (bias_mode 1), NT form only; act: 0 none, 1 relu.  `problems` is a HOST array. */
/* second half of a split-K product whose K slices ran as fp32 problems of tell_gemm_grouped (skinny decode-step GEMMs):
   out[M,N] = act((sum_s partial[s][M][N] + bias) * alpha); partial slice s at partial + s * split_stride floats;
   bias fp32 [N] or NULL; act 0 none / 1 relu / 2 gelu(erf); N and ldc multiples of 4. */
int tell_splitk_reduce(const float* partial, int splits, long split_stride, int M, int N, const float* bias, int act,
                       float alpha, void* out, long ldc, int out_dtype, tell_stream_t stream);
/* the same with a device-side row count: rows >= *m_dev (may be NULL) are neither read nor written */
int tell_splitk_reduce2(const float* partial, int splits, long split_stride, int M, int N, const float* bias, int act,
                       float alpha, void* out, long ldc, int out_dtype, const int* m_dev, tell_stream_t stream);

typedef struct tell_gemm_problem {
  const void* A; long lda;
  const void* B; long ldb;
  void* C; long ldc;
  int M, N, K;
  int trans_a, trans_b;
  int out_dtype;          /* of C: TELL_F32 / TELL_BF16 */
  int accumulate;         /* C += */
  float alpha;
  const float* bias;      /* fp32 [N] or NULL */
  int bias_mode;          /* 0 none, 1 per column */
  int act;
  float* asum;            /* fp32 [M] or NULL */
  float asum_scale;
  const int* lim_dev;     /* device int32 or NULL: the number of VALID ROWS of the batch dimension - rows of A (and C) in the
                           * nt / nn forms (M: tiles past it are skipped, their C rows stay untouched), the reduction
                           * length in the tn form (K: rows past it of both K-major operands are not read).  The adaptive
                           * softmax's tails work on fixed-capacity buffers with a device-side count (adaptive.py:61-76
                           * without the mask.any() / nonzero() host syncs). */
} tell_gemm_problem;
int tell_gemm_grouped(int n, const tell_gemm_problem* problems, tell_stream_t stream);

/* ---- casts / transposes / weight norm -------------------------------------- */
int tell_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long n, tell_stream_t stream);
/* dst = (dst_dtype)(src * *scale_dev) (scale_dev NULL = 1; dst may alias src for fp32): the fp32 flat gradient on its way
 * to the data-parallel exchange, weighted by the rank's share of the global token count - what scaling the loss
 * before backward() does in the eager schedule (new functionality; ref loop callback_apex_trainer.py:208-247) */
int tell_scale_cast(const float* src, void* dst, int dst_dtype, long n, const float* scale_dev, tell_stream_t stream);
/* dst_t[c][r] = src[r][c] * row_scale[r]; optional dst_plain[r][c] = same value (either may be NULL) */
int tell_transpose(const void* src, long ld_src, int src_dtype, void* dst_t, long ld_t, void* dst_plain,
                   long ld_p, int dst_dtype, const float* row_scale, int rows, int cols, tell_stream_t stream);
/* weight norm, tell/modules/linear.py:33 (torch weight_norm dim=0): scale[r] = g[r]/||v[r]||, norms[r] = ||v[r]|| */
int tell_wn_rowscale(const float* g, const float* v, int rows, int cols, float* scale, float* norms,
                     tell_stream_t stream);
/* the whole working weight in one pass: w[r,:] = g[r] * v[r,:] / ||v[r,:]|| (bf16 or fp32) and norms[r] */
int tell_wn_weight(const float* g, const float* v, int rows, int cols, void* w, int out_dtype, float* norms,
                   tell_stream_t stream);
int tell_wn_backward(const float* dW, const float* g, const float* v, const float* norms, int rows, int cols,
                     float* dg, float* dv, tell_stream_t stream);
/* The same two passes over n GehringLinears in ONE launch (per 32 tensors).  Every array argument is a HOST array of
   n entries: device pointers (g [rows], v [rows, cols], w [rows, cols], norms [rows], dW / dv [rows, cols],
   dg [rows]) or sizes. */
int tell_wn_weight_multi(int n, const void* const* g, const void* const* v, void* const* w, void* const* norms,
                         const int* rows, const int* cols, int out_dtype, tell_stream_t stream);
int tell_wn_backward_multi(int n, const void* const* dW, const void* const* g, const void* const* v,
                           const void* const* norms, const int* rows, const int* cols, void* const* dg,
                           void* const* dv, tell_stream_t stream);
/* store (host int[n], may be NULL): tensor j's dg / dv are written (beta = 0) instead of accumulated - for gradients
 * whose only producer in a step is this launch and that tell_bertadam_step2 was told not to zero (keep_grad). */
int tell_wn_backward_multi2(int n, const void* const* dW, const void* const* g, const void* const* v,
                            const void* const* norms, const int* rows, const int* cols, void* const* dg,
                            void* const* dv, const int* store, tell_stream_t stream);

/* ---- elementwise ------------------------------------------------------------ */
/* nn.GLU, decoder_faces_objects.py:194-195,259-261: h = [a | gate] */
int tell_glu_fwd(const void* h, void* y, long rows, int C, int dtype, tell_stream_t stream);
int tell_glu_bwd(const void* h, const void* dy, void* dh, long rows, int C, int dtype, tell_stream_t stream);
/* F.dropout, decoder_faces_objects.py:106,257 */
int tell_dropout(const void* x, void* y, long n, float p, uint32_t seed, uint32_t salt, int dtype,
                 tell_stream_t stream);
/* bias gradients: out[c] (+)= sum_r x[r][c] */
int tell_colsum_chunks(int rows);
int tell_colsum(const void* x, long ld, int rows, int C, int dtype, float* out, int accumulate,
                const int* m_dev, float scale, float* workspace, tell_stream_t stream);
/* F.relu backward (decoder_faces_objects.py:360): dx = dy * (y > 0) */
int tell_relu_bwd(const void* dy, const void* y, void* dx, long n, int dtype, tell_stream_t stream);
int tell_axpy(const void* x, void* y, long n, float alpha, int dtype, tell_stream_t stream);
/* out = x0 + ... + x[n_in-1] (dense, same shape; unused pointers NULL): the gradient fan-in autograd performs with
 * n-1 pairwise adds where one activation feeds several branches (decoder_faces_objects.py:271-352) */
int tell_sum_n(const void* x0, const void* x1, const void* x2, const void* x3, const void* x4, const void* x5,
               const void* x6, const void* x7, int n_in, void* out, long n, int dtype, tell_stream_t stream);
int tell_fill_f32(float* x, long n, float value, tell_stream_t stream);
int tell_sum_f32(const float* x, int n, const int* m_dev, float* out, int accumulate, tell_stream_t stream);
/* X.index_select(0, idx) / inverse, softmax.py:184-189 */
int tell_gather_rows(const void* src, long ld_src, const int* idx, const int* count_dev, int cap, void* dst,
                     long ld_dst, int C, int dtype, tell_stream_t stream);
int tell_scatter_add_rows(const void* src, long ld_src, const int* idx, const int* count_dev, int cap, void* dst,
                          long ld_dst, int C, int dtype, tell_stream_t stream);
/* NaN-padded face/object rows -> mask + zeros, transformer_faces_objects.py:373-379 */
int tell_nan_rows(const float* x, int rows, int C, void* y, int out_dtype, uint8_t* mask, tell_stream_t stream);
/* softmax(bert_weight)-weighted sum of the 25 RoBERTa layers, transformer_faces_objects.py:355-364 */
int tell_mix_fwd(const void* H, const float* w, int L, long n, void* out, int dtype, tell_stream_t stream);
int tell_mix_bwd(const void* H, const void* dOut, int L, long n, float* partial, int n_blocks, int dtype,
                 tell_stream_t stream);

/* gw[l] += softmax(w)[l] * (d[l] - sum_j softmax(w)[j] d[j]), d = column sums of tell_mix_bwd's partial [n_blocks, L]:
 * the gradient of the mixing logits, one launch (25 scalars). */
int tell_mix_wgrad(const float* partial, int n_blocks, int L, const float* w, float* gw, tell_stream_t stream);
/* out[0] = x[0] / (ln 2 * n_valid[0]): summed cross entropy (nats) -> bits per target token, transformer_faces_objects.py:85-88
 * (and the gradient of the sum from the gradient of the loss). */
int tell_loss_bits(const float* x, const int* n_valid, float* out, tell_stream_t stream);

/* ---- beam search bookkeeping (the loop of transformer_faces_objects.py:443-494 widened to K hypotheses)
 * tk int32 / lp fp32 [B,K,K]: the K best continuations of every hypothesis (tell_adaptive_logprob_topk); per sample the
 * K best of cum[parent] + lp / temperature (a finished hypothesis continues with pad at no cost; lowest index wins a
 * tie); in place: cum fp32 [B,K], finished uint8 [B,K], seqs int64 [B,K,L] (column step + 1 written), lps fp32
 * [B,K,L-1] (column step); out: cur int64 [B*K] next input tokens, rows int64 [B*K] the row each survivor descends from.
 * back (optional): the ancestor table int32 [n_back <= 31][B*K] of the DynamicConv rings (tell_dynconv_step), composed in
 * place with this step's parents: new back[0][r] = rows[r], new back[j][r] = old back[j-1][rows[r]] - the reference's
 * reorder_incremental_state (dynamic.py:338-342) without moving a row.  counter (optional): device int32 that a captured
 * decode step reads as its position offset (tell_set_pos_step_ptr); set to `step`, the offset of step + 1.  step_dev
 * (optional): the launch is PART of a captured step - the step index is *step_dev + 1 instead of `step`. */
int tell_beam_update(const int* tk, const float* lp, float* cum, uint8_t* finished, long* seqs, float* lps, long* cur,
                     long* rows, int B, int K, int L, int step, int pad, int eos, float inv_temp, int* back, int n_back,
                     int* counter, const int* step_dev, tell_stream_t stream);
/* buf[i][p][r][:] <- buf[i][p][rows[r]][:] in place for n <= 8 bf16 buffers [planes[i], M, 1024] (HOST arrays); rows[r]
 * must lie inside r's group of K consecutive rows (dynamic.py:338-342 reorder_incremental_state, all layers at once) - for
 * input buffers kept in time order (the layer-by-layer fp32 step); the rings of tell_dynconv_step are never moved. */
int tell_reorder_rows(int n, void* const* bufs, const int* planes, const long* rows, int M, int C, int K,
                      tell_stream_t stream);

/* ---- LayerNorm: y = LN(res + dropout(x)), decoder_faces_objects.py:263-266,367-372 */
int tell_layernorm_fwd(const void* x, long ld_x, const void* res, long ld_r, const float* gamma,
                       const float* beta, void* y, long ld_y, float* mean, float* rstd, int rows, int C,
                       float eps, float p, uint32_t seed, uint32_t salt, int dtype, tell_stream_t stream);
int tell_layernorm_bwd_blocks(int rows);
int tell_layernorm_bwd(const void* dy, long ld_dy, const void* x, long ld_x, const void* res, long ld_r,
                       const float* gamma, const float* mean, const float* rstd, void* dx, long ld_dx,
                       void* dres, long ld_dres, int dres_accumulate, float* dgamma, float* dbeta,
                       int dparam_accumulate, float* partial, int rows, int C, float p, uint32_t seed,
                       uint32_t salt, int dtype, tell_stream_t stream);

/* ---- batched small launches of the training step (csrc/multi.hip) ------------------------------------------------
 * n column sums in one launch: dst0[i][c] += sum_r src[i][r][c] for c < w0[i], dst1[i][c - w0[i]] += ... for the rest
 * (fp32, deterministic order).  The jobs: LayerNorm gamma / beta gradients from the per-row-block partials that
 * tell_layernorm_bwd / tell_layernorm_cat_bwd leave when called with dgamma == NULL ([blocks][2C] each), and the
 * bias_k / bias_v gradients of the context attentions (multi_head.py:355-374 backward: per-batch rows [B][2E]).
 * All arrays are HOST arrays of length n. */
int tell_colsum_multi(int n, const void* const* src, const long* ld, const int* rows, const int* width, const int* w0,
                      void* const* dst0, void* const* dst1, tell_stream_t stream);
/* n bf16 transposes in one launch, dst[i][c][r] = src[i][r][c]: the per-step W^T copies of the backward pass's
 * large input-gradient GEMMs (fc2, context_fc, the stacked article K|V projection). */
int tell_transpose_multi(int n, const void* const* src, const long* ld_src, void* const* dst, const long* ld_dst,
                         const int* rows, const int* cols, tell_stream_t stream);
/* out = add + dropout(x) with tell_dropout's mask (same seed / salt / element index): the gradient of a block input
 * that is both the residual and, through the input dropout, the branch input (decoder_faces_objects.py:256-266). */
int tell_dropout_add(const void* x, const void* add, void* out, long n, float p, uint32_t seed, uint32_t salt, int dtype,
                     tell_stream_t stream);
/* ---- the generation step (transformer_faces_objects.py:443-494, decoder_faces_objects.py:224-352 at T = 1), csrc/decode.hip
 * At M = batch x beam <= 128 rows every linear layer is a weight-streaming problem; these three replace the training
 * kernels on that path (12 launches per decoder layer instead of ~24).
 *
 * tell_skinny_linear: out[p] = (act(prologue(in[p]) . w[p]^T + bias[p])) * scale + residual, n_prob <= 4 problems of one
 * shape per launch (HOST arrays of n_prob pointers), bf16 weights [N (2N with GLU), K], M <= 1024, K % 256 == 0.
 *   pro 3 / 4 (round 5): the LayerNorm FOLDED into the product - in = the bf16 PRE-norm rows, w = the weights scaled by the
 *   LayerNorm's gamma along K, gamma[p] / beta[p] carry problem p's fp32 vectors s [K / seg][N (2N)] and c [N (2N)] with
 *   s[g][n] = sum_k w[n][k] over segment g, c[n] = sum_k W[n][k] beta[k]; the kernel gathers mean / rstd of its rows (per
 *   `seg` columns for pro 4: K / seg in {1, 2, 4}, one problem) from its own operands and applies
 *   LN(x) . W^T = rstd (x . w^T) - rstd mean s + c in the epilogue; stats_out (pro 3) receives (mean, rstd); act 0 or 2.
 *   pro 0: in bf16 [M,K].  pro 1: in fp32 [M,K] (a pre-norm `residual + branch`), LayerNorm(gamma[0], beta[0], eps) first;
 *   stats_out (optional, [M][2] fp32) receives (mean, rstd) per row.  pro 2: one LayerNorm per `seg` columns of in
 *   (gamma[s], beta[s], K / seg <= 4): the four LayerNorms that end the context block feeding context_fc.  pro 1 / 2 are
 *   one extra launch into ws (bf16 [M,K], required then, NULL otherwise); all problems must read the same input; rows /
 *   segments of 1024 .. 4096 columns.
 *   act 0 none, 1 ReLU, 2 GLU (gate rows at n + N, fairseq's linear1 + F.glu, dynamic.py / decoder_faces_objects.py:229-233).
 *   residual: res (bf16 [M,N]), LayerNorm(res_raw) rebuilt from res_stats ([M][2]), res_gamma, res_beta, res_f32 (fp32
 *   [M,N]); each may be NULL.
 *   out bf16 or (out_f32) fp32 [M,N]; out2 (optional, bf16): columns n >= out2_from of the result once more, at
 *   out2[m][p * N + n - out2_from] for problem p (the softmax head: cluster logits in fp32 and the tails' projected inputs in bf16 from one
 *   launch).
 *   split_ws (optional, split_ws_bytes >= 64 KB + the partial tiles; 16-byte aligned, ZEROED ONCE when allocated and then
 *   owned by the launches of ONE stream): lets a single-problem launch with few column tiles and a long reduction (N <=
 *   1536, K >= 2048: context_fc, fc2) share the reduction of a tile between 2 / 4 workgroups that combine inside the launch
 *   (write-through partial tiles, an agent-scope arrival counter per tile - never reset -, the last arrival sums the
 *   slices in slice order: deterministic).  NULL = every workgroup owns its whole reduction.  Taken only with option "sk_split"
 *   = 1 (default 0: with the operands staged through LDS the unsplit form is as fast). */
int tell_skinny_linear(int n_prob, const void* const* in, long ld_in, int pro, const void* const* gamma,
                       const void* const* beta, int seg, float eps, float* stats_out, void* ws, const void* const* w,
                       long ldw, const void* const* bias, int act, float scale, const void* res, long ld_res,
                       const float* res_raw, long ld_res_raw, const float* res_stats, const float* res_gamma,
                       const float* res_beta, const float* res_f32, long ld_res_f32, void* out2, long ld_out2,
                       int out2_from, void* const* out, long ld_out, int out_f32, int M, int N, int K, void* split_ws,
                       long split_ws_bytes, tell_stream_t stream);
/* y (bf16) = LayerNorm(x) of fp32 rows [M, C], C = 1024 .. 4096; stats_out (optional): (mean, rstd) per row [M][2]. */
int tell_layernorm_rows(const float* x, long ld_x, const float* gamma, const float* beta, float eps, void* y, long ld_y,
                        float* stats_out, int M, int C, tell_stream_t stream);
/* Front half of the step's token embedding (adaptive.py:61-76 + positional.py:167-211 at T = 1): cat[m] (bf16 [M, ktot])
 * = the token's band-table row placed at columns off_b .. off_b + dim_b - 1 of its band b (zeros elsewhere), pos_out[m]
 * (fp32 [M,E]) = the sinusoid row of this step (pad -> row pos_pad; offset start_pos + the captured step's device
 * counter).  scale * (cat . [proj_0 | proj_1 | ..]^T) + pos_out - one tell_skinny_linear - is the embedding.
 * tables / lo / hi / dim / off: HOST arrays of nb <= 4 bands (ids lo_b .. hi_b - 1). */
/* the step's embedding as a LOOKUP in a pre-projected table (generation: the weights do not move): table fp32 [V, E] =
 * embed_scale * proj_band . table_band[v] for every token (adaptive.py:61-76, built once by the caller);
 * out[m] = bf16(table[ids[m]] + sinusoid[position of this step]) (positional.py:167-211; pad -> row pos_pad).  First kernel
 * of a captured decode step: reads / publishes the position counter like tell_embed_gather_step. */
int tell_embed_lookup_step(const long* ids, int M, const float* table, int V, const float* pos_table, int pos_rows,
                           int pos_pad, int start_pos, void* out, int E, tell_stream_t stream);
int tell_embed_gather_step(const long* ids, int M, int nb, const void* const* tables, const int* lo, const int* hi,
                           const int* dim, const int* off, void* cat, int ktot, const float* pos_table, int pos_rows,
                           int pos_pad, int start_pos, float* pos_out, int E, tell_stream_t stream);
/* DynamicConv1dTBC with an input buffer, one step (dynamic.py:85-120, :285-336 at T = 1): x [M,C] bf16, wt [H*K, C] bf16
 * (weight_linear, no bias), y [M,C] bf16.  C = H * 64, K <= 32.  hist [K][M][C] bf16 is a RING of K planes indexed by
 * time: the input of step s lives in plane s mod K, in the slot its hypothesis had at step s; planes never written hold
 * zeros (cleared by the caller per caption).  t = index of this step (+ the registered decode position counter while a
 * hipGraph of the step is recorded, tell_set_pos_step_ptr); the step reads planes t-1 .. t-K+1 and writes x into plane
 * t mod K.  back (optional): int32 [>= K-1][M], back[j-1][m] = slot, j steps ago, of the hypothesis now in slot m (beam
 * search, maintained by tell_beam_update); NULL = m itself. */
int tell_dynconv_step(const void* x, void* hist, const void* wt, void* y, int M, int C, int H, int K, int t,
                      const int* back, tell_stream_t stream);
/* MultiHeadAttention at Tq = 1 against n_ctx <= 4 static key / value caches in one launch (multi_head.py:330-352,
 * :376-475): HOST arrays of n_ctx entries; q[c] / out[c] [B, H*64] with row strides q_sb / o_sb, element (b,s,h,d) of
 * k[c] at k + s*k_ss + (b / beams)*k_sb + h*64 + d (the `beams` hypotheses of a sample - rows b*beams + j - share its
 * cache), mask[c] [B / beams, S[c]] uint8 or NULL, bias_k[c] / bias_v[c] [H*64] (:355-364) or NULL, has_zero: the zero
 * row (:416-421).  bf16, q pre-scaled, S <= 2048 (S = 0 allowed with a bias / zero row: the empty context :349-374). */
/* The same attention over a PACKED cache, on the matrix cores (the generation loop owns its cache's layout):
 *   both operands in MFMA FRAGMENT ORDER (a wave-wide 16-byte load reads one contiguous KB); the learned bias_k / bias_v row
 *      and the zero row (multi_head.py:355-364, :416-421) are keys S and S + 1, keys up to Sp (a multiple of 32) are zero;
 *   kc [B/beams, H, Sp/16, 2, 64, 8] bf16 keys: per tile of 16 keys and half c of the head width, lane l holds elements
 *      c * 32 + (l >> 4) * 8 .. + 7 of key l & 15;
 *   vt [B/beams, H, Sp/32, 4, 64, 8] bf16 values TRANSPOSED: per block of 32 keys and tile rt of 16 dimensions, lane l holds
 *      dimension rt * 16 + (l & 15) of the keys 4 g + j (j < 4) and 16 + 4 g + j - 4 (j >= 4), g = l >> 4 - the keys whose scores
 *      an MFMA accumulator leaves in k-group g (host mirror: transform-and-tell_amd/decode.py PackedKV.fill);
 *   mask [B/beams, Sp] uint8, 1 = masked (context padding and every position past S + 1).
 * q[c] bf16 [B, H*64] projected and scaled, out[c] bf16 [B, H*64]; head width 64; the beams hypotheses of a sample are
 * columns of one MFMA (any beams).  HOST arrays of n_ctx <= 4 entries. */
int tell_attn_decode_packed(int n_ctx, const void* const* q, const long* q_sb, const void* const* kc, const void* const* vt,
                            const void* const* mask, const int* Sp, void* const* out, const long* o_sb, int B, int H,
                            int beams, tell_stream_t stream);
int tell_attn_decode(int n_ctx, const void* const* q, const long* q_sb, const void* const* k, const long* k_ss,
                     const long* k_sb, const long* k_sh, const void* const* v, const long* v_ss, const long* v_sb,
                     const long* v_sh, const void* const* mask, const void* const* bias_k, const void* const* bias_v,
                     int has_zero, const int* S, void* const* out, const long* o_sb, int B, int H, int beams,
                     tell_stream_t stream);
/* n LayerNorms over ONE residual in one launch each way - the end of a decoder layer's context block
   (decoder_faces_objects.py:283-352): y[:, i*C:(i+1)*C] = LayerNorm_i(res + dropout_p(x_i)), mean / rstd [n, rows].
   Backward: dx_i (entries may be NULL), dres = sum_i dz_i (may be NULL), dgamma_i / dbeta_i ACCUMULATED; partial is a
   workspace of n * tell_layernorm_bwd_blocks(rows) * 2 * C floats.  x, gamma, beta, dx, dgamma, dbeta, salts are HOST
   arrays of n (<= 8) entries; every x_i / dx_i shares one row stride.  bf16, C = 512 or 1024, 16-byte aligned rows. */
int tell_layernorm_cat_fwd(int n, const void* const* x, long ld_x, const void* res, long ld_r,
                           const float* const* gamma, const float* const* beta, void* y, long ld_y, float* mean,
                           float* rstd, int rows, int C, float eps, float p, uint32_t seed, const uint32_t* salts,
                           int dtype, tell_stream_t stream);
int tell_layernorm_cat_bwd(int n, const void* dcat, long ld_dcat, const void* const* x, long ld_x, const void* res,
                           long ld_r, const float* const* gamma, const float* mean, const float* rstd,
                           void* const* dx, long ld_dx, void* dres, long ld_dres, float* const* dgamma,
                           float* const* dbeta, float* partial, int rows, int C, float p, uint32_t seed,
                           const uint32_t* salts, int dtype, tell_stream_t stream);

/* ---- LSTM decoder of the GloVe/LSTM baseline (tell/models/decoder_flattened_lstm.py) ----
 * nn.LSTMCell (:20-26, :160-161): g1 = x W_ih^T + b_ih and g2 = h W_hh^T + b_hh come from tell_gemm_nt ([B,4H], chunk
 * order i f g o); this applies the gate non-linearities and the state update.  gates: [B,4H] fp32, activated (saved). */
int tell_lstm_cell_fwd(const void* g1, const void* g2, const void* c_prev, void* h, void* c, float* gates, int B, int H,
                       int dtype, tell_stream_t stream);
/* dh / dc may be NULL; dgates ([B,4H], the gradient of both g1 and g2) and dc_prev are written */
int tell_lstm_cell_bwd(const void* dh, const void* dc, const float* gates, const void* c, const void* c_prev,
                       void* dgates, void* dc_prev, int B, int H, int dtype, tell_stream_t stream);
/* AttentionLayer.forward (:40-60) between its two projections: scores[l,b] = <src[l,b,:], x[b,:]>, key-padding mask
 * ([B,L] uint8, may be NULL), softmax over l, ctx[b,:] = sum_l probs[l,b] src[l,b,:].  src element (l,b,d) at
 * l*src_stride_l + b*src_stride_b + d; probs: [L,B] fp32 (the returned attention scores, saved); L <= 1024. */
int tell_dot_attn_fwd(const void* src, long src_stride_l, long src_stride_b, const void* x, const unsigned char* mask,
                      void* ctx, float* probs, int L, int B, int D, int dtype, tell_stream_t stream);
/* gradient w.r.t. the projected query x and - dsrc != NULL, contiguous [L,B,D] - w.r.t. the source states (needed when
 * they are the trainable `weigh_bert` mix of the RoBERTa layers, expt/3_lstm_roberta) */
int tell_dot_attn_bwd(const void* src, long src_stride_l, long src_stride_b, const float* probs, const void* dctx,
                      const void* x, void* dx, void* dsrc, int L, int B, int D, int dtype, tell_stream_t stream);
/* torch.tanh around output_proj (:62) */
int tell_tanh_fwd(const void* x, void* y, long n, int dtype, tell_stream_t stream);
int tell_tanh_bwd(const void* dy, const void* y, void* dx, long n, int dtype, tell_stream_t stream);

/* ---- DynamicConv1dTBC core, tell/modules/convolutions/dynamic.py:285-336 (T x B x C)
 * taps = softmax_K(logits) (:302-304), DropConnect (:305), causal K-tap weighted sum;
 * replaces the band-matrix build + bmm (:318-335).  taps: [T*B*H, K] fp32 (saved for backward). */
int tell_dynconv_fwd(const void* x, const void* logits, void* y, float* taps, int T, int B, int H, int K,
                     int R, float p, uint32_t seed, uint32_t salt, int dtype, tell_stream_t stream);
int tell_dynconv_bwd(const void* x, const void* dy, const float* taps, void* dx, int dx_accumulate,
                     void* dlogits, int T, int B, int H, int K, int R, float p, uint32_t seed, uint32_t salt,
                     int dtype, tell_stream_t stream);
/* The core of the decoder's conv block as one launch (decoder_faces_objects.py:259-261 GLU + dynamic.py:300-336):
 * h1 [T*B, 2E] bf16 = linear1's output (a | gate), w_tap [H*K, E] bf16 = DynamicConv.weight_linear (no bias) ->
 * gl = GLU(h1) [T*B, E], y [T*B, E], taps [T*B*H, K] fp32 (softmax of the tap logits, before DropConnect: what
 * tell_dynconv_bwd reads).  Replaces tell_glu_fwd + the tap-logit tell_gemm_nt + tell_dynconv_fwd; the logits stay
 * fp32 on the chip.  Returns 1 (nothing launched) for shapes it does not take: bf16 only, T <= 32, K <= 32, head
 * width 64, E = 1024. */
int tell_dynconv_block_fwd(const void* h1, const void* w_tap, void* gl, void* y, float* taps, int T, int B, int H,
                           int K, float p, uint32_t seed, uint32_t salt, tell_stream_t stream);

/* ---- MultiHeadAttention core, tell/modules/attention/multi_head.py:376-475
 * element (b,h,t,d) of q at q + t*q_st + b*q_sb + h*D + d (k, v, out likewise);
 * bias_k/bias_v (:355-364) and the zero row (:416-421) are virtual keys S, S+1;
 * mask [B,S] uint8 (:442-458); fp32 softmax (:460-462); prob dropout (:463). */
int tell_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse, const uint8_t* mask,
                  const void* bias_k, const void* bias_v, int B, int H, int Tq, int S, int D, long q_st,
                  long q_sb, long k_ss, long k_sb, long v_ss, long v_sb, long o_st, long o_sb, int has_zero,
                  float p, uint32_t seed, uint32_t salt, int dtype, tell_stream_t stream);
/* dbias_k / dbias_v: fp32 [B, H*D] per-sample partials of the bias_k / bias_v gradients (the caller sums over B), each
   with row stride H*D - or, when dbias_v == dbias_k + H*D, the two column halves of one [B, 2*H*D] buffer. */
int tell_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout,
                  const float* lse, const uint8_t* mask, const void* bias_k, const void* bias_v, void* dq,
                  void* dk, void* dv, float* dbias_k, float* dbias_v, int B, int H, int Tq, int S, int D,
                  long q_st, long q_sb, long k_ss, long k_sb, long v_ss, long v_sb, long o_st, long o_sb,
                  int has_zero, float p, uint32_t seed, uint32_t salt, int dtype, tell_stream_t stream);

/* head-averaged attention weights [B,Tq,S'] for need_weights (multi_head.py:478-482), eval/demo only */
int tell_attn_avg_weights(const void* q, const void* k, const float* lse, const uint8_t* mask,
                          const void* bias_k, float* w, int B, int H, int Tq, int S, int D, long q_st, long q_sb,
                          long k_ss, long k_sb, int has_zero, int dtype, tell_stream_t stream);

/* ---- adaptive input embedding / adaptive softmax ----------------------------
 * tell_adaptive_partition: device-side replacement of the boolean-mask / nonzero()
 * logic of adaptive.py:64-74 and softmax.py:144-167 (adapt_target). */
int tell_adaptive_partition(const long* ids, int N, const int* cutoffs_host, int n_bands, int pad_idx,
                            int* band_rows, int* band_local, int* band_count, int* slot, int* head_target,
                            int* n_valid, tell_stream_t stream);
/* out = scale * band_out[slot] + sinusoid[pos], positional.py:167-211,231-268, sum_text_field_embedder.py:117-118 */
int tell_embed_finalize(const void* band_out, const int* slot, const long* ids, const float* pos_table,
                        int pos_rows, void* out, int B, int T, int E, float scale, int pos_pad, int start_pos,
                        int tbc, int dtype, tell_stream_t stream);
int tell_embed_finalize_bwd(const void* dout, const int* slot, void* dband, int B, int T, int E, float scale,
                            int tbc, int dtype, tell_stream_t stream);
int tell_embed_table_grad(const void* drows, long ld, const int* local, const int* count_dev, int cap,
                          float* demb, int dim, int padding_idx, int dtype, tell_stream_t stream);
/* F.cross_entropy(ignore_index, reduction='sum') per cluster, adaptive_loss.py:55-60 */
int tell_ce_fwd(const float* logits, long ld, int M, int V, const int* targets, const int* row_idx,
                const int* m_dev, int ignore_index, float* lse, float* loss, tell_stream_t stream);
int tell_ce_bwd(const float* logits, long ld, int M, int V, const int* targets, const int* row_idx,
                const int* m_dev, int ignore_index, const float* lse, const float* gscale_dev, void* dlogits,
                long ld_d, int dtype, tell_stream_t stream);
/* get_log_prob + topk(1), softmax.py:193-222, transformer_faces_objects.py:443-464 */
/* the k (<= 8) best (token, log-prob) pairs per row, best first - what a beam of k needs from each hypothesis;
 * `tokens`, `lps`: [rows, k] */
int tell_adaptive_logprob_topk(const float* head, long ld_head, int c0, int n_tails, const float* tail0, long ld0,
                               int n0, const float* tail1, long ld1, int n1, const float* tail2, long ld2, int n2,
                               int rows, int k, int* tokens, float* lps, tell_stream_t stream);
int tell_adaptive_logprob_argmax(const float* head, long ld_head, int c0, int n_tails, const float* tail0,
                                 long ld0, int n0, const float* tail1, long ld1, int n1, const float* tail2,
                                 long ld2, int n2, int rows, float* log_probs, long ld_lp, int* token,
                                 float* token_lp, tell_stream_t stream);

/* per-token bookkeeping of the greedy decode loop (transformer_faces_objects.py:443-494) for all B rows in one launch:
   unfinished rows record tok / lp * inv_temp at step i, rows emitting eos are marked finished (done_step = i + 1),
   cur = tok for the next step.  tok int32 [B], lp fp32 [B], finished uint8 [B], ids int64 [B, ld_ids], lps fp32
   [B, ld_lps], done_step int64 [B], cur int64 [B].  counter (optional): device int32 that a captured decode step reads
   as its position offset (tell_set_pos_step_ptr); set to i, the offset of step i + 1.  step_dev (optional): the launch is
   PART of a captured step - the step index is *step_dev + 1 instead of i. */
int tell_greedy_update(const int* tok, const float* lp, uint8_t* finished, long* ids, long ld_ids, float* lps,
                       long ld_lps, long* done_step, long* cur, int B, int i, int eos, float inv_temp, int* counter,
                       const int* step_dev, tell_stream_t stream);

/* ---- BertAdam (config.yaml:126-149), flat fp32 buffers, tensors CHUNK-aligned */
int tell_opt_chunk(void);
/* shadow_bf16 (may be NULL): bf16 copy of the updated parameters, same flat layout - the working weights of the
 * next forward; zero_grad: clear the gradient buffer in the same pass (callback_apex_trainer.py:214);
 * grad_wire_bf16 (may be NULL): the step's gradient in bf16, same flat layout - what a data-parallel exchange with
 * bf16 on the wire leaves behind; it is then READ instead of `grad` (which is still the buffer that gets cleared). */
int tell_bertadam_step(float* param, float* grad, float* m, float* v, const int* chunk_tensor,
                       const long* chunk_begin, long n_chunks, int n_tensors, float* partial, float* norms,
                       const float* lr_dev, float b1, float b2, float eps, float wd, float max_norm,
                       float grad_scale, void* shadow_bf16, int zero_grad, int* skip, const void* grad_wire_bf16,
                       int* step_dev, float lr_base, float warmup, float t_total, tell_stream_t stream);
/* The same step with keep_grad (device int32[n_tensors], may be NULL): tensors with keep_grad[t] != 0 are NOT zeroed by
 * zero_grad - their single weight-gradient product of the next backward pass stores over them (accumulate = 0), which
 * saves the 4 B / parameter zero write here and the 4 B / parameter read there.  The host side (training/optimizers.py,
 * ops.wgrad_target) decides per tensor from who wrote what in an observed first step. */
int tell_bertadam_step2(float* param, float* grad, float* m, float* v, const int* chunk_tensor,
                        const long* chunk_begin, long n_chunks, int n_tensors, float* partial, float* norms,
                        const float* lr_dev, float b1, float b2, float eps, float wd, float max_norm,
                        float grad_scale, void* shadow_bf16, int zero_grad, int* skip, const void* grad_wire_bf16,
                        int* step_dev, float lr_base, float warmup, float t_total, const int* keep_grad,
                        tell_stream_t stream);
/* step_dev (device int32, may be NULL): the count of updates APPLIED so far.  When given, the library first writes
 * *lr_dev = lr_base * warmup_linear(*step_dev / t_total, warmup) (t_total <= 0: lr_base) and the update kernel
 * increments *step_dev only when the step is not skipped - a skipped batch costs no tick of the schedule, exactly as in
 * the reference, where it never reaches optimizer.step().  NULL: the caller filled *lr_dev itself. */
/* skip (int[2], may be NULL): skip[0] != 0 -> the step leaves parameters / moments untouched (gradient still cleared)
 * and skip[1] counts such steps; tell_loss_flag sets skip[0] = !isfinite(loss) (the NaN-loss skip of
 * callback_apex_trainer.py:225-227 without a host sync), the norm pass ORs in 2 for a non-finite gradient (what apex
 * amp O2's overflow check does). */
int tell_loss_flag(const float* loss, int* skip, tell_stream_t stream);

/* ---- ResNet-152 trunk helpers, tell/models/resnet.py:92-108 (NHWC) ----------- */
/* ToTensor + Normalize(mean, std) of the dataset readers (nytimes_faces_ner_matched.py:67-69) on the device:
 * uint8 [B,H,W,3] -> float32 [B,3,H,W] */
int tell_image_normalize(const uint8_t* x, float* y, int B, int H, int W, float m0, float m1, float m2, float s0,
                         float s1, float s2, tell_stream_t stream);
int tell_nchw_to_nhwc(const float* x, void* y, int B, int C, int H, int W, int out_dtype, tell_stream_t stream);
/* fp32 NCHW (C <= 4) -> bf16 NHWC with FOUR channels per pixel (missing ones zero): the input of the implicit 7x7 stem
 * (resnet.py:92-96) - tell_conv_bn_stats / tell_conv_bn_act / tell_conv_bias_act called with Cin = 4, KH = KW = 7,
 * stride 2, pad 3 and w [Cout, 256] laid out as 8 kernel rows (the 8th zero) x 8 window columns (the FIRST zero: the
 * window of output column ow starts at input column 2 ow - 4) x 4 channels (the 4th zero); W must be even. */
int tell_nchw_to_nhwc4(const float* x, void* y, int B, int C, int H, int W, tell_stream_t stream);
/* conv (1x1 / 3x3, stride 1 / 2, Cin = 64 * 2^n; or the 7x7 stem, see tell_nchw_to_nhwc4) as an IMPLICIT GEMM on the matrix cores + the statistics of the
 * train-mode BatchNorm behind it (resnet.py:92-108: torchvision Bottleneck conv -> bn; BN in batch-stat mode per
 * callback_apex_trainer.py:259): no im2col matrix, no statistics pass over the activation.  x [B,H,W,Cin] bf16,
 * w [Cout, KH*KW*Cin] bf16, y [B*OH*OW, Cout] bf16 raw conv output; mean / invstd [Cout] (+ running stats update), or
 * mean == NULL for the convolution alone.  workspace: 2 * ceil(B*OH*OW / 64) * Cout floats; zero_page: >= 16 zero
 * bytes, 16-byte aligned (source of the padding ring). */
int tell_conv_bn_stats(const void* x, const void* w, void* y, int B, int H, int W, int Cin, int KH, int KW, int stride,
                       int pad, int OH, int OW, int Cout, float eps, float momentum, float* mean, float* invstd,
                       float* running_mean, float* running_var, float* workspace, const void* zero_page,
                       tell_stream_t stream);
/* conv -> train-mode BatchNorm (-> + residual) (-> ReLU) of a Bottleneck (resnet.py:92-108 via torchvision; the frozen
 * trunk runs in train mode, callback_apex_trainer.py:259): the implicit-GEMM convolution of tell_conv_bn_stats, then the
 * combine of the per-row-chunk statistics AND the normalisation as ONE launch (two when the activation has more than
 * 128 row chunks).  y [B*OH*OW, Cout] bf16 receives the finished activation; residual (same shape) or NULL;
 * running_mean / running_var get the momentum update.  workspace: 2 * ceil(M / 64) * Cout + 258 * Cout floats (chunk statistics, mean / invstd, and the
 * super-chunk statistics of the > 128-chunk case: one parallel merge launch in front of the fused finish + apply). */
int tell_conv_bn_act(const void* x, const void* w, void* y, int B, int H, int W, int Cin, int KH, int KW, int stride,
                     int pad, int OH, int OW, int Cout, float eps, float momentum, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, const void* residual, int relu, float* workspace,
                     const void* zero_page, tell_stream_t stream);
/* the same block with the trunk in eval mode (running statistics): BatchNorm folded into the convolution by the host
 * (w' = w * gamma / sqrt(running_var + eps), bias = beta - running_mean * that): y = act(conv(x, w') + bias [+ residual])
 * in ONE launch; relu = 1 with a residual gives relu(conv + bias + residual). */
int tell_conv_bias_act(const void* x, const void* w, void* y, int B, int H, int W, int Cin, int KH, int KW, int stride,
                       int pad, int OH, int OW, int Cout, const float* bias, const void* residual, int relu,
                       const void* zero_page, tell_stream_t stream);
int tell_im2col(const void* x, void* col, int B, int H, int W, int Cin, int KH, int KW, int stride, int pad,
                int OH, int OW, int Kp, int dtype, tell_stream_t stream);
/* im2col of y = relu(BatchNorm(x)): the producer's normalisation is applied while gathering (resnet.py Bottleneck:
 * conv1 -> bn1 -> relu -> conv2); padding taps are zeros of the post-activation tensor. */
int tell_im2col_bn(const void* x, void* col, int B, int H, int W, int Cin, int KH, int KW, int stride, int pad,
                   int OH, int OW, const float* mean, const float* invstd, const float* gamma, const float* beta,
                   int relu, int dtype, tell_stream_t stream);
long tell_bn_chunks(long M);
/* conv (as GEMM over NHWC rows) + the batch statistics of the BatchNorm2d that follows it, in one pass: the GEMM
 * epilogue reduces each output tile's columns, a small second kernel combines the tiles (resnet.py:94-108 with
 * the trunk in train mode).  bf16 only; workspace: 2 * ceil(M/64) * N floats. */
int tell_gemm_bn_stats(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                       float eps, float momentum, float* mean, float* invstd, float* running_mean,
                       float* running_var, float* workspace, tell_stream_t stream);
int tell_bn_stats(const void* x, long M, int C, float eps, float momentum, float* mean, float* invstd,
                  float* running_mean, float* running_var, float* workspace, int dtype, tell_stream_t stream);
int tell_bn_apply(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                  const void* residual, void* y, long M, int C, int relu, int dtype, tell_stream_t stream);
int tell_maxpool3x3s2(const void* x, void* y, int B, int H, int W, int C, int OH, int OW, int dtype,
                      tell_stream_t stream);

/* ---- RoBERTa input embedding (fairseq roberta.large, call site transformer_faces_objects.py:352) */
int tell_roberta_embed(const long* ids, int B, int S, int pad, const void* word, const void* posemb, int* pos_ws,
                       void* out, int E, int dtype, tell_stream_t stream);

int tell_mask_rows(void* x, const uint8_t* mask, long rows, int C, int dtype, tell_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TELL_HIP_H */
