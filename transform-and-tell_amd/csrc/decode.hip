// The generation step (transformer_faces_objects.py:443-494: one new token per caption and step) as a short chain of
// weight-streaming kernels.
//
// At M = B x beam <= 128 rows every linear layer of the decoder is a GEMV-class problem: 2-8 MB of weights read once,
// a few MFLOP.  Run through the training GEMMs a decode step was 107 launches of 5-25 us - 16-64 workgroups walking
// dependent K tiles, split-K partials folded by a second launch, GLU / dropout / LayerNorm / concatenation launches in
// between - at 5-8 % of the HBM roofline.  Here a layer is 12 launches:
//   skinny_mfma_kernel   out[M,N] = epilogue(in[M,K] . W[N,K]^T): a workgroup owns 32 rows x 4-16 output columns and the
//                        whole reduction (no partial sums in memory, deterministic); its 4 waves split K.  Rounds 3-5:
//                        MFMA fragments loaded straight from memory (no LDS staging, no barrier before the first MFMA);
//                        round 6: every wave stages 8 rows x 128 bytes per load instruction through its own slice of
//                        LDS - the fragment-order access kept the address unit, not the memory, busy (STAGED, below).
//                        Epilogues: bias, scale, ReLU, GLU (the gate column j + N travels with column j), residual
//                        (bf16 rows, fp32 rows, or LayerNorm of fp32 rows rebuilt from row statistics), bf16 or fp32
//                        output (+ a bf16 copy of the trailing columns).  Up to 4 problems per launch (the query /
//                        output projections of a layer's context attentions).
//   ln_rows_kernel       LayerNorm of the fp32 `residual + branch` rows a producer left behind, to bf16, per 1024-column
//                        segment (the four LayerNorms that end the context block, decoder_faces_objects.py:283-352, are
//                        one launch).  Doing this inside the consumer's staging was measured: every one of its 256-512
//                        workgroups then normalises all rows again (~1000 VALU instructions per wave and chunk,
//                        +6 us per 1024 columns) - the 3 us launch wins.
//   dynconv_step_kernel  the DynamicConv1dTBC step (dynamic.py:85-120 with an input buffer): tap logits of a head
//                        (K x 1024 dot products), softmax over the taps, the K-tap sum over the buffered rows and the
//                        buffer shift.
//   attn_decode_kernel   the 2 / 4 one-query context attentions against the projected K/V cache, one launch.
// A first version of the linear ran on the VALU (v_dot2_f32_bf16) with the activation rows staged in LDS by every
// workgroup: 256 workgroups re-reading 64-256 KB of rows each made it L2-bound (10-30 us per launch at 32-128 rows);
// the register-direct MFMA form measures 5-8 us at 32 rows and 5-17 us at 128.
#include "common.h"
#include "options.h"
#include "../../include/tell_hip.h"

#define SK_MAXP 4
struct SkinnyArgs {
  const void* in[SK_MAXP];              // bf16 [M,K]
  const uint16_t* w[SK_MAXP];           // bf16 [N (2N with GLU), K]
  const float* bias[SK_MAXP];           // fp32 [N (2N)] or null
  void* out[SK_MAXP];                   // bf16 or fp32 [M,N]
  const float* gamma[SK_MAXP];          // ln_rows_kernel: one per column segment
  const float* beta[SK_MAXP];
  long ld_in, ldw, ld_out, ld_res, ld_res_raw;
  float* stats_out;                     // ln_rows_kernel: (mean, rstd) per row, [M][2], or null
  const uint16_t* res;                  // epilogue residual, bf16 [M,N] - or
  const float* res_raw;                 //   LayerNorm(res_raw)[m][n] rebuilt from res_stats [M][2], res_gamma, res_beta
  const float* res_stats; const float* res_gamma; const float* res_beta;
  const float* res_f32; long ld_res_f32;   //   and / or fp32 rows [M,N]
  uint16_t* out2; long ld_out2; int out2_from;   // columns n >= out2_from also as bf16 at out2[m][n - out2_from], or null
  int cn;                                  // mfma kernel: columns per workgroup (4 / 8 / 16)
  float eps, scale;
  int M, N, K, seg, out_f32;
  // FOLDED LayerNorm (pro 3 / 4): `in` holds the bf16 PRE-norm rows, w the weights scaled by gamma along K; the kernel
  // gathers the row statistics from its own A fragments and the epilogue applies
  //   LN(x) . W^T = rstd (x . W'^T) - rstd mean s + c,   W' = W gamma,  s[n] = sum_k W'[n][k],  c[n] = sum_k W[n][k] beta[k]
  // per segment of `seg` columns (the context block's n LayerNorms feeding context_fc: s is [K / seg][N], c their sum)
  const float* fold_s[SK_MAXP];
  const float* fold_c[SK_MAXP];
  long out2_prob;                          // out2 column offset per problem (n_prob outputs side by side), elements
  // in-launch split of the reduction over `ksplit` workgroups per output tile (round 6): fp32 partial tiles in `sk_slab`
  // ([tile][slice][MT * 16 (* 2 with GLU)]), arrivals counted in sk_cnt[tile] (monotonic: never reset)
  int ksplit; float* sk_slab; unsigned* sk_cnt;
#ifdef TELL_PROBES
  // probe build: per-workgroup wall-clock stamps of ONE launch slot ([1 + workgroups][4] u64: entry 0 = the launch's shape;
  // then start, end of the K loop, end, hardware id) - tools/probes/skinny_stamps.py
  unsigned long long* stamp;
#endif
};

typedef __attribute__((ext_vector_type(2))) __bf16 sk_bf16x2;
// Register arrays that live across a conditional (the prefetch of the next chunk) are native vectors: arrays of HIP's
// uint4 / float4 structs assigned under a condition are not promoted to registers by the compiler (they go to scratch).
typedef uint32_t sk_u4 __attribute__((ext_vector_type(4)));
typedef float sk_f4 __attribute__((ext_vector_type(4)));
template <typename VA, typename VB>
__device__ __forceinline__ float sk_dot8(const VA& a, const VB& b, float acc) {
#if __has_builtin(__builtin_amdgcn_fdot2_f32_bf16)
  // (elements copied out first: __builtin_bit_cast applied directly to an ext-vector element expression reads .x for all)
  const uint32_t a0 = a.x, a1 = a.y, a2 = a.z, a3 = a.w, b0 = b.x, b1 = b.y, b2 = b.z, b3 = b.w;
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(sk_bf16x2, a0), __builtin_bit_cast(sk_bf16x2, b0), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(sk_bf16x2, a1), __builtin_bit_cast(sk_bf16x2, b1), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(sk_bf16x2, a2), __builtin_bit_cast(sk_bf16x2, b2), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(sk_bf16x2, a3), __builtin_bit_cast(sk_bf16x2, b3), acc, false);
#else
  const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    acc = fmaf(__uint_as_float(aw[i] << 16), __uint_as_float(bw[i] << 16), acc);
    acc = fmaf(__uint_as_float(aw[i] & 0xffff0000u), __uint_as_float(bw[i] & 0xffff0000u), acc);
  }
#endif
  return acc;
}

// 64-lane sum on the VALU (DPP butterflies inside each row of 16 lanes, then the four row totals through SGPRs): the
// LayerNorm prologues and the tap softmax issue dozens of these per workgroup; through ds_bpermute (__shfl_xor) each
// is a chain of six LDS round trips.
template <int CTRL>
__device__ __forceinline__ float sk_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float sk_wave_sum(float v) {
  v += sk_dpp<0xB1>(v);        // quad_perm [1,0,3,2]
  v += sk_dpp<0x4E>(v);        // quad_perm [2,3,0,1]
  v += sk_dpp<0x141>(v);       // row_half_mirror
  v += sk_dpp<0x140>(v);       // row_mirror
  const int i = __builtin_bit_cast(int, v);
  return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16))) +
         (__builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48)));
}

// ------------------------------------------------------------------ skinny linear on the matrix cores, register-direct
// Up to 16 output columns per workgroup, the reduction split over its 4 waves; a wave loads its A (activation rows) and B
// (weight rows) fragments of mfma_f32_16x16x32_bf16 straight from memory in the instruction's own layout - lane l holds
// 8 consecutive k of row / column (l & 15) at k-group (l >> 4) - so there is no LDS staging, no barrier before the
// first MFMA, and the whole K range of a wave can be in flight at once: one round trip, then RT x K/128 MFMAs, one LDS
// fold of the 4 partial tiles, the epilogue.  (A and B use the same k placement, so the product does not depend on
// the hardware's k order inside the instruction; C/D: column = lane & 15, row = (lane >> 4) * 4 + register.)
// A workgroup covers 32 rows (blockIdx.z: row group) x cn <= 16 columns: with few column tiles (N = 1024: 64) a tile
// is shared out over 2 or 4 workgroups by COLUMNS (cn = 8 / 4; the other lanes of the B fragment repeat a column and
// their results are dropped) - every workgroup still owns its outputs entirely.  (Splitting the reduction over
// workgroups instead, with a last-arriver fix-up, was measured: the device-scope fence it needs writes back the XCD's
// L2 - 25 us per launch against 6.)
// U: k-steps of 32 per batch of loads (two batches are in flight).
typedef __attribute__((ext_vector_type(8))) __bf16 sk_bf16x8;
// what an output element's epilogue reads from memory: requested BEFORE the reduction (round 5).  Fetched behind it, the
// bias / residual / LayerNorm-rebuild operands were one more dependent memory round trip at the tail of every launch of
// the step's chain (~35 launches).
struct SkPre { float b0, b1, res, raw, mean, rstd, gam, bet, r32; };
__device__ __forceinline__ SkPre skinny_preload(const SkinnyArgs& p, int prob, int m, int n, int act) {
  SkPre q = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float* bias = p.bias[prob];
  if (bias) { q.b0 = bias[n]; if (act == 2) q.b1 = bias[p.N + n]; }
  // (the bf16 residual stays as the 16 bits the load delivers: shifting it into a float HERE made the compiler wait for
  //  the load - and, loads returning in order, for the whole first batch of operand loads issued before it - in front of
  //  the K loop; the epilogue shifts)
  if (p.res) q.res = __uint_as_float((uint32_t)p.res[(long)m * p.ld_res + n]);
  if (p.res_raw) {
    q.raw = p.res_raw[(long)m * p.ld_res_raw + n]; q.mean = p.res_stats[m * 2]; q.rstd = p.res_stats[m * 2 + 1];
    q.gam = p.res_gamma[n]; q.bet = p.res_beta[n];
  }
  if (p.res_f32) q.r32 = p.res_f32[(long)m * p.ld_res_f32 + n];
  return q;
}
__device__ __forceinline__ void skinny_epilogue(const SkinnyArgs& p, int prob, int m, int n, float v, float g, int act,
                                                const SkPre* pre = nullptr) {
  const int N = p.N;
  const float* bias = p.bias[prob];
  if (pre) {
    v += pre->b0;
    if (act == 1) v = fmaxf(v, 0.f);
    if (act == 2) v = tell_glu(v, g + pre->b1);
    v *= p.scale;
    {                                                                // (bf16 bits -> float; 0 stays 0.  The empty asm pins the
      uint32_t rb = __float_as_uint(pre->res);                       //  shift HERE: the optimizer otherwise moves it back up to
      asm volatile("" : "+v"(rb));                                   //  the load, and the wait with it)
      v += __uint_as_float(rb << 16);
    }
    if (p.res_raw) v += (pre->raw - pre->mean) * pre->rstd * pre->gam + pre->bet;
    v += pre->r32;
  } else {
  if (bias) v += bias[n];
  if (act == 1) v = fmaxf(v, 0.f);
  if (act == 2) {
    if (bias) g += bias[N + n];
    v = tell_glu(v, g);
  }
  v *= p.scale;
  if (p.res) v += __uint_as_float((uint32_t)p.res[(long)m * p.ld_res + n] << 16);
  if (p.res_raw)
    v += (p.res_raw[(long)m * p.ld_res_raw + n] - p.res_stats[m * 2]) * p.res_stats[m * 2 + 1] * p.res_gamma[n] + p.res_beta[n];
  if (p.res_f32) v += p.res_f32[(long)m * p.ld_res_f32 + n];
  }
  if (p.out2 && n >= p.out2_from) p.out2[(long)m * p.ld_out2 + prob * p.out2_prob + n - p.out2_from] = f2bf(v);
  if (p.out_f32) static_cast<float*>(p.out[prob])[(long)m * p.ld_out + n] = v;
  else static_cast<uint16_t*>(p.out[prob])[(long)m * p.ld_out + n] = f2bf(v);
}

// NW: waves of the workgroup = slices of the reduction.  (Round 5 measured NW = 8 for K = 4096 - context_fc, fc2: two
// batches of loads per wave instead of four - same box: greedy step 494.5 -> 507.3 us, beam 4 822 -> 834; only 4 is built.)
// SPLIT (round 6): the reduction of an output tile shared by p.ksplit workgroups (blockIdx.z = row group * ksplit + slice).
// The K = 4096 linears of the step (context_fc, fc2: N = 1024 = 64 column tiles) ran 256 workgroups of 4 columns x the WHOLE
// reduction - every workgroup pulls all 32 x 4096 activation elements (256 KB) through its CU for 32 KB of weights, and
// takes 11-14 us where the K = 1024 layers take 5.  Split four ways a workgroup owns 16 columns x 1024 of K: 64 KB of
// activations, one batch of loads per wave.  The combine follows cdna_hip_programming.md 6 G16 / the split-K recipe:
// partial tiles leave as write-through (sc1) stores, every wave drains them (s_waitcnt vmcnt(0)), one lane takes a ticket
// from the tile's agent-scope counter, the workgroup that draws the last ticket of the launch reads all slabs back with
// sc1 loads IN SLICE ORDER (deterministic sum whatever the arrival order) and runs the epilogue.  The counter is monotonic
// (ticket % ksplit == ksplit - 1 marks the last arrival; every launch adds exactly ksplit): nothing to zero between
// launches or graph replays.  With the folded LayerNorm a slice is exactly one segment (K / ksplit == seg): its row
// statistics are complete inside the workgroup and its partial tile is already corrected.
//
// STAGED (round 6, second half): the operands reach the matrix cores through LDS.  Loading an MFMA fragment straight from a
// row-major matrix makes every quarter-wave of a 16-byte load touch 16 ROWS x 16 bytes - the texture addresser spends a
// lookup per row where a lookup can deliver 64 bytes, and that, not HBM, the L2 or latency, is what the K loop waited for:
// per-workgroup wall-clock stamps (tools/probes/skinny_stamps.py, profiles/r06_skinny_stamps_*.txt) show all 256 workgroups
// of a launch alive from its first to its last microsecond (dispatch ramp 0.2-0.3 us, launch-to-launch gap 1.5-2 us) with
// 4 us (32 rows) / 9-11 us (128 rows) inside the K loop = 20-30 GB/s per CU; staggering the column tiles' walks through K
// changed nothing (not an L2 hot spot: fc1 12.79 -> 12.83 us); the same loads issued as one contiguous KB per wave
// instruction (wrong results, probe build) took the launches from 5.2-5.8 to 3.9-4.7 us at 32 rows and from 12.8-16.2 to
// 7.1-10.9 us at 128.  So a wave now reads 8 rows x 128 bytes per instruction (whole cache lines: lane l = row l >> 3,
// 16-byte chunk l & 7) into registers, U stages of 64 k in flight, drops a landed stage into ITS OWN slice of LDS
// (ds_write_b128, chunk position XORed with the row's low 3 bits) and reads the fragments back in the instruction's layout
// (ds_read_b128; the XOR makes both the 8-lane write groups and the 16-lane read groups of MI355X_MICROARCH.md's LDS table
// conflict-free).  No barrier: a wave stages only what it consumes.  Activation rows and the tile's weight rows are the
// same thing to the stage (R = MT + 16 NB rows).  The k -> fragment-slot assignment and the order of the accumulation are
// those of the direct form: results are bit-identical.
template <int RT, int ACT, int U, bool FOLD = false, int NW = 4, bool SPLIT = false, bool STAGED = false>
__global__ __launch_bounds__(64 * NW) void skinny_mfma_kernel(SkinnyArgs p) {
  constexpr int NB = ACT == 2 ? 2 : 1, MT = RT * 16, CW = 16 * NB, NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char sk_smem[];
  float* red = reinterpret_cast<float*>(sk_smem);                      // [NW waves][MT][CW]
  float* wst = red + NW * MT * CW;                                     // FOLD: [NW waves][MT][2] sum, sum of squares of the wave's k range
  float* rst = wst + NW * MT * 2;                                      // FOLD: [MT][4 segments][2] mean, rstd
  const int tid = threadIdx.x, prob = blockIdx.y, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
#ifdef TELL_PROBES
  unsigned long long st0 = 0, st1 = 0, st_sync = 0, st_calc = 0;
  if (p.stamp) st0 = wall_clock64();
  auto stamp_out = [&]() {
    if (!p.stamp || tid != 0) return;
    const long wg = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    unsigned long long* s = p.stamp + (1 + wg) * 4;
    s[0] = st0; s[1] = st1; s[2] = wall_clock64();
    // word 3: XCC_ID (4 bits) | HW_ID CU / SH / SE bits (8 .. 15) | ticks from the end of the K loop to the barrier behind the partial
    // tiles (12 bits) | ticks from there to the last epilogue value computed, stores not yet issued (12 bits)
    const unsigned long long hw = __builtin_amdgcn_s_getreg((31 << 11) | 4 /* HW_ID, 32 bits */);
    const unsigned long long d1 = st_sync > st1 ? (st_sync - st1 < 4095 ? st_sync - st1 : 4095) : 0;
    const unsigned long long d2 = st_calc > st_sync ? (st_calc - st_sync < 4095 ? st_calc - st_sync : 4095) : 0;
    s[3] = (hw & 0xff00ull) | ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20 /* XCC_ID, 4 bits */) << 32) | (d1 << 36) | (d2 << 48);
    if (wg == 0) {
      p.stamp[0] = gridDim.x | ((unsigned long long)gridDim.y << 16) | ((unsigned long long)gridDim.z << 32);
      p.stamp[1] = (unsigned long long)p.M | ((unsigned long long)p.N << 20) | ((unsigned long long)p.K << 40);
      p.stamp[2] = RT | (ACT << 8) | (U << 16) | ((int)FOLD << 24) | ((int)SPLIT << 25) | (NW << 26);
      p.stamp[3] = p.cn;
    }
  };
#endif
  const int ksplit = SPLIT ? p.ksplit : 1, slice = SPLIT ? (int)blockIdx.z % ksplit : 0;
  const int cn = p.cn, n0 = blockIdx.x * cn, m0 = (SPLIT ? (int)blockIdx.z / ksplit : (int)blockIdx.z) * MT, M = p.M, N = p.N;
  const int K = SPLIT ? p.K / ksplit : p.K;                            // this workgroup's share of the reduction
  const int kw = K / NW, nbatch = kw / (32 * U), kbeg = slice * K + wave * kw;
  const uint16_t* X = static_cast<const uint16_t*>(p.in[prob]);
  const uint16_t* W = p.w[prob];
  const uint16_t* ap[RT];
  const uint16_t* bp[NB];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int m = m0 + rt * 16 + lr;
    ap[rt] = X + (long)(m < M ? m : M - 1) * p.ld_in + kbeg + lg * 8;
  }
  const int cl = n0 + (lr < cn ? lr : 0);                              // (lanes past cn repeat column n0: dropped)
  const int col = cl < N ? cl : N - 1;                                  // (columns past N: computed, never stored)
#pragma unroll
  for (int b = 0; b < NB; ++b) bp[b] = W + (long)(b * N + col) * p.ldw + kbeg + lg * 8;
  typedef float c4 __attribute__((ext_vector_type(4)));
  c4 acc[RT][NB];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[rt][b] = c4{0.f, 0.f, 0.f, 0.f};
  sk_u4 fa0[U][RT], fb0[U][NB], fa1[U][RT], fb1[U][NB];
  auto load = [&](sk_u4 (&fa)[U][RT], sk_u4 (&fb)[U][NB], int batch) __attribute__((always_inline)) {
    const int k = batch * 32 * U;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // (round 5: non-temporal loads for the weights here and for the cached K / V in attn_decode_kernel - each byte is read
      //  by one workgroup, once per step - measured: greedy step 486.6-487.5 us plain, 483.9-489.4 nt; not kept)
      // (round 6 measured the B loads of the lanes past cn masked off instead of repeating column n0 - if the address unit's
      //  time per wave instruction were what bounds the cn = 4 launches, that would have removed a quarter of it: context_fc
      //  14.2 -> 15.5 us, fc2 11.3 -> 12.1, linear2 5.0 -> 5.6 at 32 rows (same box within 1 % on the unchanged shapes). Not kept.)
#pragma unroll
      for (int b = 0; b < NB; ++b) fb[u][b] = *reinterpret_cast<const sk_u4*>(bp[b] + k + u * 32);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) fa[u][rt] = *reinterpret_cast<const sk_u4*>(ap[rt] + k + u * 32);
    }
  };
  // FOLD: the row statistics come off the matrix cores too.  X . 1 (B = a fragment of bf16 ones) leaves sum_k x[m][k] in
  // every column of row m; X . X^T (the A fragment passed as B as well: both operands use the same k placement) leaves
  // sum_k x[m][k]^2 on the diagonal.  Two more 16x16x32 MFMAs per A fragment on a pipe that has one to do, instead of 24
  // VALU operations per fragment (first version: 3 us of arithmetic in front of context_fc's 4096-column rows).
  c4 fsum[RT], fsq[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) { fsum[rt] = c4{0.f, 0.f, 0.f, 0.f}; fsq[rt] = c4{0.f, 0.f, 0.f, 0.f}; }
  const sk_u4 ones4 = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  auto compute = [&](const auto& fa, const auto& fb) __attribute__((always_inline)) {
    constexpr int UU = sizeof(fa) / sizeof(fa[0]);                      // k-steps of 32 held by the fragments
    if constexpr (FOLD) {
#pragma unroll
      for (int u = 0; u < UU; ++u)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const sk_bf16x8 av = __builtin_bit_cast(sk_bf16x8, fa[u][rt]);
          fsum[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(sk_bf16x8, ones4), fsum[rt], 0, 0, 0);
          fsq[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, av, fsq[rt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int u = 0; u < UU; ++u)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const sk_u4 bw = fb[u][b];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const sk_u4 aw = fa[u][rt];
          acc[rt][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sk_bf16x8, aw), __builtin_bit_cast(sk_bf16x8, bw),
                                                               acc[rt][b], 0, 0, 0);
        }
      }
  };
  // ---- STAGED: stage = 64 k of the wave's R rows = NL wave loads of 8 rows x 128 bytes
  constexpr int R = MT + 16 * NB, NL = R / 8, SB = R * 128;
  sk_u4 stg[STAGED ? U : 1][STAGED ? NL : 1];
  const int ns = kw / 64;                                              // stages of this wave
  const int lrow = lane >> 3, lch = lane & 7;
  unsigned char* const sbuf = sk_smem + (STAGED ? (long)wave * SB : 0);   // ONE buffer: a wave's LDS instructions execute in order
  const int woff = lrow * 128 + ((lch ^ lrow) << 4);                   // + i * 1024: where this lane's 16 bytes of load i go
  const int roff0 = lr * 128 + ((lg ^ (lr & 7)) << 4);                 // + row tile * 2048: fragment of k-step 0 ...
  const int roff1 = lr * 128 + (((4 + lg) ^ (lr & 7)) << 4);           //   ... and of k-step 1 of the stage
  auto stage_load = [&](sk_u4 (&st)[STAGED ? NL : 1], int sidx) __attribute__((always_inline)) {
    const int k = kbeg + sidx * 64 + lch * 8;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int row = i * 8 + lrow;
      if (i * 8 < MT) {                                                // (MT is a multiple of 8: a load is all activations or all weights)
        const int m = m0 + row;
        st[i] = *reinterpret_cast<const sk_u4*>(X + (long)(m < M ? m : M - 1) * p.ld_in + k);
      } else {
        const int c = row - MT, cc = c & 15;
        const int cq = n0 + (cc < cn ? cc : 0);
        st[i] = *reinterpret_cast<const sk_u4*>(W + (long)((c >> 4) * N + (cq < N ? cq : N - 1)) * p.ldw + k);
      }
    }
  };
  auto stage_write = [&](const sk_u4 (&st)[STAGED ? NL : 1], int) __attribute__((always_inline)) {
    unsigned char* const buf = sbuf;
#pragma unroll
    for (int i = 0; i < NL; ++i) *reinterpret_cast<sk_u4*>(buf + i * 1024 + woff) = st[i];
  };
  auto stage_compute = [&](int) __attribute__((always_inline)) {
    unsigned char* const buf = sbuf;
    if constexpr (RT >= 8) {                                           // (register budget: one k-step's fragments at a time)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int ro = ks ? roff1 : roff0;
        sk_u4 ga[1][RT], gb[1][NB];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) ga[0][rt] = *reinterpret_cast<const sk_u4*>(buf + rt * 2048 + ro);
#pragma unroll
        for (int b = 0; b < NB; ++b) gb[0][b] = *reinterpret_cast<const sk_u4*>(buf + (MT + b * 16) * 128 + ro);
        compute(ga, gb);
      }
    } else {
      sk_u4 ga[2][RT], gb[2][NB];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        ga[0][rt] = *reinterpret_cast<const sk_u4*>(buf + rt * 2048 + roff0);
        ga[1][rt] = *reinterpret_cast<const sk_u4*>(buf + rt * 2048 + roff1);
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        gb[0][b] = *reinterpret_cast<const sk_u4*>(buf + (MT + b * 16) * 128 + roff0);
        gb[1][b] = *reinterpret_cast<const sk_u4*>(buf + (MT + b * 16) * 128 + roff1);
      }
      compute(ga, gb);
    }
  };
  if constexpr (!STAGED) load(fa0, fb0, 0);
  // FOLD: the epilogue's s / c values of this thread's column (o & 15 == tid & 15 for every o it handles) are requested
  // now - behind the reduction they were a dependent global round trip at the very end of the launch
  float pre_s[4][NB], pre_c[NB];
  if constexpr (FOLD) {
    const int n_pre = n0 + (tid & 15) < N ? n0 + (tid & 15) : N - 1;
    const int nsg = K / p.seg;                                          // (SPLIT: the slice covers segments slice * nsg ..)
#pragma unroll
    for (int e = 0; e < NB; ++e) {
      pre_c[e] = p.fold_c[prob][e * N + n_pre];
#pragma unroll
      for (int sg = 0; sg < 4; ++sg) pre_s[sg][e] = sg < nsg ? p.fold_s[prob][(long)(sg + slice * nsg) * N * NB + e * N + n_pre] : 0.f;
    }
  }
  constexpr int EPI = MT * 16 / NT;                                    // output elements per thread
  constexpr bool PRE = EPI <= 8;
  SkPre pre[PRE ? EPI : 1];
  if constexpr (PRE) {
#pragma unroll
    for (int e = 0; e < EPI; ++e) {
      const int o = tid + e * NT, r = o >> 4, c = o & 15;
      const int m = m0 + r < M ? m0 + r : M - 1, n = n0 + c < N ? n0 + c : N - 1;
      pre[e] = skinny_preload(p, prob, m, n, ACT);
    }
  }
  // STAGED: the epilogue operands above were requested FIRST, the operand stages follow, every stage load unconditional
  // (the launcher picks U so that the wave's stage count is a multiple of it): s_waitcnt vmcnt counts loads in issue
  // order, and the compiler, which must assume the fewest younger loads any path may have issued, waits for far more
  // than the stage it needs as soon as a stage load sits under a condition (first version: vmcnt(5) in front of stage 0 -
  // the whole prologue drained before the first MFMA).
  if constexpr (STAGED) {
#pragma unroll
    for (int d = 0; d < U; ++d) stage_load(stg[d], d);
  }
  if constexpr (STAGED) {
    for (int s0 = U; s0 < ns; s0 += U) {                               // every group but the last: consume a stage, refill its registers
#pragma unroll
      for (int d = 0; d < U; ++d) {
        stage_write(stg[d], d & 1);                                    // (the landed stage leaves its registers ...
        stage_load(stg[d], s0 + d);                                    //  ... to the loads of the stage U further on)
        stage_compute(d & 1);
      }
    }
#pragma unroll
    for (int d = 0; d < U; ++d) {                                      // the last U stages: nothing left to request
      stage_write(stg[d], d & 1);
      stage_compute(d & 1);
    }
    __syncthreads();                                                   // (the reduction buffer below lies over the stages)
  } else {
  int b = 0;
  for (; b + 1 < nbatch; b += 2) {
    load(fa1, fb1, b + 1);
    compute(fa0, fb0);
    if (b + 2 < nbatch) load(fa0, fb0, b + 2);
    compute(fa1, fb1);
  }
  if (b < nbatch) compute(fa0, fb0);
  }
#ifdef TELL_PROBES
  if (p.stamp) {                                                       // (the stamp waits for the accumulators: the K loop's loads have landed)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) asm volatile("" ::"v"(acc[rt][0]));
    asm volatile("s_nop 4" ::: "memory");
    st1 = wall_clock64();
  }
#endif
  // ---- fold the 4 waves' partial tiles, epilogue
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((long)wave * MT + rt * 16 + lg * 4 + r) * CW + nb * 16 + lr] = acc[rt][nb][r];
  if constexpr (FOLD) {
    // C layout: column = lane & 15, row = (lane >> 4) * 4 + register.  Row sums: any column (column 0: lanes 0, 16, 32,
    // 48 hold rows lg * 4 + r); sums of squares: the diagonal (row == column: the lane with (lr >> 2) == lg, register lr & 3)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      if (lr == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) wst[((long)wave * MT + rt * 16 + lg * 4 + r) * 2] = fsum[rt][r];
      }
      if ((lr >> 2) == lg) {
        const int r = lr & 3;
        const float q = r == 0 ? fsq[rt][0] : r == 1 ? fsq[rt][1] : r == 2 ? fsq[rt][2] : fsq[rt][3];
        wst[((long)wave * MT + rt * 16 + lr) * 2 + 1] = q;
      }
    }
  }
  __syncthreads();
#ifdef TELL_PROBES
  if (p.stamp) st_sync = wall_clock64();
#endif
  const int nseg = FOLD ? K / p.seg : 1, wps = NW / nseg;              // segments of the row, waves per segment
  // (1 / seg as a factor where seg is a power of two - then the product IS the quotient, bit for bit: two divisions per
  //  closed segment and output element were a third of the folded epilogue's instructions)
  const bool seg_pow2 = (p.seg & (p.seg - 1)) == 0;
  const float inv_seg = 1.f / (float)p.seg;
  (void)rst;
  const int lw = __builtin_ctz(wps);                                   // log2 of the waves per segment
  float sv[SPLIT ? EPI : 1], sg2[SPLIT && ACT == 2 ? EPI : 1];
#pragma unroll
  for (int e = 0; e < EPI; ++e) {
    const int o = tid + e * NT;
    const int r = o >> 4, c = o & 15, m = m0 + r, n = n0 + c;
    if constexpr (!SPLIT) { if (c >= cn || m >= M || n >= N) continue; }
    float v, g = 0.f;
    if constexpr (FOLD) {
      v = SPLIT ? 0.f : pre_c[0];                                      // (SPLIT: c is added once, by the combining workgroup)
      if constexpr (ACT == 2) g = SPLIT ? 0.f : pre_c[NB - 1];
      // one unrolled walk over the waves: partial dot products and row statistics of a segment add up over its waves, the
      // segment closes behind its last wave.  (Until round 6 the statistics were a stage of their own - 128 threads, a second
      // barrier - and the walk two nested run-time loops: 2.2-3 us of epilogue where the plain form takes 0.7.)
      float dv = 0.f, dg = 0.f, sa = 0.f, sb = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        dv += red[((long)w * MT + r) * CW + c];
        if constexpr (ACT == 2) dg += red[((long)w * MT + r) * CW + 16 + c];
        sa += wst[((long)w * MT + r) * 2]; sb += wst[((long)w * MT + r) * 2 + 1];
        if (((w + 1) & (wps - 1)) == 0) {
          const int sg = w >> lw;
          const float mu = seg_pow2 ? __fmul_rn(sa, inv_seg) : sa / (float)p.seg;
          const float ex2 = seg_pow2 ? __fmul_rn(sb, inv_seg) : sb / (float)p.seg;   // (rounded like the quotient: no contraction)
          float var = ex2 - mu * mu;
          var = var > 0.f ? var : 0.f;
          const float rs = rsqrtf(var + p.eps);
          const float ps0 = sg == 0 ? pre_s[0][0] : sg == 1 ? pre_s[1][0] : sg == 2 ? pre_s[2][0] : pre_s[3][0];
          v += rs * (dv - mu * ps0);
          if constexpr (ACT == 2) {
            const float ps1 = sg == 0 ? pre_s[0][NB - 1] : sg == 1 ? pre_s[1][NB - 1] : sg == 2 ? pre_s[2][NB - 1] : pre_s[3][NB - 1];
            g += rs * (dg - mu * ps1);
          }
          if (p.stats_out && nseg == 1 && c == 0 && blockIdx.x == 0 && prob == 0 && m < M) {
            p.stats_out[m * 2] = mu; p.stats_out[m * 2 + 1] = rs;
          }
          dv = dg = sa = sb = 0.f;
        }
      }
    } else {
      v = (red[((long)0 * MT + r) * CW + c] + red[((long)1 * MT + r) * CW + c]) +
          (red[((long)2 * MT + r) * CW + c] + red[((long)3 * MT + r) * CW + c]);
      if constexpr (NW == 8)
        v += (red[((long)4 * MT + r) * CW + c] + red[((long)5 * MT + r) * CW + c]) +
             (red[((long)6 * MT + r) * CW + c] + red[((long)7 * MT + r) * CW + c]);
      if constexpr (ACT == 2) {
        g = (red[((long)0 * MT + r) * CW + 16 + c] + red[((long)1 * MT + r) * CW + 16 + c]) +
            (red[((long)2 * MT + r) * CW + 16 + c] + red[((long)3 * MT + r) * CW + 16 + c]);
        if constexpr (NW == 8)
          g += (red[((long)4 * MT + r) * CW + 16 + c] + red[((long)5 * MT + r) * CW + 16 + c]) +
               (red[((long)6 * MT + r) * CW + 16 + c] + red[((long)7 * MT + r) * CW + 16 + c]);
      }
    }
#ifdef TELL_PROBES
    if (p.stamp && e == EPI - 1) { asm volatile("" ::"v"(v)); st_calc = wall_clock64(); }
#endif
    if constexpr (SPLIT) { sv[e] = v; if constexpr (ACT == 2) sg2[e] = g; }
    else skinny_epilogue(p, prob, m, n, v, g, ACT, PRE ? &pre[e] : nullptr);
  }
#ifdef TELL_PROBES
  if constexpr (!SPLIT) stamp_out();
#endif
  if constexpr (SPLIT) {
    // ---- publish this slice's partial tile (write-through), take a ticket; the last arrival combines
    const long tile = ((long)(blockIdx.z / ksplit) * gridDim.y + prob) * gridDim.x + blockIdx.x;
    constexpr int SLAB = MT * 16 * NB;
    float* slab = p.sk_slab + (tile * ksplit + slice) * SLAB;
#pragma unroll
    for (int e = 0; e < EPI; ++e) {
      __hip_atomic_store(slab + tid + e * NT, sv[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if constexpr (ACT == 2) __hip_atomic_store(slab + MT * 16 + tid + e * NT, sg2[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // EVERY storing wave drains its stores ...
    __syncthreads();                                                    // ... before ONE lane announces the slice
    int* flag = reinterpret_cast<int*>(sk_smem);                        // (red is dead: every wave passed the barrier above)
    if (tid == 0) {
      const unsigned t = __hip_atomic_fetch_add(p.sk_cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *flag = ((t % (unsigned)ksplit) == (unsigned)ksplit - 1u) ? 1 : 0;
    }
    __syncthreads();
#ifdef TELL_PROBES
    if (*flag == 0) { stamp_out(); return; }
#else
    if (*flag == 0) return;
#endif
    const float* slabs = p.sk_slab + tile * ksplit * SLAB;
#pragma unroll
    for (int e = 0; e < EPI; ++e) {
      const int o = tid + e * NT;
      const int r = o >> 4, c = o & 15, m = m0 + r, n = n0 + c;
      float v = FOLD ? pre_c[0] : 0.f, g = (FOLD && ACT == 2) ? pre_c[NB - 1] : 0.f;
      for (int sl = 0; sl < ksplit; ++sl) {                              // slice order: the sum does not depend on who came last
        v += __hip_atomic_load(slabs + (long)sl * SLAB + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (ACT == 2) g += __hip_atomic_load(slabs + (long)sl * SLAB + MT * 16 + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (c >= cn || m >= M || n >= N) continue;
      skinny_epilogue(p, prob, m, n, v, g, ACT, PRE ? &pre[e] : nullptr);
    }
#ifdef TELL_PROBES
    stamp_out();
#endif
  }
}
#ifdef TELL_PROBES
// probe build: every skinny launch takes the next slot of the caller's stamp buffer (options sk_stamp_ptr = device address,
// sk_stamp_slots; a slot = SK_STAMP_SLOT u64s); a captured launch keeps its slot, so a graph replay refreshes it.
constexpr long SK_STAMP_SLOT = (1 + 2048) * 4;
static unsigned long long* skinny_next_stamp() {
  static long base_seen = 0, next = 0;
  const long base = tell_opt(PROBE_SK_STAMP_PTR), slots = tell_opt(PROBE_SK_STAMP_SLOTS);
  if (base != base_seen) { base_seen = base; next = 0; }
  if (!base || next >= slots) return nullptr;
  return reinterpret_cast<unsigned long long*>(base) + (next++) * SK_STAMP_SLOT;
}
#define SK_STAMP(a) (a).stamp = skinny_next_stamp()
#else
#define SK_STAMP(a)
#endif
// LDS of a launch: the reduction buffers; STAGED: or the waves' stage buffers, which they lie over
template <int RT, int ACT, bool FOLD, int NW, bool STAGED>
constexpr size_t skinny_smem() {
  constexpr int MT = RT * 16, NB = ACT == 2 ? 2 : 1;
  constexpr size_t red = (size_t)NW * MT * 16 * NB * 4 + (FOLD ? (size_t)(NW * MT * 2 + MT * 8) * 4 : 0);
  constexpr size_t stages = STAGED ? (size_t)NW * (MT + 16 * NB) * 128 : 0;
  return red > stages ? red : stages;
}
// the stages in flight of the STAGED form per row tiles (registers: stages x (MT + 16 NB) / 8 x 4)
// (0: the launch keeps the direct form - the wave's stage count must be a multiple of the stages in flight)
static inline int skinny_stage_count(const SkinnyArgs& a, int nw) {
  return tell_opt(OPT_SK_STAGED) && a.K / a.ksplit % (64 * nw) == 0 ? a.K / a.ksplit / (64 * nw) : 0;
}

template <int RT, int ACT, int U, bool FOLD = false, int NW = 4, bool STAGED = false>
static int skinny_mfma_launch(SkinnyArgs a, int n_prob, hipStream_t stream) {
  if constexpr (!STAGED && (NW == 4)) {
    const int ns = skinny_stage_count(a, NW);
    if constexpr (RT >= 8) {
      if (ns && ns % 2 == 0) return skinny_mfma_launch<RT, ACT, 2, FOLD, NW, true>(a, n_prob, stream);
    } else {
      // (128 rows = four row groups = 1024 workgroups, two resident per CU with these registers: a two-stage variant at
      //  118 registers - four per CU, the whole grid resident at once - measured the same, beam-4 step 489.8 vs 488.4 us.  So did
  //  64 / 32 COLUMNS per workgroup where a launch has 1024 / 512 tiles - the activation rows staged once for four column
  //  tiles, 256 workgroups, half the bytes per CU: with two stages in flight 479.9 us against 475.7, with all four 488.9
  //  against 476.6; 128 greedy rows 642-645 against 633.  Not kept: what makes a 128-row launch take 9-12 us where 32 rows
  //  take 5-7 is neither residency nor the bytes through a CU nor the number of dependent round trips.  Nor where the row
  //  groups of a column tile run: dispatch index L -> tile (L / 8G) * 8 + L % 8, row group (L / 8) % G puts the G groups of
  //  a tile on ONE XCD within 8 G consecutive dispatches - HBM traffic of the beam-4 step unchanged to five digits (956.24
  //  MB), step 484 -> 498 us.)
      // (K = 4096 - sixteen stages per wave, two dependent rounds of eight - as EIGHT waves x K / 8 with all eight stages
      //  of a wave in flight: 256 registers and spills, greedy step 363.5 -> 386.3 us, beam 4 474.9 -> 496.9; not kept)
      if (ns && ns % 8 == 0) return skinny_mfma_launch<RT, ACT, 8, FOLD, NW, true>(a, n_prob, stream);
      if (ns && ns % 4 == 0) return skinny_mfma_launch<RT, ACT, 4, FOLD, NW, true>(a, n_prob, stream);
    }
  }
  constexpr int MT = RT * 16;
  constexpr size_t smem = skinny_smem<RT, ACT, FOLD, NW, STAGED>();
  static_assert(smem <= 160 * 1024, "skinny_linear: LDS");
  const int groups = (a.M + MT - 1) / MT;
  const long tiles = (long)((a.N + 15) / 16) * n_prob * groups;
  a.cn = tiles >= 192 ? 16 : (tiles >= 96 ? 8 : 4);          // at least ~256 workgroups where the layer has the columns
  auto kern = skinny_mfma_kernel<RT, ACT, U, FOLD, NW, false, STAGED>;
  static bool attr_done = false;
  if (!attr_done && smem > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  SK_STAMP(a);
  hipLaunchKernelGGL(kern, dim3((a.N + a.cn - 1) / a.cn, n_prob, groups), dim3(64 * NW), smem, stream, a);
  return tell_check_launch("skinny_linear (mfma)");
}
// the split form: cn = 16, grid.z = row groups x ksplit
template <int RT, int ACT, int U, bool FOLD, bool STAGED = false>
static int skinny_split_launch(SkinnyArgs a, hipStream_t stream) {
  if constexpr (!STAGED) {
    const int ns = skinny_stage_count(a, 4);
    if (ns && ns % 8 == 0) return skinny_split_launch<RT, ACT, 8, FOLD, true>(a, stream);
    if (ns && ns % 4 == 0) return skinny_split_launch<RT, ACT, 4, FOLD, true>(a, stream);
  }
  constexpr int MT = RT * 16;
  constexpr size_t smem = skinny_smem<RT, ACT, FOLD, 4, STAGED>();
  const int groups = (a.M + MT - 1) / MT;
  auto kern = skinny_mfma_kernel<RT, ACT, U, FOLD, 4, true, STAGED>;
  static bool attr_done = false;
  if (!attr_done && smem > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  SK_STAMP(a);
  hipLaunchKernelGGL(kern, dim3((a.N + a.cn - 1) / a.cn, 1, groups * a.ksplit), dim3(256), smem, stream, a);
  return tell_check_launch("skinny_linear (mfma, split reduction)");
}
// (Warming the NEXT launch's weights from inside a launch - a dword per line of the successor's row blocks, read by the
// workgroups whose dispatch index puts them on the XCD that will want them (workgroup i runs on XCD (i + 6) % 8, launch after
// launch: tools/probes/skinny_stamps.py) - was built and measured, tools/probes/skinny_cold.py: a launch whose weights are
// cold costs 0.9-1.9 us more than with them in its L2 (linear2 3.7 -> 4.7 us, fc1 3.7 -> 5.1, fc2 7.6 -> 9.2 at 32 rows),
// but every launch warming its successor made the chain SLOWER: 4.65 -> 5.3, 5.1 -> 5.8, 9.1 -> 9.5 us per launch, greedy
// step 378 -> 395 us, beam 4 531 -> 545; 64-byte instead of 128-byte strides the same.  Removed.)
template <int RT, int U>
static int skinny_split_dispatch(const SkinnyArgs& a, int act, hipStream_t stream, bool fold) {
  if (fold) return skinny_split_launch<RT, 0, U, true>(a, stream);                 // (context_fc behind its LayerNorms)
  if (act == 0) return skinny_split_launch<RT, 0, U, false>(a, stream);
  return skinny_split_launch<RT, 1, U, false>(a, stream);
}

template <int RT, int U>
static int skinny_mfma_dispatch(const SkinnyArgs& a, int n_prob, int act, hipStream_t stream, bool fold = false) {
  if (fold) {                                                         // (query projections, context_fc: act 0; linear1: GLU)
    if (act == 0) return skinny_mfma_launch<RT, 0, U, true>(a, n_prob, stream);
    if (act == 2) return skinny_mfma_launch<RT, 2, U, true>(a, n_prob, stream);
    tell_set_error("skinny_linear: the folded LayerNorm prologue comes with act 0 or 2");
    return TELL_ERR_ARG;
  }
  if (act == 0) return skinny_mfma_launch<RT, 0, U>(a, n_prob, stream);
  if (act == 1) return skinny_mfma_launch<RT, 1, U>(a, n_prob, stream);
  return skinny_mfma_launch<RT, 2, U>(a, n_prob, stream);
}

// LayerNorm of fp32 rows, one per `span` columns, to bf16 (the prologue as its own launch where the rows do not fit the
// in-kernel form: M > 32).  grid (M, nseg), one workgroup per row segment.
__global__ __launch_bounds__(256) void ln_rows_kernel(const float* __restrict__ x, long ld_x, SkinnyArgs p, int span,
                                                      uint16_t* __restrict__ y, long ld_y) {
  __shared__ float redw[8];
  const int m = blockIdx.x, s = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = x + (long)m * ld_x + (long)s * span;
  const int nv = span / 1024;                       // float4 per thread (span: 1024 .. 4096)
  sk_f4 v[4], gg[4], bb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (j < nv) {
      v[j] = *reinterpret_cast<const sk_f4*>(row + tid * 4 + j * 1024);
      gg[j] = *reinterpret_cast<const sk_f4*>(p.gamma[s] + tid * 4 + j * 1024);
      bb[j] = *reinterpret_cast<const sk_f4*>(p.beta[s] + tid * 4 + j * 1024);
    }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (j < nv) sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  sum = sk_wave_sum(sum);
  if (lane == 0) redw[wave] = sum;
  __syncthreads();
  const float mean = ((redw[0] + redw[1]) + (redw[2] + redw[3])) / (float)span;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (j < nv) {
      const float a0 = v[j].x - mean, a1 = v[j].y - mean, a2 = v[j].z - mean, a3 = v[j].w - mean;
      sq += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
  sq = sk_wave_sum(sq);
  if (lane == 0) redw[4 + wave] = sq;
  __syncthreads();
  const float rstd = rsqrtf(((redw[4] + redw[5]) + (redw[6] + redw[7])) / (float)span + p.eps);
  if (tid == 0 && p.stats_out && gridDim.y == 1) { p.stats_out[m * 2] = mean; p.stats_out[m * 2 + 1] = rstd; }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (j < nv) {
      uint2 o;
      o.x = (uint32_t)f2bf((v[j].x - mean) * rstd * gg[j].x + bb[j].x) | ((uint32_t)f2bf((v[j].y - mean) * rstd * gg[j].y + bb[j].y) << 16);
      o.y = (uint32_t)f2bf((v[j].z - mean) * rstd * gg[j].z + bb[j].z) | ((uint32_t)f2bf((v[j].w - mean) * rstd * gg[j].w + bb[j].w) << 16);
      *reinterpret_cast<uint2*>(y + (long)m * ld_y + (long)s * span + tid * 4 + j * 1024) = o;
    }
}

// LayerNorm of fp32 pre-norm rows [M, C] to bf16 (the decoder output of a generation step, in front of the softmax head).
extern "C" int tell_layernorm_rows(const float* x, long ld_x, const float* gamma, const float* beta, float eps, void* y,
                                   long ld_y, float* stats_out, int M, int C, hipStream_t stream) {
  TELL_REQUIRE(M > 0 && C % 1024 == 0 && C <= 4096 && ld_x % 4 == 0 && ld_y % 4 == 0, "layernorm_rows: C = 1024 .. 4096");
  SkinnyArgs a = {};
  a.gamma[0] = gamma; a.beta[0] = beta; a.eps = eps; a.stats_out = stats_out;
  hipLaunchKernelGGL(ln_rows_kernel, dim3(M, 1), dim3(256), 0, stream, x, ld_x, a, C, static_cast<uint16_t*>(y), ld_y);
  return tell_check_launch("layernorm_rows");
}

// out[p] = epilogue(prologue(in[p]) . w[p]^T) for n_prob <= 4 problems of one shape (host arrays of n_prob pointers).
//   pro 0: in bf16 [M,K];  1: in fp32 [M,K], LayerNorm(gamma[0], beta[0]) first (stats_out: optional [M][2]);
//       2: in fp32 [M,K], one LayerNorm per `seg` columns first (gamma[s], beta[s]; K / seg <= 4).
//      pro 1 / 2 are one extra launch into `ws` (bf16 [M,K], required then) ahead of the GEMM launch; all problems
//      must read the same input; rows / segments of 1024 .. 4096 columns.
//   act 0 none, 1 relu, 2 GLU (w [2N,K], bias [2N]: out = (a + b_a) * sigmoid(g + b_g), gate rows at n + N)
//   out = (act(acc + bias)) * scale + residual;  residual: res (bf16 [M,N]), LayerNorm(res_raw) from res_stats, res_f32
//   out2 (optional): columns n >= out2_from of the result a second time, as bf16, at out2[m][n - out2_from]
extern "C" int tell_skinny_linear(int n_prob, const void* const* in, long ld_in, int pro, const void* const* gamma,
                                  const void* const* beta, int seg, float eps, float* stats_out, void* ws,
                                  const void* const* w, long ldw, const void* const* bias, int act, float scale,
                                  const void* res, long ld_res, const float* res_raw, long ld_res_raw,
                                  const float* res_stats, const float* res_gamma, const float* res_beta,
                                  const float* res_f32, long ld_res_f32, void* out2, long ld_out2, int out2_from,
                                  void* const* out, long ld_out, int out_f32, int M, int N, int K, void* split_ws,
                                  long split_ws_bytes, hipStream_t stream) {
  TELL_REQUIRE(n_prob >= 1 && n_prob <= SK_MAXP && M >= 1 && M <= 1024 && N >= 1 && K >= 256, "skinny_linear: bad shape");
  TELL_REQUIRE(K % 256 == 0 && ldw % 8 == 0 && ld_in % 8 == 0, "skinny_linear: K % 256, 16-byte rows");
  TELL_REQUIRE(pro >= 0 && pro <= 4 && act >= 0 && act <= 2, "skinny_linear: bad mode");
  const bool fold = pro >= 3;
  if (fold) {
    TELL_REQUIRE(gamma && beta, "skinny_linear: the folded prologue needs the s / c vectors");
    if (pro == 3) seg = K;
    TELL_REQUIRE(seg > 0 && K % seg == 0 && (K / seg == 1 || K / seg == 2 || K / seg == 4), "skinny_linear: folded LayerNorm over 1, 2 or 4 segments");
    TELL_REQUIRE(pro == 3 || n_prob == 1, "skinny_linear: segmented fold: one problem");
  }
  TELL_REQUIRE(pro != 2 || (seg > 0 && seg % 256 == 0 && K % seg == 0 && K / seg <= SK_MAXP), "skinny_linear: bad segments");
  TELL_REQUIRE(!res_raw || (res_stats && res_gamma && res_beta), "skinny_linear: res_raw needs statistics and affine");
  SkinnyArgs a;
  for (int i = 0; i < SK_MAXP; ++i) {
    const int j = i < n_prob ? i : 0;
    a.in[i] = in[j]; a.w[i] = static_cast<const uint16_t*>(w[j]); a.bias[i] = bias ? static_cast<const float*>(bias[j]) : nullptr;
    a.out[i] = out[j];
    a.gamma[i] = a.beta[i] = nullptr;
  }
  for (int i = 0; i < SK_MAXP; ++i) {
    const int j = i < n_prob ? i : 0;
    a.fold_s[i] = fold ? static_cast<const float*>(gamma[j]) : nullptr;
    a.fold_c[i] = fold ? static_cast<const float*>(beta[j]) : nullptr;
    if (fold) TELL_REQUIRE(a.fold_s[i] && a.fold_c[i], "skinny_linear: folded prologue without s / c");
  }
  a.out2_prob = N;
  const int nseg = pro == 2 ? K / seg : (pro == 1 ? 1 : 0);
  for (int s = 0; s < nseg; ++s) {
    TELL_REQUIRE(gamma && beta && gamma[s] && beta[s], "skinny_linear: LayerNorm prologue without gamma / beta");
    a.gamma[s] = static_cast<const float*>(gamma[s]); a.beta[s] = static_cast<const float*>(beta[s]);
  }
  a.ld_in = ld_in; a.ldw = ldw; a.ld_out = ld_out; a.ld_res = ld_res; a.ld_res_raw = ld_res_raw;
  a.stats_out = stats_out; a.res = static_cast<const uint16_t*>(res); a.res_raw = res_raw; a.res_stats = res_stats;
  a.res_gamma = res_gamma; a.res_beta = res_beta; a.res_f32 = res_f32; a.ld_res_f32 = ld_res_f32;
  a.out2 = static_cast<uint16_t*>(out2); a.ld_out2 = ld_out2; a.out2_from = out2_from; a.eps = eps; a.scale = scale; a.M = M; a.N = N; a.K = K;
  a.seg = seg > 0 ? seg : K; a.out_f32 = out_f32; a.cn = 16;
  a.ksplit = 1; a.sk_slab = nullptr; a.sk_cnt = nullptr;

  if (pro == 1 || pro == 2) {
    TELL_REQUIRE((pro == 1 ? K : seg) % 1024 == 0 && (pro == 1 ? K : seg) <= 4096, "skinny_linear: LayerNorm over 1024 .. 4096 columns");
    TELL_REQUIRE(ws, "skinny_linear: the LayerNorm prologue needs ws [M,K] bf16");
    for (int i = 1; i < n_prob; ++i) TELL_REQUIRE(in[i] == in[0], "skinny_linear: separate prologue: one shared input");
    const int span = pro == 1 ? K : seg;
    hipLaunchKernelGGL(ln_rows_kernel, dim3(M, K / span), dim3(256), 0, stream, static_cast<const float*>(in[0]), ld_in,
                       a, span, static_cast<uint16_t*>(ws), (long)K);
    int rc = tell_check_launch("skinny_linear (LayerNorm rows)");
    if (rc) return rc;
    for (int i = 0; i < SK_MAXP; ++i) a.in[i] = ws;
    a.ld_in = K; a.stats_out = nullptr;
  }
  // rows per workgroup: 32 (more workgroups) unless 128-row groups alone already fill the chip - then the weight
  // fragments of a column tile are loaded once for 128 rows instead of four times
  // (64 rows per workgroup - RT = 4, U = 4 - where 128 is taken: beam 4 711 -> 705 us, 128 greedy rows 866 -> 853; everywhere
  //  above 64 rows: 784 / 926 us.  Not instantiated.)
  // ---- the reduction shared by several workgroups per tile (skinny_mfma_kernel SPLIT): one problem, few column tiles, a
  // long reduction - context_fc / fc2 of the step (N = 1024, K = 4096).  Needs the caller's workspace: the first 64 KB
  // are arrival counters (zero once, at allocation; never reset), the rest holds the partial tiles.
  // MEASURED (MI355X, hot operands, fc2 shape): 32 rows 11.3 -> 9.5 us with 4 slices x 16 columns; 128 rows as ONE 128-row
  // group per tile 13.1 -> 22.4 us (the activation bytes per workgroup do not shrink there - 128 rows x 1024 = the 32 rows
  // x 4096 of the unsplit form - and the partial tiles are 8 KB each); 64 rows as two 32-row groups: fc2 12.1 -> 12.2, the
  // folded context_fc 13.3 -> 18.6 us (512 workgroups, two per CU, 1 MB of partial tiles).  Taken up to 32 rows: fc2 11.4
  // -> 9.4 us, context_fc behind its folded LayerNorms 12.7 -> 10.6 us; greedy step 433 -> 418 us same box.
  // option sk_split: 1 = K / 1024 slices x 16 columns; 2 = 2 slices x 8 columns (A/B aid).
  const long sk_opt = tell_opt(OPT_SK_SPLIT);
  if (split_ws && sk_opt && n_prob == 1 && M <= 32 && act != 2 && !stats_out && (pro == 0 || pro == 4) &&
      K >= 2048 && K % 1024 == 0 && (N + 15) / 16 <= 96) {
    int ks = K >= 4096 ? 4 : 2;
    a.cn = 16;
    if (sk_opt == 2) { ks = 2; a.cn = 8; }
    const int mt = 32, groups = (M + mt - 1) / mt;
    const long tiles = (long)((N + a.cn - 1) / a.cn) * groups;
    const long need = 65536 + tiles * ks * mt * 16 * 4;
    if ((K / ks) % 1024 == 0 && (pro != 4 || (K / ks) % a.seg == 0) && tiles <= 8192 && need <= split_ws_bytes &&
        (reinterpret_cast<uintptr_t>(split_ws) & 15) == 0) {
      a.ksplit = ks;
      a.sk_cnt = static_cast<unsigned*>(split_ws) + (ks == 2 ? 8192 : 0);
      a.sk_slab = reinterpret_cast<float*>(static_cast<char*>(split_ws) + 65536);
      return skinny_split_dispatch<2, 8>(a, act, stream, fold);
    }
    a.cn = 16;
  }
  // Rows per workgroup: 32.  (Rounds 3-5, direct fragment loads: 128-row workgroups wherever they alone filled the chip - the
  // weight fragments of a column tile loaded once for 128 rows instead of four times.  With the operands staged through LDS
  // the activation rows are cheap and the 128-row workgroup - 2 stages of 18 loads in flight, a 128 x 16 reduction through
  // LDS, 8 outputs per thread - loses everywhere: beam-4 step 512.5 us with the old rule, 486.9 with 32 rows everywhere,
  // 686.9 with 128 everywhere; 128 greedy rows 666 -> 642 us; 64-row workgroups (4 stages of 10 loads) above 32 rows: beam 4
  // 491.8 -> 570.8 us, 128 greedy rows 643 -> 722 - not kept.  Option sk_rows = 128 keeps the tall form reachable.)
  const bool tall = tell_opt(OPT_SK_ROWS) == 128 && M > 64;
  if (tall) {
    // option sk_tall_waves = 8: the 128-row workgroup as 8 waves x K / 8 (both batches of a wave's loads in flight at once,
    // twice the waves per CU to hide them) instead of 4 x K / 4.  MEASURED (round 6): alone q-proj x4 15.6 -> 15.0 us, out-proj
    // x4 12.3 -> 12.0, fc1 12.3 -> 11.7 - and the beam-4 step, where the weights are cold, 644 -> 672 us.  Default 4.
    if (tell_opt(OPT_SK_TALL_WAVES) == 8 && K % 512 == 0 && act != 2) {
      if (fold) return skinny_mfma_launch<8, 0, 2, true, 8>(a, n_prob, stream);
      return act == 1 ? skinny_mfma_launch<8, 1, 2, false, 8>(a, n_prob, stream) : skinny_mfma_launch<8, 0, 2, false, 8>(a, n_prob, stream);
    }
    return skinny_mfma_dispatch<8, 2>(a, n_prob, act, stream, fold);
  }
  return K % 1024 == 0 ? skinny_mfma_dispatch<2, 8>(a, n_prob, act, stream, fold) : skinny_mfma_dispatch<2, 2>(a, n_prob, act, stream, fold);
}

// ------------------------------------------------------------------ DynamicConv step (T = 1, K-plane ring of past inputs)
// x [M, C] bf16 (the GLU output of this step), wt [H*K, C] bf16 (weight_linear), hist [K planes][M][C] bf16.
//   logits[k] = x[m,:] . wt[h*K + k,:]; taps = softmax_k(logits)            (dynamic.py:300-304, eval: no DropConnect)
//   y[m, h*64 + d] = sum_k taps[k] * window[k][m, h*64 + d],  window = the K-1 previous inputs, then x   (:306-336, causal)
//   the input buffer takes x                                                              (:95-99)
// Round 6 - nothing is moved any more.  Rounds 3-5 kept the buffer as K-1 rows in time order: every step SHIFTED it (K-2
// planes re-written per layer) and beam search physically re-ordered its rows by parent (tell_reorder_rows: every plane
// read and written again) - 27 of the 100 MB a beam-4 step wrote.  Now the buffer is a RING of K planes indexed by time
// (the row of step s lives in plane s mod K: a step writes ONE plane - the one plane it does not read, hence K planes
// for K-1 past rows) and a hypothesis finds its past through an ANCESTOR table: back[j-1][m] = the slot, j steps ago, of
// the hypothesis that now sits in slot m (null: m itself - greedy decoding).  The reference's contract is only
// reorder_incremental_state (dynamic.py:338-342): the beam bookkeeping launch composes the table with this step's parents
// instead (tell_beam_update).  Planes that were never written hold zeros (the caller clears the ring per caption): the
// taps that reach before the start of the caption multiply zeros, which is what the reference's narrowing does.
// t = index of this step = t_host + *step_dev (step_dev: the registered decode position counter while a hipGraph of the
// step is recorded - one captured launch then serves every position, like the embedder's sinusoid row).
// One workgroup per (head, R rows).  The tap logits come off the matrix cores (rounds 3-5: K x C dot products per row on
// the VALU against K x C weights staged through LDS, 64 KB per workgroup at K = 31, with a 64-lane sum per tap): the 4
// waves split C, load their mfma_f32_16x16x32_bf16 fragments straight from memory (A: the R rows, B: the head's K <= 32
// tap rows as two column tiles) and fold the partial tiles through LDS; every past row's load is issued before the first
// MFMA; softmax by half-waves (lane = tap); then each thread sums its R / 4 channels over the K taps.
// KB: taps the unrolled loops cover (4 / 8 / 16 / 32 >= K)
template <int R, int NK, bool HAS_BACK, int KB>
__global__ __launch_bounds__(256) void dynconv_step_kernel(const uint16_t* __restrict__ x, uint16_t* hist,
                                                           const uint16_t* __restrict__ wt, uint16_t* __restrict__ y,
                                                           int M, int H, int K, int t_host,
                                                           const uint32_t* __restrict__ step_dev,
                                                           const int* __restrict__ back) {
  constexpr int C = NK * 128, VEC = R / 4, TPR = 64 / VEC;             // channels per thread, threads per row
  __shared__ float red[4][R][33];
  __shared__ float prob[R][32];
  const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.y * R;
  // ---- what does not depend on the step index is requested FIRST (the history loads further down wait for `t`, a value this
  // launch has to load, and, issued in front, held everything else back behind that round trip): the ancestor slots of this
  // thread's row - oldest, so that the history loads wait for them alone -, then the operands of the tap logits
  const int r = tid / TPR, c0 = (tid % TPR) * VEC, m = m0 + r;           // tap-sum role: row r, channels c0 .. of the head
  const bool live = m < M;
  const int mm = live ? m : M - 1;
  int slots[HAS_BACK ? KB - 1 : 1];
  if constexpr (HAS_BACK) {
    const int* bk = back + mm;
#pragma clang loop unroll(full)
    for (int j = 1; j < KB; ++j) {
      slots[j - 1] = *bk;
      bk = j + 1 < K ? bk + M : bk;
    }
  }
  // ---- tap logits of the R rows: [16 (R live), C] . [32 (K live), C]^T, C split over the waves
  const int ma = m0 + (lr < R ? lr : R - 1);
  const uint16_t* ap = x + (long)(ma < M ? ma : M - 1) * C + wave * (C / 4) + lg * 8;
  const uint16_t* bp[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int tap = ct * 16 + lr;
    bp[ct] = wt + (long)(h * K + (tap < K ? tap : K - 1)) * C + wave * (C / 4) + lg * 8;
  }
  sk_u4 fa[NK], fb[2][NK];
  constexpr bool two = KB > 16;                                        // (K <= 16: the second column tile has no live tap)
#pragma unroll
  for (int i = 0; i < NK; ++i) {
    fb[0][i] = *reinterpret_cast<const sk_u4*>(bp[0] + i * 32);
    fb[1][i] = sk_u4{0u, 0u, 0u, 0u};
    if (two) fb[1][i] = *reinterpret_cast<const sk_u4*>(bp[1] + i * 32);
    fa[i] = *reinterpret_cast<const sk_u4*>(ap + i * 32);
  }
  const int t = t_host + (step_dev ? (int)*step_dev : 0);
  const int ch = h * 64 + c0;
  // (plain 32-bit registers: lo = channels 0 / 0-1, hi = channels 2-3 when a thread owns four)
  uint32_t hlo[KB - 1], hhi[KB - 1];
  const long plane = (long)M * C;
  auto fetch = [](const uint16_t* src, uint32_t& lo, uint32_t& hi) __attribute__((always_inline)) {
    if constexpr (VEC == 1) lo = (uint32_t)*src;
    else if constexpr (VEC == 2) lo = *reinterpret_cast<const uint32_t*>(src);
    else { const uint2 w = *reinterpret_cast<const uint2*>(src); lo = w.x; hi = w.y; }
  };
  // (branch-free: a dead tap j >= K re-reads the row of tap K - 1 and gets weight 0 below, a dead row m >= M reads row M - 1
  //  and stores nothing - 31 guarded loads made the register allocator spill every fragment around 31 branches)
  int pl = (t - 1) % K;
  pl = pl < 0 ? pl + K : pl;
#pragma clang loop unroll(full)
  for (int j = 1; j < KB; ++j) {
    int slot = mm;
    if constexpr (HAS_BACK) slot = slots[j - 1];
    hhi[j - 1] = 0u;
    fetch(hist + pl * plane + (long)slot * C + ch, hlo[j - 1], hhi[j - 1]);
    const bool more = j + 1 < K;
    pl = more ? (pl == 0 ? K - 1 : pl - 1) : pl;
  }
  uint32_t clo = 0u, chi = 0u;
  fetch(x + (long)mm * C + ch, clo, chi);
  {
    typedef float c4 __attribute__((ext_vector_type(4)));
    c4 acc[2] = {c4{0.f, 0.f, 0.f, 0.f}, c4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sk_bf16x8, fa[i]), __builtin_bit_cast(sk_bf16x8, fb[0][i]),
                                                       acc[0], 0, 0, 0);
      if (two)
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sk_bf16x8, fa[i]), __builtin_bit_cast(sk_bf16x8, fb[1][i]),
                                                         acc[1], 0, 0, 0);
    }
    // C layout: column (tap) = lane & 15, row = (lane >> 4) * 4 + register
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (lg * 4 + q < R) red[wave][lg * 4 + q][ct * 16 + lr] = acc[ct][q];
  }
  __syncthreads();
  // ---- softmax over the taps: a half-wave per row, lane = tap
  for (int o = tid; o < R * 32; o += 256) {
    const int rr = o >> 5, k = o & 31;
    float l = k < K ? (red[0][rr][k] + red[1][rr][k]) + (red[2][rr][k] + red[3][rr][k]) : -INFINITY;
    float mx = l;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 32));
    const float e = k < K ? __expf(l - mx) : 0.f;
    float den = e;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) den += __shfl_xor(den, off, 32);
    prob[rr][k] = e / den;
  }
  __syncthreads();
  if (!live) return;
  float out[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) out[v] = 0.f;
  auto elem = [](uint32_t lo, uint32_t hi, int v) __attribute__((always_inline)) -> float {
    const uint32_t word = v < 2 ? lo : hi;
    return __uint_as_float((VEC == 1 || (v & 1) == 0) ? (word << 16) : (word & 0xffff0000u));
  };
  // window[K - 1 - j] = the input j steps ago
#pragma clang loop unroll(full)
  for (int j = 1; j < KB; ++j) {
    const float pk = j < K ? prob[r][j < K ? K - 1 - j : 0] : 0.f;
#pragma unroll
    for (int v = 0; v < VEC; ++v) out[v] = fmaf(pk, elem(hlo[j - 1], hhi[j - 1], v), out[v]);
  }
  {
    const float pk = prob[r][K - 1];
#pragma unroll
    for (int v = 0; v < VEC; ++v) out[v] = fmaf(pk, elem(clo, chi, v), out[v]);
  }
  int pw = t % K;
  pw = pw < 0 ? pw + K : pw;
  uint16_t* hd = hist + pw * plane + (long)m * C + ch;
  uint16_t* yd = y + (long)m * C + ch;
  if constexpr (VEC == 1) {
    *hd = (uint16_t)clo;
    *yd = f2bf(out[0]);
  } else if constexpr (VEC == 2) {
    *reinterpret_cast<uint32_t*>(hd) = clo;
    *reinterpret_cast<uint32_t*>(yd) = (uint32_t)f2bf(out[0]) | ((uint32_t)f2bf(out[1]) << 16);
  } else {
    *reinterpret_cast<uint2*>(hd) = make_uint2(clo, chi);
    *reinterpret_cast<uint2*>(yd) = make_uint2((uint32_t)f2bf(out[0]) | ((uint32_t)f2bf(out[1]) << 16),
                                               (uint32_t)f2bf(out[2]) | ((uint32_t)f2bf(out[3]) << 16));
  }
}
template <int R>
static int dynconv_step_launch(const uint16_t* x, uint16_t* hist, const uint16_t* wt, uint16_t* y, int M, int C, int H, int K,
                               int t, const int* back, hipStream_t stream) {
  const dim3 grid(H, (M + R - 1) / R), block(256);
#define DCS2(NK_, KB_) do { if (back) hipLaunchKernelGGL((dynconv_step_kernel<R, NK_, true, KB_>), grid, block, 0, stream, x, hist, wt, y, M, H, K, t, g_tell_pos_step, back); \
                     else hipLaunchKernelGGL((dynconv_step_kernel<R, NK_, false, KB_>), grid, block, 0, stream, x, hist, wt, y, M, H, K, t, g_tell_pos_step, back); } while (0)
#define DCS(NK_) do { if (K <= 4) DCS2(NK_, 4); else if (K <= 8) DCS2(NK_, 8); else if (K <= 16) DCS2(NK_, 16); else DCS2(NK_, 32); } while (0)
  switch (C / 128) {
    case 4: DCS(4); break;
    case 8: DCS(8); break;
    default: DCS(16); break;
  }
#undef DCS
#undef DCS2
  return tell_check_launch("dynconv_step");
}
extern "C" int tell_dynconv_step(const void* x, void* hist, const void* wt, void* y, int M, int C, int H, int K, int t,
                                 const int* back, hipStream_t stream) {
  TELL_REQUIRE(M > 0 && H > 0 && C == H * 64 && K >= 2 && K <= 32 && (C == 512 || C == 1024 || C == 2048), "dynconv_step: head width 64, 2 <= K <= 32, C = 512 / 1024 / 2048");
  TELL_REQUIRE(t >= 0 || g_tell_pos_step, "dynconv_step: step index");
  // rows per workgroup: enough workgroups to cover the chip at 32 rows (16 heads x 8), fewer re-reads of the head's tap
  // weights where the rows are many
  const uint16_t* xp = (const uint16_t*)x; uint16_t* hp = (uint16_t*)hist; const uint16_t* wp = (const uint16_t*)wt; uint16_t* yp = (uint16_t*)y;
  if ((long)H * ((M + 3) / 4) <= 256) return dynconv_step_launch<4>(xp, hp, wp, yp, M, C, H, K, t, back, stream);
  if ((long)H * ((M + 7) / 8) <= 512) return dynconv_step_launch<8>(xp, hp, wp, yp, M, C, H, K, t, back, stream);
  return dynconv_step_launch<16>(xp, hp, wp, yp, M, C, H, K, t, back, stream);
}

// ------------------------------------------------------------------ one-query attention over up to 4 cached contexts
// multi_head.py:376-475 at Tq = 1 against static keys / values (:330-352): scores = K q (q already scaled), key-padding
// mask, fp32 softmax, P V.  The 32-row MFMA tile of the training kernel carries one real row here; this is a
// bandwidth problem (the cached K and V of every (row, head) are read once per step), so: one workgroup per
// (row, head, context), 8 lanes share a key (16 bytes of its 128 each - a wave load covers 8 keys = 1 KB contiguous),
// scores parked in LDS, two passes.  The learned bias_k / bias_v row (:355-364) and the zero row (:416-421) are two more
// keys after the S cached ones, never masked.  The `beams` hypotheses of a sample (rows b*beams + j) share its cache.
#define AD_MAXS 2048
struct AttnDecCtx {
  const uint16_t *q, *k, *v; uint16_t* out; const uint8_t* mask;
  const uint16_t *bias_k, *bias_v;      // [H*64] or null
  long q_sb, k_ss, k_sb, k_sh, v_ss, v_sb, v_sh, o_sb;    // k_sh / v_sh: elements between the heads of a key (64: row-major [.., H*64])
  int S, has_zero;
};
struct AttnDecArgs { AttnDecCtx c[SK_MAXP]; int B, H, beams; };

// NQ: hypotheses of one sample served by a workgroup (beam search: the NQ rows b*beams + j read the SAME cached keys
// and values - loaded once, used NQ times).
// Round 6: ONE pass.  Rounds 3-5 walked the keys twice - scores into LDS (33 KB at four hypotheses), a block-wide softmax,
// then the values - i.e. 2 x 4 dependent trips of loads per workgroup with barriers in between: a chain of ~8 memory round
// trips for 128 KB of cache, 2.5 TB/s at four hypotheses per workgroup (3.8 with one) whatever the arithmetic cost (halving
// the VALU work changed nothing).  Now every group of 8 lanes that shares a key keeps its OWN running (max, sum, output) over
// the keys it sees - the flash-decoding split, across lanes instead of workgroups: a trip loads K AND V of its keys (8 loads
// in flight per lane), no score ever leaves the registers, nothing synchronises until the end, where the 8 key slots of a wave
// merge through shuffles and the 4 waves through LDS.  The learned bias_k / bias_v key and the zero key are two more keys of
// wave 0's first slot.
template <int NQ>
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnDecArgs g) {
  const AttnDecCtx& p = g.c[blockIdx.y];
  __shared__ float part_o[NQ][NQ <= 2 ? 4 : 32][64];           // per wave (NQ <= 2) / per (wave, key slot): output, (max, sum)
  __shared__ float part_ml[NQ][NQ <= 2 ? 4 : 32][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int groups = (g.beams + NQ - 1) / NQ;                 // workgroups per (sample, head)
  const int h = blockIdx.x % g.H, bg = blockIdx.x / g.H, bs = bg / groups, j0 = (bg % groups) * NQ;
  const int nq = g.beams - j0 < NQ ? g.beams - j0 : NQ;       // live hypotheses of this workgroup
  const int b0 = bs * g.beams + j0;                            // first row
  const int ks = lane >> 3, dc = lane & 7, S = p.S;
  uint4 q[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i)
    q[i] = *reinterpret_cast<const uint4*>(p.q + (long)(b0 + (i < nq ? i : 0)) * p.q_sb + h * 64 + dc * 8);
  const uint16_t* kb = p.k + (long)bs * p.k_sb + (long)h * p.k_sh + dc * 8;
  const uint16_t* vb = p.v + (long)bs * p.v_sb + (long)h * p.v_sh + dc * 8;
  const uint8_t* mk = p.mask ? p.mask + (long)bs * S : nullptr;
  const int S8 = (S + 7) & ~7;
  typedef float ad_f2 __attribute__((ext_vector_type(2)));
  float m[NQ], l[NQ];
  ad_f2 o[NQ][4];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    m[i] = -INFINITY; l[i] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[i][e] = ad_f2{0.f, 0.f};
  }
  auto unpack2 = [](const uint4& w, ad_f2 (&f)[4]) __attribute__((always_inline)) {
    f[0] = ad_f2{__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u)};
    f[1] = ad_f2{__uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u)};
    f[2] = ad_f2{__uint_as_float(w.z << 16), __uint_as_float(w.z & 0xffff0000u)};
    f[3] = ad_f2{__uint_as_float(w.w << 16), __uint_as_float(w.w & 0xffff0000u)};
  };
  // Scores are kept in the exp2 domain (d * log2 e: one multiply, then v_exp_f32 directly).  A TRIP of AD_G keys is absorbed
  // at once: the running maximum moves once per trip (one rescale of the state for AD_G keys).  PMC on the first one-pass
  // version (profiles/r06_pmc_attn_decode.txt): the kernel was INSTRUCTION-bound - 100 instructions per key and lane with one
  // hypothesis, 200 with four (issue slots 78 % busy at beam 4), memory latency 770-1050 cycles per request and hidden.
  constexpr float LOG2E = 1.4426950408889634f;
  constexpr int AD_G = 4;
  // act[u]: WAVE-uniform - key group u of the trip has any key at all (the short contexts - 4 faces, 49 regions - fill one
  // or two of a trip's four groups: the others cost a scalar branch instead of ~100 instructions)
  auto absorb = [&](const uint4 (&kr)[AD_G], const uint4 (&vr)[AD_G], const bool (&live)[AD_G], const bool (&act)[AD_G])
      __attribute__((always_inline)) {
    float sc[NQ][AD_G], mn[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) mn[i] = m[i];
#pragma unroll
    for (int u = 0; u < AD_G; ++u) {
      if (act[u]) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          float d = sk_dot8(q[i], kr[u], 0.f);
          d += sk_dpp<0xB1>(d); d += sk_dpp<0x4E>(d); d += sk_dpp<0x141>(d);   // over the 8 lanes that share the key
          sc[i][u] = live[u] ? d * LOG2E : -INFINITY;
          mn[i] = fmaxf(mn[i], sc[i][u]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NQ; ++i) sc[i][u] = -INFINITY;
      }
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      // (mn = -inf: nothing live so far - keep the empty state; m = -inf, mn finite: exp2(-inf) = 0)
      const float a = mn[i] == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(m[i] - mn[i]);
      const ad_f2 a2 = {a, a};
      l[i] *= a;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[i][e] *= a2;
      m[i] = mn[i];
    }
#pragma unroll
    for (int u = 0; u < AD_G; ++u) {
      if (!act[u]) continue;
      ad_f2 vf[4];
      unpack2(vr[u], vf);
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const float pr = sc[i][u] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(sc[i][u] - mn[i]);
        l[i] += pr;
        const ad_f2 p2 = {pr, pr};
#pragma unroll
        for (int e = 0; e < 4; ++e) o[i][e] = __builtin_elementwise_fma(p2, vf[e], o[i][e]);
      }
    }
  };
  // a trip = AD_G key groups of 32 keys (8 per wave): the K and V loads of a trip (2 x AD_G per lane) fly together.
  // Branch-free: a key past S re-reads key S - 1 and is not live.
  const bool has_mask = mk != nullptr;
  const uint8_t* mkp = has_mask ? mk : reinterpret_cast<const uint8_t*>(p.q);   // (no mask: any readable byte, ignored)
  const int kss = (int)p.k_ss, vss = (int)p.v_ss;                               // (S * stride < 2^31: checked by the launcher)
  for (int s0 = wave * 8; s0 < S8; s0 += 32 * AD_G) {
    uint4 kr[AD_G], vr[AD_G];
    bool live[AD_G], act[AD_G];
#pragma unroll
    for (int u = 0; u < AD_G; ++u) {
      const int s = s0 + u * 32 + ks;
      const int sc_ = s < S ? s : S - 1;
      act[u] = s0 + u * 32 < S8;                                                // (s0 is a multiple of 8: wave-uniform)
      kr[u] = make_uint4(0u, 0u, 0u, 0u); vr[u] = make_uint4(0u, 0u, 0u, 0u);
      live[u] = false;
      if (act[u]) {
        kr[u] = *reinterpret_cast<const uint4*>(kb + sc_ * kss);
        vr[u] = *reinterpret_cast<const uint4*>(vb + sc_ * vss);
        const uint8_t mb = mkp[has_mask ? sc_ : 0];
        live[u] = (s < S) & !(has_mask & (mb != 0));
      }
    }
    absorb(kr, vr, live, act);
  }
  {  // the two virtual keys (multi_head.py:355-364, :416-421), never masked: wave 0, key slot 0
    const bool mine = wave == 0 && ks == 0;
    if (wave == 0) {
      uint4 kr[AD_G], vr[AD_G];
      bool live[AD_G], act[AD_G];
#pragma unroll
      for (int u = 0; u < AD_G; ++u) { kr[u] = make_uint4(0u, 0u, 0u, 0u); vr[u] = make_uint4(0u, 0u, 0u, 0u); live[u] = false; act[u] = u < 2; }
      if (p.bias_k) {
        kr[0] = *reinterpret_cast<const uint4*>(p.bias_k + h * 64 + dc * 8);
        if (p.bias_v) vr[0] = *reinterpret_cast<const uint4*>(p.bias_v + h * 64 + dc * 8);
        live[0] = mine;
      }
      live[1] = mine && p.has_zero != 0;                   // the zero key: score 0, value 0
      absorb(kr, vr, live, act);
    }
  }
  // ---- merge: every (wave, key slot) that saw a key parks its state in LDS; NQ x 64 threads fold the slots per output
  // element.  (First version: three rounds of shuffles per wave - 30 ds_bpermute per hypothesis - then the 4 waves through LDS:
  // ~600 instructions per wave, more than a short context's whole trip.)  Waves without keys (a 4-face context has one
  // live wave) only pass the barrier.
  const int n_waves = S8 > 24 ? 4 : (S8 > 16 ? 3 : (S8 > 8 ? 2 : 1));      // waves with at least one key (wave 0 always: virtual keys)
  // One or two hypotheses per workgroup: only 64 / 128 threads fold, so the 8 key slots of a wave merge by shuffles first
  // and one slot per wave goes through LDS (measured with one hypothesis: 17.2 us this way, 21.8 us with 32 slots in LDS;
  // with four: 36.8 against 28.6 us the other way round).
  constexpr bool WAVE_MERGE = NQ <= 2;
  if (wave < n_waves) {
    if constexpr (WAVE_MERGE) {
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) {
          const float m2 = __shfl_xor(m[i], off), l2 = __shfl_xor(l[i], off);
          const float mn = fmaxf(m[i], m2);
          const float a = mn == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(m[i] - mn), b = mn == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(m2 - mn);
          l[i] = l[i] * a + l2 * b;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float ox = __shfl_xor(o[i][e].x, off), oy = __shfl_xor(o[i][e].y, off);
            o[i][e] = ad_f2{o[i][e].x * a + ox * b, o[i][e].y * a + oy * b};
          }
          m[i] = mn;
        }
      }
    }
    if (!WAVE_MERGE || ks == 0) {
      const int slot = WAVE_MERGE ? wave : wave * 8 + ks;
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        float* po = &part_o[i][slot][dc * 8];
#pragma unroll
        for (int e = 0; e < 4; ++e) *reinterpret_cast<ad_f2*>(po + 2 * e) = o[i][e];
        if (dc == 0) { part_ml[i][slot][0] = m[i]; part_ml[i][slot][1] = l[i]; }
      }
    }
  }
  __syncthreads();
  const int n_slots = WAVE_MERGE ? n_waves : n_waves * 8;
  for (int t = tid; t < NQ * 64; t += 256) {
    const int i = t >> 6, d = t & 63;
    if (i >= nq) continue;
    float mn = -INFINITY;
    for (int sl = 0; sl < n_slots; ++sl) mn = fmaxf(mn, part_ml[i][sl][0]);
    float lt = 0.f, v = 0.f;
    for (int sl = 0; sl < n_slots; ++sl) {
      const float a = mn == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(part_ml[i][sl][0] - mn);
      lt = fmaf(part_ml[i][sl][1], a, lt);
      v = fmaf(part_o[i][sl][d], a, v);
    }
    p.out[(long)(b0 + i) * p.o_sb + h * 64 + d] = f2bf(lt > 0.f ? v / lt : 0.f);
  }
}

// n_ctx <= 4 contexts of one decode step in one launch (host arrays of n_ctx entries).  bf16, head width 64, Tq = 1.
// q[c] [B, H*64] (row stride q_sb[c]), k[c] / v[c]: element (b / beams, s, h, d) at k + s*k_ss + (b/beams)*k_sb + h*k_sh + d
// (k_sh / v_sh NULL: 64 - heads side by side in a row of E; the generation loop keeps its cache HEAD-MAJOR, [B, H, S, 64]:
// the S keys a workgroup walks are then one contiguous 64 KB block instead of 128-byte pieces a whole [B, 2E] row apart),
// mask[c] [B / beams, S[c]] uint8 or null, bias_k[c] / bias_v[c] [H*64] or null, out[c] [B, H*64] (row stride o_sb[c]).
extern "C" int tell_attn_decode(int n_ctx, const void* const* q, const long* q_sb, const void* const* k, const long* k_ss,
                                const long* k_sb, const long* k_sh, const void* const* v, const long* v_ss, const long* v_sb,
                                const long* v_sh, const void* const* mask, const void* const* bias_k,
                                const void* const* bias_v, int has_zero, const int* S, void* const* out, const long* o_sb,
                                int B, int H, int beams, hipStream_t stream) {
  TELL_REQUIRE(n_ctx >= 1 && n_ctx <= SK_MAXP && B > 0 && H > 0 && beams >= 1 && B % beams == 0, "attn_decode: 1-4 contexts");
  AttnDecArgs g;
  g.B = B; g.H = H; g.beams = beams;
  for (int c = 0; c < SK_MAXP; ++c) {
    const int j = c < n_ctx ? c : 0;
    TELL_REQUIRE(S[j] >= 0 && S[j] <= AD_MAXS, "attn_decode: S <= 2048");
    TELL_REQUIRE(S[j] + (bias_k && bias_k[j] ? 1 : 0) + has_zero >= 1, "attn_decode: no keys");
    TELL_REQUIRE(q_sb[j] % 8 == 0 && k_ss[j] % 8 == 0 && k_sb[j] % 8 == 0 && v_ss[j] % 8 == 0 && v_sb[j] % 8 == 0,
                 "attn_decode: 16-byte aligned rows");
    TELL_REQUIRE(S[j] == 0 || ((long)S[j] * k_ss[j] < (1L << 31) && (long)S[j] * v_ss[j] < (1L << 31) && k_ss[j] >= 0 && v_ss[j] >= 0),
                 "attn_decode: key offsets of a (sample, head) must fit 31 bits");
    g.c[c].q = (const uint16_t*)q[j]; g.c[c].k = (const uint16_t*)k[j]; g.c[c].v = (const uint16_t*)v[j];
    g.c[c].out = (uint16_t*)out[j]; g.c[c].mask = mask ? (const uint8_t*)mask[j] : nullptr;
    g.c[c].q_sb = q_sb[j]; g.c[c].k_ss = k_ss[j]; g.c[c].k_sb = k_sb[j]; g.c[c].v_ss = v_ss[j]; g.c[c].v_sb = v_sb[j];
    g.c[c].k_sh = k_sh ? k_sh[j] : 64; g.c[c].v_sh = v_sh ? v_sh[j] : 64;
    TELL_REQUIRE(g.c[c].k_sh % 8 == 0 && g.c[c].v_sh % 8 == 0, "attn_decode: 16-byte aligned heads");
    g.c[c].o_sb = o_sb[j]; g.c[c].S = S[j]; g.c[c].has_zero = has_zero ? 1 : 0;
    g.c[c].bias_k = bias_k ? (const uint16_t*)bias_k[j] : nullptr; g.c[c].bias_v = bias_v ? (const uint16_t*)bias_v[j] : nullptr;
  }
  const int samples = B / beams;
  if (beams == 1) hipLaunchKernelGGL((attn_decode_kernel<1>), dim3(B * H, n_ctx), dim3(256), 0, stream, g);
  else if (beams == 2) hipLaunchKernelGGL((attn_decode_kernel<2>), dim3(samples * H, n_ctx), dim3(256), 0, stream, g);
  else hipLaunchKernelGGL((attn_decode_kernel<4>), dim3(samples * ((beams + 3) / 4) * H, n_ctx), dim3(256), 0, stream, g);
  return tell_check_launch("attn_decode");
}

// ------------------------------------------------------------------ one-query attention over a PACKED cache, matrix cores
// Round 6, second half.  PMC showed the VALU kernel above instruction-bound (52 instructions per key and lane with one
// hypothesis, 122 with four, for 32 bytes of cache; memory latency 770-1050 cycles per request, hidden).  The generation loop
// owns the layout of its cache, so it can store what the matrix cores want to read:
//   kc   [Bs, H, Sp / 16, 2, 64, 8]  keys; key rows S and S + 1 are the learned bias_k row and the zero row (multi_head.py:355-364,
//                         :416-421) - virtual keys are ordinary keys here; keys up to Sp (a multiple of 32) are zero.  FRAGMENT
//                         ORDER: per tile of 16 keys and half c of the head width, lane l's 16 bytes = elements c * 32 + (l >> 4) * 8
//                         .. + 7 of key l & 15 - the A operand of the score MFMA, one contiguous KB per wave load
//   vt   [Bs, H, Sp / 32, 4, 64, 8]  values TRANSPOSED (a lane's 16 bytes = 8 keys of one dimension: the A operand of O^T = V^T
//                         P^T), fragment order as well: per block of 32 keys and tile rt of 16 dimensions, lane l holds dimension
//                         rt * 16 + (l & 15) of the keys 4 g + j (j < 4) and 16 + 4 g + j - 4 (j >= 4), g = l >> 4 - exactly the 8 keys
//                         whose scores the QK^T accumulators leave in the lane of k-group g, so the probabilities go from the
//                         accumulator of one MFMA into the B operand of the next without leaving their lane
//                         (first version: kc [Bs, H, Sp, 64] / vt [Bs, H, 64, Sp] row-major - every quarter-wave of a fragment
//                         load then touches 16 rows x 16 bytes, the address unit delivers 16 bytes per lookup instead of 64:
//                         the skinny linears' finding, tools/probes/skinny_stamps.py; 22.8 us per launch at beam 4)
//   mask [Bs, Sp] uint8   1 = masked (padding of the context, and everything past S + 1)
// Per 32 keys a wave issues 4 + 4 sixteen-byte loads per lane, 4 MFMAs for the scores (A = K tile, B = the hypotheses as
// columns: up to 16 of them for the price of one), ~60 VALU instructions of online softmax in the exp2 domain on 8 scores per
// lane, 4 MFMAs for P V: ~0.8 instructions per cache byte against 3.8.  Waves of a workgroup take blocks of 32 keys round
// robin and merge through LDS.
struct AttnPkCtx { const uint16_t *q, *kc, *vt; const uint8_t* mask; uint16_t* out; long q_sb, o_sb; int Sp; };
struct AttnPkArgs { AttnPkCtx c[SK_MAXP]; int B, H, beams; };

__global__ __launch_bounds__(256) void attn_decode_packed_kernel(AttnPkArgs g) {
  const AttnPkCtx& p = g.c[blockIdx.y];
  __shared__ float po[4][64][17];                              // per wave: O^T [d][hypothesis] (+1: bank spread)
  __shared__ float pml[4][16][2];
  typedef float c4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lg = lane >> 4;
  const int groups = (g.beams + 15) / 16;
  const int h = blockIdx.x % g.H, bg = blockIdx.x / g.H, bs = bg / groups, j0 = (bg % groups) * 16;
  const int nq = g.beams - j0 < 16 ? g.beams - j0 : 16;
  const int b0 = bs * g.beams + j0;
  const int Sp = p.Sp, nblk = Sp >> 5;
  constexpr float LOG2E = 1.4426950408889634f;
  sk_u4 qf[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    qf[c] = sk_u4{0u, 0u, 0u, 0u};
    if (lr < nq) qf[c] = *reinterpret_cast<const sk_u4*>(p.q + (long)(b0 + lr) * p.q_sb + h * 64 + c * 32 + lg * 8);
  }
  // both caches in fragment order (layouts above): every 16-byte wave load below is one contiguous KB
  const uint16_t* kbase = p.kc + (long)(bs * g.H + h) * Sp * 64 + lane * 8;
  const uint16_t* vbase = p.vt + (long)(bs * g.H + h) * 64 * Sp + lane * 8;
  const uint8_t* mbase = p.mask + (long)bs * Sp + lg * 4;
  c4 acc_o[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) acc_o[rt] = c4{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;
  // (Measured and not kept, fragment-order cache: TWO blocks of a wave in flight, refill loads unconditional - 156 registers, two waves
  //  per SIMD: beam-4 step 476 -> 487 us; EIGHT waves per workgroup - the article's 17 blocks in 2-3 iterations per wave instead
  //  of 4-5: 488 us.)
  // (Measured and not kept: PAIRS of adjacent blocks per wave and iteration - the 16-byte loads of a V^T row then cover whole
  //  128-byte lines, 16 loads per lane in flight - 22.6-24.9 -> 24.5-26.1 us per launch at beam 4, 20.4 -> 21.5 with one hypothesis.)
  sk_u4 kf[2][2], vf[4];
  uint32_t mw[2];
  auto load = [&](int blk) __attribute__((always_inline)) {
    const uint16_t* kp = kbase + (long)blk * 2048;
    const uint16_t* vp = vbase + (long)blk * 2048;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      kf[t][0] = *reinterpret_cast<const sk_u4*>(kp + t * 1024);
      kf[t][1] = *reinterpret_cast<const sk_u4*>(kp + t * 1024 + 512);
      mw[t] = *reinterpret_cast<const uint32_t*>(mbase + blk * 32 + t * 16);
    }
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) vf[rt] = *reinterpret_cast<const sk_u4*>(vp + rt * 512);
  };
  if (wave < nblk) load(wave);
  for (int blk = wave; blk < nblk; blk += 4) {
    c4 sc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      sc[t] = c4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 2; ++c)
        sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sk_bf16x8, kf[t][c]), __builtin_bit_cast(sk_bf16x8, qf[c]),
                                                        sc[t], 0, 0, 0);
    }
    const sk_u4 v0 = vf[0], v1 = vf[1], v2 = vf[2], v3 = vf[3];
    const uint32_t m0 = mw[0], m1 = mw[1];
    if (blk + 4 < nblk) load(blk + 4);                          // the next block of this wave flies under the softmax
    // scores of this lane: keys 4 lg + r (tile 0) and 16 + 4 lg + r (tile 1) for hypothesis lr
    float v[8];
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = ((m0 >> (8 * r)) & 0xffu) ? -INFINITY : sc[0][r] * LOG2E;
      v[4 + r] = ((m1 >> (8 * r)) & 0xffu) ? -INFINITY : sc[1][r] * LOG2E;
      mx = fmaxf(mx, fmaxf(v[r], v[4 + r]));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mn = fmaxf(m, mx);
    const float a = mn == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(m - mn);
    l *= a;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) acc_o[rt] *= a;
    m = mn;
    float pr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pr[j] = v[j] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(v[j] - mn);
      l += pr[j];
    }
    sk_u4 pf;
    pf.x = (uint32_t)f2bf(pr[0]) | ((uint32_t)f2bf(pr[1]) << 16);
    pf.y = (uint32_t)f2bf(pr[2]) | ((uint32_t)f2bf(pr[3]) << 16);
    pf.z = (uint32_t)f2bf(pr[4]) | ((uint32_t)f2bf(pr[5]) << 16);
    pf.w = (uint32_t)f2bf(pr[6]) | ((uint32_t)f2bf(pr[7]) << 16);
    acc_o[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sk_bf16x8, v0), __builtin_bit_cast(sk_bf16x8, pf), acc_o[0], 0, 0, 0);
    acc_o[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sk_bf16x8, v1), __builtin_bit_cast(sk_bf16x8, pf), acc_o[1], 0, 0, 0);
    acc_o[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sk_bf16x8, v2), __builtin_bit_cast(sk_bf16x8, pf), acc_o[2], 0, 0, 0);
    acc_o[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sk_bf16x8, v3), __builtin_bit_cast(sk_bf16x8, pf), acc_o[3], 0, 0, 0);
  }
  // ---- the row sums over the 4 k-groups, then the waves through LDS
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  const int n_waves = nblk < 4 ? nblk : 4;
  if (wave < n_waves) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) po[wave][rt * 16 + lg * 4 + r][lr] = acc_o[rt][r];
    if (lg == 0) { pml[wave][lr][0] = m; pml[wave][lr][1] = l; }
  }
  __syncthreads();
  for (int t = tid; t < nq * 64; t += 256) {
    const int qi = t >> 6, d = t & 63;
    float mn = -INFINITY;
    for (int w = 0; w < n_waves; ++w) mn = fmaxf(mn, pml[w][qi][0]);
    float lt = 0.f, o = 0.f;
    for (int w = 0; w < n_waves; ++w) {
      const float a = mn == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(pml[w][qi][0] - mn);
      lt = fmaf(pml[w][qi][1], a, lt);
      o = fmaf(po[w][d][qi], a, o);
    }
    p.out[(long)(b0 + qi) * p.o_sb + h * 64 + d] = f2bf(lt > 0.f ? o / lt : 0.f);
  }
}
// n_ctx <= 4 contexts of one decode step in one launch over packed caches (layouts above; HOST arrays of n_ctx entries):
// q[c] bf16 [B, H*64] (row stride q_sb[c], already scaled), kc[c] / vt[c] / mask[c] for B / beams samples, Sp[c] % 32 == 0,
// out[c] bf16 [B, H*64] (row stride o_sb[c]).  Head width 64.  The `beams` hypotheses of a sample are columns of one MFMA.
extern "C" int tell_attn_decode_packed(int n_ctx, const void* const* q, const long* q_sb, const void* const* kc,
                                       const void* const* vt, const void* const* mask, const int* Sp, void* const* out,
                                       const long* o_sb, int B, int H, int beams, hipStream_t stream) {
  TELL_REQUIRE(n_ctx >= 1 && n_ctx <= SK_MAXP && B > 0 && H > 0 && beams >= 1 && B % beams == 0, "attn_decode_packed: 1-4 contexts");
  AttnPkArgs g;
  g.B = B; g.H = H; g.beams = beams;
  for (int c = 0; c < SK_MAXP; ++c) {
    const int j = c < n_ctx ? c : 0;
    TELL_REQUIRE(Sp[j] >= 32 && Sp[j] % 32 == 0 && q_sb[j] % 8 == 0 && mask[j] && kc[j] && vt[j], "attn_decode_packed: Sp % 32, masks");
    TELL_REQUIRE(((reinterpret_cast<uintptr_t>(kc[j]) | reinterpret_cast<uintptr_t>(vt[j]) | reinterpret_cast<uintptr_t>(q[j])) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(mask[j]) & 3) == 0, "attn_decode_packed: 16-byte aligned caches");
    g.c[c].q = (const uint16_t*)q[j]; g.c[c].kc = (const uint16_t*)kc[j]; g.c[c].vt = (const uint16_t*)vt[j];
    g.c[c].mask = (const uint8_t*)mask[j]; g.c[c].out = (uint16_t*)out[j]; g.c[c].q_sb = q_sb[j]; g.c[c].o_sb = o_sb[j];
    g.c[c].Sp = Sp[j];
  }
  const int samples = B / beams, groups = (beams + 15) / 16;
  hipLaunchKernelGGL(attn_decode_packed_kernel, dim3(samples * groups * H, n_ctx), dim3(256), 0, stream, g);
  return tell_check_launch("attn_decode_packed");
}

// ------------------------------------------------------------------ the step's token embedding, front half
// adaptive.py:61-76 at T = 1: a token's embedding is proj_band . table_band[id - lo] with band-dependent width.  As ONE
// skinny linear: row m of `cat` holds the token's table row in its band's column range (zeros elsewhere), the weight is
// [proj_0 | proj_1 | ...] along K.  This kernel builds `cat` (bf16 [M, ktot]) and gathers the sinusoid rows
// (positional.py:167-211: pad -> row pad_idx, else pad_idx + 1 + start_pos + device step) as the fp32 residual.
struct EmbedStepArgs {
  const uint16_t* table[4]; int lo[4], hi[4], dim[4], off[4];
  int nb, ktot, E, pos_rows, pos_pad, start_pos;
};
__global__ __launch_bounds__(256) void embed_gather_step_kernel(const long* __restrict__ ids, EmbedStepArgs p,
                                                                uint16_t* __restrict__ cat,
                                                                const float* __restrict__ pos_table,
                                                                float* __restrict__ pos_out,
                                                                const uint32_t* step, const uint32_t* next) {
  const int m = blockIdx.x, tid = threadIdx.x;
  // in-graph bookkeeping (api.hip g_tell_pos_next): this step's offset is in `next`; block 0 publishes it as the counter
  // the later kernels of the step read
  const uint32_t* sp = next ? next : step;
  const int sv = sp ? (int)*sp : 0;
  if (next && m == 0 && tid == 0) *const_cast<uint32_t*>(step) = (uint32_t)sv;
  const long id = ids[m];
  int band = -1;
  for (int b = 0; b < p.nb; ++b)
    if (id >= p.lo[b] && id < p.hi[b]) band = b;
  uint16_t* dst = cat + (long)m * p.ktot;
  const uint16_t* src = band >= 0 ? p.table[band] + (id - p.lo[band]) * p.dim[band] : nullptr;
  const int c0 = band >= 0 ? p.off[band] : 0, c1 = band >= 0 ? c0 + p.dim[band] : 0;
  for (int c = tid * 8; c < p.ktot; c += 2048) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (c >= c0 && c < c1) v = *reinterpret_cast<const uint4*>(src + (c - c0));
    *reinterpret_cast<uint4*>(dst + c) = v;
  }
  int pos = id == p.pos_pad ? p.pos_pad : p.pos_pad + 1 + p.start_pos + sv;
  if (pos >= p.pos_rows) pos = p.pos_rows - 1;
  for (int c = tid * 4; c < p.E; c += 1024)
    *reinterpret_cast<float4*>(pos_out + (long)m * p.E + c) = *reinterpret_cast<const float4*>(pos_table + (long)pos * p.E + c);
}
// ids [M] int64; nb <= 4 bands: tables[b] bf16 [hi_b - lo_b, dim_b] (dim_b % 8 == 0), band b covers ids lo_b .. hi_b - 1
// and columns off_b .. off_b + dim_b - 1 of cat [M, ktot] (HOST arrays); pos_out fp32 [M, E].
extern "C" int tell_embed_gather_step(const long* ids, int M, int nb, const void* const* tables, const int* lo, const int* hi,
                                      const int* dim, const int* off, void* cat, int ktot, const float* pos_table,
                                      int pos_rows, int pos_pad, int start_pos, float* pos_out, int E, hipStream_t stream) {
  TELL_REQUIRE(M > 0 && nb >= 1 && nb <= 4 && ktot % 8 == 0 && E % 4 == 0, "embed_gather_step: bad shape");
  EmbedStepArgs a;
  for (int b = 0; b < 4; ++b) {
    const int j = b < nb ? b : 0;
    TELL_REQUIRE(dim[j] % 8 == 0 && off[j] % 8 == 0 && off[j] + dim[j] <= ktot, "embed_gather_step: 16-byte bands inside cat");
    a.table[b] = static_cast<const uint16_t*>(tables[j]); a.lo[b] = lo[j]; a.hi[b] = hi[j]; a.dim[b] = dim[j]; a.off[b] = off[j];
  }
  a.nb = nb; a.ktot = ktot; a.E = E; a.pos_rows = pos_rows; a.pos_pad = pos_pad; a.start_pos = start_pos;
  hipLaunchKernelGGL(embed_gather_step_kernel, dim3(M), dim3(256), 0, stream, ids, a, static_cast<uint16_t*>(cat), pos_table,
                     pos_out, g_tell_pos_step, g_tell_pos_step ? g_tell_pos_next : nullptr);
  return tell_check_launch("embed_gather_step");
}

// The same front half against a PRE-PROJECTED table (round 6): during generation the weights do not move, so
// scale * proj_band . table_band[v] is computed ONCE per vocabulary entry ([V, E] fp32, three GEMMs when a decode graph is
// built) and the step's embedding is a lookup: out[m] = bf16(table[id[m]] + sinusoid[position]) - one 5 us launch instead of
// the gather (5 us) + a skinny linear that streamed [proj_0 | proj_1 | proj_2] (6 MB, two thirds of it against zeros) in 12 us.
// Same roles for the position counter as tell_embed_gather_step (first kernel of the captured step).
__global__ __launch_bounds__(256) void embed_lookup_step_kernel(const long* __restrict__ ids, const float* __restrict__ table,
                                                                int V, const float* __restrict__ pos_table, int pos_rows,
                                                                int pos_pad, int start_pos, uint16_t* __restrict__ out, int E,
                                                                const uint32_t* step, const uint32_t* next) {
  const int m = blockIdx.x, tid = threadIdx.x;
  const uint32_t* sp = next ? next : step;
  const int sv = sp ? (int)*sp : 0;
  if (next && m == 0 && tid == 0) *const_cast<uint32_t*>(step) = (uint32_t)sv;
  long id = ids[m];
  int pos = id == pos_pad ? pos_pad : pos_pad + 1 + start_pos + sv;
  if (pos >= pos_rows) pos = pos_rows - 1;
  const bool known = id >= 0 && id < V;
  const float* row = table + (known ? id : 0) * (long)E;
  const float* prow = pos_table + (long)pos * E;
  for (int c = tid * 4; c < E; c += 1024) {
    float4 a = known ? *reinterpret_cast<const float4*>(row + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 b = *reinterpret_cast<const float4*>(prow + c);
    uint2 o;
    o.x = (uint32_t)f2bf(a.x + b.x) | ((uint32_t)f2bf(a.y + b.y) << 16);
    o.y = (uint32_t)f2bf(a.z + b.z) | ((uint32_t)f2bf(a.w + b.w) << 16);
    *reinterpret_cast<uint2*>(out + (long)m * E + c) = o;
  }
}
// ids [M] int64; table fp32 [V, E] = scale * projected embedding of every token; out bf16 [M, E].  E % 4 == 0.
extern "C" int tell_embed_lookup_step(const long* ids, int M, const float* table, int V, const float* pos_table, int pos_rows,
                                      int pos_pad, int start_pos, void* out, int E, hipStream_t stream) {
  TELL_REQUIRE(M > 0 && V > 0 && E % 4 == 0 && (reinterpret_cast<uintptr_t>(table) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(out) & 7) == 0, "embed_lookup_step: bad shape");
  hipLaunchKernelGGL(embed_lookup_step_kernel, dim3(M), dim3(256), 0, stream, ids, table, V, pos_table, pos_rows, pos_pad,
                     start_pos, static_cast<uint16_t*>(out), E, g_tell_pos_step, g_tell_pos_step ? g_tell_pos_next : nullptr);
  return tell_check_launch("embed_lookup_step");
}

// ------------------------------------------------------------------ beam search: one step's bookkeeping
// What the host loop does per token after the top-k head (scoring = sum of token log-probs, a finished
// hypothesis has one continuation: pad at no cost): per sample the K best of the K x K candidates
// cum[parent] + lp[parent][m] (lowest flat index wins a tie), the surviving sequences / per-token log-probs gathered
// by parent and extended, the parent ROW of every surviving hypothesis for the state reorder, the next input tokens.
// One 64-thread workgroup per sample; K <= 8.  Thirty-five elementwise / gather / top-k launches per token before.
__global__ __launch_bounds__(64) void beam_update_kernel(const int* __restrict__ tk, const float* __restrict__ lp,
                                                         float* __restrict__ cum, uint8_t* __restrict__ finished,
                                                         long* __restrict__ seqs, float* __restrict__ lps,
                                                         long* __restrict__ cur, long* __restrict__ rows, int K, int L,
                                                         int step_host, int pad, int eos, float inv_temp, int* back, int n_back,
                                                         int M, int* counter, const int* step_dev) {
  const int step = step_dev ? *step_dev + 1 : step_host;      // (in a captured step: the registered counter holds step - 1)
  __shared__ long s_seq[8 * 256];
  __shared__ float s_lp[8 * 256];
  __shared__ int s_parent[8], s_tok[8];
  __shared__ float s_top[8], s_dlp[8];
  __shared__ uint8_t s_fin[8];
  __shared__ int s_back[31 * 8];
  const int b = blockIdx.x, t = threadIdx.x;
  // the ancestor table of the DynamicConv rings (dynconv_step_kernel): this sample's K columns, before the update
  if (back)
    for (int e = t; e < n_back * K; e += 64) s_back[e] = back[(long)(e / K) * M + b * K + e % K];
  // the histories of the sample's K hypotheses into LDS FIRST: they do not depend on the selection below, whose K rounds of
  // shuffles then run under these loads (round 6: the launch was a chain of four dependent memory round trips + two loops with
  // an integer division per element; 15 us at K = 4)
  // (whole rows, L and L - 1 columns: the trip counts must not hang on `step`, a value this launch first has to load - with
  //  step + 2 columns the history loads queued behind that round trip; 15.7 -> 13.1 us)
  const int Lp = L - 1, nc = step + 2 < L ? step + 2 : L, ncp = step + 1 < Lp ? step + 1 : Lp;
  for (int c = t; c < K * L; c += 64) s_seq[c] = seqs[(long)b * K * L + c];
  for (int c = t; c < K * Lp; c += 64) s_lp[c] = lps[(long)b * K * Lp + c];
  const int j = t / K, m = t % K;
  float score = -INFINITY;
  int token = pad;
  bool was_fin = false;
  if (t < K * K) {
    was_fin = finished[b * K + j] != 0;
    const float l = was_fin ? (m == 0 ? 0.f : -INFINITY) : lp[((long)b * K + j) * K + m] * inv_temp;
    token = was_fin ? pad : tk[((long)b * K + j) * K + m];
    score = cum[b * K + j] + l;
  }
  bool taken = false;
  for (int r = 0; r < K; ++r) {
    float bv = taken ? -INFINITY : score;
    int bi = taken || t >= K * K ? 0x7fffffff : t;
    if (bv != bv) { bv = -INFINITY; }                       // (a NaN score never wins)
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (t == bi) {
      taken = true;
      s_parent[r] = j; s_top[r] = score;
      s_tok[r] = token;
      s_fin[r] = (was_fin || token == eos) ? 1 : 0;
      s_dlp[r] = was_fin ? 0.f : score - cum[b * K + j];
    }
  }
  __syncthreads();
  // gather the surviving histories by parent (through LDS: the permutation is in place)
  // (only columns 0 .. step + 1 - everything behind them is still the initial padding in every row; a hypothesis that
  //  descends from its own slot keeps its row: only the new column is written)
  for (int r = 0; r < K; ++r) {
    const int pr = s_parent[r];
    if (pr != r) {
      for (int c = t; c < nc; c += 64) seqs[(long)b * K * L + r * L + c] = s_seq[pr * L + c];
      for (int c = t; c < ncp; c += 64) lps[(long)b * K * Lp + r * Lp + c] = s_lp[pr * Lp + c];
    }
  }
  __syncthreads();
  if (t < K) {
    seqs[(long)b * K * L + t * L + step + 1] = (long)s_tok[t];
    if (step < Lp) lps[(long)b * K * Lp + t * Lp + step] = s_dlp[t];
  }
  if (t < K) {
    cum[b * K + t] = s_top[t];
    finished[b * K + t] = s_fin[t];
    cur[b * K + t] = s_tok[t];
    rows[b * K + t] = (long)b * K + s_parent[t];
  }
  // slot r now holds a child of slot parent[r]: one step ago it sat there, j steps ago where the parent sat j - 1 steps ago
  if (back)
    for (int e = t; e < n_back * K; e += 64) {
      const int j = e / K, r = e % K;
      back[(long)j * M + b * K + r] = j == 0 ? b * K + s_parent[r] : s_back[(j - 1) * K + s_parent[r]];
    }
  if (counter && b == 0 && t == 0) *counter = step;            // position offset of the NEXT replay of a captured step: (step + 1) - 1
}
// tk int32 / lp fp32 [B,K,K] (the K best continuations of every hypothesis, best first), cum fp32 [B,K], finished uint8
// [B,K], seqs int64 [B,K,L], lps fp32 [B,K,L-1] - all updated in place -, cur int64 [B*K] (next input tokens), rows
// int64 [B*K] (row each surviving hypothesis descends from).  K <= 8, L <= 256.
// back (optional): the ancestor table int32 [n_back <= 31][B*K] of the DynamicConv rings (tell_dynconv_step), composed in
// place with this step's parents - the reference's reorder_incremental_state (dynamic.py:338-342) without moving a row.
// counter (optional): device int32 that the captured decode step reads as its position offset; set to `step` (= the
// offset of step + 1 for a graph captured at step 1).
extern "C" int tell_beam_update(const int* tk, const float* lp, float* cum, uint8_t* finished, long* seqs, float* lps,
                                long* cur, long* rows, int B, int K, int L, int step, int pad, int eos, float inv_temp,
                                int* back, int n_back, int* counter, const int* step_dev, hipStream_t stream) {
  TELL_REQUIRE(B > 0 && K >= 1 && K <= 8 && L >= 2 && L <= 256 && (step_dev || (step >= 0 && step + 1 < L)), "beam_update: K <= 8, L <= 256");
  TELL_REQUIRE(!back || (n_back >= 1 && n_back <= 31), "beam_update: ancestor table of 1 .. 31 steps");
  hipLaunchKernelGGL(beam_update_kernel, dim3(B), dim3(64), 0, stream, tk, lp, cum, finished, seqs, lps, cur, rows, K, L,
                     step, pad, eos, inv_temp, back, back ? n_back : 0, B * K, counter, step_dev);
  return tell_check_launch("beam_update");
}

// Rows of the DynamicConv input buffers follow their hypotheses (dynamic.py:338-342 reorder_incremental_state):
// buf[p][r][:] <- buf[p][rows[r]][:] for every plane p of up to 8 buffers [planes, M, C] bf16, in place - rows[r] stays
// inside r's group of K consecutive rows (a sample's hypotheses), so a workgroup that owns (plane, group) reads its K
// rows into registers and writes them back permuted.  C == 1024.
struct ReorderArgs { uint16_t* buf[8]; int plane0[9]; int n; };
__global__ __launch_bounds__(128) void reorder_rows_kernel(ReorderArgs a, const long* __restrict__ rows, int M, int C, int K) {
  int pl = blockIdx.x, bi = 0;
  while (bi + 1 < a.n && pl >= a.plane0[bi + 1]) ++bi;
  pl -= a.plane0[bi];
  const int g = blockIdx.y, t = threadIdx.x;
  uint16_t* base = a.buf[bi] + ((long)pl * M + (long)g * K) * C + t * 8;
  sk_u4 v[8];
#pragma unroll
  for (int r = 0; r < 8; ++r)
    if (r < K) v[r] = *reinterpret_cast<const sk_u4*>(base + (long)r * C);
#pragma unroll
  for (int r = 0; r < 8; ++r)
    if (r < K) {
      const int src = (int)(rows[g * K + r] - (long)g * K);
      sk_u4 w = v[0];
#pragma unroll
      for (int q = 1; q < 8; ++q)
        if (q == src) w = v[q];
      *reinterpret_cast<sk_u4*>(base + (long)r * C) = w;
    }
}
extern "C" int tell_reorder_rows(int n, void* const* bufs, const int* planes, const long* rows, int M, int C, int K,
                                 hipStream_t stream) {
  TELL_REQUIRE(n >= 1 && n <= 8 && C == 1024 && K >= 1 && K <= 8 && M % K == 0, "reorder_rows: C = 1024, K <= 8 rows per group");
  ReorderArgs a;
  a.n = n; a.plane0[0] = 0;
  for (int i = 0; i < 8; ++i) {
    a.buf[i] = static_cast<uint16_t*>(bufs[i < n ? i : 0]);
    a.plane0[i + 1] = a.plane0[i] + (i < n ? planes[i] : 0);
  }
  if (a.plane0[n] == 0) return TELL_OK;
  hipLaunchKernelGGL(reorder_rows_kernel, dim3(a.plane0[n], M / K), dim3(128), 0, stream, a, rows, M, C, K);
  return tell_check_launch("reorder_rows");
}
