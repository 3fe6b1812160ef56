// Multi-head cross/self attention on the CDNA4 matrix cores
// (tell/modules/attention/multi_head.py:288-486; also RoBERTa self-attention).
//
//   scores = q.k^T (q pre-scaled by the projection GEMM), key-padding mask -> -inf,
//   the learned bias_k/bias_v row and the all-zero row (add_bias_kv / add_zero_attn,
//   :355-374, :416-427) are *virtual* keys S and S+1 synthesised while staging a
//   tile - K/V are never concatenated or copied; fp32 online softmax (:460-462);
//   dropout on the probabilities (:463); out = P.V.
//
// Every contraction runs through one primitive, mma32: a 32x32 output tile from
// two LDS tiles stored [row][k] with k contiguous (bf16: v_mfma_f32_32x32x16_bf16
// fed by ds_read_b128; f32: v_mfma_f32_32x32x2_f32 fed by ds_read_b32).  Tiles are
// staged so that the contraction index is contiguous (V and K^T/Q^T/dO^T tiles
// are transposed on the way into LDS).  Scores are produced TRANSPOSED
// (S^T[key][q] = K.Q^T) so that each lane owns one query column: the softmax
// row-reduction is 16 in-register values + one cross-half wavefront shuffle.
#include "common.h"
#include "options.h"
#include <type_traits>

template <typename T, int D> struct ACfg {
  static constexpr int VEC = Elem<T>::VEC;
  static constexpr int PAD = sizeof(T) == 2 ? 8 : 1;
  static constexpr int DS = D + PAD;          // stride of [32 rows][D] tiles
  static constexpr int SS = 32 + PAD;         // stride of [rows][32] tiles
  static constexpr int DP = D < 32 ? 32 : D;  // rows of [d][32] tiles (zero padded)
  static constexpr int DF = DP / 32;
};

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {   // -> v_cvt_pk_bf16_f32 (RNE)
  f32x2_t f = {lo, hi};
  bf16x2_t h = __builtin_convertvector(f, bf16x2_t);
  return *reinterpret_cast<unsigned*>(&h);
}
template <typename T, int KD>
__device__ __forceinline__ f32x16 mma32(f32x16 acc, const T* a, int as, const T* b, int bs, int lane) {
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int ks = 0; ks < KD; ks += 16) {
      bf16x8 av = *reinterpret_cast<const bf16x8*>(a + (lane & 31) * as + ks + ((lane >> 5) << 3));
      bf16x8 bv = *reinterpret_cast<const bf16x8*>(b + (lane & 31) * bs + ks + ((lane >> 5) << 3));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int ks = 0; ks < KD; ks += 2) {
      float av = a[(lane & 31) * as + ks + (lane >> 5)];
      float bv = b[(lane & 31) * bs + ks + (lane >> 5)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
  }
  return acc;
}
// row index (A-tile row) held in accumulator register r of this lane
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Stage 32 rows x D of a row-major source into tile[row][D] and/or tileT[d][row].
// rowptr(row) returns the row's base pointer or nullptr (-> zeros).  The load itself is unconditional
// (nullptr rows read `dummy`, any valid 16-byte-aligned address) and the zero is selected afterwards:
// loads inside divergent branches make hipcc serialise them with s_waitcnt vmcnt(0).
template <typename T, int D, typename RowPtr>
__device__ __forceinline__ void stage32(T* tile, int stride, T* tileT, int strideT, int tid, int nthr,
                                        RowPtr rowptr, const T* dummy) {
  constexpr int VEC = Elem<T>::VEC, CPR = D / VEC;
  for (int c = tid; c < 32 * CPR; c += nthr) {
    const int row = c / CPR, ch = c % CPR;
    const T* p = rowptr(row);
    const bool ok = p != nullptr;
    uint4 v = *reinterpret_cast<const uint4*>((ok ? p : dummy) + (ok ? ch * VEC : 0));
    if (!ok) v = make_uint4(0, 0, 0, 0);
    if (tile) {
      if constexpr (sizeof(T) == 2) {
        *reinterpret_cast<uint4*>(tile + row * stride + ch * VEC) = v;
      } else {
        float* q = reinterpret_cast<float*>(tile) + row * stride + ch * VEC;
        q[0] = __uint_as_float(v.x); q[1] = __uint_as_float(v.y);
        q[2] = __uint_as_float(v.z); q[3] = __uint_as_float(v.w);
      }
    }
    if (tileT) {
      const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
      for (int k = 0; k < VEC; ++k) tileT[(ch * VEC + k) * strideT + row] = e[k];
    }
  }
}
// stage32 split in two for software prefetch by ONE wave (64 lanes, 32 rows x D): load32 issues the tile's global loads
// into registers (rows without a source read `dummy` and are zeroed at the store), store32 writes them to tile[row][D]
// and / or tileT[d][row] later - the loads of key tile i+1 are in flight while tile i is being computed.
template <typename T, int D> struct Tile32Regs { uint4 v[(32 * D / Elem<T>::VEC + 63) / 64]; unsigned ok; };
template <typename T, int D, typename RowPtr>
__device__ __forceinline__ void load32(Tile32Regs<T, D>& r, int lane, RowPtr rowptr, const T* dummy) {
  constexpr int VEC = Elem<T>::VEC, CPR = D / VEC, N = (32 * CPR + 63) / 64;
  r.ok = 0u;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int c = lane + 64 * i, row = c / CPR, ch = c % CPR;
    const T* p = c < 32 * CPR ? rowptr(row) : nullptr;
    const bool ok = p != nullptr;
    r.v[i] = *reinterpret_cast<const uint4*>((ok ? p : dummy) + (ok ? ch * VEC : 0));
    r.ok |= ok ? 1u << i : 0u;
  }
}
template <typename T, int D>
__device__ __forceinline__ void store32(const Tile32Regs<T, D>& r, T* tile, int stride, T* tileT, int strideT, int lane) {
  constexpr int VEC = Elem<T>::VEC, CPR = D / VEC, N = (32 * CPR + 63) / 64;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int c = lane + 64 * i, row = c / CPR, ch = c % CPR;
    if (c >= 32 * CPR) continue;
    const uint4 v = (r.ok >> i) & 1u ? r.v[i] : make_uint4(0, 0, 0, 0);
    if (tile) {
      if constexpr (sizeof(T) == 2) {
        *reinterpret_cast<uint4*>(tile + row * stride + ch * VEC) = v;
      } else {
        float* q = reinterpret_cast<float*>(tile) + row * stride + ch * VEC;
        q[0] = __uint_as_float(v.x); q[1] = __uint_as_float(v.y);
        q[2] = __uint_as_float(v.z); q[3] = __uint_as_float(v.w);
      }
    }
    if (tileT) {
      const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
      for (int k = 0; k < VEC; ++k) tileT[(ch * VEC + k) * strideT + row] = e[k];
    }
  }
}
template <typename T>
__device__ __forceinline__ void zero_lds(T* p, int n, int tid, int nthr) {
  for (int i = tid; i < n; i += nthr) p[i] = (T)0;
}

struct AttnArgs {
  const void *q, *k, *v;          // element (b,h,t,d) at q + t*q_st + b*q_sb + h*D + d, etc.
  void* out;
  float* lse;                     // [B*H, Tq]
  const uint8_t* mask;            // [B, S] key padding (1 = masked) or null
  const void *bias_k, *bias_v;    // [H*D] in the compute dtype, or null
  long q_st, q_sb, k_ss, k_sb, v_ss, v_sb, o_st, o_sb;
  int B, H, Tq, S;
  int has_bias, has_zero;
  uint32_t thr; float inv_keep; uint32_t seed, salt;
  const uint32_t* step;      // replayable dropout (common.h tell_step_salt); NULL in eager mode
  // backward only
  const void *dout, *o;           // same layout as out
  void *dq, *dk, *dv;             // same layouts as q, k, v
  float *dbias_k, *dbias_v;       // [B, H*D] fp32 partials (summed over b by the caller), row stride dbias_ld
  long dbias_ld;
};

// ------------------------------------------------------------------ forward
template <typename T, int D, int NW>
__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(AttnArgs p) {
  const uint32_t salt_eff = tell_step_salt(p.salt, p.step);   // hoisted: one scalar load per kernel
  using C_ = ACfg<T, D>;
  constexpr int DS = C_::DS, SS = C_::SS, DP = C_::DP, DF = C_::DF;
  constexpr int WAVE_ELEMS = 2 * 32 * DS + DP * SS + 32 * SS;   // Qs, Ks, Vt, Ps
  static_assert(WAVE_ELEMS * sizeof(T) >= (2 + 16 * DF) * 64 * sizeof(float), "combine buffer must fit");
  __shared__ __attribute__((aligned(16))) T smem[NW * WAVE_ELEMS];
  __shared__ float key_bias[NW][32];                 // 0 / -inf per key of the wave's current tile

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
  const int QB = (p.Tq + 31) / 32;
  const int KSPLIT = (QB == 1) ? NW : 1;         // decoder / generation: waves split the keys
  const int qb = (QB == 1) ? 0 : blockIdx.x * NW + wave;
  const int ks = (QB == 1) ? wave : 0;
  const bool active = qb < QB;
  const int q0 = qb * 32;
  const int S_total = p.S + p.has_bias + p.has_zero;
  const int nkt = (S_total + 31) / 32;
  const int iters = (nkt + KSPLIT - 1) / KSPLIT;

  T* Qs = smem + wave * WAVE_ELEMS;
  T* Ks = Qs + 32 * DS;
  T* Vt = Ks + 32 * DS;
  T* Ps = Vt + DP * SS;
  const T* qg = static_cast<const T*>(p.q);
  const T* kg = static_cast<const T*>(p.k);
  const T* vg = static_cast<const T*>(p.v);
  const T* bk = static_cast<const T*>(p.bias_k);
  const T* bv = static_cast<const T*>(p.bias_v);

  // every global load of the prologue is issued before the first use of any of them: Q, then this wave's first K / V
  // tile (one round trip, not three)
  Tile32Regs<T, D> rq;
  load32<T, D>(rq, lane, [&](int row) -> const T* {
    int t = q0 + row;
    return (active && t < p.Tq) ? qg + t * p.q_st + b * p.q_sb + (long)h * D : nullptr;
  }, qg);

  f32x16 o[DF];
#pragma unroll
  for (int f = 0; f < DF; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[f][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int qi = lane & 31;

  // this wave's key tiles are prefetched one iteration ahead (registers), so their global latency hides under the
  // previous tile's MFMAs and softmax instead of standing in front of every tile
  Tile32Regs<T, D> rk, rv;
  auto fetch = [&](int kt_) __attribute__((always_inline)) {
    const int s0_ = kt_ * 32;
    load32<T, D>(rk, lane, [&](int row) -> const T* {
      int s = s0_ + row;
      if (s < p.S) return kg + s * p.k_ss + b * p.k_sb + (long)h * D;
      if (s == p.S && p.has_bias) return bk + (long)h * D;
      return nullptr;
    }, qg);
    load32<T, D>(rv, lane, [&](int row) -> const T* {
      int s = s0_ + row;
      if (s < p.S) return vg + s * p.v_ss + b * p.v_sb + (long)h * D;
      if (s == p.S && p.has_bias) return bv + (long)h * D;
      return nullptr;
    }, qg);
  };
  if (active && ks < nkt) fetch(ks);
  store32<T, D>(rq, Qs, DS, (T*)nullptr, 0, lane);
  if (DP > D) zero_lds(Vt, DP * SS, lane, 64);   // rows d >= D stay zero
  for (int it = 0; it < iters; ++it) {
    const int kt = it * KSPLIT + ks;
    const bool tv = active && kt < nkt;
    const int s0 = kt * 32;
    __syncthreads();                              // previous tile fully consumed
    if (tv) {
      store32<T, D>(rk, Ks, DS, (T*)nullptr, 0, lane);
      store32<T, D>(rv, (T*)nullptr, 0, Vt, SS, lane);
      if (kt + KSPLIT < nkt) fetch(kt + KSPLIT);
      if (lane < 32) {                              // key-padding mask + range, branch-free in the softmax
        const int s = s0 + lane;
        bool ok = s < S_total;
        if (ok && s < p.S && p.mask) ok = p.mask[(long)b * p.S + s] == 0;
        key_bias[wave][lane] = ok ? 0.f : -INFINITY;
      }
    }
    __syncthreads();
    float pr[16];
    if (tv) {
      f32x16 st;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = 0.f;
      st = mma32<T, D>(st, Ks, DS, Qs, DS, lane);            // S^T[key][q]
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[r] += key_bias[wave][acc_row(r, lane)];
        mx = fmaxf(mx, st[r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_new);
      float ls = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pr[r] = (st[r] == -INFINITY) ? 0.f : __expf(st[r] - m_new);
        ls += pr[r];
      }
      ls += __shfl_xor(ls, 32, 64);
      l_run = l_run * alpha + ls;
      m_run = m_new;
#pragma unroll
      for (int f = 0; f < DF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[f][r] *= alpha;
      const int t = q0 + qi;
      if (p.thr) {                                            // one hash per aligned key pair when the row length is even
        const uint64_t base = ((uint64_t)bh * p.Tq + t) * S_total + s0;
        if ((S_total & 1) == 0) {
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            float k0, k1;
            tell_keep2(p.seed, salt_eff, base + acc_row(r, lane), p.thr, p.inv_keep, k0, k1);
            pr[r] *= k0;
            pr[r + 1] *= k1;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) pr[r] *= tell_keep(p.seed, salt_eff, base + acc_row(r, lane), p.thr, p.inv_keep);
        }
      }
      // P[q][key], key contiguous: a register group holds 4 consecutive keys -> one 8-byte LDS write (bf16)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        T* pp = Ps + qi * SS + 8 * g + 4 * (lane >> 5);
        if constexpr (sizeof(T) == 2) {
          uint2 w;
          w.x = pack_bf16x2(pr[4 * g], pr[4 * g + 1]);
          w.y = pack_bf16x2(pr[4 * g + 2], pr[4 * g + 3]);
          *reinterpret_cast<uint2*>(pp) = w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) Elem<T>::st(pp + e, pr[4 * g + e]);
        }
      }
    }
    __syncthreads();
    if (tv) {
#pragma unroll
      for (int f = 0; f < DF; ++f) o[f] = mma32<T, 32>(o[f], Vt + f * 32 * SS, SS, Ps, SS, lane);  // O^T[d][q]
    }
  }
  __syncthreads();

  // ---- combine the key splits (only when QB == 1) and write out
  float* comb = reinterpret_cast<float*>(smem + wave * WAVE_ELEMS);
  if (KSPLIT > 1) {
    comb[0 * 64 + lane] = m_run;
    comb[1 * 64 + lane] = l_run;
#pragma unroll
    for (int f = 0; f < DF; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) comb[(2 + f * 16 + r) * 64 + lane] = o[f][r];
  }
  __syncthreads();
  if (KSPLIT > 1) {
    if (wave != 0) return;
    float mm = m_run;
    for (int w = 1; w < NW; ++w) {
      const float* cw = reinterpret_cast<const float*>(smem + w * WAVE_ELEMS);
      mm = fmaxf(mm, cw[lane]);
    }
    float ltot = 0.f;
    f32x16 acc[DF];
#pragma unroll
    for (int f = 0; f < DF; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
    for (int w = 0; w < NW; ++w) {
      const float* cw = reinterpret_cast<const float*>(smem + w * WAVE_ELEMS);
      const float mw = cw[lane];
      const float sc = (mw == -INFINITY) ? 0.f : __expf(mw - mm);
      ltot += cw[64 + lane] * sc;
#pragma unroll
      for (int f = 0; f < DF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] += cw[(2 + f * 16 + r) * 64 + lane] * sc;
    }
    m_run = mm; l_run = ltot;
#pragma unroll
    for (int f = 0; f < DF; ++f) o[f] = acc[f];
  }
  if (!active) return;
  const int t = q0 + qi;
  if (t >= p.Tq) return;
  const float inv_l = l_run > 0.f ? 1.f / l_run : 0.f;
  T* og = static_cast<T*>(p.out) + t * p.o_st + b * p.o_sb + (long)h * D;
#pragma unroll
  for (int f = 0; f < DF; ++f)
#pragma unroll
    for (int g = 0; g < 4; ++g) {                     // 4 consecutive d per register group -> one 8-byte store (bf16)
      const int d0 = f * 32 + 8 * g + 4 * (lane >> 5);
      if (d0 >= D) continue;
      if constexpr (sizeof(T) == 2) {
        uint2 w;
        w.x = pack_bf16x2(o[f][4 * g] * inv_l, o[f][4 * g + 1] * inv_l);
        w.y = pack_bf16x2(o[f][4 * g + 2] * inv_l, o[f][4 * g + 3] * inv_l);
        *reinterpret_cast<uint2*>(og + d0) = w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) Elem<T>::st(og + d0 + e, o[f][4 * g + e] * inv_l);
      }
    }
  if (lane < 32 && p.lse) p.lse[(long)bh * p.Tq + t] = l_run > 0.f ? m_run + __logf(l_run) : INFINITY;
}

// ------------------------------------------------------------------ forward, long sequences (Tq >= 128, D = 64)
// RoBERTa self-attention shape.  One workgroup = 4 waves = 128 query rows of one (b,h); the 64-key
// K tile and the transposed V tile are staged ONCE per workgroup (all 256 threads, next tile
// prefetched into registers under the MFMAs) and shared by the 4 waves; each wave keeps its own
// Q tile, probability tile and O^T accumulators.  Per 64 keys and wave: 16 MFMA 32x32x16.

template <typename T, bool DROP>
__global__ __launch_bounds__(256) void attn_fwd_tile64_kernel(AttnArgs p) {
  const uint32_t salt_eff = tell_step_salt(p.salt, p.step);   // hoisted: one scalar load per kernel
  constexpr int D = 64, KT = 64, NW = 4;
  constexpr int VEC = Elem<T>::VEC;
  constexpr int PAD = sizeof(T) == 2 ? 8 : 1;
  constexpr int DS = 64 + PAD;                       // stride of every [rows][64] tile
  constexpr int CPR = D / VEC;                       // 16-byte chunks per row
  constexpr int CPT = KT * CPR / 256;                // chunks per thread per tile (bf16 2, f32 4)
  __shared__ __attribute__((aligned(16))) T Ks[KT * DS];
  __shared__ __attribute__((aligned(16))) T Vt[D * DS];
  __shared__ __attribute__((aligned(16))) T Qs[NW][32 * DS];
  __shared__ __attribute__((aligned(16))) T Ps[NW][32 * DS];
  __shared__ float key_bias[KT];                     // 0 for a valid key, -inf for padded / out-of-range keys
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
  const int QB = (p.Tq + 31) / 32;
  const int qb = blockIdx.x * NW + wave;
  const bool active = qb < QB;
  const int q0 = qb * 32;
  const int S_total = p.S + p.has_bias + p.has_zero;
  const int nkt = (S_total + KT - 1) / KT;
  const T* qg = static_cast<const T*>(p.q);
  const T* kg = static_cast<const T*>(p.k);
  const T* vg = static_cast<const T*>(p.v);
  const T* bk = static_cast<const T*>(p.bias_k);
  const T* bv = static_cast<const T*>(p.bias_v);

  stage32<T, D>(Qs[wave], DS, (T*)nullptr, 0, lane, 64, [&](int row) -> const T* {
    int t = q0 + row;
    return (active && t < p.Tq) ? qg + t * p.q_st + b * p.q_sb + (long)h * D : nullptr;
  }, qg);

  // chunk -> (key row, 16-byte column).  bf16: a thread owns the SAME column of two adjacent keys so
  // the transposed V store packs key pairs into 32-bit words; f32: plain round-robin.
  auto chunk_row = [&](int i) -> int { return sizeof(T) == 2 ? 2 * (tid / CPR) + i : (tid + 256 * i) / CPR; };
  auto chunk_col = [&](int i) -> int { return sizeof(T) == 2 ? tid % CPR : (tid + 256 * i) % CPR; };
  u32x4 rk[CPT], rv[CPT];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
#define LOAD_TILE(KTI)                                                                          \
  {                                                                                             \
    _Pragma("unroll") for (int i = 0; i < CPT; ++i) {                                           \
      const int s = (KTI) * KT + chunk_row(i), ch = chunk_col(i);                               \
      const bool real = s < p.S, isb = (s == p.S) && p.has_bias;                                \
      const T* kp = real ? kg + s * p.k_ss + b * p.k_sb + (long)h * D : (isb ? bk + (long)h * D : qg); \
      const T* vp = real ? vg + s * p.v_ss + b * p.v_sb + (long)h * D : (isb ? bv + (long)h * D : qg); \
      const int off = (real || isb) ? ch * VEC : 0;                                             \
      rk[i] = *reinterpret_cast<const u32x4*>(kp + off);   /* unconditional; zeroed at the LDS store */ \
      rv[i] = *reinterpret_cast<const u32x4*>(vp + off);                                        \
    }                                                                                           \
  }
#define TILE_ROW_OK(KTI, I) ((((KTI) * KT + chunk_row(I)) < p.S) || ((((KTI) * KT + chunk_row(I)) == p.S) && p.has_bias))
  LOAD_TILE(0)

  f32x16 o[2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[f][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int qi = lane & 31;
  T* myP = Ps[wave];

  for (int kt = 0; kt < nkt; ++kt) {
    const int s0 = kt * KT;
    __syncthreads();                                  // every wave is done with the previous tile
    // ---- registers -> LDS: K rows as they are, V transposed (Vt[d][key])
#pragma unroll
    for (int i = 0; i < CPT; ++i)
      if (!TILE_ROW_OK(kt, i)) { rk[i] = zero4; rv[i] = zero4; }        // virtual zero row / past the end
    if constexpr (sizeof(T) == 2) {
      const int key = chunk_row(0), ch = chunk_col(0);
#pragma unroll
      for (int i = 0; i < CPT; ++i) *reinterpret_cast<u32x4*>(Ks + (key + i) * DS + ch * VEC) = rk[i];
#pragma unroll
      for (int w = 0; w < 4; ++w) {                   // 8 d-values, 2 per 32-bit register
        const unsigned a0 = rv[0][w], a1 = rv[1][w];
        const unsigned lo = (a0 & 0xffffu) | (a1 << 16);          // d = ch*8 + 2w    : keys (key, key+1)
        const unsigned hi = (a0 >> 16) | (a1 & 0xffff0000u);      // d = ch*8 + 2w + 1
        *reinterpret_cast<unsigned*>(Vt + (ch * VEC + 2 * w) * DS + key) = lo;
        *reinterpret_cast<unsigned*>(Vt + (ch * VEC + 2 * w + 1) * DS + key) = hi;
      }
    } else {
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        const int key = chunk_row(i), ch = chunk_col(i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          Ks[key * DS + ch * VEC + e] = __uint_as_float(rk[i][e]);
          Vt[(ch * VEC + e) * DS + key] = __uint_as_float(rv[i][e]);
        }
      }
    }
    if (tid < KT) {                                   // key-padding mask + range check, once per tile
      const int s = s0 + tid;
      bool ok = s < S_total;
      if (ok && s < p.S && p.mask) ok = p.mask[(long)b * p.S + s] == 0;
      key_bias[tid] = ok ? 0.f : -INFINITY;
    }
    __syncthreads();
    if (kt + 1 < nkt) LOAD_TILE(kt + 1)               // next tile streams in under the MFMAs
    if (active) {
      f32x16 st[2];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[f][r] = 0.f;
        st[f] = mma32<T, D>(st[f], Ks + f * 32 * DS, DS, Qs[wave], DS, lane);     // S^T[key][q]
      }
      float mx = -INFINITY;
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          st[f][r] += key_bias[f * 32 + acc_row(r, lane)];
          mx = fmaxf(mx, st[f][r]);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;        // all keys masked so far: exp(-inf - 0) = 0
      const float alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_new);
      float ls = 0.f;
      const int t = q0 + qi;
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __expf(st[f][r] - m_safe);                  // masked keys: exp(-inf) = 0, branch-free
          ls += pv;
          st[f][r] = pv;
        }
      if constexpr (DROP) {
        // element index of P[bh][t][key]; this lane's keys come in aligned pairs (r, r+1), so with an even
        // row length one hash decides two probabilities (common.h tell_keep2)
        const uint64_t base = ((uint64_t)bh * p.Tq + t) * S_total + s0 + 4 * (lane >> 5);
        if ((S_total & 1) == 0) {
#pragma unroll
          for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              float k0, k1;
              tell_keep2(p.seed, salt_eff, base + (f * 32 + (r & 3) + 8 * (r >> 2)), p.thr, p.inv_keep, k0, k1);
              st[f][r] *= k0;
              st[f][r + 1] *= k1;
            }
        } else {
#pragma unroll
          for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              st[f][r] *= tell_keep(p.seed, salt_eff, base + (f * 32 + (r & 3) + 8 * (r >> 2)), p.thr, p.inv_keep);
        }
      }
      ls += __shfl_xor(ls, 32, 64);
      l_run = l_run * alpha + ls;
      m_run = m_new;
      if (__any(alpha != 1.f)) {                                       // wave-uniform: the running max moved
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[f][r] *= alpha;
      }
      // P[q][key]: registers r = 4g..4g+3 are 4 consecutive keys -> one 8-byte (bf16) store
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int k0 = f * 32 + 8 * g + 4 * (lane >> 5);
          if constexpr (sizeof(T) == 2) {
            uint2 w;
            w.x = pack_bf16x2(st[f][4 * g], st[f][4 * g + 1]);
            w.y = pack_bf16x2(st[f][4 * g + 2], st[f][4 * g + 3]);
            *reinterpret_cast<uint2*>(myP + qi * DS + k0) = w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) myP[qi * DS + k0 + e] = st[f][4 * g + e];
          }
        }
    }
    __syncthreads();                                  // P visible to the whole wave (and block-uniform)
    if (active) {
#pragma unroll
      for (int f = 0; f < 2; ++f) o[f] = mma32<T, KT>(o[f], Vt + f * 32 * DS, DS, myP, DS, lane);  // O^T[d][q]
    }
  }
#undef LOAD_TILE
#undef TILE_ROW_OK
  if (!active) return;
  const int t = q0 + qi;
  if (t >= p.Tq) return;
  const float inv_l = l_run > 0.f ? 1.f / l_run : 0.f;
  T* og = static_cast<T*>(p.out) + t * p.o_st + b * p.o_sb + (long)h * D;
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int g = 0; g < 4; ++g) {                     // 4 consecutive d per register group
      const int d0 = f * 32 + 8 * g + 4 * (lane >> 5);
#pragma unroll
      for (int e = 0; e < 4; ++e) Elem<T>::st(og + d0 + e, o[f][4 * g + e] * inv_l);
    }
  if (lane < 32 && p.lse) p.lse[(long)bh * p.Tq + t] = l_run > 0.f ? m_run + __logf(l_run) : INFINITY;
}

// ------------------------------------------------------------------ long sequences, bf16: probabilities stay in registers
// Same tiling as attn_fwd_tile64_kernel (4 waves x 32 queries share 64-key K/V tiles) with three changes:
//  * Q fragments live in registers for the whole kernel (they are the B operand of S^T = K Q^T);
//  * the S^T accumulator layout already IS the B operand layout of O^T = V^T P^T: lane (q, hh) holds, for every
//    16-key MFMA step, 8 keys {4hh..4hh+3, 8+4hh..8+4hh+3} - any key order works for a reduction as long as the
//    A operand uses the same one - so exp(S) is packed to bf16 in place and never visits LDS;
//  * V is staged row-major exactly as loaded and its transposed fragments come from ds_read_b64_tr_b16 (two
//    4-key reads per fragment = the same key sets), so the transposing LDS store is gone as well.
// K/V tiles are double-buffered: one barrier per 64 keys instead of three.
typedef __attribute__((ext_vector_type(4))) short attn_s16x4;
typedef __attribute__((address_space(3))) attn_s16x4* attn_lds_s16x4_ptr;
__device__ __forceinline__ bf16x8 attn_tr_frag(const uint16_t* p0, const uint16_t* p1) {
  const attn_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((attn_lds_s16x4_ptr)p0);
  const attn_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((attn_lds_s16x4_ptr)p1);
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

template <bool DROP, int OCC = 2>
__global__ __launch_bounds__(256, OCC) void attn_fwd_reg_kernel(AttnArgs p) {
  const uint32_t salt_eff = tell_step_salt(p.salt, p.step);   // hoisted: one scalar load per kernel
  using T = uint16_t;
  constexpr int D = 64, KT = 64, NW = 4;
  constexpr int KS = 72;                             // K tile row stride (144 B): conflict-free ds_read_b128
  constexpr int VS = 96;                             // V tile row stride (192 B): 64-byte skew for the tr reads
  __shared__ __attribute__((aligned(16))) T Ks[2][KT * KS];
  __shared__ __attribute__((aligned(16))) T Vs[2][KT * VS];
  __shared__ __attribute__((aligned(16))) float key_bias[2][KT];
  __shared__ int tile_masked[2];                     // any key of the staged tile masked / past the end?
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  typedef __attribute__((ext_vector_type(4))) float f32x4;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, qi = lane & 31;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
  const int QB = (p.Tq + 31) / 32;
  const int qb = blockIdx.x * NW + wave;
  const bool active = qb < QB;
  const int q0 = qb * 32, t = q0 + qi;
  const int S_total = p.S + p.has_bias + p.has_zero;
  const int nkt = (S_total + KT - 1) / KT;
  const T* qg = static_cast<const T*>(p.q);
  const T* kg = static_cast<const T*>(p.k);
  const T* vg = static_cast<const T*>(p.v);
  const T* bk = static_cast<const T*>(p.bias_k);
  const T* bv = static_cast<const T*>(p.bias_v);
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  bf16x8 qf[4];                                      // Q[t][16 ks + 8 hh .. + 7]
  {
    const bool ok = active && t < p.Tq;
    const T* qp = ok ? qg + t * p.q_st + b * p.q_sb + (long)h * D : qg;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 v = *reinterpret_cast<const u32x4*>(qp + (ok ? 16 * ks + 8 * hh : 0));
      if (!ok) v = zero4;
      qf[ks] = __builtin_bit_cast(bf16x8, v);
    }
  }

  u32x4 rk[2], rv[2];                                // chunk c = tid + 256 i: key c >> 3, 16-byte column c & 7
#define RLOAD_TILE(KTI)                                                                         \
  {                                                                                             \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                             \
      const int c = tid + 256 * i, s = (KTI) * KT + (c >> 3), ch = c & 7;                       \
      const bool real = s < p.S, isb = (s == p.S) && p.has_bias;                                \
      const T* kp = real ? kg + s * p.k_ss + b * p.k_sb + (long)h * D : (isb ? bk + (long)h * D : qg); \
      const T* vp = real ? vg + s * p.v_ss + b * p.v_sb + (long)h * D : (isb ? bv + (long)h * D : qg); \
      const int off = (real || isb) ? ch * 8 : 0;                                               \
      rk[i] = *reinterpret_cast<const u32x4*>(kp + off);   /* unconditional; zeroed at the LDS store */ \
      rv[i] = *reinterpret_cast<const u32x4*>(vp + off);                                        \
    }                                                                                           \
  }
#define RSTORE_TILE(KTI, BUF)                                                                   \
  {                                                                                             \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                             \
      const int c = tid + 256 * i, key = c >> 3, ch = c & 7, s = (KTI) * KT + key;              \
      const bool ok = s < p.S || (s == p.S && p.has_bias);       /* virtual zero row / past the end -> zeros */ \
      *reinterpret_cast<u32x4*>(&Ks[BUF][key * KS + ch * 8]) = ok ? rk[i] : zero4;              \
      *reinterpret_cast<u32x4*>(&Vs[BUF][key * VS + ch * 8]) = ok ? rv[i] : zero4;              \
    }                                                                                           \
    if (tid < KT) {                                                                             \
      const int s = (KTI) * KT + tid;                                                           \
      bool ok = s < S_total;                                                                    \
      if (ok && s < p.S && p.mask) ok = p.mask[(long)b * p.S + s] == 0;                         \
      key_bias[BUF][tid] = ok ? 0.f : -INFINITY;                                                \
      const bool any_masked = __ballot(!ok) != 0;             /* tid < 64 is exactly wave 0 */   \
      if (tid == 0) tile_masked[BUF] = any_masked;                                              \
    }                                                                                           \
  }
  RLOAD_TILE(0)
  RSTORE_TILE(0, 0)
  __syncthreads();

  f32x16 o[2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[f][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  // per-lane LDS origins: K fragment row (lane&31), k chunk 8 hh; V^T fragment via tr read: key row 4 hh +
  // ((lane&15)>>2), d column 16 ((lane>>4)&1) + 4 (lane&3)
  const int k_off = qi * KS + 8 * hh;
  const int v_off = (4 * hh + ((lane & 15) >> 2)) * VS + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1, s0 = kt * KT;
    if (kt + 1 < nkt) RLOAD_TILE(kt + 1)              // streams in under the MFMAs below
    if (active) {
      const T* Kb = &Ks[buf][k_off];
      const T* Vb = &Vs[buf][v_off];
      f32x16 st[2];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[f][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kb + f * 32 * KS + 16 * ks);
          st[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st[f], 0, 0, 0);   // S^T[key][q]
        }
      }
      // block-uniform: padding sits at the end of a sequence, so most tiles of a padded batch have nothing to add
      if (__builtin_amdgcn_readfirstlane(tile_masked[buf])) {
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 kb4 = *reinterpret_cast<const f32x4*>(&key_bias[buf][f * 32 + 8 * g + 4 * hh]);
#pragma unroll
            for (int e = 0; e < 4; ++e) st[f][4 * g + e] += kb4[e];
          }
      }
      float mx = -INFINITY;
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[f][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;        // all keys masked so far: exp(-inf - 0) = 0
      const float alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_new);
      typedef __attribute__((ext_vector_type(2))) float f32x2;
      constexpr float LOG2E = 1.4426950408889634f;
      const float neg_m = -m_safe * LOG2E;
      f32x2 ls2 = {0.f, 0.f};                          // pairs: the scale/shift and the row sum are packed fp32 ops
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 s2 = {st[f][r], st[f][r + 1]};
          const f32x2 e2 = s2 * LOG2E + neg_m;         // exp(s - m) = exp2(s log2e - m log2e)
          const f32x2 pv = {__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1])};
          ls2 += pv;
          st[f][r] = pv[0];
          st[f][r + 1] = pv[1];
        }
      float ls = ls2[0] + ls2[1];
      // Dropout only SELECTS here (kept probability or 0); the 1/(1-p) factor is applied once to the output row.
      if constexpr (DROP) {
        const uint64_t base = ((uint64_t)bh * p.Tq + t) * S_total + s0 + 4 * hh;
        if ((S_total & 3) == 0 && tell_keep_row_ok(base >> 2, 16)) {
          // the lane's keys come in aligned quads: 4 hh + 8 m + {0..3}, m = 0..7 -> quad offsets 0, 2, 4, ... 14:
          // one running hash input, += two quad strides per step, four decisions per hash
          const TellKeepRow row = tell_keep_row(p.seed, salt_eff, base >> 2);    // base is a multiple of 4 here
          uint32_t x = row.x0;
#pragma unroll
          for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
              bool k0, k1, k2, k3;
              tell_keep4_bits(x, row.y, p.thr, k0, k1, k2, k3);   // elements base + f*32 + 8*(r>>2) + {0,1,2,3}
              st[f][r] = k0 ? st[f][r] : 0.f;
              st[f][r + 1] = k1 ? st[f][r + 1] : 0.f;
              st[f][r + 2] = k2 ? st[f][r + 2] : 0.f;
              st[f][r + 3] = k3 ? st[f][r + 3] : 0.f;
              x += 2u * TELL_QUAD_STRIDE;
            }
        } else if ((S_total & 1) == 0) {
#pragma unroll
          for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              float k0, k1;
              tell_keep2(p.seed, salt_eff, base + (f * 32 + (r & 3) + 8 * (r >> 2)), p.thr, p.inv_keep, k0, k1);
              st[f][r] = k0 != 0.f ? st[f][r] : 0.f;
              st[f][r + 1] = k1 != 0.f ? st[f][r + 1] : 0.f;
            }
        } else {
#pragma unroll
          for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              st[f][r] = tell_keep(p.seed, salt_eff, base + (f * 32 + (r & 3) + 8 * (r >> 2)), p.thr, p.inv_keep) != 0.f
                             ? st[f][r] : 0.f;
        }
      }
      ls += __shfl_xor(ls, 32, 64);
      l_run = l_run * alpha + ls;
      m_run = m_new;
      if (__any(alpha != 1.f)) {                                       // wave-uniform: the running max moved
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[f][r] *= alpha;
      }
      // O^T[d][q] += V^T[d][keys] P^T[keys][q], 16 keys per MFMA: registers 8j..8j+7 of sub-tile f
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          typedef __attribute__((ext_vector_type(4))) unsigned int pk4;
          const pk4 w = {pack_bf16x2(st[f][8 * j], st[f][8 * j + 1]), pack_bf16x2(st[f][8 * j + 2], st[f][8 * j + 3]),
                         pack_bf16x2(st[f][8 * j + 4], st[f][8 * j + 5]), pack_bf16x2(st[f][8 * j + 6], st[f][8 * j + 7])};
          const bf16x8 pf = __builtin_bit_cast(bf16x8, w);
          const T* vrow = Vb + (f * 32 + j * 16) * VS;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const bf16x8 vf = attn_tr_frag(vrow + dt * 32, vrow + 8 * VS + dt * 32);
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dt], 0, 0, 0);
          }
        }
    }
    if (kt + 1 < nkt) RSTORE_TILE(kt + 1, buf ^ 1)     // buffer buf^1 was last read before the previous barrier
    __syncthreads();
  }
#undef RLOAD_TILE
#undef RSTORE_TILE
  if (!active || t >= p.Tq) return;
  const float inv_l = (l_run > 0.f ? 1.f / l_run : 0.f) * (DROP ? p.inv_keep : 1.f);
  T* og = static_cast<T*>(p.out) + t * p.o_st + b * p.o_sb + (long)h * D;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {                     // 4 consecutive d per register group -> one 8-byte store
      const int d0 = dt * 32 + 8 * g + 4 * hh;
      uint2 w;
      w.x = pack_bf16x2(o[dt][4 * g] * inv_l, o[dt][4 * g + 1] * inv_l);
      w.y = pack_bf16x2(o[dt][4 * g + 2] * inv_l, o[dt][4 * g + 3] * inv_l);
      *reinterpret_cast<uint2*>(og + d0) = w;
    }
  if (lane < 32 && p.lse) p.lse[(long)bh * p.Tq + t] = l_run > 0.f ? m_run + __logf(l_run) : INFINITY;
}

// ------------------------------------------------------------------ self-attention of the article encoder (bf16)
// fairseq's RoBERTa self-attention as transformer_faces_objects.py:352-353 runs it (T = S = 512, 16 heads of 64, key
// padding mask, attention dropout in train mode) is the second largest kernel of the step after the encoder GEMMs and it
// is VALU-bound: the softmax + dropout arithmetic of a 64-key tile takes ~4x the cycles of its 16 MFMAs.  Same tiling and
// register-resident probabilities as attn_fwd_reg_kernel, specialised to what this caller guarantees - no bias_k / zero
// rows, S a multiple of 64 (every tile full) - and rebuilt around the VALU budget:
//   * tile loads are `uniform base + per-lane 32-bit offset` (the per-tile part of every address lives in scalar
//     registers: zero vector instructions per tile for addressing; the generic kernel spent ~80 on row classification
//     and 64-bit pointer arithmetic), LDS store addresses are immediates of a 2x unrolled loop;
//   * the running maximum is only moved when some row's tile maximum exceeds it by more than 8 (wave-uniform test):
//     after the first tiles the 32-multiply rescale of the output accumulators and the exp of the correction factor
//     disappear; probabilities are then bounded by e^8 instead of 1, which bf16 / fp32 accumulation carry without loss
//     of relative precision, and the log-sum-exp is exact either way;
//   * three-input maxima; three waves per SIMD (168 registers, launch bounds) instead of two.
// Dropout masks are the generic kernel's (same element indices, same hash): bit-identical selections.
#define ATTN_SELF_THR 8.0f
// DMA = true: the K / V tiles go global -> LDS by `global_load_lds_dwordx4` (no staging registers, no ds_write, the
// issuing wave does not wait): rows are 128 bytes back to back (the DMA writes lane-linear images) and bank conflicts are
// avoided by an XOR swizzle applied on the SOURCE side - the lane that fills 16-byte position c of row r fetches chunk
// c ^ (r & 7) - which the fragment reads undo; the tile is complete after `s_waitcnt vmcnt(0)` + the tile's barrier.
// MEASURED (opt-in, TELL_ATTN_DMA=1): bit-identical results, 90.5 us against 88.2 with dropout, 66.9 against 64.8
// without - the 14 us the no-staging ablation removes are the tile loads themselves, not the register round trip.
typedef __attribute__((address_space(3))) void* attn_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* attn_glb_ptr_t;
template <bool DROP, int ABL = 0, bool DMA = false>
__global__ __launch_bounds__(256, 3) void attn_self_fwd_kernel(AttnArgs p) {
  const uint32_t salt_eff = tell_step_salt(p.salt, p.step);
  using T = uint16_t;
  constexpr int D = 64, KT = 64, NW = 4;
  constexpr int KS = DMA ? 64 : 72, VS = DMA ? 64 : 96;      // register-staged: padded rows as attn_fwd_reg_kernel
  __shared__ __attribute__((aligned(16))) T Ks[2][KT * KS];
  __shared__ __attribute__((aligned(16))) T Vs[2][KT * VS];
  __shared__ __attribute__((aligned(16))) float key_bias[2][KT];
  __shared__ int tile_masked[2];
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  typedef __attribute__((ext_vector_type(4))) float f32x4;
  typedef __attribute__((ext_vector_type(2))) float f32x2;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, qi = lane & 31;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
  const int QB = (p.Tq + 31) / 32;
  const int qb = blockIdx.x * NW + wave;
  const bool active = qb < QB;
  const int t = qb * 32 + qi;
  const int nkt = p.S / KT;
  const T* qg = static_cast<const T*>(p.q);
  const T* kb = static_cast<const T*>(p.k) + (long)b * p.k_sb + (long)h * D;      // uniform
  const T* vb = static_cast<const T*>(p.v) + (long)b * p.v_sb + (long)h * D;
  const uint8_t* mrow = p.mask ? p.mask + (long)b * p.S : nullptr;

  bf16x8 qf[4];
  {
    const bool ok = active && t < p.Tq;
    const T* qp = ok ? qg + (long)t * p.q_st + (long)b * p.q_sb + (long)h * D : qg;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 v = *reinterpret_cast<const u32x4*>(qp + (ok ? 16 * ks + 8 * hh : 0));
      if (!ok) v = zero4;
      qf[ks] = __builtin_bit_cast(bf16x8, v);
    }
  }
  // staging: chunk c = tid + 256 i covers key c >> 3, 16-byte column c & 7
  // (DMA: instruction 2 wave + i fills rows 8 (2 wave + i) .. + 7; lane l fills position l & 7 of row .. + (l >> 3))
  unsigned koff[2], voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 256 * i;
    if constexpr (DMA) {
      const int row = 8 * (2 * wave + i) + (lane >> 3), ch = (lane & 7) ^ (row & 7);
      koff[i] = (unsigned)(row * (int)p.k_ss + ch * 8);
      voff[i] = (unsigned)(row * (int)p.v_ss + ch * 8);
    } else {
      koff[i] = (unsigned)((c >> 3) * (int)p.k_ss + (c & 7) * 8);
      voff[i] = (unsigned)((c >> 3) * (int)p.v_ss + (c & 7) * 8);
    }
  }
  const int kw0 = (tid >> 3) * KS + (tid & 7) * 8, vw0 = (tid >> 3) * VS + (tid & 7) * 8;   // chunk i: + 32 * stride
  u32x4 rk[2], rv[2];
  auto rload = [&](int kt) __attribute__((always_inline)) {
    const T* kt_b = kb + (long)kt * KT * p.k_ss;      // scalar
    const T* vt_b = vb + (long)kt * KT * p.v_ss;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      rk[i] = *reinterpret_cast<const u32x4*>(kt_b + koff[i]);
      rv[i] = *reinterpret_cast<const u32x4*>(vt_b + voff[i]);
    }
  };
  auto dma = [&](int kt, auto buf_c) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_c)::value;
    const T* kt_b = kb + (long)kt * KT * p.k_ss;      // scalar
    const T* vt_b = vb + (long)kt * KT * p.v_ss;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_global_load_lds((attn_glb_ptr_t)(kt_b + koff[i]), (attn_lds_ptr_t)&Ks[BUF][8 * (2 * wave + i) * KS], 16, 0, 0);
      __builtin_amdgcn_global_load_lds((attn_glb_ptr_t)(vt_b + voff[i]), (attn_lds_ptr_t)&Vs[BUF][8 * (2 * wave + i) * VS], 16, 0, 0);
    }
  };
  auto rstore = [&](int kt, auto buf_c) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_c)::value;
    if constexpr (!DMA) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        *reinterpret_cast<u32x4*>(&Ks[BUF][kw0 + i * 32 * KS]) = rk[i];
        *reinterpret_cast<u32x4*>(&Vs[BUF][vw0 + i * 32 * VS]) = rv[i];
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the next tile have landed
    }
    if (tid < KT) {                                   // exactly wave 0
      const bool ok = mrow ? mrow[kt * KT + tid] == 0 : true;
      key_bias[BUF][tid] = ok ? 0.f : -INFINITY;
      const bool any_masked = __ballot(!ok) != 0;
      if (tid == 0) tile_masked[BUF] = any_masked;
    }
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  if constexpr (DMA) dma(0, B0{}); else rload(0);
  rstore(0, B0{});
  __syncthreads();

  f32x16 o[2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[f][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int k_off = qi * KS + 8 * hh;
  const int v_off = (4 * hh + ((lane & 15) >> 2)) * VS + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  // DMA image: element (row, col) lives at row * 64 + (((col >> 3) ^ (row & 7)) << 3) + (col & 7)
  int k_sw[4], v_sw[2][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) k_sw[ks] = qi * 64 + (((2 * ks + hh) ^ (qi & 7)) << 3);        // rows qi + 32 f
#pragma unroll
  for (int half = 0; half < 2; ++half)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int r = 8 * half + 4 * hh + ((lane & 15) >> 2), c = dt * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
      v_sw[half][dt] = r * 64 + (((c >> 3) ^ (r & 7)) << 3) + (c & 7);                            // rows r + 32 f + 16 j
    }
  const uint64_t row_base = ((uint64_t)bh * p.Tq + t) * (uint64_t)p.S + 4 * hh;
  constexpr float LOG2E = 1.4426950408889634f;

  auto tile = [&](int kt, auto buf_c) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_c)::value;
    if ((ABL & 2) == 0 && kt + 1 < nkt) {                               // streams in under the MFMAs below
      if constexpr (DMA) {
        if constexpr (BUF == 0) dma(kt + 1, B1{}); else dma(kt + 1, B0{});
      } else {
        rload(kt + 1);
      }
    }
    if (active) {
      const T* Kb = &Ks[BUF][k_off];
      const T* Vb = &Vs[BUF][v_off];
      f32x16 st[2];
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[f][r] = 0.f;
      // the two 32-key halves alternate: a 32x32x16 MFMA issues every 32 cycles but its accumulator is only back after
      // 64 - one chain of four dependent MFMAs per half, interleaved, keeps the pipe busy
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const bf16x8 kf = DMA ? *reinterpret_cast<const bf16x8*>(&Ks[BUF][f * 32 * 64 + k_sw[ks]])
                                : *reinterpret_cast<const bf16x8*>(Kb + f * 32 * KS + 16 * ks);
          st[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st[f], 0, 0, 0);   // S^T[key][q]
        }
      if (__builtin_amdgcn_readfirstlane(tile_masked[BUF])) {
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 kb4 = *reinterpret_cast<const f32x4*>(&key_bias[BUF][f * 32 + 8 * g + 4 * hh]);
#pragma unroll
            for (int e = 0; e < 4; ++e) st[f][4 * g + e] += kb4[e];
          }
      }
      float mx = fmaxf(st[0][0], st[0][1]);
#pragma unroll
      for (int r = 2; r < 16; r += 2) mx = fmaxf(fmaxf(mx, st[0][r]), st[0][r + 1]);      // v_max3_f32
#pragma unroll
      for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, st[1][r]), st[1][r + 1]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      // move the running maximum only when a row outgrows it by more than the threshold (NaN compares - an all-masked
      // history, -inf - -inf - count as "move": the update below is the exact one)
      const float m_cand = fmaxf(m_run, mx);
      if (__any(!(m_cand - m_run <= ATTN_SELF_THR))) {
        const float alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_cand);
        m_run = m_cand;
        l_run *= alpha;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[f][r] *= alpha;
      }
      const float neg_m = (m_run == -INFINITY) ? 0.f : -m_run * LOG2E;
      f32x2 ls2 = {0.f, 0.f};
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 s2 = {st[f][r], st[f][r + 1]};
          const f32x2 e2 = s2 * LOG2E + neg_m;
          const f32x2 pv = {__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1])};
          ls2 += pv;
          st[f][r] = pv[0];
          st[f][r + 1] = pv[1];
        }
      float ls = ls2[0] + ls2[1];
      if constexpr (DROP) {
        const uint64_t base = row_base + (uint64_t)kt * KT;
        if (tell_keep_row_ok(base >> 2, 16)) {
          const TellKeepRow row = tell_keep_row(p.seed, salt_eff, base >> 2);    // S % 64 == 0: base is a multiple of 4
          uint32_t x = row.x0;
#pragma unroll
          for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
              bool k0, k1, k2, k3;
              tell_keep4_bits(x, row.y, p.thr, k0, k1, k2, k3);
              st[f][r] = k0 ? st[f][r] : 0.f;
              st[f][r + 1] = k1 ? st[f][r + 1] : 0.f;
              st[f][r + 2] = k2 ? st[f][r + 2] : 0.f;
              st[f][r + 3] = k3 ? st[f][r + 3] : 0.f;
              x += 2u * TELL_QUAD_STRIDE;
            }
        } else {
#pragma unroll
          for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              float k0, k1;
              tell_keep2(p.seed, salt_eff, base + (f * 32 + (r & 3) + 8 * (r >> 2)), p.thr, p.inv_keep, k0, k1);
              st[f][r] = k0 != 0.f ? st[f][r] : 0.f;
              st[f][r + 1] = k1 != 0.f ? st[f][r + 1] : 0.f;
            }
        }
      }
      ls += __shfl_xor(ls, 32, 64);
      l_run += ls;
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          typedef __attribute__((ext_vector_type(4))) unsigned int pk4;
          const pk4 w = {pack_bf16x2(st[f][8 * j], st[f][8 * j + 1]), pack_bf16x2(st[f][8 * j + 2], st[f][8 * j + 3]),
                         pack_bf16x2(st[f][8 * j + 4], st[f][8 * j + 5]), pack_bf16x2(st[f][8 * j + 6], st[f][8 * j + 7])};
          const bf16x8 pf = __builtin_bit_cast(bf16x8, w);
          const T* vrow = Vb + (f * 32 + j * 16) * VS;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const T* vt = &Vs[BUF][(f * 32 + j * 16) * 64];
            const bf16x8 vf = DMA ? attn_tr_frag(vt + v_sw[0][dt], vt + v_sw[1][dt])
                                  : attn_tr_frag(vrow + dt * 32, vrow + 8 * VS + dt * 32);
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dt], 0, 0, 0);
          }
        }
    }
    if ((ABL & 2) == 0 && kt + 1 < nkt) {
      if constexpr (BUF == 0) rstore(kt + 1, B1{}); else rstore(kt + 1, B0{});
    }
    if ((ABL & 1) == 0) __syncthreads();
  };
  for (int kt = 0; kt < nkt; kt += 2) {
    tile(kt, B0{});
    if (kt + 1 < nkt) tile(kt + 1, B1{});
  }
  if (!active || t >= p.Tq) return;
  const float inv_l = (l_run > 0.f ? 1.f / l_run : 0.f) * (DROP ? p.inv_keep : 1.f);
  T* og = static_cast<T*>(p.out) + (long)t * p.o_st + (long)b * p.o_sb + (long)h * D;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d0 = dt * 32 + 8 * g + 4 * hh;
      uint2 w;
      w.x = pack_bf16x2(o[dt][4 * g] * inv_l, o[dt][4 * g + 1] * inv_l);
      w.y = pack_bf16x2(o[dt][4 * g + 2] * inv_l, o[dt][4 * g + 3] * inv_l);
      *reinterpret_cast<uint2*>(og + d0) = w;
    }
  if (lane < 32 && p.lse) p.lse[(long)bh * p.Tq + t] = l_run > 0.f ? m_run + __logf(l_run) : INFINITY;
}

// MFMA operand fragments out of a K-MAJOR tile (tile[k][m], row stride ld, bf16) by gfx950's LDS transpose read: the lane
// of row m = m0 + (lane & 31) gets k = ks + 8 (lane >> 5) .. + 7 - the same k order as a row-major operand read by
// mma32, so the two can meet in one MFMA.  (csrc/gemm.hip gemm_tx_body has the addressing rule.)
__device__ __forceinline__ bf16x8 attn_km_frag(const uint16_t* tile, int ld, int m0, int ks, int lane) {
  const uint16_t* p = tile + (ks + 8 * (lane >> 5) + ((lane & 15) >> 2)) * ld + m0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  return attn_tr_frag(p, p + 4 * ld);
}
// acc[m][n] += sum_k a_km[k][m0 + m] * b[n][k]  (k < 32)
__device__ __forceinline__ f32x16 mma32_kmA(f32x16 acc, const uint16_t* a_km, int ald, int m0, const uint16_t* b, int bs, int lane) {
#pragma unroll
  for (int ks = 0; ks < 32; ks += 16) {
    const bf16x8 av = attn_km_frag(a_km, ald, m0, ks, lane);
    const bf16x8 bv = *reinterpret_cast<const bf16x8*>(b + (lane & 31) * bs + ks + ((lane >> 5) << 3));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
  }
  return acc;
}
// acc[m][n] += sum_k a[m][k] * b_km[k][n0 + n]  (k < 32)
__device__ __forceinline__ f32x16 mma32_kmB(f32x16 acc, const uint16_t* a, int as, const uint16_t* b_km, int bld, int n0, int lane) {
#pragma unroll
  for (int ks = 0; ks < 32; ks += 16) {
    const bf16x8 av = *reinterpret_cast<const bf16x8*>(a + (lane & 31) * as + ks + ((lane >> 5) << 3));
    const bf16x8 bv = attn_km_frag(b_km, bld, n0, ks, lane);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
  }
  return acc;
}

// ------------------------------------------------------------------ backward
// One workgroup per (b,h); its NW waves split the key tiles.  Query blocks of 32 are
// the outer loop (Q-side tiles shared by the waves); dQ is reduced across waves in LDS;
// each key tile is owned by exactly one wave, so dK/dV need no atomics.
// bf16 with 64-wide heads (round 4): the transposed copies Qt / dOt / Kt - written two bytes at a time, 30 KB of the 108 KB
// a four-wave workgroup held - are gone: dV^T, dK^T and dQ take their K-major operands out of the ROW-major Q / dO / K
// tiles by ds_read_b64_tr_b16 (attn_km_frag).  77 KB of LDS and <= 256 registers: TWO workgroups per CU, so the 512
// (b, h) workgroups of the article attention are resident at once and each SIMD has a second wave to run while one waits
// (this kernel is a chain of LDS round trips and barriers: 102 us for 134 MB at one wave per SIMD).
template <typename T, int D, int NW>
__global__ __launch_bounds__(64 * NW, (sizeof(T) == 2 && D == 64 && NW == 4) ? 2 : 1) void attn_bwd_kernel(AttnArgs p) {
  const uint32_t salt_eff = tell_step_salt(p.salt, p.step);   // hoisted: one scalar load per kernel
  using C_ = ACfg<T, D>;
  constexpr int DS = C_::DS, SS = C_::SS, DP = C_::DP, DF = C_::DF;
  constexpr bool TR = sizeof(T) == 2 && D == 64;                // K-major operands by transpose reads: no Qt / dOt / Kt
  constexpr int SHARED_ELEMS = 2 * 32 * DS + (TR ? 0 : 2 * DP * SS);       // Qs, dOs, (Qt, dOt)
  constexpr int WAVE_ELEMS = 2 * 32 * DS + (TR ? 0 : DP * SS) + 3 * 32 * SS;  // Ks, Vs, (Kt), PdT, dST, dS
  static_assert(WAVE_ELEMS * sizeof(T) >= 16 * DF * 64 * sizeof(float), "dQ combine buffer must fit");
  __shared__ __attribute__((aligned(16))) T smem[SHARED_ELEMS + NW * WAVE_ELEMS];
  __shared__ float lse_s[32], delta_s[32];
  __shared__ float key_ok[NW][32];                   // 1 / 0 per key of the wave's current tile

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
  const int QB = (p.Tq + 31) / 32;
  const int S_total = p.S + p.has_bias + p.has_zero;
  const int nkt = (S_total + 31) / 32;
  const int iters = (nkt + NW - 1) / NW;
  const int qi = lane & 31;

  T* Qs = smem;
  T* dOs = Qs + 32 * DS;
  T* Qt = dOs + 32 * DS;                             // (TR: no transposed copies - the names alias the wave areas, unused)
  T* dOt = Qt + (TR ? 0 : DP * SS);
  T* Ks = smem + SHARED_ELEMS + wave * WAVE_ELEMS;
  T* Vs = Ks + 32 * DS;
  T* Kt = Vs + 32 * DS;
  T* PdT = Kt + (TR ? 0 : DP * SS);
  T* dST = PdT + 32 * SS;
  T* dSs = dST + 32 * SS;
  const T* qg = static_cast<const T*>(p.q);
  const T* kg = static_cast<const T*>(p.k);
  const T* vg = static_cast<const T*>(p.v);
  const T* og = static_cast<const T*>(p.o);
  const T* dog = static_cast<const T*>(p.dout);
  const T* bk = static_cast<const T*>(p.bias_k);
  const T* bv = static_cast<const T*>(p.bias_v);

  if (DP > D) {
    zero_lds(Qt, 2 * DP * SS, tid, 64 * NW);
    zero_lds(Kt, DP * SS, lane, 64);
  }

  for (int qb = 0; qb < QB; ++qb) {
    const int q0 = qb * 32;
    __syncthreads();                              // all waves done with the previous q block
    // Prologue in ONE global round trip: this wave's first K / V tile, then per thread one 16-byte chunk each of Q, dO
    // and O (row c / CPR, chunk c % CPR) - Q and dO go to LDS (row-major and transposed), delta[q] = <dO[q], O[q]>
    // folds over the CPR lanes of a row.  (Staged one after the other, with delta as a loop of per-row wave
    // reductions, the prologue was ~12 dependent round trips: most of the 29 us a 4-key context cost.)
    Tile32Regs<T, D> rk, rv;
    auto fetch = [&](int kt_) __attribute__((always_inline)) {
      const int s0_ = kt_ * 32;
      load32<T, D>(rk, lane, [&](int row) -> const T* {
        int s = s0_ + row;
        if (s < p.S) return kg + s * p.k_ss + b * p.k_sb + (long)h * D;
        if (s == p.S && p.has_bias) return bk + (long)h * D;
        return nullptr;
      }, qg);
      load32<T, D>(rv, lane, [&](int row) -> const T* {
        int s = s0_ + row;
        if (s < p.S) return vg + s * p.v_ss + b * p.v_sb + (long)h * D;
        if (s == p.S && p.has_bias) return bv + (long)h * D;
        return nullptr;
      }, qg);
    };
    if (wave < nkt) fetch(wave);
    {
      constexpr int VEC = Elem<T>::VEC, CPR = D / VEC;
      static_assert((32 * CPR) % 64 == 0 && CPR <= 16 && (CPR & (CPR - 1)) == 0, "whole waves, rows inside a wave");
      for (int c = tid; c < 32 * CPR; c += 64 * NW) {
        const int row = c / CPR, ch = c % CPR, t = q0 + row;
        const bool ok = t < p.Tq;
        const long qo = ok ? t * p.q_st + b * p.q_sb + (long)h * D + ch * VEC : 0;
        const long oo = ok ? t * p.o_st + b * p.o_sb + (long)h * D + ch * VEC : 0;
        uint4 vq = *reinterpret_cast<const uint4*>(qg + qo);
        uint4 vdo = *reinterpret_cast<const uint4*>((ok ? dog : qg) + oo);
        uint4 vo = *reinterpret_cast<const uint4*>((ok ? og : qg) + oo);
        const float lse_row = p.lse[(long)bh * p.Tq + (ok ? t : 0)];
        if (!ok) vq = vdo = vo = make_uint4(0, 0, 0, 0);
        float fdo[VEC], fo[VEC];
        unpack16(vdo, fdo, (const T*)nullptr);
        unpack16(vo, fo, (const T*)nullptr);
        float part = 0.f;
#pragma unroll
        for (int k = 0; k < VEC; ++k) part += fdo[k] * fo[k];
#pragma unroll
        for (int o = CPR / 2; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        if (ch == 0) {
          delta_s[row] = part;
          lse_s[row] = ok ? lse_row : INFINITY;
        }
        if constexpr (sizeof(T) == 2) {
          *reinterpret_cast<uint4*>(Qs + row * DS + ch * VEC) = vq;
          *reinterpret_cast<uint4*>(dOs + row * DS + ch * VEC) = vdo;
        } else {
          float* q1 = reinterpret_cast<float*>(Qs) + row * DS + ch * VEC;
          float* q2 = reinterpret_cast<float*>(dOs) + row * DS + ch * VEC;
          q1[0] = __uint_as_float(vq.x); q1[1] = __uint_as_float(vq.y); q1[2] = __uint_as_float(vq.z); q1[3] = __uint_as_float(vq.w);
          q2[0] = __uint_as_float(vdo.x); q2[1] = __uint_as_float(vdo.y); q2[2] = __uint_as_float(vdo.z); q2[3] = __uint_as_float(vdo.w);
        }
        if constexpr (!TR) {
          const T* eq = reinterpret_cast<const T*>(&vq);
          const T* ed = reinterpret_cast<const T*>(&vdo);
#pragma unroll
          for (int k = 0; k < VEC; ++k) {
            Qt[(ch * VEC + k) * SS + row] = eq[k];
            dOt[(ch * VEC + k) * SS + row] = ed[k];
          }
        }
      }
    }
    f32x16 dq[DF];
#pragma unroll
    for (int f = 0; f < DF; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[f][r] = 0.f;

    // (key tiles prefetched one iteration ahead, as in the forward kernel)
    for (int it = 0; it < iters; ++it) {
      const int kt = it * NW + wave;
      const bool tv = kt < nkt;
      const int s0 = kt * 32;
      __syncthreads();
      if (tv) {
        store32<T, D>(rk, Ks, DS, TR ? (T*)nullptr : Kt, SS, lane);
        store32<T, D>(rv, Vs, DS, (T*)nullptr, 0, lane);
        if (kt + NW < nkt) fetch(kt + NW);
        if (lane < 32) {
          const int s = s0 + lane;
          bool ok = s < S_total;
          if (ok && s < p.S && p.mask) ok = p.mask[(long)b * p.S + s] == 0;
          key_ok[wave][lane] = ok ? 1.f : 0.f;
        }
      }
      __syncthreads();
      if (tv) {
        f32x16 st, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
        st = mma32<T, D>(st, Ks, DS, Qs, DS, lane);           // S^T[key][q]
        dp = mma32<T, D>(dp, Vs, DS, dOs, DS, lane);          // dPd^T[key][q] = V.dO^T
        const float lse = lse_s[qi], delta = delta_s[qi];
        const int t = q0 + qi;
        // dropout factors of the lane's 16 keys: they come in aligned pairs (4 hh + 8 (r >> 2) + {0,1}, {2,3}) when
        // the row length is even - one hash per pair instead of one per element
        float keepv[16], dsv[16];
        if (p.thr) {
          const uint64_t base = ((uint64_t)bh * p.Tq + t) * S_total + s0;
          if ((S_total & 1) == 0) {
#pragma unroll
            for (int r = 0; r < 16; r += 2)
              tell_keep2(p.seed, salt_eff, base + acc_row(r, lane), p.thr, p.inv_keep, keepv[r], keepv[r + 1]);
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) keepv[r] = tell_keep(p.seed, salt_eff, base + acc_row(r, lane), p.thr, p.inv_keep);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) keepv[r] = 1.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kk = acc_row(r, lane);
          const bool ok = key_ok[wave][kk] != 0.f && t < p.Tq;
          const float pv = ok ? __expf(st[r] - lse) : 0.f;
          const float keep = keepv[r];
          const float ds = pv * (dp[r] * keep - delta);
          Elem<T>::st(PdT + kk * SS + qi, pv * keep);         // Pd^T[key][q]
          Elem<T>::st(dST + kk * SS + qi, ds);                // dS^T[key][q]
          dsv[r] = ds;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {                         // dS[q][key]: 4 consecutive keys per register group
          T* pd = dSs + qi * SS + 8 * g + 4 * (lane >> 5);
          if constexpr (sizeof(T) == 2) {
            uint2 w;
            w.x = pack_bf16x2(dsv[4 * g], dsv[4 * g + 1]);
            w.y = pack_bf16x2(dsv[4 * g + 2], dsv[4 * g + 3]);
            *reinterpret_cast<uint2*>(pd) = w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) Elem<T>::st(pd + e, dsv[4 * g + e]);
          }
        }
      }
      __syncthreads();
      if (tv) {
#pragma unroll
        for (int f = 0; f < DF; ++f) {
          // dV^T[d][key] and dK^T[d][key]: with the operands in this order a lane owns ONE key row and, per register
          // group, 4 consecutive d - its gradients leave as 8-byte (bf16) / 16-byte (fp32) pieces of that row instead
          // of 32 two-byte stores each
          f32x16 dv, dk;
#pragma unroll
          for (int r = 0; r < 16; ++r) { dv[r] = 0.f; dk[r] = 0.f; }
          if constexpr (TR) {
            dv = mma32_kmA(dv, dOs, DS, f * 32, PdT, SS, lane);               // dO is [q][d]: k = q, m = d
            dk = mma32_kmA(dk, Qs, DS, f * 32, dST, SS, lane);
            dq[f] = mma32_kmB(dq[f], dSs, SS, Ks, DS, f * 32, lane);          // K is [key][d]: k = key, n = d
          } else {
            dv = mma32<T, 32>(dv, dOt + f * 32 * SS, SS, PdT, SS, lane);
            dk = mma32<T, 32>(dk, Qt + f * 32 * SS, SS, dST, SS, lane);
            dq[f] = mma32<T, 32>(dq[f], dSs, SS, Kt + f * 32 * SS, SS, lane);  // dQ[q][d]
          }
          const int s = s0 + (lane & 31);
          const bool real = s < p.S, isb = s == p.S && p.has_bias;
          T* pk = static_cast<T*>(p.dk) + (real ? s * p.k_ss + b * p.k_sb : 0) + (long)h * D;
          T* pv_ = static_cast<T*>(p.dv) + (real ? s * p.v_ss + b * p.v_sb : 0) + (long)h * D;
          float* gk = p.dbias_k ? p.dbias_k + (long)b * p.dbias_ld + (long)h * D : nullptr;
          float* gv = p.dbias_v ? p.dbias_v + (long)b * p.dbias_ld + (long)h * D : nullptr;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int d0 = f * 32 + 8 * g + 4 * (lane >> 5);
            if (d0 >= D) continue;
            float vk[4], vv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { vk[e] = dk[4 * g + e]; vv[e] = dv[4 * g + e]; }
            if (real) {
              if (qb != 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { vk[e] += Elem<T>::ld(pk + d0 + e); vv[e] += Elem<T>::ld(pv_ + d0 + e); }
              }
              if constexpr (sizeof(T) == 2) {
                uint2 wk, wv;
                wk.x = pack_bf16x2(vk[0], vk[1]); wk.y = pack_bf16x2(vk[2], vk[3]);
                wv.x = pack_bf16x2(vv[0], vv[1]); wv.y = pack_bf16x2(vv[2], vv[3]);
                *reinterpret_cast<uint2*>(pk + d0) = wk;
                *reinterpret_cast<uint2*>(pv_ + d0) = wv;
              } else {
                *reinterpret_cast<float4*>(pk + d0) = make_float4(vk[0], vk[1], vk[2], vk[3]);
                *reinterpret_cast<float4*>(pv_ + d0) = make_float4(vv[0], vv[1], vv[2], vv[3]);
              }
            } else if (isb) {
              float4* qk = reinterpret_cast<float4*>(gk + d0);
              float4* qv = reinterpret_cast<float4*>(gv + d0);
              if (qb == 0) {
                *qk = make_float4(vk[0], vk[1], vk[2], vk[3]);
                *qv = make_float4(vv[0], vv[1], vv[2], vv[3]);
              } else {
                const float4 ok_ = *qk, ov_ = *qv;
                *qk = make_float4(ok_.x + vk[0], ok_.y + vk[1], ok_.z + vk[2], ok_.w + vk[3]);
                *qv = make_float4(ov_.x + vv[0], ov_.y + vv[1], ov_.z + vv[2], ov_.w + vv[3]);
              }
            }
          }
        }
      }
    }
    // ---- reduce dQ over the waves and store
    __syncthreads();
    float* comb = reinterpret_cast<float*>(smem + SHARED_ELEMS + wave * WAVE_ELEMS);
#pragma unroll
    for (int f = 0; f < DF; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) comb[(f * 16 + r) * 64 + lane] = dq[f][r];
    __syncthreads();
    for (int item = wave; item < DF * 16; item += NW) {
      float s = 0.f;
      for (int w = 0; w < NW; ++w)
        s += reinterpret_cast<const float*>(smem + SHARED_ELEMS + w * WAVE_ELEMS)[item * 64 + lane];
      const int f = item / 16, r = item % 16;
      const int t = q0 + acc_row(r, lane), d = f * 32 + (lane & 31);
      if (t < p.Tq && d < D)
        Elem<T>::st(static_cast<T*>(p.dq) + t * p.q_st + b * p.q_sb + (long)h * D + d, s);
    }
  }
}

// ------------------------------------------------------------------ head-averaged weights (need_weights, eval only)
// w[b,t,s] = (1/H) sum_h exp(q_h . k_h[s] - lse[b,h,t])   (multi_head.py:478-482)
template <typename T>
__global__ __launch_bounds__(256) void attn_avg_weights_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                               const float* __restrict__ lse,
                                                               const uint8_t* __restrict__ mask,
                                                               const T* __restrict__ bias_k,
                                                               float* __restrict__ w, int B, int H, int Tq,
                                                               int S, int D, long q_st, long q_sb, long k_ss,
                                                               long k_sb, int has_bias, int has_zero) {
  const int t = blockIdx.x, b = blockIdx.y;
  const int S_total = S + has_bias + has_zero;
  for (int s = threadIdx.x; s < S_total; s += 256) {
    float acc = 0.f;
    const bool masked = s < S && mask && mask[(long)b * S + s];
    if (!masked) {
      for (int h = 0; h < H; ++h) {
        const T* qp = q + t * q_st + b * q_sb + (long)h * D;
        const T* kp = s < S ? k + s * k_ss + b * k_sb + (long)h * D
                            : (s == S && has_bias ? bias_k + (long)h * D : nullptr);
        float dot = 0.f;
        if (kp) for (int d = 0; d < D; ++d) dot += Elem<T>::ld(qp + d) * Elem<T>::ld(kp + d);
        acc += __expf(dot - lse[((long)b * H + h) * Tq + t]);
      }
    }
    w[((long)b * Tq + t) * S_total + s] = acc / H;
  }
}
extern "C" int tell_attn_avg_weights(const void* q, const void* k, const float* lse, const uint8_t* mask,
                                     const void* bias_k, float* w, int B, int H, int Tq, int S, int D,
                                     long q_st, long q_sb, long k_ss, long k_sb, int has_zero, int dtype,
                                     hipStream_t stream) {
  if (B * Tq <= 0) return TELL_OK;
  dim3 grid(Tq, B);
  if (dtype == TELL_BF16)
    hipLaunchKernelGGL((attn_avg_weights_kernel<uint16_t>), grid, dim3(256), 0, stream, (const uint16_t*)q, (const uint16_t*)k, lse, mask, (const uint16_t*)bias_k, w, B, H, Tq, S, D, q_st, q_sb, k_ss, k_sb, bias_k ? 1 : 0, has_zero);
  else
    hipLaunchKernelGGL((attn_avg_weights_kernel<float>), grid, dim3(256), 0, stream, (const float*)q, (const float*)k, lse, mask, (const float*)bias_k, w, B, H, Tq, S, D, q_st, q_sb, k_ss, k_sb, bias_k ? 1 : 0, has_zero);
  return tell_check_launch("attn_avg_weights");
}

// ------------------------------------------------------------------ C ABI
static int fill_args(AttnArgs& a, const void* q, const void* k, const void* v, void* out, float* lse,
                     const uint8_t* mask, const void* bias_k, const void* bias_v, int B, int H, int Tq,
                     int S, int D, long q_st, long q_sb, long k_ss, long k_sb, long v_ss, long v_sb,
                     long o_st, long o_sb, int has_zero, float p, uint32_t seed, uint32_t salt, int dtype) {
  TELL_REQUIRE(D == 64 || D == 16, "attention: head_dim must be 64 (or 16 for the reduced-size tests)");
  TELL_REQUIRE(B > 0 && H > 0 && Tq > 0 && S >= 0, "attention: bad sizes");
  TELL_REQUIRE((bias_k == nullptr) == (bias_v == nullptr), "attention: bias_k and bias_v go together");
  TELL_REQUIRE(S + (bias_k ? 1 : 0) + has_zero > 0, "attention: no keys");
  TELL_REQUIRE(p >= 0.f && p < 1.f, "attention: dropout p must be in [0,1)");
  const int vec = dtype == TELL_BF16 ? 8 : 4;
  TELL_REQUIRE(q_st % vec == 0 && q_sb % vec == 0 && k_ss % vec == 0 && k_sb % vec == 0 &&
               v_ss % vec == 0 && v_sb % vec == 0, "attention: strides must be multiples of a 16-byte chunk");
  TELL_REQUIRE(((uintptr_t)q & 15) == 0 && ((uintptr_t)k & 15) == 0 && ((uintptr_t)v & 15) == 0,
               "attention: q/k/v must be 16-byte aligned");
  a.q = q; a.k = k; a.v = v; a.out = out; a.lse = lse; a.mask = mask; a.bias_k = bias_k; a.bias_v = bias_v;
  a.q_st = q_st; a.q_sb = q_sb; a.k_ss = k_ss; a.k_sb = k_sb; a.v_ss = v_ss; a.v_sb = v_sb;
  a.o_st = o_st; a.o_sb = o_sb; a.B = B; a.H = H; a.Tq = Tq; a.S = S;
  a.has_bias = bias_k ? 1 : 0; a.has_zero = has_zero;
  a.thr = p > 0.f ? tell_drop_threshold(p) : 0u; a.inv_keep = 1.f / (1.f - p); a.seed = seed; a.salt = salt; a.step = g_tell_rng_step;
  a.dout = nullptr; a.o = nullptr; a.dq = a.dk = a.dv = nullptr; a.dbias_k = a.dbias_v = nullptr; a.dbias_ld = 0;
  return TELL_OK;
}

extern "C" int tell_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse,
                             const uint8_t* mask, const void* bias_k, const void* bias_v, int B, int H,
                             int Tq, int S, int D, long q_st, long q_sb, long k_ss, long k_sb, long v_ss,
                             long v_sb, long o_st, long o_sb, int has_zero, float p, uint32_t seed,
                             uint32_t salt, int dtype, hipStream_t stream) {
  AttnArgs a;
  int rc = fill_args(a, q, k, v, out, lse, mask, bias_k, bias_v, B, H, Tq, S, D, q_st, q_sb, k_ss, k_sb,
                     v_ss, v_sb, o_st, o_sb, has_zero, p, seed, salt, dtype);
  if (rc) return rc;
  const int QB = (Tq + 31) / 32;
  if (D == 64 && QB >= 4) {            // long sequences: shared 64-key tiles
    dim3 grid((QB + 3) / 4, B * H);
    const int old_path = (int)tell_opt(OPT_ATTN_TILE64);   // A/B aid
    const bool self_env = tell_opt(OPT_ATTN_SELF) != 0;      // A/B aid
    if (dtype == TELL_BF16 && !old_path && self_env && !a.has_bias && !a.has_zero && S % 64 == 0 &&
        (long)63 * k_ss + 64 < (1L << 31) && (long)63 * v_ss + 64 < (1L << 31)) {
      const bool dma_env = tell_opt(OPT_ATTN_DMA) == 1;      // A/B aid
      if (dma_env) {
        if (a.thr) hipLaunchKernelGGL((attn_self_fwd_kernel<true, 0, true>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((attn_self_fwd_kernel<false, 0, true>), grid, dim3(256), 0, stream, a);
      } else if (a.thr) {
        // (round 4 measured the keep decisions as lane masks from their own launch: 126.6 us against 78.2 us hashing in
        //  place - tools/probes/rejected/README; removed from the library in round 5)
        hipLaunchKernelGGL((attn_self_fwd_kernel<true>), grid, dim3(256), 0, stream, a);
      } else hipLaunchKernelGGL((attn_self_fwd_kernel<false>), grid, dim3(256), 0, stream, a);
      return tell_check_launch("attn_self_fwd");
    }
    if (dtype == TELL_BF16 && !old_path) {
      const int occ = (int)tell_opt(OPT_ATTN_OCC);     // A/B aid
      if (occ == 3) {
        if (a.thr) hipLaunchKernelGGL((attn_fwd_reg_kernel<true, 3>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((attn_fwd_reg_kernel<false, 3>), grid, dim3(256), 0, stream, a);
      } else if (a.thr) hipLaunchKernelGGL((attn_fwd_reg_kernel<true>), grid, dim3(256), 0, stream, a);
      else hipLaunchKernelGGL((attn_fwd_reg_kernel<false>), grid, dim3(256), 0, stream, a);
      return tell_check_launch("attn_fwd_reg");
    } else if (dtype == TELL_BF16) {
      if (a.thr) hipLaunchKernelGGL((attn_fwd_tile64_kernel<uint16_t, true>), grid, dim3(256), 0, stream, a);
      else hipLaunchKernelGGL((attn_fwd_tile64_kernel<uint16_t, false>), grid, dim3(256), 0, stream, a);
    } else {
      if (a.thr) hipLaunchKernelGGL((attn_fwd_tile64_kernel<float, true>), grid, dim3(256), 0, stream, a);
      else hipLaunchKernelGGL((attn_fwd_tile64_kernel<float, false>), grid, dim3(256), 0, stream, a);
    }
    return tell_check_launch("attn_fwd_tile64");
  }
  if (dtype == TELL_BF16) {
    constexpr int NW = 4;
    dim3 grid(QB == 1 ? 1 : (QB + NW - 1) / NW, B * H);
    if (D == 64) hipLaunchKernelGGL((attn_fwd_kernel<uint16_t, 64, NW>), grid, dim3(64 * NW), 0, stream, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<uint16_t, 16, NW>), grid, dim3(64 * NW), 0, stream, a);
  } else {
    constexpr int NW = 4;
    dim3 grid(QB == 1 ? 1 : (QB + NW - 1) / NW, B * H);
    if (D == 64) hipLaunchKernelGGL((attn_fwd_kernel<float, 64, NW>), grid, dim3(64 * NW), 0, stream, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<float, 16, NW>), grid, dim3(64 * NW), 0, stream, a);
  }
  return tell_check_launch("attn_fwd");
}

extern "C" int tell_attn_bwd(const void* q, const void* k, const void* v, const void* out,
                             const void* dout, const float* lse, const uint8_t* mask, const void* bias_k,
                             const void* bias_v, void* dq, void* dk, void* dv, float* dbias_k,
                             float* dbias_v, int B, int H, int Tq, int S, int D, long q_st, long q_sb,
                             long k_ss, long k_sb, long v_ss, long v_sb, long o_st, long o_sb,
                             int has_zero, float p, uint32_t seed, uint32_t salt, int dtype,
                             hipStream_t stream) {
  AttnArgs a;
  int rc = fill_args(a, q, k, v, const_cast<void*>(out), const_cast<float*>(lse), mask, bias_k, bias_v, B,
                     H, Tq, S, D, q_st, q_sb, k_ss, k_sb, v_ss, v_sb, o_st, o_sb, has_zero, p, seed, salt,
                     dtype);
  if (rc) return rc;
  TELL_REQUIRE(!bias_k || (dbias_k && dbias_v), "attention bwd: dbias buffers required with bias rows");
  a.dout = dout; a.o = out; a.dq = dq; a.dk = dk; a.dv = dv; a.dbias_k = dbias_k; a.dbias_v = dbias_v;
  // the two partial buffers may be the halves of ONE [B, 2 H D] buffer (dbias_v == dbias_k + H D): one column-sum
  // launch then folds both bias gradients
  a.dbias_ld = (dbias_k && dbias_v == dbias_k + (long)H * D) ? 2L * H * D : (long)H * D;
  dim3 grid(B * H);
  if (dtype == TELL_BF16) {
    // every wave of the workgroup owns 21.5 KB of LDS tiles (4 waves: 105 KB - one workgroup per CU, two rounds for
    // B*H = 512): short contexts take only the waves that have a key tile to work on, so that all of B*H is resident
    const int nkt = (S + (bias_k ? 1 : 0) + has_zero + 31) / 32;
    if (D == 64 && nkt <= 1) hipLaunchKernelGGL((attn_bwd_kernel<uint16_t, 64, 1>), grid, dim3(64), 0, stream, a);
    else if (D == 64 && nkt <= 4) hipLaunchKernelGGL((attn_bwd_kernel<uint16_t, 64, 2>), grid, dim3(128), 0, stream, a);
    else if (D == 64) hipLaunchKernelGGL((attn_bwd_kernel<uint16_t, 64, 4>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((attn_bwd_kernel<uint16_t, 16, 4>), grid, dim3(256), 0, stream, a);
  } else {
    if (D == 64) hipLaunchKernelGGL((attn_bwd_kernel<float, 64, 2>), grid, dim3(128), 0, stream, a);
    else hipLaunchKernelGGL((attn_bwd_kernel<float, 16, 2>), grid, dim3(128), 0, stream, a);
  }
  return tell_check_launch("attn_bwd");
}
