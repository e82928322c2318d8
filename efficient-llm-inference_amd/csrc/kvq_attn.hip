// kvq_attn.hip — single-query (decode) attention that reads the INT8 / packed-INT4 KV store
// directly: SURVEY §8(f) N1, second form ("a decode-attention kernel that reads INT8/INT4
// directly removes the O(T) per-step traffic" of to_past_key_values, ops.py:345-355, and of the
// fp16 copy HF attention then reads, benchmarker.py:470-471).
//
// What the reference computes per decode step and layer: dequantise every stored token
// (k_int * sk[t], v_int * sv[t] -> fp16), cat the new token's exact fp16 k/v, and run
// softmax(q K^T * sm_scale) V. Here the per-token scales factor out of both products:
//   s[t]   = sm_scale * sk[t] * sum_d q[d] * k_int[t,d]
//   out[d] = sum_t p[t] * sv[t] * v_int[t,d]  (+ the exact new token as one more softmax term)
// so the store is read ONCE at 1 or 0.5 byte per element and no fp16 copy of the cache exists.
// Numerics: fp32 accumulation of exact integer x fp16 products; the reference's intermediate
// rounding of the dequantised value to fp16 is skipped (<= 2^-11 relative per element), so the
// result is within fp16 tolerance of the reference, not bit-identical (test tolerance 2e-3).
//
// Split-T flash decoding:
//   decode_attn_partial_k       VALU kernel (any supported head_dim, <= 8 query heads per kv head):
//                               grid (split, kv head, batch), 256 threads, TS tokens per workgroup, all
//                               Hq/Hkv query heads of the kv head in one pass (K/V read once); 16
//                               elements per lane per token: 16-byte (INT8) / 8-byte (INT4) loads
//   decode_attn_partial_mfma_k  head_dim 64 / 128 with 3..16 query heads per kv head: both products on the
//                               matrix cores, one wave per 128-token split (see its header below)
//   decode_attn_stream_mfma_k   the same tile arithmetic for larger batches (more 64-token tiles than wave slots):
//                               one wave walks several tiles, online softmax, ONE partial per wave; the registers
//                               of the tile being reduced are re-requested for the tile after next (ROLL)
//   all write (m, l) and acc[D] per (batch, query head, split) to the caller's workspace
//   decode_attn_merge_fast_k    grid (query head, batch): log-sum-exp merge of the splits and of the exact new
//                               token, every operand requested up front (<= 256 splits); two extra workgroups
//                               quantise the new token into slot T of the stores (a decode step = two launches)
//   decode_attn_merge_k         the same arithmetic as a chain of dependent loads (any number of splits)
//   decode_attn_fused_mfma_k    opt-in single launch (in-workgroup merge + arrival ticket): measured slower
#include <atomic>
#include <type_traits>

#include "kvq_common.h"

// cache policy of the K / V row loads (buffer-load aux immediate: 2 = non-temporal, what ships; `make calib_attn`
// builds may override it for A-B runs)
#ifndef KVQ_ATTN_KV_AUX
#define KVQ_ATTN_KV_AUX 2
#endif

namespace kvq {

constexpr int kAttnBlock = 256;
constexpr int kAttnMaxTS = 1024;  // tokens per workgroup (LDS: NQ * TS floats)
constexpr int kAttnUnroll = 4;    // tokens in flight per lane

struct AttnArgs {
  const void* q;
  int64_t q_sb, q_sh;  // elements
  const uint8_t* k;
  int64_t k_sb, k_sh, k_st;  // bytes
  const float* k_scale;
  const uint8_t* v;
  int64_t v_sb, v_sh, v_st;
  const float* v_scale;
  const void* kn;
  int64_t kn_sb, kn_sh;
  const void* vn;
  int64_t vn_sb, vn_sh;
  void* out;
  int64_t o_sb, o_sh;
  float* ws;        // [B*Hq*nsplit][2] (m, l), then at acc_off [B*Hq*nsplit][D]
  int64_t acc_off;  // floats
  float sm_scale;
  uint32_t B, Hq, Hkv, T, D, TS, nsplit, nq;
  int32_t lpt_shift;  // log2(D / 16): lanes per token
  int32_t dtype;      // KVQ_F16 | KVQ_BF16 (q, k_new, v_new, out)
  int32_t mfma;       // host: the MFMA partial kernel serves this call
  uint32_t stream_tpw;  // host: > 0 = the streaming MFMA kernel, that many 64-token tiles per wave
  uint32_t lds;         // host: 1 = the LDS-staged streaming kernel (decode_attn_lds_mfma_k) serves the streaming plan
  // kvq_decode_step_dev: the stored-token count lives in DEVICE memory (a captured HIP graph replays
  // the same launch for every decode step). T above is then the host's upper bound: it sizes the grid and
  // the workspace; workgroups past the real count exit, the merge reads ceil(T / TS) partials.
  const int32_t* t_dev;
  // In-launch merge (round 4; host: attn_fold_plan): != nullptr = the LDS-staged kernel's waves store their partials
  // write-through, take a ticket on arrive[b * Hkv + hk], and the wave that draws the last one merges the kv head's query
  // heads itself (merge_group_one_wave) — no merge launch. The words are zero when the launch starts (the host call's
  // hipMemsetAsync) and zero again when it ends (the merging wave resets its word).
  uint32_t* arrive;
  uint32_t ws_bytes;      // bytes of `ws` the merge may address (buffer descriptor range; host: < 2 GiB)
  int32_t fold_has_new;   // the exact new token's term is part of the merge (kn / vn valid either way)
};

// stored tokens this launch attends: the host's count, or the device word of a graph-replayed step
// (clamped to the host's bound: a count the caller let run past it can make the result wrong, never the accesses)
__device__ inline uint32_t live_tokens(const AttnArgs& a) {
  if (!a.t_dev) return a.T;
  const uint32_t t = (uint32_t)__builtin_nontemporal_load(a.t_dev);
  return t < a.T ? t : a.T;
}

__device__ inline f16x2 bits_h2(uint32_t u) {
  f16x2 h;
  __builtin_memcpy(&h, &u, 4);
  return h;
}
__device__ inline uint32_t h2_bits(f16x2 h) {
  uint32_t u;
  __builtin_memcpy(&u, &h, 4);
  return u;
}
__device__ inline float load_elem(const void* p, int64_t i, int dtype) {
  const uint16_t b = reinterpret_cast<const uint16_t*>(p)[i];
  return dtype == KVQ_F16 ? Elem<KVQ_F16>::widen(b) : Elem<KVQ_BF16>::widen(b);
}

// 16 consecutive query elements as 8 f16 pairs (bf16 queries are converted: exact for the normal
// f16 range). PERM4: pair order of the INT4 path, per 8 elements (q0,q2) (q4,q6) (q1,q3) (q5,q7).
template <bool PERM4>
__device__ inline void convert_q16(const u32x4 a, const u32x4 b, int dtype, f16x2 (&qv)[8]) {
  uint32_t w[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  if (dtype == KVQ_BF16) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f16x2 h = {(f16)__uint_as_float(w[j] << 16), (f16)__uint_as_float(w[j] & 0xFFFF0000u)};
      w[j] = h2_bits(h);
    }
  }
  if constexpr (PERM4) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const uint32_t w0 = w[4 * g], w1 = w[4 * g + 1], w2 = w[4 * g + 2], w3 = w[4 * g + 3];
      w[4 * g + 0] = __builtin_amdgcn_perm(w1, w0, 0x05040100u);  // (q0, q2)
      w[4 * g + 1] = __builtin_amdgcn_perm(w3, w2, 0x05040100u);  // (q4, q6)
      w[4 * g + 2] = __builtin_amdgcn_perm(w1, w0, 0x07060302u);  // (q1, q3)
      w[4 * g + 3] = __builtin_amdgcn_perm(w3, w2, 0x07060302u);  // (q5, q7)
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) qv[j] = bits_h2(w[j]);
}

// Raw bytes of 16 stored elements of one token row.
template <int BITS>
struct Raw16;
template <>
struct Raw16<8> {
  u32x4 w;
  __device__ inline void load(const uint8_t* p) { w = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
  // exact f16 pairs of the int8 values: byte ^ 0x80 = q + 128; 0x6400 | u is the f16 1024 + u
  __device__ inline void to_h2(f16x2 (&kp)[8]) const {
    const f16x2 bias = {(f16)1152.0f, (f16)1152.0f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t x = w[i] ^ 0x80808080u;
      kp[2 * i] = bits_h2(__builtin_amdgcn_perm(0x64646464u, x, 0x04010400u)) - bias;      // (e0, e1)
      kp[2 * i + 1] = bits_h2(__builtin_amdgcn_perm(0x64646464u, x, 0x04030402u)) - bias;  // (e2, e3)
    }
  }
  // q + 128 as floats (bias folded by the caller)
  __device__ inline void to_f32_biased(float (&u)[16]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t x = w[i] ^ 0x80808080u;
      u[4 * i] = (float)(x & 0xFFu);
      u[4 * i + 1] = (float)((x >> 8) & 0xFFu);
      u[4 * i + 2] = (float)((x >> 16) & 0xFFu);
      u[4 * i + 3] = (float)(x >> 24);
    }
  }
  static constexpr float kBias = 128.0f;
  static constexpr int kBytes = 16;
};
template <>
struct Raw16<4> {
  u32x2 w;
  __device__ inline void load(const uint8_t* p) { w = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p)); }
  // nibble = q + 8, even element in the HIGH nibble (ops.py:61-63); pair order per 8 elements:
  // (e0,e2) (e4,e6) (e1,e3) (e5,e7) — convert_q16<true> arranges the query the same way
  __device__ inline void to_h2(f16x2 (&kp)[8]) const {
    const f16x2 bias = {(f16)1032.0f, (f16)1032.0f};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint32_t hi = (w[i] >> 4) & 0x0F0F0F0Fu;  // elements 0, 2, 4, 6
      const uint32_t lo = w[i] & 0x0F0F0F0Fu;         // elements 1, 3, 5, 7
      kp[4 * i + 0] = bits_h2(__builtin_amdgcn_perm(0x64646464u, hi, 0x04010400u)) - bias;
      kp[4 * i + 1] = bits_h2(__builtin_amdgcn_perm(0x64646464u, hi, 0x04030402u)) - bias;
      kp[4 * i + 2] = bits_h2(__builtin_amdgcn_perm(0x64646464u, lo, 0x04010400u)) - bias;
      kp[4 * i + 3] = bits_h2(__builtin_amdgcn_perm(0x64646464u, lo, 0x04030402u)) - bias;
    }
  }
  __device__ inline void to_f32_biased(float (&u)[16]) const {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint32_t hi = (w[i] >> 4) & 0x0F0F0F0Fu;
      const uint32_t lo = w[i] & 0x0F0F0F0Fu;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        u[8 * i + 2 * b] = (float)((hi >> (8 * b)) & 0xFFu);
        u[8 * i + 2 * b + 1] = (float)((lo >> (8 * b)) & 0xFFu);
      }
    }
  }
  static constexpr float kBias = 8.0f;
  static constexpr int kBytes = 8;
};

template <int CTRL>
__device__ inline uint32_t dpp_u32_attn(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
// sum over aligned groups of 2^logw consecutive lanes (logw <= 4), every lane gets the sum
__device__ inline float group_fadd(float v, int logw) {
  if (logw > 0) v += __uint_as_float(dpp_u32_attn<0xB1>(__float_as_uint(v)));
  if (logw > 1) v += __uint_as_float(dpp_u32_attn<0x4E>(__float_as_uint(v)));
  if (logw > 2) v += __uint_as_float(dpp_u32_attn<0x141>(__float_as_uint(v)));
  if (logw > 3) v += __uint_as_float(dpp_u32_attn<0x140>(__float_as_uint(v)));
  return v;
}

// Exchanges across the four 16-lane rows of a wave WITHOUT the LDS crossbar (__shfl_xor is ds_bpermute: an LDS round
// trip on the wave's critical path): v_permlane16_swap / v_permlane32_swap of a value with a copy of itself leave
// {own, partner} in the two results for the partner 16 (32) lanes away; max and + are commutative, so every lane
// computes the same bits a __shfl_xor butterfly would.
__device__ inline float xor16_max(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ inline float xor32_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ inline float xor16_add(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ inline float xor32_add(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// max over the wave, in every lane: DPP butterflies inside a 16-lane row, lane swaps across the rows
__device__ inline float wave_fmax(float v) {
  v = fmaxf(v, __uint_as_float(dpp_u32_attn<0xB1>(__float_as_uint(v))));   // lane ^ 1
  v = fmaxf(v, __uint_as_float(dpp_u32_attn<0x4E>(__float_as_uint(v))));   // lane ^ 2
  v = fmaxf(v, __uint_as_float(dpp_u32_attn<0x141>(__float_as_uint(v))));  // row_half_mirror
  v = fmaxf(v, __uint_as_float(dpp_u32_attn<0x140>(__float_as_uint(v))));  // row_mirror
  return xor32_max(xor16_max(v));
}
// sum over the wave, in every lane (same exchanges as wave_fmax: no LDS round trips; every lane adds in the same order)
__device__ inline float wave_fsum(float v) {
  v += __uint_as_float(dpp_u32_attn<0xB1>(__float_as_uint(v)));
  v += __uint_as_float(dpp_u32_attn<0x4E>(__float_as_uint(v)));
  v += __uint_as_float(dpp_u32_attn<0x141>(__float_as_uint(v)));
  v += __uint_as_float(dpp_u32_attn<0x140>(__float_as_uint(v)));
  return xor32_add(xor16_add(v));
}

// Workgroup timeline (one round trip to HBM on the critical path when TS == TL * kAttnUnroll):
// issue the first K tile, the first V tile, the scales and the query at once; scores while V is
// still in flight; softmax in LDS; weighted V sum; token lanes -> one row per wave -> workspace.
template <int KBITS, int VBITS, int NQ>
__global__ __launch_bounds__(kAttnBlock) void decode_attn_partial_k(const AttnArgs a) {
  __shared__ float s_p[NQ][kAttnMaxTS];  // raw scores, then p[t] * sv[t]
  __shared__ float s_ks[kAttnMaxTS];     // sk[t] * sm_scale
  __shared__ float s_vs[kAttnMaxTS];     // sv[t]
  __shared__ float s_acc[kAttnBlock / kWave][NQ][256];
  __shared__ float s_w[NQ];              // sum_t p[t] * sv[t]  (bias fold)
  const uint32_t tid = threadIdx.x;
  const uint32_t split = blockIdx.x, hk = blockIdx.y, b = blockIdx.z;
  const uint32_t t0 = split * a.TS;
  const uint32_t T = live_tokens(a);
  if (t0 >= T) return;  // device-side T (graph replay): this split holds nothing yet
  const uint32_t nt = T - t0 < a.TS ? T - t0 : a.TS;
  const uint32_t lpt = 1u << a.lpt_shift;
  const uint32_t ld = tid & (lpt - 1u);    // which 16-element slice of D
  const uint32_t tl = tid >> a.lpt_shift;  // token lane
  const uint32_t TL = kAttnBlock >> a.lpt_shift;
  const uint32_t step = TL * kAttnUnroll;  // tokens per loop iteration

  const uint8_t* kb = a.k + (int64_t)b * a.k_sb + (int64_t)hk * a.k_sh + (int64_t)t0 * a.k_st +
                      (int64_t)ld * Raw16<KBITS>::kBytes;
  const uint8_t* vb = a.v + (int64_t)b * a.v_sb + (int64_t)hk * a.v_sh + (int64_t)t0 * a.v_st +
                      (int64_t)ld * Raw16<VBITS>::kBytes;
  // Every request of the prologue goes out before anything waits (one round trip): first K tile,
  // first V tile, query, scales. Rows past the split's end are CLAMPED to its last row instead of
  // skipped (no branches between the loads); their scores are never stored and their P never read.
  Raw16<KBITS> kraw[kAttnUnroll];
  Raw16<VBITS> vraw[kAttnUnroll];
#pragma unroll
  for (int u = 0; u < kAttnUnroll; ++u) {
    const uint32_t i = u * TL + tl;
    kraw[u].load(kb + (int64_t)(i < nt ? i : nt - 1u) * a.k_st);
  }
#pragma unroll
  for (int u = 0; u < kAttnUnroll; ++u) {
    const uint32_t i = u * TL + tl;
    vraw[u].load(vb + (int64_t)(i < nt ? i : nt - 1u) * a.v_st);
  }
  u32x4 qraw[NQ][2];
#pragma unroll
  for (int h = 0; h < NQ; ++h) {
    const uint32_t hh = (uint32_t)h < a.nq ? (uint32_t)h : 0u;  // padded heads read head 0, zeroed below
    const char* qp = reinterpret_cast<const char*>(a.q) +
                     ((int64_t)b * a.q_sb + (int64_t)(hk * a.nq + hh) * a.q_sh + (int64_t)ld * 16) * 2;
    qraw[h][0] = *reinterpret_cast<const u32x4*>(qp);
    qraw[h][1] = *(reinterpret_cast<const u32x4*>(qp) + 1);
  }
  {
    float ksv[kAttnMaxTS / kAttnBlock], vsv[kAttnMaxTS / kAttnBlock];
#pragma unroll
    for (int r = 0; r < kAttnMaxTS / kAttnBlock; ++r) {
      const uint32_t i = r * kAttnBlock + tid;
      const uint32_t ic = i < nt ? i : nt - 1u;
      ksv[r] = a.k_scale[t0 + ic];
      vsv[r] = a.v_scale[t0 + ic];
    }
#pragma unroll
    for (int r = 0; r < kAttnMaxTS / kAttnBlock; ++r) {
      const uint32_t i = r * kAttnBlock + tid;
      if (i < nt) {
        s_ks[i] = ksv[r] * a.sm_scale;
        s_vs[i] = vsv[r];
      }
    }
  }

  // ---- phase A: raw scores q . k_int ------------------------------------------------------
  {
    f16x2 qv[NQ][8];
#pragma unroll
    for (int h = 0; h < NQ; ++h) {
      if ((uint32_t)h < a.nq) {
        convert_q16<KBITS == 4>(qraw[h][0], qraw[h][1], a.dtype, qv[h]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[h][j] = f16x2{(f16)0.0f, (f16)0.0f};
      }
    }
    for (uint32_t base = 0; base < nt; base += step) {  // uniform trip count
      const bool more = base + step < nt;
      Raw16<KBITS> nxt[kAttnUnroll];
      if (more) {  // next tile in flight while this one is reduced
#pragma unroll
        for (int u = 0; u < kAttnUnroll; ++u) {
          const uint32_t i = base + step + u * TL + tl;
          nxt[u].load(kb + (int64_t)(i < nt ? i : nt - 1u) * a.k_st);
        }
      }
#pragma unroll
      for (int u = 0; u < kAttnUnroll; ++u) {
        const uint32_t i = base + u * TL + tl;
        f16x2 kp[8];
        kraw[u].to_h2(kp);
#pragma unroll
        for (int h = 0; h < NQ; ++h) {
          float s = 0.0f;
#pragma unroll
          for (int j = 0; j < 8; ++j) s = __builtin_amdgcn_fdot2(kp[j], qv[h][j], s, false);
          s = group_fadd(s, a.lpt_shift);
          if (ld == 0u && i < nt) s_p[h][i] = s;
        }
      }
      if (more) {
#pragma unroll
        for (int u = 0; u < kAttnUnroll; ++u) kraw[u] = nxt[u];
      }
    }
  }
  __syncthreads();

  // ---- phase B: per-head softmax over this split (one wave per head, round robin) -----------
  {
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    for (uint32_t h = wave; h < a.nq; h += kAttnBlock / kWave) {
      float m = -INFINITY;
      for (uint32_t i = lane; i < nt; i += kWave) {
        const float sc = s_p[h][i] * s_ks[i];
        s_p[h][i] = sc;
        m = fmaxf(m, sc);
      }
      m = wave_fmax(m);
      float l = 0.0f, wsum = 0.0f;
      for (uint32_t i = lane; i < nt; i += kWave) {
        const float p = __expf(s_p[h][i] - m);
        const float pv = p * s_vs[i];
        l += p;
        wsum += pv;
        s_p[h][i] = pv;
      }
      l = wave_fsum(l);
      wsum = wave_fsum(wsum);
      if (lane == 0u) {
        s_w[h] = wsum;
        float* o = a.ws + (((int64_t)b * a.Hq + hk * a.nq + h) * a.nsplit + split) * 2;
        o[0] = m;
        o[1] = l;
      }
    }
  }
  __syncthreads();

  // ---- phase C: acc[h][d] = sum_t (p sv)[t] * (v_int[t,d] + bias) ----------------------------
  float acc[NQ][16];
#pragma unroll
  for (int h = 0; h < NQ; ++h)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[h][j] = 0.0f;
  for (uint32_t base = 0; base < nt; base += step) {
    const bool more = base + step < nt;
    Raw16<VBITS> nxt[kAttnUnroll];
    if (more) {
#pragma unroll
      for (int u = 0; u < kAttnUnroll; ++u) {
        const uint32_t i = base + step + u * TL + tl;
        nxt[u].load(vb + (int64_t)(i < nt ? i : nt - 1u) * a.v_st);
      }
    }
#pragma unroll
    for (int u = 0; u < kAttnUnroll; ++u) {
      const uint32_t i = base + u * TL + tl;
      if (i < nt) {
        float uf[16];
        vraw[u].to_f32_biased(uf);
#pragma unroll
        for (int h = 0; h < NQ; ++h) {
          const float p = s_p[h][i];
#pragma unroll
          for (int j = 0; j < 16; ++j) acc[h][j] = fmaf(uf[j], p, acc[h][j]);
        }
      }
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < kAttnUnroll; ++u) vraw[u] = nxt[u];
    }
  }
  // token lanes of one wave -> every lane holds the wave's sum, then the 4 waves through LDS.
  // Inside a 16-lane row: DPP rotations (one VALU op each); across rows: two lane exchanges per
  // value, issued back to back (the uniform branches stay OUTSIDE the value loops so that they pipeline).
#define KVQ_ROW_ROR(N)                                                                                     \
  _Pragma("unroll") for (int h = 0; h < NQ; ++h) _Pragma("unroll") for (int j = 0; j < 16; ++j)           \
      acc[h][j] += __uint_as_float(dpp_u32_attn<0x120 + N>(__float_as_uint(acc[h][j])));
  if (lpt <= 8u) { KVQ_ROW_ROR(8) }
  if (lpt <= 4u) { KVQ_ROW_ROR(4) }
  if (lpt <= 2u) { KVQ_ROW_ROR(2) }
  if (lpt <= 1u) { KVQ_ROW_ROR(1) }
#undef KVQ_ROW_ROR
#pragma unroll
  for (int h = 0; h < NQ; ++h) {
    float t[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) t[j] = __shfl_xor(acc[h][j], 16);
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[h][j] += t[j];
#pragma unroll
    for (int j = 0; j < 16; ++j) t[j] = __shfl_xor(acc[h][j], 32);
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[h][j] += t[j];
  }
  {
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    if (lane < lpt) {
#pragma unroll
      for (int h = 0; h < NQ; ++h)
#pragma unroll
        for (int j = 0; j < 16; ++j) s_acc[wave][h][lane * 16 + j] = acc[h][j];
    }
  }
  __syncthreads();
  for (uint32_t idx = tid; idx < a.nq * a.D; idx += kAttnBlock) {
    const uint32_t h = idx / a.D, d = idx - h * a.D;
    const float v = s_acc[0][h][d] + s_acc[1][h][d] + s_acc[2][h][d] + s_acc[3][h][d] -
                    Raw16<VBITS>::kBias * s_w[h];
    a.ws[a.acc_off + (((int64_t)b * a.Hq + hk * a.nq + h) * a.nsplit + split) * a.D + d] = v;
  }
}

// ---------------------------------------------------------------------------- MFMA variant
// Grouped-query heads (3..16 query heads per kv head, head_dim 64 or 128): the per-element VALU cost of
// the kernel above grows with the group size (one dot2 / fma per element and head) and it ends up
// VALU-bound (llama 4:1 grouping: 2.1 TB/s at batch 8). Here both products run on the matrix
// cores as 16x16x32 f16 MFMAs with the heads padded to 16 columns; the VALU only converts
// int8 / nibbles to f16 (exactly) and transposes V bytes in registers.
//   ONE WAVE per workgroup, TC tokens, no barriers between phases.
//   S = K Q^T : A = K tile (row = token x, k = 8 of the lane group's d), B = Q^T (col = head x),
//               C: lane (x, g) holds head x, tokens 16 i + 4 g + r  (i = tile, r = 0..3)
//   O = P V   : A = P, taken from the lane's OWN score registers of tiles 2s, 2s+1 (the k order
//               of a 32-token step is defined as tokens 32 s + 4 g + j, 32 s + 16 + 4 g + (j-4), so
//               no lane movement); B = V with col x <-> d = 8 x + n for MFMA n = 0..7, so the lane
//               reads 8 contiguous bytes (4 for INT4) of 8 token rows and transposes them 8x8 in
//               registers; C: lane (x, g) holds heads 4 g + r of d = 8 x + n.
// P is scaled by sv[t] / max sv before the f16 pack (range), the output by max sv afterwards.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline f16x8 pack_h8(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  u32x4 w = {a, b, c, d};
  f16x8 h;
  __builtin_memcpy(&h, &w, 16);
  return h;
}
// bytes (u0..u3) of x, each meaning value u - BIAS -> two f16 pairs, exact
template <int BIAS>
__device__ inline void bytes_to_h4(uint32_t x, uint32_t& lo, uint32_t& hi) {
  const f16x2 bias = {(f16)(1024.0f + BIAS), (f16)(1024.0f + BIAS)};
  lo = h2_bits(bits_h2(__builtin_amdgcn_perm(0x64646464u, x, 0x04010400u)) - bias);
  hi = h2_bits(bits_h2(__builtin_amdgcn_perm(0x64646464u, x, 0x04030402u)) - bias);
}
// 4x4 byte transpose: rows a0..a3 -> columns c0..c3 (c_n = byte n of every row)
__device__ inline void transpose4x4(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t (&c)[4]) {
  const uint32_t t0 = __builtin_amdgcn_perm(a1, a0, 0x05010400u);
  const uint32_t t1 = __builtin_amdgcn_perm(a1, a0, 0x07030602u);
  const uint32_t t2 = __builtin_amdgcn_perm(a3, a2, 0x05010400u);
  const uint32_t t3 = __builtin_amdgcn_perm(a3, a2, 0x07030602u);
  c[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u);
  c[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
  c[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
  c[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
}

// INT8 K without any conversion: the stored bytes ARE the A operand of v_mfma_i32_16x16x64_i8 (lane (x, g): row x,
// k = 16 g .. 16 g + 15 of a 64-wide k-block = the 16 bytes the K load of chunk c already holds). The query becomes
// two int8 planes per head, q ~= aq (q1 + q2 / 254) with aq = max|q| / 127 (error <= max|q| / 64,516 per element,
// finer than the f16 input's own 2^-11 for all but the smallest elements), so S = aq (K.q1 + K.q2 / 254) with both
// products exact in int32. Saves the byte -> f16 conversion of K (5 vector instructions per 4 elements).
typedef int i32x4 __attribute__((ext_vector_type(4)));
struct QPlanes {
  i32x4 p1, p2;
};
// 16 query values = 8 dwords of f16 pairs (already in element order) -> 16 bytes per plane
__device__ inline QPlanes quantize_q16(const uint32_t (&w)[8], const float inv) {
  QPlanes o;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    uint32_t b1 = 0u, b2 = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f16x2 h = bits_h2(w[2 * d + (j >> 1)]);
      const float t = (float)h[j & 1] * inv;
      const float q1 = rintf(t);
      const float q2 = rintf((t - q1) * 254.0f);
      b1 |= ((uint32_t)(int)q1 & 0xFFu) << (8 * j);
      b2 |= ((uint32_t)(int)q2 & 0xFFu) << (8 * j);
    }
    o.p1[d] = (int)b1;
    o.p2[d] = (int)b2;
  }
  return o;
}
__device__ inline float absmax_h2(const uint32_t wd) {
  const f16x2 h = bits_h2(wd & 0x7FFF7FFFu);
  return fmaxf((float)h[0], (float)h[1]);
}

// One wave's tile of TC tokens starting at t0 (nt valid ones, 1 <= nt <= TC) of kv head hk, batch row b:
// on return lane (x, g) holds m (log2 domain) and l of head x, and acc[c][r] = head 4 g + r, element
// d = DVN x + e(c) (see attn_tile_store) of sum_t p[t] sv[t] v_int[t, d] / svmax. s_ks / s_vs: TC floats of
// LDS each, private to this wave.
// (Requesting EVERY V step together with the K rows — one round trip for the whole tile — was measured at batch 1,
// where the launch is a single round of waves and registers are free: 10.5 us vs 9.6 us for the partial kernel,
// profiles/r02s_attn_b1_vfirst.txt. The staggered requests below stay.)
template <int KBITS, int VBITS, int TC, int HD, bool KI8 = false>
struct AttnTile {
  static_assert(HD == 64 || HD == 128, "head_dim of the MFMA kernel");
  static_assert(!KI8 || KBITS == 8, "the int8 MFMA path is for INT8 keys");
  static constexpr int NT = TC / 16;   // 16-token score tiles
  static constexpr int NS = TC / 32;   // 32-token P V steps
  static constexpr int KS = HD / 32;   // k-steps of the score product
  static constexpr int DVN = HD / 16;  // MFMAs (d values per lane) of a P V step
  // K row bytes per lane group and load: 16 (8 for INT4 at head_dim 64); loads per row and lane
  static constexpr int CBK = (HD * KBITS / 8) / 4 < 16 ? 8 : 16;
  static constexpr int NL = (HD * KBITS / 8) / (4 * CBK);
  static constexpr int EPC = CBK * 8 / KBITS;  // elements per chunk
  static constexpr int SPL = EPC / 8;          // k-steps per load
  static_assert(NL * SPL == KS, "k-step bookkeeping");
  static constexpr int VB = DVN * VBITS / 8;   // V bytes per lane and row: 8, 4 or 2
  float m, l, svmax;
  f32x4 acc[DVN];

  __device__ __forceinline__ void run(const AttnArgs& a, const uint32_t b, const uint32_t hk, const uint32_t t0, const uint32_t nt,
                                      float* s_ks, float* s_vs) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t x = lane & 15u, g = lane >> 4;

  // K / V rows are read with BUFFER loads: descriptor (uniform base of this split's rows, size =
  // the split's valid bytes) + a 32-bit offset (constant lane part + uniform row part: one v_add per
  // load instead of 64-bit address arithmetic); rows past the split's end are out of the descriptor's
  // range and read as zeros instead of being clamped or branched around (their scores are masked to
  // -inf, their P is exactly 0), and nothing outside the split's rows is ever touched.
  const bool full = nt == (uint32_t)TC;  // uniform
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(a.v + (int64_t)b * a.v_sb + (int64_t)hk * a.v_sh + (int64_t)t0 * a.v_st), 0,
      (int)(nt * (uint32_t)a.v_st), 0x00020000);
  const uint32_t v_lane = 4u * g * (uint32_t)a.v_st + (uint32_t)VB * x;
  // V rows of a 32-token step, raw bytes (two dwords hold up to 8 bytes)
  struct VRaw {
    uint32_t w0, w1;
  };
  auto load_v_step = [&](int s, VRaw (&dst)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t row = 32 * s + 16 * (j >> 2) + (j & 3);  // + 4 g from the lane offset
      // the row offset rides in the VECTOR offset (one v_add): that is the operand the range check covers
      const uint32_t off = v_lane + row * (uint32_t)a.v_st;
      if constexpr (VB == 8) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(v_rsrc, off, 0, KVQ_ATTN_KV_AUX);
        dst[j].w0 = v[0];
        dst[j].w1 = v[1];
      } else if constexpr (VB == 4) {
        dst[j].w0 = __builtin_amdgcn_raw_buffer_load_b32(v_rsrc, off, 0, KVQ_ATTN_KV_AUX);
        dst[j].w1 = 0u;
      } else {
        dst[j].w0 = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(v_rsrc, off, 0, KVQ_ATTN_KV_AUX);
        dst[j].w1 = 0u;
      }
    }
  };
  // mixed mode (INT8 K, INT4 V): every later step fits the registers the K rows free (3 waves per SIMD
  // either way); the other kind pairs keep one step of look-ahead
  // (requesting EVERY V row up front, behind the K rows, measured slower at batch 1 — 10.3 vs 9.7 us per launch:
  // the V requests queue in front of other waves' K rows, which the score product waits for)
  constexpr bool V_EARLY = KBITS == 8 && VBITS == 4;
  VRaw vr[V_EARLY ? NS : 2][8];

  // ---- S = K Q^T: tile i, token row x -----------------------------------------------------------
  f32x4 sc[NT];
  f16x8 qb[KS];
  svmax = 0.0f;
  {
    const uint8_t* kb = a.k + (int64_t)b * a.k_sb + (int64_t)hk * a.k_sh + (int64_t)t0 * a.k_st;
    uint32_t raw[NT][NL][CBK / 4];
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(kb), 0, (int)(nt * (uint32_t)a.k_st), 0x00020000);
    const uint32_t k_lane = x * (uint32_t)a.k_st + (uint32_t)CBK * g;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
#pragma unroll
      for (int c = 0; c < NL; ++c) {
        const uint32_t off = k_lane + 16 * i * (uint32_t)a.k_st + 4 * CBK * c;
        if constexpr (CBK == 16) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, off, 0, KVQ_ATTN_KV_AUX);
#pragma unroll
          for (int j = 0; j < 4; ++j) raw[i][c][j] = v[j];
        } else {
          const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(k_rsrc, off, 0, KVQ_ATTN_KV_AUX);
          raw[i][c][0] = v[0];
          raw[i][c][1] = v[1];
        }
      }
    }
    load_v_step(0, vr[0]);
    // query and scales are REQUESTED here too, before anything waits: one round trip for all of it.
    // k-step (c, wi) of this lane group covers d = EPC (4 c + g) + 8 wi + j (INT4: j in pair order)
    uint32_t w[4 * KS];
    {
      const uint32_t hx = x < a.nq ? x : 0u;  // padded heads read head 0 and are zeroed below
      const char* qp = reinterpret_cast<const char*>(a.q) + ((int64_t)b * a.q_sb + (int64_t)(hk * a.nq + hx) * a.q_sh) * 2;
#pragma unroll
      for (int c = 0; c < NL; ++c)
#pragma unroll
        for (int wi = 0; wi < SPL; ++wi) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(qp + (EPC * (4 * c + g) + 8 * wi) * 2);
#pragma unroll
          for (int j = 0; j < 4; ++j) w[4 * (c * SPL + wi) + j] = v[j];
        }
    }
    constexpr int SR = (TC + kWave - 1) / kWave;  // TC = 32: the upper half of the wave idles here
    float ksv[SR], vsv[SR];
#pragma unroll
    for (int r = 0; r < SR; ++r) {
      const uint32_t i = r * kWave + lane;
      const uint32_t ic = i < nt ? i : nt - 1u;
      ksv[r] = a.k_scale[t0 + ic];
      vsv[r] = a.v_scale[t0 + ic];
    }
    // scores are kept in the log2 domain (one v_exp_f32 per probability); V scales are staged
    // already divided by the split's largest one (the f16 pack of P needs the range, see header)
#pragma unroll
    for (int r = 0; r < SR; ++r) {
      const uint32_t i = r * kWave + lane;
      if (i >= nt) vsv[r] = 0.0f;
      svmax = fmaxf(svmax, vsv[r]);
    }
    svmax = wave_fmax(svmax);
    const float svn = svmax > 0.0f ? 1.0f / svmax : 0.0f;
#pragma unroll
    for (int r = 0; r < SR; ++r) {
      const uint32_t i = r * kWave + lane;
      if (TC >= kWave || i < (uint32_t)TC) {
        s_ks[i] = ksv[r] * (a.sm_scale * 1.44269504088896341f);
        s_vs[i] = vsv[r] * svn;
      }
    }
    // ---- Q^T operands: head x, the 8 d's of this lane group per k-step (zeros for padded heads) ----
    {
      if (a.dtype == KVQ_BF16) {
#pragma unroll
        for (int j = 0; j < 4 * KS; ++j) {
          const f16x2 h = {(f16)__uint_as_float(w[j] << 16), (f16)__uint_as_float(w[j] & 0xFFFF0000u)};
          w[j] = h2_bits(h);
        }
      }
      if constexpr (KBITS == 4) {
#pragma unroll
        for (int c = 0; c < KS; ++c) {
          const uint32_t w0 = w[4 * c], w1 = w[4 * c + 1], w2 = w[4 * c + 2], w3 = w[4 * c + 3];
          w[4 * c + 0] = __builtin_amdgcn_perm(w1, w0, 0x05040100u);  // (q0, q2)
          w[4 * c + 1] = __builtin_amdgcn_perm(w3, w2, 0x05040100u);  // (q4, q6)
          w[4 * c + 2] = __builtin_amdgcn_perm(w1, w0, 0x07060302u);  // (q1, q3)
          w[4 * c + 3] = __builtin_amdgcn_perm(w3, w2, 0x07060302u);  // (q5, q7)
        }
      }
      if (x >= a.nq) {
#pragma unroll
        for (int j = 0; j < 4 * KS; ++j) w[j] = 0u;
      }
#pragma unroll
      for (int c = 0; c < KS; ++c) qb[c] = pack_h8(w[4 * c], w[4 * c + 1], w[4 * c + 2], w[4 * c + 3]);
    }
    QPlanes qi[KI8 ? NL : 1];
    float aq = 1.0f;
    if constexpr (KI8) {  // the head's two int8 planes (w is zero for padded heads: planes 0, aq irrelevant)
      float qm = 0.0f;
#pragma unroll
      for (int j = 0; j < 4 * KS; ++j) qm = fmaxf(qm, absmax_h2(w[j]));
      qm = xor32_max(xor16_max(qm));
      aq = qm > 0.0f ? qm / 127.0f : 1.0f;
      const float inv = qm > 0.0f ? 127.0f / qm : 0.0f;
#pragma unroll
      for (int c = 0; c < NL; ++c) {
        uint32_t wc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wc[j] = w[8 * c + j];
        qi[c] = quantize_q16(wc, inv);
      }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      f32x4 c4 = {0.0f, 0.0f, 0.0f, 0.0f};
      if constexpr (KI8) {
        i32x4 c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < NL; ++c) {
          const i32x4 ka = {(int)raw[i][c][0], (int)raw[i][c][1], (int)raw[i][c][2], (int)raw[i][c][3]};
          c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ka, qi[c].p1, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ka, qi[c].p2, c2, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) c4[r] = fmaf((float)c2[r], 1.0f / 254.0f, (float)c1[r]) * aq;
        sc[i] = c4;
        continue;
      }
#pragma unroll
      for (int c = 0; c < NL; ++c) {
        if constexpr (KBITS == 8) {  // two words = 8 elements = one k-step
#pragma unroll
          for (int wi = 0; wi < SPL; ++wi) {
            uint32_t h[4];
            bytes_to_h4<128>(raw[i][c][2 * wi] ^ 0x80808080u, h[0], h[1]);
            bytes_to_h4<128>(raw[i][c][2 * wi + 1] ^ 0x80808080u, h[2], h[3]);
            c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(pack_h8(h[0], h[1], h[2], h[3]), qb[c * SPL + wi], c4, 0, 0, 0);
          }
        } else {  // one word = 8 elements = one k-step
#pragma unroll
          for (int wi = 0; wi < SPL; ++wi) {
            uint32_t h[4];
            bytes_to_h4<8>((raw[i][c][wi] >> 4) & 0x0F0F0F0Fu, h[0], h[1]);  // (e0,e2) (e4,e6)
            bytes_to_h4<8>(raw[i][c][wi] & 0x0F0F0F0Fu, h[2], h[3]);         // (e1,e3) (e5,e7)
            c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(pack_h8(h[0], h[1], h[2], h[3]), qb[c * SPL + wi], c4, 0, 0, 0);
          }
        }
      }
      sc[i] = c4;
    }
    // the K registers are free now: request the remaining V steps ahead of the softmax
    if constexpr (V_EARLY) {
#pragma unroll
      for (int s = 1; s < NS; ++s) load_v_step(s, vr[s]);
    }
  }
  // this wave's own LDS writes above are read below by other lanes of the SAME wave: the LDS queue of a
  // wave is in order, so only the compiler has to be kept from moving the reads up
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // ---- softmax over this split for head x (log2 domain); P scaled by sv / max sv ------------------
  m = -INFINITY;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const f32x4 ks = *reinterpret_cast<const f32x4*>(&s_ks[16 * i + 4 * g]);
#pragma unroll
    for (int r = 0; r < 4; ++r) sc[i][r] *= ks[r];
  }
  if (!full) {  // uniform: only a ragged last tile pays for the masks
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if ((uint32_t)(16 * i + 4 * g + r) >= nt) sc[i][r] = -INFINITY;
  }
#pragma unroll
  for (int i = 0; i < NT; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) m = fmaxf(m, sc[i][r]);
  }
  m = xor32_max(xor16_max(m));
  l = 0.0f;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const f32x4 sv = *reinterpret_cast<const f32x4*>(&s_vs[16 * i + 4 * g]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = __builtin_amdgcn_exp2f(sc[i][r] - m);  // tokens past nt: exp2(-inf) = 0
      l += p;
      sc[i][r] = p * sv[r];
    }
  }
  l = xor32_add(xor16_add(l));

  // ---- O = P V: column x <-> d = DVN x + e(c) for MFMA c ---------------------------------------------
#pragma unroll
  for (int n = 0; n < DVN; ++n) acc[n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if constexpr (!V_EARLY) {
        if (s + 1 < NS) load_v_step(s + 1, vr[(s + 1) & 1]);
      }
      const int vi = V_EARLY ? s : (s & 1);  // constant after unrolling
      // the lane's 8 token rows of this step as byte images, 4 elements per dword:
      //   head_dim 128: img[0] = elements 0..3 (INT4: 0,2,4,6), img[1] = 4..7 (INT4: 1,3,5,7)
      //   head_dim 64:  img[0] = elements 0..3 (INT4: 0,2,1,3)
      uint32_t img[DVN / 4][8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t w0 = vr[vi][j].w0, w1 = vr[vi][j].w1;
        if constexpr (HD == 128 && VBITS == 8) {
          img[0][j] = w0 ^ 0x80808080u;
          img[1][j] = w1 ^ 0x80808080u;
        } else if constexpr (HD == 128) {
          img[0][j] = (w0 >> 4) & 0x0F0F0F0Fu;
          img[1][j] = w0 & 0x0F0F0F0Fu;
        } else if constexpr (VBITS == 8) {
          img[0][j] = w0 ^ 0x80808080u;
        } else {
          img[0][j] = ((w0 >> 4) & 0x0F0Fu) | ((w0 & 0x0F0Fu) << 16);
        }
      }
      const f16x8 pa = pack_h8(Elem<KVQ_F16>::pack2(sc[2 * s][0], sc[2 * s][1]), Elem<KVQ_F16>::pack2(sc[2 * s][2], sc[2 * s][3]),
                               Elem<KVQ_F16>::pack2(sc[2 * s + 1][0], sc[2 * s + 1][1]),
                               Elem<KVQ_F16>::pack2(sc[2 * s + 1][2], sc[2 * s + 1][3]));
      constexpr int BIAS = VBITS == 8 ? 128 : 8;
#pragma unroll
      for (int half = 0; half < DVN / 4; ++half) {
        uint32_t ca[4], cb[4];
        transpose4x4(img[half][0], img[half][1], img[half][2], img[half][3], ca);  // tokens j = 0..3
        transpose4x4(img[half][4], img[half][5], img[half][6], img[half][7], cb);  // tokens j = 4..7
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          uint32_t h[4];
          bytes_to_h4<BIAS>(ca[n], h[0], h[1]);
          bytes_to_h4<BIAS>(cb[n], h[2], h[3]);
          acc[4 * half + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa, pack_h8(h[0], h[1], h[2], h[3]), acc[4 * half + n], 0, 0, 0);
        }
      }
    }
  }

  }

  // acc values of row r (head 4 g + r) in element order: o8[e] for d = DVN x + e
  __device__ __forceinline__ void ordered(const int r, const float scale, float (&o8)[DVN]) const {
#pragma unroll
    for (int c = 0; c < DVN; ++c) {
      int e = c;
      if constexpr (VBITS == 4 && HD == 128) e = c < 4 ? 2 * c : 2 * (c - 4) + 1;
      if constexpr (VBITS == 4 && HD == 64) e = c == 1 ? 2 : (c == 2 ? 1 : c);
      o8[e] = acc[c][r] * scale;
    }
  }
};

template <int KBITS, int VBITS, int TC, int HD, bool KI8 = false>
__global__ __launch_bounds__(kWave) void decode_attn_partial_mfma_k(const AttnArgs a) {
  typedef AttnTile<KBITS, VBITS, TC, HD, KI8> Tile;
  constexpr int DVN = Tile::DVN;
  __shared__ __attribute__((aligned(16))) float s_ks[TC];
  __shared__ __attribute__((aligned(16))) float s_vs[TC];
  const uint32_t lane = threadIdx.x;
  const uint32_t x = lane & 15u, g = lane >> 4;
  const uint32_t split = blockIdx.x, hk = blockIdx.y, b = blockIdx.z;
  const uint32_t t0 = split * TC;
  const uint32_t T = live_tokens(a);
  if (t0 >= T) return;  // device-side T (graph replay): this split holds nothing yet
  const uint32_t nt = T - t0 < (uint32_t)TC ? T - t0 : (uint32_t)TC;
  Tile tile;
  tile.run(a, b, hk, t0, nt, s_ks, s_vs);
  // ---- workspace: (m, l) per head, acc[heads][D] --------------------------------------------------
  if (g == 0u && x < a.nq) {
    float* o = a.ws + (((int64_t)b * a.Hq + hk * a.nq + x) * a.nsplit + split) * 2;
    o[0] = tile.m * 0.693147180559945309f;  // back to the natural-log domain the merge kernel works in
    o[1] = tile.l;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t h = 4 * g + r;
    if (h < a.nq) {
      float o8[DVN];
      tile.ordered(r, tile.svmax, o8);
      float* dst = a.ws + a.acc_off + (((int64_t)b * a.Hq + hk * a.nq + h) * a.nsplit + split) * a.D + DVN * x;
#pragma unroll
      for (int c = 0; c < DVN; c += 4) *reinterpret_cast<f32x4*>(dst + c) = f32x4{o8[c], o8[c + 1], o8[c + 2], o8[c + 3]};
    }
  }
}

// ---------------------------------------------------------------------------- streaming variant
// Larger batches (more tiles than wave slots): ONE wave walks `tpw` consecutive TC-token tiles of its
// (batch row, kv head) with the NEXT tile's K / V rows and scales requested before the current tile is
// reduced (two register sets, ping-pong), carries the softmax online across its tiles
//   m' = max(m, m_tile), alpha = 2^(m - m'), l = l alpha + sum p, acc = acc alpha (svref / svref') + P~ V
// and writes ONE partial at the end: a wave's loads and arithmetic overlap, the per-wave start-up (query
// operands, descriptors) is paid once per tpw tiles, and the split partials shrink by tpw.
// Same operand layouts as AttnTile (see its header). P is scaled by sv[t] / svref with svref the largest
// V scale seen so far (non-decreasing), the accumulator carries the matching 1 / svref.
// ROLL: the registers of the tile being reduced are re-requested piece by piece for the tile after next as soon as
// each piece has been consumed (K rows per 16-token group after its score MFMAs, V rows after their byte images are
// built, scales after they are staged): close to two tiles per wave stay in flight instead of one, which is what
// the bytes in flight the HBM needs at this bandwidth ask for (Little's law: 6 TB/s x ~5 us under load = 30 MB; one
// 12 KB tile per wave x 2048 waves is 24 MB). Batch 8, 16 K tokens: 49.4 -> 44.8 us per call.
// (Also measured: the INT4 V tile as 16-byte loads redistributed through an LDS image — 4 load instructions instead
// of 16: 2 us faster with the arithmetic removed, nothing with it, slower together with ROLL; not kept.)
// TG = 4 (int8 keys, at most 4 query heads per kv head — Llama-3 / Mistral grouping, 64-token tiles): the score product
// as ONE 16 x 16 output per int8 plane for the whole tile instead of four with 12 of 16 columns padding. Column j = (token
// group j >> 2, head j & 3): the MFMA of token group tg and d-half c multiplies that group's K rows by a query operand that
// is zero outside the four columns of tg, and all eight accumulate into the same registers (block-diagonal over the
// contraction: the same sixteen MFMAs per tile). Every lane then holds 4 real scores (head x & 3, tokens 16 (x >> 2) + 4 g
// + q) instead of 16 of which 12 are padding: a quarter of the softmax arithmetic (scale, max, exp2, sum, V-scale, f16
// pack); P reaches the P·V operand layout (row = head, 8 tokens per lane) by three DPP row rotations of two registers.
template <int KBITS, int VBITS, int TC, int HD, bool KI8 = false, bool ROLL = false, int TG = 1>
struct AttnStream {
  static_assert(TG == 1 || (TG == 4 && KI8 && TC == 64), "TG = 4: int8 keys, 64-token tiles");
  typedef AttnTile<KBITS, VBITS, TC, HD> TL;
  static constexpr int NT = TL::NT, NS = TL::NS, KS = TL::KS, DVN = TL::DVN, CBK = TL::CBK, NL = TL::NL, SPL = TL::SPL, VB = TL::VB;
  static constexpr int SR = (TC + kWave - 1) / kWave;
  struct Raw {  // one tile in flight: K rows, V rows, scales — as loaded
    uint32_t k[NT][NL][CBK / 4];
    uint32_t v[NS][8][VB == 8 ? 2 : 1];
    float ks[SR], vs[SR];
  };
  f16x8 qb[KS];
  QPlanes qi[KI8 ? NL : 1];  // INT8 K through the int8 MFMA: the query as two int8 planes, aq their scale
  QPlanes qt[TG == 4 ? 4 : 1][TG == 4 ? NL : 1];  // TG = 4: qi masked to the columns of token group tg
  float aq;
  float m, l, svref;
  f32x4 acc[DVN];

  __device__ __forceinline__ void init(const AttnArgs& a, const uint32_t b, const uint32_t hk) {
    const uint32_t lane = threadIdx.x & 63u, x = lane & 15u, g = lane >> 4;
    uint32_t w[4 * KS];
    const uint32_t xh = TG == 4 ? (x & 3u) : x;  // the head this lane's column stands for
    const uint32_t hx = xh < a.nq ? xh : 0u;     // padded heads read head 0 and are zeroed below
    const char* qp = reinterpret_cast<const char*>(a.q) + ((int64_t)b * a.q_sb + (int64_t)(hk * a.nq + hx) * a.q_sh) * 2;
#pragma unroll
    for (int c = 0; c < NL; ++c)
#pragma unroll
      for (int wi = 0; wi < SPL; ++wi) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(qp + (TL::EPC * (4 * c + g) + 8 * wi) * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) w[4 * (c * SPL + wi) + j] = v[j];
      }
    if (a.dtype == KVQ_BF16) {
#pragma unroll
      for (int j = 0; j < 4 * KS; ++j) {
        const f16x2 h = {(f16)__uint_as_float(w[j] << 16), (f16)__uint_as_float(w[j] & 0xFFFF0000u)};
        w[j] = h2_bits(h);
      }
    }
    if constexpr (KBITS == 4) {
#pragma unroll
      for (int c = 0; c < KS; ++c) {
        const uint32_t w0 = w[4 * c], w1 = w[4 * c + 1], w2 = w[4 * c + 2], w3 = w[4 * c + 3];
        w[4 * c + 0] = __builtin_amdgcn_perm(w1, w0, 0x05040100u);  // (q0, q2)
        w[4 * c + 1] = __builtin_amdgcn_perm(w3, w2, 0x05040100u);  // (q4, q6)
        w[4 * c + 2] = __builtin_amdgcn_perm(w1, w0, 0x07060302u);  // (q1, q3)
        w[4 * c + 3] = __builtin_amdgcn_perm(w3, w2, 0x07060302u);  // (q5, q7)
      }
    }
    if (xh >= a.nq) {
#pragma unroll
      for (int j = 0; j < 4 * KS; ++j) w[j] = 0u;
    }
#pragma unroll
    for (int c = 0; c < KS; ++c) qb[c] = pack_h8(w[4 * c], w[4 * c + 1], w[4 * c + 2], w[4 * c + 3]);
    aq = 1.0f;
    if constexpr (KI8) {
      float qm = 0.0f;
#pragma unroll
      for (int j = 0; j < 4 * KS; ++j) qm = fmaxf(qm, absmax_h2(w[j]));
      qm = xor32_max(xor16_max(qm));
      aq = qm > 0.0f ? qm / 127.0f : 1.0f;
      const float inv = qm > 0.0f ? 127.0f / qm : 0.0f;
#pragma unroll
      for (int c = 0; c < NL; ++c) {
        uint32_t wc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wc[j] = w[8 * c + j];
        qi[c] = quantize_q16(wc, inv);
      }
      if constexpr (TG == 4) {
        const i32x4 zero = {0, 0, 0, 0};
#pragma unroll
        for (int tg = 0; tg < 4; ++tg)
#pragma unroll
          for (int c = 0; c < NL; ++c) {
            qt[tg][c].p1 = (x >> 2) == (uint32_t)tg ? qi[c].p1 : zero;
            qt[tg][c].p2 = (x >> 2) == (uint32_t)tg ? qi[c].p2 : zero;
          }
      }
    }
    m = -INFINITY;
    l = 0.0f;
    svref = 0.0f;
#pragma unroll
    for (int n = 0; n < DVN; ++n) acc[n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }

  // where a tile's rows come from: buffer descriptors over its valid rows (nt == 0: every load returns zeros and
  // touches no memory — the "tile" past a wave's last one)
  struct Src {
    __amdgpu_buffer_rsrc_t k, v;
    uint32_t t0, nt;
  };
  __device__ __forceinline__ Src src(const AttnArgs& a, const uint32_t b, const uint32_t hk, const uint32_t t0, const uint32_t nt) const {
    Src s;
    s.k = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.k + (int64_t)b * a.k_sb + (int64_t)hk * a.k_sh + (int64_t)t0 * a.k_st), 0,
                                            (int)(nt * (uint32_t)a.k_st), 0x00020000);
    s.v = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.v + (int64_t)b * a.v_sb + (int64_t)hk * a.v_sh + (int64_t)t0 * a.v_st), 0,
                                            (int)(nt * (uint32_t)a.v_st), 0x00020000);
    s.t0 = t0;
    s.nt = nt;
    return s;
  }
  // the pieces of a tile's request; nothing waits here
  __device__ __forceinline__ void issue_k(const AttnArgs& a, const Src& s, const int i, Raw& r) const {  // 16-token group i
    const uint32_t lane = threadIdx.x & 63u, x = lane & 15u, g = lane >> 4;
    const uint32_t k_lane = x * (uint32_t)a.k_st + (uint32_t)CBK * g;
#pragma unroll
    for (int c = 0; c < NL; ++c) {
      const uint32_t off = k_lane + 16 * i * (uint32_t)a.k_st + 4 * CBK * c;
      if constexpr (CBK == 16) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(s.k, off, 0, KVQ_ATTN_KV_AUX);
#pragma unroll
        for (int j = 0; j < 4; ++j) r.k[i][c][j] = v[j];
      } else {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(s.k, off, 0, KVQ_ATTN_KV_AUX);
        r.k[i][c][0] = v[0];
        r.k[i][c][1] = v[1];
      }
    }
  }
  __device__ __forceinline__ void issue_v(const AttnArgs& a, const Src& s, const int sidx, Raw& r) const {  // 32-token step sidx
    const uint32_t lane = threadIdx.x & 63u, x = lane & 15u, g = lane >> 4;
    const uint32_t v_lane = 4u * g * (uint32_t)a.v_st + (uint32_t)VB * x;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t row = 32 * sidx + 16 * (j >> 2) + (j & 3);  // + 4 g from the lane offset
      const uint32_t off = v_lane + row * (uint32_t)a.v_st;
      if constexpr (VB == 8) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(s.v, off, 0, KVQ_ATTN_KV_AUX);
        r.v[sidx][j][0] = v[0];
        r.v[sidx][j][1] = v[1];
      } else if constexpr (VB == 4) {
        r.v[sidx][j][0] = __builtin_amdgcn_raw_buffer_load_b32(s.v, off, 0, KVQ_ATTN_KV_AUX);
      } else {
        r.v[sidx][j][0] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(s.v, off, 0, KVQ_ATTN_KV_AUX);
      }
    }
  }
  __device__ __forceinline__ void issue_scales(const AttnArgs& a, const Src& s, Raw& r) const {
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int q = 0; q < SR; ++q) {
      const uint32_t i = q * kWave + lane;
      const uint32_t ic = i < s.nt ? i : (s.nt ? s.nt - 1u : 0u);
      r.ks[q] = a.k_scale[s.t0 + ic];
      r.vs[q] = i < s.nt ? a.v_scale[s.t0 + ic] : 0.0f;
    }
  }
  __device__ __forceinline__ void issue(const AttnArgs& a, const Src& s, Raw& r) const {  // the whole tile
#pragma unroll
    for (int i = 0; i < NT; ++i) issue_k(a, s, i, r);
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) issue_v(a, s, sidx, r);
    issue_scales(a, s, r);
  }

  // fold one loaded tile into the running (m, l, acc); s_ks / s_vs: TC floats each, s_al: 16 floats (this wave's)
  // nx (ROLL): the tile after next, whose rows take over r's registers as this tile's pieces are consumed
  __device__ __forceinline__ void consume(const AttnArgs& a, const uint32_t nt, Raw& r, const Src& nx, float* s_ks, float* s_vs, float* s_al) {
    const uint32_t lane = threadIdx.x & 63u, x = lane & 15u, g = lane >> 4;
#ifdef KVQ_ATTN_CALIB  // calibration build (`make calib_attn`, never shipped): every loaded word is consumed, nothing is
    {                  // computed — the time of the kernel's own load pattern and prefetch depth (inexact results)
      uint32_t xr = 0u;
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int c = 0; c < NL; ++c)
#pragma unroll
          for (int j = 0; j < CBK / 4; ++j) xr ^= r.k[i][c][j];
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) xr ^= r.v[s][j][0] ^ (VB == 8 ? r.v[s][j][VB == 8 ? 1 : 0] : 0u);
      float fs = 0.0f;
#pragma unroll
      for (int q = 0; q < SR; ++q) fs += r.ks[q] + r.vs[q];
      acc[0][0] += __uint_as_float(xr & 0x007FFFFFu) + fs;
      m = 0.0f;
      l = 1.0f;
      svref = 1.0f;
      (void)x; (void)g; (void)s_ks; (void)s_vs; (void)s_al; (void)nt;
      if constexpr (ROLL) issue(a, nx, r);
      return;
    }
#endif
    const bool full = nt == (uint32_t)TC;  // uniform
    // ---- scales: K side with sm_scale log2(e) folded in, V side relative to the running reference ----------
    float svmax = 0.0f;
#pragma unroll
    for (int q = 0; q < SR; ++q) svmax = fmaxf(svmax, r.vs[q]);
    svmax = wave_fmax(svmax);
    const float svnew = fmaxf(svref, svmax);
    const float svn = svnew > 0.0f ? 1.0f / svnew : 0.0f;
    const float ratio = svnew > 0.0f ? svref * svn : 1.0f;  // <= 1; the first tile's accumulator is 0 anyway
    svref = svnew;
#pragma unroll
    for (int q = 0; q < SR; ++q) {
      const uint32_t i = q * kWave + lane;
      if (TC >= kWave || i < (uint32_t)TC) {
        s_ks[i] = r.ks[q] * (a.sm_scale * 1.44269504088896341f);
        s_vs[i] = r.vs[q] * svn;
      }
    }
    if constexpr (ROLL) issue_scales(a, nx, r);
    uint32_t pp[NS][4];  // P of this tile as the P·V operand: 8 tokens of head x per lane and 32-token step, f16 pairs
    float alpha;
    if constexpr (TG == 4) {
      // ---- S = K Q^T, one output for the tile: column x = (token group x >> 2, head x & 3) ---------------------------
#ifndef KVQ_ATTN_SKIP  // calibration bit mask: 1 = no score MFMAs, 2 = no P·V MFMAs, 4 = no V byte -> f16 conversion, 8 = no exp2
#define KVQ_ATTN_SKIP 0
#endif
#ifndef KVQ_TG4_CHAINS  // calibration (`make calib_tg4`): 2 = two accumulators per plane (dependent chains of 4 MFMAs instead of 8)
#define KVQ_TG4_CHAINS 1
#endif
      i32x4 c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0};
#if KVQ_TG4_CHAINS == 2
      i32x4 c1b = {0, 0, 0, 0}, c2b = {0, 0, 0, 0};
#endif
#pragma unroll
      for (int tg = 0; tg < 4; ++tg) {
#pragma unroll
        for (int c = 0; c < NL; ++c) {
          const i32x4 ka = {(int)r.k[tg][c][0], (int)r.k[tg][c][1], (int)r.k[tg][c][2], (int)r.k[tg][c][3]};
#if KVQ_TG4_CHAINS == 2
          if (tg & 1) {
            c1b = __builtin_amdgcn_mfma_i32_16x16x64_i8(ka, qt[tg][c].p1, c1b, 0, 0, 0);
            c2b = __builtin_amdgcn_mfma_i32_16x16x64_i8(ka, qt[tg][c].p2, c2b, 0, 0, 0);
            continue;
          }
#endif
#if KVQ_ATTN_SKIP & 1  // calibration (`make calib_attn_skip`, inexact): the score MFMAs replaced by one VALU op per operand word
          c1 ^= ka;
          c2 += qt[tg][c].p2;
          continue;
#endif
          c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ka, qt[tg][c].p1, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ka, qt[tg][c].p2, c2, 0, 0, 0);
        }
        if constexpr (ROLL) issue_k(a, nx, tg, r);
      }
#if KVQ_TG4_CHAINS == 2
      c1 += c1b;
      c2 += c2b;
#endif
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // s_ks / s_vs written above by this wave's lanes
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // ---- online softmax (log2 domain): 4 scores per lane, tokens tb .. tb + 3 of head x & 3 -------------------------
      const uint32_t tb = 16u * (x >> 2) + 4u * g;
      const f32x4 ks = *reinterpret_cast<const f32x4*>(&s_ks[tb]);
      float s4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) s4[q] = fmaf((float)c2[q], 1.0f / 254.0f, (float)c1[q]) * aq * ks[q];
      if (!full) {  // uniform: only a ragged last tile pays for the masks
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (tb + (uint32_t)q >= nt) s4[q] = -INFINITY;
      }
      float mt = fmaxf(fmaxf(s4[0], s4[1]), fmaxf(s4[2], s4[3]));
      // across the head's 16 lanes: the four token groups of the row (rotations by 8 and 4 lanes), the four rows
      mt = fmaxf(mt, __uint_as_float(dpp_u32_attn<0x128>(__float_as_uint(mt))));
      mt = fmaxf(mt, __uint_as_float(dpp_u32_attn<0x124>(__float_as_uint(mt))));
      mt = xor32_max(xor16_max(mt));
      const float mnew = fmaxf(m, mt);                 // finite: every tile holds >= 1 token
      alpha = __builtin_amdgcn_exp2f(m - mnew);        // first tile: 2^(-inf) = 0
      const f32x4 sv = *reinterpret_cast<const f32x4*>(&s_vs[tb]);
      float lt = 0.0f, pv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#if KVQ_ATTN_SKIP & 8
        const float p = fminf(fabsf(s4[q] - mnew), 1.0f);
#else
        const float p = __builtin_amdgcn_exp2f(s4[q] - mnew);  // tokens past nt: 2^(-inf) = 0
#endif
        lt += p;
        pv[q] = p * sv[q];
      }
      lt += __uint_as_float(dpp_u32_attn<0x128>(__float_as_uint(lt)));
      lt += __uint_as_float(dpp_u32_attn<0x124>(__float_as_uint(lt)));
      lt = xor32_add(xor16_add(lt));
      l = l * alpha + lt;
      m = mnew;
      // P·V operand of lane x (row = head x for x < 4; the other rows are padding): token group i comes from lane x + 4 i
      // of the row (row_ror:n hands lane i the value of lane i - n)
      const uint32_t p01 = Elem<KVQ_F16>::pack2(pv[0], pv[1]), p23 = Elem<KVQ_F16>::pack2(pv[2], pv[3]);
      pp[0][0] = p01;
      pp[0][1] = p23;
      pp[0][2] = dpp_u32_attn<0x12C>(p01);
      pp[0][3] = dpp_u32_attn<0x12C>(p23);
      pp[1][0] = dpp_u32_attn<0x128>(p01);
      pp[1][1] = dpp_u32_attn<0x128>(p23);
      pp[1][2] = dpp_u32_attn<0x124>(p01);
      pp[1][3] = dpp_u32_attn<0x124>(p23);
    } else {
    // ---- S = K Q^T ------------------------------------------------------------------------------------------
    f32x4 sc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      f32x4 c4 = {0.0f, 0.0f, 0.0f, 0.0f};
      if constexpr (KI8) {
        i32x4 c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < NL; ++c) {
          const i32x4 ka = {(int)r.k[i][c][0], (int)r.k[i][c][1], (int)r.k[i][c][2], (int)r.k[i][c][3]};
          c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ka, qi[c].p1, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ka, qi[c].p2, c2, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) c4[q] = fmaf((float)c2[q], 1.0f / 254.0f, (float)c1[q]) * aq;
        sc[i] = c4;
        if constexpr (ROLL) issue_k(a, nx, i, r);
        continue;
      }
#pragma unroll
      for (int c = 0; c < NL; ++c) {
#pragma unroll
        for (int wi = 0; wi < SPL; ++wi) {
          uint32_t h[4];
          if constexpr (KBITS == 8) {
            bytes_to_h4<128>(r.k[i][c][2 * wi] ^ 0x80808080u, h[0], h[1]);
            bytes_to_h4<128>(r.k[i][c][2 * wi + 1] ^ 0x80808080u, h[2], h[3]);
          } else {
            bytes_to_h4<8>((r.k[i][c][wi] >> 4) & 0x0F0F0F0Fu, h[0], h[1]);  // (e0,e2) (e4,e6)
            bytes_to_h4<8>(r.k[i][c][wi] & 0x0F0F0F0Fu, h[2], h[3]);         // (e1,e3) (e5,e7)
          }
          c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(pack_h8(h[0], h[1], h[2], h[3]), qb[c * SPL + wi], c4, 0, 0, 0);
        }
      }
      sc[i] = c4;
      if constexpr (ROLL) issue_k(a, nx, i, r);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // s_ks / s_vs written above by this wave's lanes
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- online softmax (log2 domain) -------------------------------------------------------------------------
    float mt = -INFINITY;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const f32x4 ks = *reinterpret_cast<const f32x4*>(&s_ks[16 * i + 4 * g]);
#pragma unroll
      for (int q = 0; q < 4; ++q) sc[i][q] *= ks[q];
    }
    if (!full) {  // uniform: only a ragged last tile pays for the masks
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if ((uint32_t)(16 * i + 4 * g + q) >= nt) sc[i][q] = -INFINITY;
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) mt = fmaxf(mt, sc[i][q]);
    }
    mt = xor32_max(xor16_max(mt));
    const float mnew = fmaxf(m, mt);                          // finite: every tile holds >= 1 token
    alpha = __builtin_amdgcn_exp2f(m - mnew);     // first tile: 2^(-inf) = 0
    float lt = 0.0f;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const f32x4 sv = *reinterpret_cast<const f32x4*>(&s_vs[16 * i + 4 * g]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float p = __builtin_amdgcn_exp2f(sc[i][q] - mnew);  // tokens past nt: 2^(-inf) = 0
        lt += p;
        sc[i][q] = p * sv[q];
      }
    }
    lt = xor32_add(xor16_add(lt));
    l = l * alpha + lt;
    m = mnew;
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      pp[sidx][0] = Elem<KVQ_F16>::pack2(sc[2 * sidx][0], sc[2 * sidx][1]);
      pp[sidx][1] = Elem<KVQ_F16>::pack2(sc[2 * sidx][2], sc[2 * sidx][3]);
      pp[sidx][2] = Elem<KVQ_F16>::pack2(sc[2 * sidx + 1][0], sc[2 * sidx + 1][1]);
      pp[sidx][3] = Elem<KVQ_F16>::pack2(sc[2 * sidx + 1][2], sc[2 * sidx + 1][3]);
    }
    }
    // the accumulator rows of this lane are heads 4 g + q: their alpha lives in lanes x = 4 g + q
    if (g == 0u) s_al[x] = alpha * ratio;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const f32x4 al = *reinterpret_cast<const f32x4*>(&s_al[4 * g]);
#pragma unroll
    for (int n = 0; n < DVN; ++n) acc[n] *= al;
    // ---- O += P V ---------------------------------------------------------------------------------------------
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      uint32_t img[DVN / 4][8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t w0 = r.v[sidx][j][0];
        if constexpr (HD == 128 && VBITS == 8) {
          img[0][j] = w0 ^ 0x80808080u;
          img[1][j] = r.v[sidx][j][VB == 8 ? 1 : 0] ^ 0x80808080u;
        } else if constexpr (HD == 128) {
          img[0][j] = (w0 >> 4) & 0x0F0F0F0Fu;
          img[1][j] = w0 & 0x0F0F0F0Fu;
        } else if constexpr (VBITS == 8) {
          img[0][j] = w0 ^ 0x80808080u;
        } else {
          img[0][j] = ((w0 >> 4) & 0x0F0Fu) | ((w0 & 0x0F0Fu) << 16);
        }
      }
      if constexpr (ROLL) issue_v(a, nx, sidx, r);
      const f16x8 pa = pack_h8(pp[sidx][0], pp[sidx][1], pp[sidx][2], pp[sidx][3]);
      constexpr int BIAS = VBITS == 8 ? 128 : 8;
#pragma unroll
      for (int half = 0; half < DVN / 4; ++half) {
        uint32_t ca[4], cb[4];
        transpose4x4(img[half][0], img[half][1], img[half][2], img[half][3], ca);  // tokens j = 0..3
        transpose4x4(img[half][4], img[half][5], img[half][6], img[half][7], cb);  // tokens j = 4..7
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          uint32_t h[4];
#if KVQ_ATTN_SKIP & 4
          h[0] = ca[n]; h[1] = cb[n]; h[2] = ca[n] >> 1; h[3] = cb[n] >> 1;
#else
          bytes_to_h4<BIAS>(ca[n], h[0], h[1]);
          bytes_to_h4<BIAS>(cb[n], h[2], h[3]);
#endif
#if KVQ_ATTN_SKIP & 2
          acc[4 * half + n][0] += __uint_as_float((h[0] ^ h[1] ^ h[2] ^ h[3] ^ pp[sidx][n]) & 0x3FFFFFFFu);
#else
          acc[4 * half + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa, pack_h8(h[0], h[1], h[2], h[3]), acc[4 * half + n], 0, 0, 0);
#endif
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the next tile's staging writes stay below this tile's LDS reads
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }

  // ---- TG = 4, scales straight from the LDS slot (the LDS-staged kernel) -------------------------------------------------
  // The same tile reduction with the wave's serial chain cut down: what a tile costs one wave per SIMD is less its arithmetic
  // (builds without any MFMA, V conversion and exp2 run 1 us faster out of 40, profiles/r03skip_*) than the LDS round trips
  // and lane exchanges between its steps. Here
  //   * the K / V scales are read as f32x4 from the slot the DMA left them in (no staging write, no wave barrier);
  //   * the V reference scale is the wave's own maximum, taken once before the first tile (`svref`, `svn` fixed by the
  //     kernel): no per-tile wave maximum, no rescale ratio;
  //   * l is summed per lane and reduced across the head's lanes once, at the end (`reduce_l`);
  //   * alpha reaches the accumulator rows by v_readlane (heads 0..3 live in lanes 0..3; rows 4..15 are padding) instead of
  //     an LDS round trip, and the rescale is skipped while no head's maximum moved;
  //   * nothing is written to LDS, so no fence closes the tile.
  // sc_slot: the slot's 64 K scales followed by its 64 V scales. Same sums as consume() up to the order l is added in.
  float svn;  // 1 / svref (0 when every V scale of the wave is 0)
  __device__ __forceinline__ void consume_direct(const AttnArgs& a, const uint32_t nt, Raw& r, const float* sc_slot) {
    static_assert(TG == 4, "the one-output score product");
    const uint32_t lane = threadIdx.x & 63u, x = lane & 15u, g = lane >> 4;
    const bool full = nt == (uint32_t)TC;  // uniform
    i32x4 c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0};
#pragma unroll
    for (int tg = 0; tg < 4; ++tg) {
#pragma unroll
      for (int c = 0; c < NL; ++c) {
        const i32x4 ka = {(int)r.k[tg][c][0], (int)r.k[tg][c][1], (int)r.k[tg][c][2], (int)r.k[tg][c][3]};
        c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ka, qt[tg][c].p1, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ka, qt[tg][c].p2, c2, 0, 0, 0);
      }
    }
    const uint32_t tb = 16u * (x >> 2) + 4u * g;
    const f32x4 ks = *reinterpret_cast<const f32x4*>(sc_slot + tb);
    const f32x4 sv = *reinterpret_cast<const f32x4*>(sc_slot + 64 + tb);
    const float kq = aq * (a.sm_scale * 1.44269504088896341f);
    float s4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) s4[q] = fmaf((float)c2[q], 1.0f / 254.0f, (float)c1[q]) * (kq * ks[q]);
    if (!full) {  // uniform: only a ragged last tile pays for the masks (its rows past nt hold an older tile's bytes)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (tb + (uint32_t)q >= nt) s4[q] = -INFINITY;
    }
    float mt = fmaxf(fmaxf(s4[0], s4[1]), fmaxf(s4[2], s4[3]));
    mt = fmaxf(mt, __uint_as_float(dpp_u32_attn<0x128>(__float_as_uint(mt))));
    mt = fmaxf(mt, __uint_as_float(dpp_u32_attn<0x124>(__float_as_uint(mt))));
    mt = xor32_max(xor16_max(mt));
    const float mnew = fmaxf(m, mt);                       // finite: every tile holds >= 1 token
    const float alpha = __builtin_amdgcn_exp2f(m - mnew);  // first tile: 2^(-inf) = 0
    m = mnew;
    float lt = 0.0f, pv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float p = __builtin_amdgcn_exp2f(s4[q] - mnew);  // tokens past nt: 2^(-inf) = 0
      lt += p;
      pv[q] = p * (full || tb + (uint32_t)q < nt ? sv[q] * svn : 0.0f);
    }
    l = l * alpha + lt;  // this lane's share of the head's sum (reduce_l() at the end)
    const uint32_t p01 = Elem<KVQ_F16>::pack2(pv[0], pv[1]), p23 = Elem<KVQ_F16>::pack2(pv[2], pv[3]);
    uint32_t pp[NS][4];
    pp[0][0] = p01;
    pp[0][1] = p23;
    pp[0][2] = dpp_u32_attn<0x12C>(p01);
    pp[0][3] = dpp_u32_attn<0x12C>(p23);
    pp[1][0] = dpp_u32_attn<0x128>(p01);
    pp[1][1] = dpp_u32_attn<0x128>(p23);
    pp[1][2] = dpp_u32_attn<0x124>(p01);
    pp[1][3] = dpp_u32_attn<0x124>(p23);
    // accumulator rows of this lane = heads 4 g + q; heads 0..3 (the real ones) keep their alpha in lanes 0..3
    f32x4 al;
#pragma unroll
    for (int q = 0; q < 4; ++q) al[q] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(alpha), q));
    if (!(al[0] == 1.0f && al[1] == 1.0f && al[2] == 1.0f && al[3] == 1.0f)) {  // uniform
#pragma unroll
      for (int n = 0; n < DVN; ++n) acc[n] *= al;
    }
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      uint32_t img[DVN / 4][8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t w0 = r.v[sidx][j][0];
        if constexpr (HD == 128 && VBITS == 8) {
          img[0][j] = w0 ^ 0x80808080u;
          img[1][j] = r.v[sidx][j][VB == 8 ? 1 : 0] ^ 0x80808080u;
        } else if constexpr (HD == 128) {
          img[0][j] = (w0 >> 4) & 0x0F0F0F0Fu;
          img[1][j] = w0 & 0x0F0F0F0Fu;
        } else if constexpr (VBITS == 8) {
          img[0][j] = w0 ^ 0x80808080u;
        } else {
          img[0][j] = ((w0 >> 4) & 0x0F0Fu) | ((w0 & 0x0F0Fu) << 16);
        }
      }
      const f16x8 pa = pack_h8(pp[sidx][0], pp[sidx][1], pp[sidx][2], pp[sidx][3]);
      constexpr int BIAS = VBITS == 8 ? 128 : 8;
#pragma unroll
      for (int half = 0; half < DVN / 4; ++half) {
        uint32_t ca[4], cb[4];
        transpose4x4(img[half][0], img[half][1], img[half][2], img[half][3], ca);  // tokens j = 0..3
        transpose4x4(img[half][4], img[half][5], img[half][6], img[half][7], cb);  // tokens j = 4..7
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          uint32_t h[4];
          bytes_to_h4<BIAS>(ca[n], h[0], h[1]);
          bytes_to_h4<BIAS>(cb[n], h[2], h[3]);
          acc[4 * half + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa, pack_h8(h[0], h[1], h[2], h[3]), acc[4 * half + n], 0, 0, 0);
        }
      }
    }
  }
  // consume_direct's per-lane shares of l -> the head's sum, in every lane of the head
  __device__ __forceinline__ void reduce_l() {
    l += __uint_as_float(dpp_u32_attn<0x128>(__float_as_uint(l)));
    l += __uint_as_float(dpp_u32_attn<0x124>(__float_as_uint(l)));
    l = xor32_add(xor16_add(l));
  }
};

// Two tiles' worth of raw rows live in registers: 64-token tiles fit 2 waves per SIMD (<= 256 VGPRs),
// 32-token tiles 3 (<= 168); the second launch-bounds argument is waves per SIMD for one-wave workgroups.
template <int KBITS, int VBITS, int TC, int HD, bool KI8 = false, bool ROLL = false>
__global__ __launch_bounds__(kWave, TC >= 64 ? 2 : 3) void decode_attn_stream_mfma_k(const AttnArgs a, const uint32_t tpw) {
  typedef AttnStream<KBITS, VBITS, TC, HD, KI8, ROLL> ST;
  constexpr int DVN = ST::DVN;
  __shared__ __attribute__((aligned(16))) float s_ks[TC];
  __shared__ __attribute__((aligned(16))) float s_vs[TC];
  __shared__ __attribute__((aligned(16))) float s_al[16];
  const uint32_t lane = threadIdx.x, x = lane & 15u, g = lane >> 4;
  const uint32_t split = blockIdx.x, hk = blockIdx.y, b = blockIdx.z;
  const uint32_t ntiles = (a.T + (uint32_t)TC - 1u) / (uint32_t)TC;
  const uint32_t first = split * tpw;
  const uint32_t last = first + tpw < ntiles ? first + tpw : ntiles;  // host: first < ntiles for every split
  ST st;
  // The wave's k-th tile, or the empty tile (nothing is read) past its last one. (Starting every wave at a different
  // tile of its range — the online softmax does not care about the order — measured 1-2 % slower: the waves' regions
  // do not alias onto the same HBM channels as they are.)
  const uint32_t n = last - first;
  auto tile = [&](uint32_t k) {
    const uint32_t tt = first + (k < n ? k : 0u);
    const uint32_t cnt = a.T - tt * (uint32_t)TC < (uint32_t)TC ? a.T - tt * (uint32_t)TC : (uint32_t)TC;
    return st.src(a, b, hk, tt * (uint32_t)TC, k < n ? cnt : 0u);
  };
  typename ST::Raw ra, rb;
  if constexpr (ROLL) {  // two tiles requested up front; every consume re-requests its registers for the tile after next
    st.issue(a, tile(0u), ra);
    st.issue(a, tile(1u), rb);
    st.init(a, b, hk);
    for (uint32_t k = 0; k < n; k += 2u) {
      st.consume(a, tile(k).nt, ra, tile(k + 2u), s_ks, s_vs, s_al);
      if (k + 1u < n) st.consume(a, tile(k + 1u).nt, rb, tile(k + 3u), s_ks, s_vs, s_al);
    }
  } else {
    st.issue(a, tile(0u), ra);
    st.init(a, b, hk);
    for (uint32_t k = 0; k < n; k += 2u) {
      const bool two = k + 1u < n;  // uniform
      if (two) st.issue(a, tile(k + 1u), rb);
      st.consume(a, tile(k).nt, ra, tile(k), s_ks, s_vs, s_al);
      if (two) {
        if (k + 2u < n) st.issue(a, tile(k + 2u), ra);
        st.consume(a, tile(k + 1u).nt, rb, tile(k + 1u), s_ks, s_vs, s_al);
      }
    }
  }
  // ---- workspace: (m, l) per head, acc[heads][D] — the layout decode_attn_merge_k reads ---------------------
  if (g == 0u && x < a.nq) {
    float* o = a.ws + (((int64_t)b * a.Hq + hk * a.nq + x) * a.nsplit + split) * 2;
    o[0] = st.m * 0.693147180559945309f;
    o[1] = st.l;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t h = 4 * g + q;
    if (h < a.nq) {
      float o8[DVN];
#pragma unroll
      for (int c = 0; c < DVN; ++c) {
        int e = c;
        if constexpr (VBITS == 4 && HD == 128) e = c < 4 ? 2 * c : 2 * (c - 4) + 1;
        if constexpr (VBITS == 4 && HD == 64) e = c == 1 ? 2 : (c == 2 ? 1 : c);
        o8[e] = st.acc[c][q] * st.svref;
      }
      float* dst = a.ws + a.acc_off + (((int64_t)b * a.Hq + hk * a.nq + h) * a.nsplit + split) * a.D + DVN * x;
#pragma unroll
      for (int c = 0; c < DVN; c += 4) *reinterpret_cast<f32x4*>(dst + c) = f32x4{o8[c], o8[c + 1], o8[c + 2], o8[c + 3]};
    }
  }
}

#if KVQ_AB
// ---------------------------------------------------------------------------- merge inside the partial launch (round 4)
// A-B builds only (attn_fold): MEASURED SLOWER than partial + merge launches (batch 8, 16 K tokens, per layer call, one box,
// profiles/r04b_*): two launches 40.1 us; write-through stores + ticket and NO merge 39.2; this code 42.0 without the agent
// acquire, 43.4 with it — the separate merge launch costs 0.9 us more than not merging at all (its dispatch overlaps the
// partial kernel's drain), one wave merging four heads behind its ticket costs 2.8.
// The separate merge launch costs a kernel boundary (1.5-1.9 us) plus its own round trip although it moves 2.4 MB. Here the
// LAST wave of a (batch row, kv head) to finish does that head group's merge itself (split-K ticket, cdna_hip_programming.md
// section 5 item 2 / Guideline 16, counter form):
//   every wave   partial stores WRITE-THROUGH (buffer stores, aux 16 = sc1: the bytes leave this XCD's L2), then
//                s_waitcnt vmcnt(0) (they have), then ONE relaxed agent-scope fetch_add on the group's word
//   last ticket  (the add returned nsplit - 1; no wave ever waits or spins) resets the word for the next launch, ONE
//                agent-scope acquire (this CU's L1 may hold nothing of the partials: nobody read them in this launch, but the
//                always-valid form is kept: the hand-off table's one-workgroup-per-CU cell is not ours), then reads every
//                partial with sc1 buffer loads (L1 bypassed) and merges
// Nothing depends on dispatch order, timing or placement: waves only ever add and leave; exactly one add per launch and
// group returns nsplit - 1 provided the word was zero at launch (the host call's memset node; the reset keeps it so
// between the launches of one host call).
#ifndef KVQ_FOLD_ACQ  // calibration (`make calib_fold`): 0 = no agent-scope acquire in front of the merge's sc1 loads
#define KVQ_FOLD_ACQ 1
#endif
__device__ __forceinline__ bool arrive_is_last(uint32_t* word, const uint32_t narrive) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every write-through store of this wave has completed
  uint32_t old = 0u;
  if ((threadIdx.x & 63u) == 0u) old = __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
  if (old != narrive - 1u) return false;
  if ((threadIdx.x & 63u) == 0u) __hip_atomic_store(word, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if KVQ_FOLD_ACQ
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the invalidate has completed before the first partial is requested
#else
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // no instruction: keeps the compiler from moving the sc1 loads above the ticket
#endif
  return true;
}

// The merge of ONE kv head's query heads (at most NH, head_dim 128, at most 16 splits) by ONE wave, every operand requested
// up front as 16-byte sc1 loads: lane = (half h2, dl): the four elements d = 4 dl .. 4 dl + 3 of the eight splits whose
// bit 2 is h2 (so that the block kernel's split groups k and k + 8 sit in one lane). Weights by v_readlane. The arithmetic is
// decode_attn_merge_fast_k's operand for operand and in its order (o_k = x_k w_k + x_(k+8) w_(k+8); t = o_0 + ... + o_7 with
// o_4 .. o_7 fetched from the other half; the new token's fma; the division): equal bits, tested against the two-launch path.
template <int NH>
__device__ __forceinline__ void merge_group_one_wave(const AttnArgs& a, const bool has_new, const uint32_t hk, const uint32_t b) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr uint32_t D = 128;
  const uint32_t lane = threadIdx.x & 63u, h2 = lane >> 5, dl = lane & 31u;
  const uint32_t nb = a.nsplit;  // 1 ... 16 (host); every split is live (no device-side token count on this path)
  const uint32_t ns = nb;
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(a.ws, 0, (int)a.ws_bytes, 0x00020000);
  u32x2 ml_raw[NH];
  u32x4 x[NH][8];
  uint16_t qraw[NH][2], kraw[2], vraw[4];
#pragma unroll
  for (int j = 0; j < NH; ++j) {
    const uint32_t hq = hk * a.nq + ((uint32_t)j < a.nq ? (uint32_t)j : a.nq - 1u);  // heads past nq: a valid row, result dropped
    const uint32_t row0 = (b * a.Hq + hq) * nb;
    ml_raw[j] = __builtin_amdgcn_raw_buffer_load_b64(wr, (row0 + (lane < nb ? lane : nb - 1u)) * 8u, 0, 16);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t sp = (uint32_t)(u & 3) + 4u * h2 + 8u * (uint32_t)(u >> 2);
      const uint32_t sc = sp < nb ? sp : nb - 1u;
      x[j][u] = __builtin_amdgcn_raw_buffer_load_b128(wr, ((uint32_t)a.acc_off + (row0 + sc) * D + 4u * dl) * 4u, 0, 16);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
      qraw[j][h] = reinterpret_cast<const uint16_t*>(a.q)[(int64_t)b * a.q_sb + (int64_t)hq * a.q_sh + 64 * h + lane];
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) kraw[h] = reinterpret_cast<const uint16_t*>(a.kn)[(int64_t)b * a.kn_sb + (int64_t)hk * a.kn_sh + 64 * h + lane];
#pragma unroll
  for (int e = 0; e < 4; ++e) vraw[e] = reinterpret_cast<const uint16_t*>(a.vn)[(int64_t)b * a.vn_sb + (int64_t)hk * a.vn_sh + 4u * dl + e];
  __builtin_amdgcn_sched_barrier(0);
  auto widen = [&](uint16_t h) { return a.dtype == KVQ_F16 ? Elem<KVQ_F16>::widen(h) : Elem<KVQ_BF16>::widen(h); };
  // the NH heads in LOCKSTEP (no branch between them: one wave's dependent reductions of one head issue under another's)
  float w[NH], w_new[NH], inv[NH];
#pragma unroll
  for (int j = 0; j < NH; ++j) {
    f32x2 mlv = {__uint_as_float(ml_raw[j][0]), __uint_as_float(ml_raw[j][1])};
    if (lane >= ns) mlv = f32x2{-INFINITY, 0.0f};
    const float p0 = wave_fsum(widen(has_new ? qraw[j][0] : (uint16_t)0) * widen(kraw[0]));
    const float p1 = wave_fsum(widen(has_new ? qraw[j][1] : (uint16_t)0) * widen(kraw[1]));
    const float s_tok = has_new ? ((p0 + p1) + (0.0f + 0.0f)) * a.sm_scale : -INFINITY;
    const float M = fmaxf(wave_fmax(mlv[0]), s_tok);
    w[j] = lane < ns ? __expf(mlv[0] - M) : 0.0f;
    const float lw = wave_fsum(lane < ns ? mlv[1] * w[j] : 0.0f);
    w_new[j] = has_new ? __expf(s_tok - M) : 0.0f;
    const float L = ((lw + 0.0f) + (0.0f + 0.0f)) + w_new[j];
    inv[j] = 1.0f / L;
  }
  float o[NH][4][4], oo[NH][4][4];  // [head][split group kk of this half: k = kk + 4 h2][element]; oo: the other half's
#pragma unroll
  for (int j = 0; j < NH; ++j) {
    float wt[16];
#pragma unroll
    for (int sp = 0; sp < 16; ++sp) wt[sp] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(w[j]), sp));
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const uint32_t k = (uint32_t)kk + 4u * h2;
      const float w_lo = h2 ? wt[kk + 4] : wt[kk], w_hi = h2 ? wt[kk + 12] : wt[kk + 8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = 0.0f;
        const float a0 = v + __uint_as_float(x[j][kk][e]) * w_lo;
        v = k < ns ? a0 : v;
        const float a1 = v + __uint_as_float(x[j][kk + 4][e]) * w_hi;
        v = k + 8u < ns ? a1 : v;
        o[j][kk][e] = v;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NH; ++j)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int e = 0; e < 4; ++e) oo[j][kk][e] = __shfl_xor(o[j][kk][e], 32);
#pragma unroll
  for (int j = 0; j < NH; ++j) {
    // lanes 0..31 hold groups 0..3 and have fetched 4..7: the block kernel's final row sum, in its order
    if (h2 == 0u && (uint32_t)j < a.nq) {
      const uint32_t hq = hk * a.nq + (uint32_t)j;
      uint16_t ob[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = 0.0f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) t += o[j][kk][e];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) t += oo[j][kk][e];
        if (has_new) t = fmaf(w_new[j], widen(vraw[e]), t);
        t *= inv[j];
        if (a.dtype == KVQ_F16) {
          const f16 hv = (f16)t;
          __builtin_memcpy(&ob[e], &hv, 2);
        } else {
          const __bf16 bv = (__bf16)t;
          __builtin_memcpy(&ob[e], &bv, 2);
        }
      }
      uint16_t* dst = reinterpret_cast<uint16_t*>(a.out) + (int64_t)b * a.o_sb + (int64_t)hq * a.o_sh + 4u * dl;
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[e] = ob[e];
    }
  }
}
#endif  // KVQ_AB (merge inside the partial launch)

// ---------------------------------------------------------------------------- LDS-staged streaming variant
// The streaming kernel above asks HBM for its rows in MFMA-operand shape: a K load instruction touches sixteen rows,
// 64 B of each (lane (x, g): 16 B at row x, byte 16 g), a V load four rows — half-line and sub-line pieces whose
// address processing (TA) and DRAM efficiency cap the loads-only build of that kernel at 5.6 TB/s where contiguous
// reads reach 7.0 (profiles/r02ab, r02ae). Here a 64-token tile — one CONTIGUOUS run of the store for a (batch row,
// kv head): 8 KiB of INT8 keys, 4 KiB of INT4 values — travels as whole 1 KiB wave requests straight into LDS
// (`buffer_load_dwordx4 ... lds`: LDS-DMA, no VGPRs, range-checked by the descriptor like every row load in this file),
// NB - 1 tiles ahead of the one being reduced in a ring of NB LDS slots, and the MFMA operand fragments are read from
// LDS in exactly the register layout AttnStream::consume already takes (same arithmetic, same operand order: results
// are bit-identical to the streaming kernel's). In flight per wave: (NB - 1) x 12.5 KiB held by LDS instead of
// registers; the K image is XOR-swizzled on the SOURCE address (LDS-DMA writes lane-linear) so that the ds_read_b128
// fragment reads are bank-conflict free: slot s of row r holds 16-byte chunk s ^ f(r),
//   INT8 keys (8 chunks per row): f(r) = (r >> 1) & 7        INT4 keys (4 chunks per row): f(r) = r & 8 ? 3 : 0
// (derivation: the four 16-lane groups of ds_read_b128, MI355X_MICROARCH.md LDS table). The V rows are read as the
// streaming kernel reads them from memory (4 or 8 bytes per lane; rows r and r + 4 share a bank half: 2-way, on an
// LDS that is far from busy). Completion: every request of a tile is an LDS-DMA, so `s_waitcnt vmcnt(OPS x (NB-1))`
// is "tile k has landed" (loads retire in order); tiles past the wave's last are requested with an empty descriptor
// (nothing is read) so that the count is the same in every iteration.
template <int KCPR>  // 16-byte chunks per stored key row: 8 (128 B: INT8 at head_dim 128) or 4 (64 B: INT4 at 128, INT8 at 64)
__device__ inline uint32_t k_swizzle(uint32_t row) {
  static_assert(KCPR == 8 || KCPR == 4, "key rows of 128 or 64 bytes");
  if constexpr (KCPR == 8) return (row >> 1) & 7u;
  else return (row & 8u) ? 3u : 0u;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {  // s_waitcnt vmcnt(N) only (expcnt / lgkmcnt fields at their maxima)
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

// MFMA operand fragments of one tile out of its LDS image (K rows swizzled as k_swizzle says, V rows plain), in
// AttnStream::Raw's register layout: lane (x, g) takes the 16 K bytes of row 16 i + x at chunk 4 c + g, and the V bytes
// of its eight token rows per 32-token step.
template <int KBITS, int VBITS, int TC, bool KI8, class RAW, int HD = 128>
__device__ __forceinline__ void read_fragments(const uint8_t* img, RAW& r) {  // RAW: AttnStream<...>::Raw of any TG
  typedef AttnStream<KBITS, VBITS, TC, HD, KI8, false> ST;
  constexpr int NT = ST::NT, NS = ST::NS, NL = ST::NL, VB = ST::VB;
  constexpr int KROW = HD * KBITS / 8, VROW = HD * VBITS / 8;
  const uint32_t lane = threadIdx.x & 63u, x = lane & 15u, g = lane >> 4;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const uint32_t row = 16u * i + x;
#pragma unroll
    for (int c = 0; c < NL; ++c) {
      const u32x4 w = *reinterpret_cast<const u32x4*>(img + row * KROW + (((4u * c + g) ^ k_swizzle<KROW / 16>(row)) << 4));
#pragma unroll
      for (int j = 0; j < 4; ++j) r.k[i][c][j] = w[j];
    }
  }
  const uint8_t* vimg = img + TC * KROW;
#pragma unroll
  for (int sidx = 0; sidx < NS; ++sidx) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t row = 32u * sidx + 16u * (j >> 2) + (j & 3) + 4u * g;
      if constexpr (VB == 8) {
        const u32x2 w = *reinterpret_cast<const u32x2*>(vimg + row * VROW + 8u * x);
        r.v[sidx][j][0] = w[0];
        r.v[sidx][j][VB == 8 ? 1 : 0] = w[1];
      } else if constexpr (VB == 4) {
        r.v[sidx][j][0] = *reinterpret_cast<const uint32_t*>(vimg + row * VROW + 4u * x);
      } else {  // head_dim 64, INT4 values: 2 bytes (4 elements) per lane and row
        static_assert(VB == 2, "V bytes per lane and row");
        r.v[sidx][j][0] = (uint32_t)*reinterpret_cast<const uint16_t*>(vimg + row * VROW + 2u * x);
      }
    }
  }
}

template <int KBITS, int VBITS, int TC, bool KI8, int NB, int TG = 1, int HD = 128>
__global__ __launch_bounds__(kWave, TC >= 64 ? 2 : 3) void decode_attn_lds_mfma_k(const AttnArgs a, const uint32_t tpw) {
  // HD = 64 (round 4; INT8 keys only: 64-byte key rows = the 4-chunk image of INT4 keys at head_dim 128; values of 64 / 32
  // bytes per row): a 64-token tile is 4 + 2 (INT4 values) or 4 + 4 requests of 1 KiB + the two scale rows, a ring slot 6.5 / 8.5 KiB
  static_assert(HD == 128 || (HD == 64 && KBITS == 8), "head_dim 128, or 64 with INT8 keys");
  typedef AttnStream<KBITS, VBITS, TC, HD, KI8, false, TG> ST;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  constexpr int DVN = ST::DVN;
  static_assert(ST::CBK == 16, "16-byte K fragments (head_dim 128, or 64 with INT8 keys)");
  constexpr int KROW = HD * KBITS / 8, VROW = HD * VBITS / 8;   // stored bytes per token row
  constexpr int KCPR = KROW / 16, VCPR = VROW / 16;              // 16-byte chunks per row
  constexpr int KOPS = TC * KROW / 1024, VOPS = TC * VROW / 1024;  // 1 KiB wave requests per tile
  constexpr int OPS = KOPS + VOPS + 2;                           // + the two scale rows
  constexpr int SLOT = TC * (KROW + VROW) + 512;                 // K image, V image, 64 + 64 scale floats
  static_assert(NB >= 1 && OPS * (NB - 1) < 64, "the ring's requests must fit the vmcnt counter");
  extern __shared__ __attribute__((aligned(16))) uint8_t ring[];  // NB slots
  __shared__ __attribute__((aligned(16))) float s_ks[TC];
  __shared__ __attribute__((aligned(16))) float s_vs[TC];
  __shared__ __attribute__((aligned(16))) float s_al[16];
  const uint32_t lane = threadIdx.x, x = lane & 15u, g = lane >> 4;
  const uint32_t split = blockIdx.x, hk = blockIdx.y, b = blockIdx.z;
  const uint32_t ntiles = (a.T + (uint32_t)TC - 1u) / (uint32_t)TC;
  // which tiles are this wave's: a contiguous run of tpw tiles (a.lds != 3), or every nsplit-th tile (a.lds == 3: the
  // waves of a (batch row, kv head) then read ADJACENT tiles at about the same time, as neighbouring workgroups of the
  // dequantise kernels do; the online softmax does not care about the order)
  const bool strided = a.lds == 3u;
  const uint32_t first = strided ? split : split * tpw;
  const uint32_t step = strided ? a.nsplit : 1u;
  uint32_t n;
  if (strided) n = first < ntiles ? (ntiles - first + step - 1u) / step : 0u;
  else n = (first + tpw < ntiles ? first + tpw : ntiles) - first;  // host: first < ntiles for every split
  const uint8_t* k_row0 = a.k + (int64_t)b * a.k_sb + (int64_t)hk * a.k_sh;
  const uint8_t* v_row0 = a.v + (int64_t)b * a.v_sb + (int64_t)hk * a.v_sh;

  // per-lane source offsets of the K requests: request j covers rows (1024 / KROW) j ..., lane l = slot l % KCPR of row
  // l / KCPR; it fetches chunk slot ^ f(row). f depends on j only through its parity (INT8) or not at all (INT4).
  uint32_t k_off[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const uint32_t row = (uint32_t)par * (1024u / KROW) + lane / KCPR;
    k_off[par] = (lane / KCPR) * (uint32_t)a.k_st + (((lane % KCPR) ^ k_swizzle<KCPR>(row)) << 4);
  }
  const uint32_t v_off = (lane / VCPR) * (uint32_t)a.v_st + ((lane % VCPR) << 4);

  auto request = [&](const uint32_t k) {  // tile k of this wave -> ring slot k % NB; nothing waits here
    const uint32_t tt = first + (k < n ? k : 0u) * step;
    const uint32_t t0 = tt * (uint32_t)TC;
    const uint32_t cnt = k < n ? (a.T - t0 < (uint32_t)TC ? a.T - t0 : (uint32_t)TC) : 0u;  // 0: empty descriptors, no traffic
    uint8_t* slot = ring + (k % NB) * SLOT;
    const __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(k_row0 + (int64_t)t0 * a.k_st), 0, (int)(cnt * (uint32_t)a.k_st), 0x00020000);
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(v_row0 + (int64_t)t0 * a.v_st), 0, (int)(cnt * (uint32_t)a.v_st), 0x00020000);
    const __amdgpu_buffer_rsrc_t ksr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.k_scale + t0), 0, (int)(cnt * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t vsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.v_scale + t0), 0, (int)(cnt * 4u), 0x00020000);
#pragma unroll
    for (int j = 0; j < KOPS; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(kr, (lds_ptr)(slot + 1024 * j), 16, k_off[j & 1], (uint32_t)(j * (1024 / KROW)) * (uint32_t)a.k_st, 0, KVQ_ATTN_KV_AUX);
#pragma unroll
    for (int j = 0; j < VOPS; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(vr, (lds_ptr)(slot + TC * KROW + 1024 * j), 16, v_off, (uint32_t)(j * (1024 / VROW)) * (uint32_t)a.v_st, 0, KVQ_ATTN_KV_AUX);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ksr, (lds_ptr)(slot + TC * (KROW + VROW)), 4, lane * 4u, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(vsr, (lds_ptr)(slot + TC * (KROW + VROW) + 256), 4, lane * 4u, 0, 0, 0);
  };

  ST st;
  constexpr int PRE = NB > 1 ? NB - 1 : 1;  // tiles requested before the loop
#pragma unroll
  for (int k = 0; k < PRE; ++k) request((uint32_t)k);
  st.init(a, b, hk);
  if constexpr (TG == 4) {
    // the largest V scale of the wave's own tokens, once: consume_direct's fixed reference (P is carried as p sv / svref <= 1)
    float vm = 0.0f;
    if (!strided) {
      const uint32_t tok0 = first * (uint32_t)TC;
      const uint32_t cnt = a.T - tok0 < n * (uint32_t)TC ? a.T - tok0 : n * (uint32_t)TC;
      const __amdgpu_buffer_rsrc_t vsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.v_scale + tok0), 0, (int)(cnt * 4u), 0x00020000);
      for (uint32_t i0 = 0; i0 < cnt; i0 += 1024u) {  // 256 floats per wave load, four loads in flight
        u32x4 w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = __builtin_amdgcn_raw_buffer_load_b128(vsr, (i0 + 256u * j + 4u * lane) * 4u, 0, 0);  // past cnt: zeros
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) vm = fmaxf(vm, __uint_as_float(w[j][e]));
      }
    } else {
      for (uint32_t k = 0; k < n; ++k) {
        const uint32_t t = (first + k * step) * (uint32_t)TC + lane;
        if (t < a.T) vm = fmaxf(vm, a.v_scale[t]);
      }
    }
    vm = wave_fmax(vm);
    st.svref = vm;
    st.svn = vm > 0.0f ? 1.0f / vm : 0.0f;
  }
  typename ST::Raw r;
  const typename ST::Src none = st.src(a, b, hk, 0u, 0u);
  for (uint32_t k = 0; k < n; ++k) {
    if constexpr (NB > 1) {
      // the slot tile k + NB - 1 goes to is the one tile k - 1 was read from: those reads have all been consumed
      // (their values fed MFMAs of the previous iteration); the compiler is kept from moving anything across
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      request(k + (uint32_t)(NB - 1));
      wait_vmcnt<OPS * (NB - 1)>();  // tile k has landed
    } else {
      wait_vmcnt<0>();  // ONE slot: tile k has landed; the next request goes out as soon as its bytes are in registers
    }
    asm volatile("" ::: "memory");
    const uint32_t t0 = (first + k * step) * (uint32_t)TC;
    const uint32_t nt = a.T - t0 < (uint32_t)TC ? a.T - t0 : (uint32_t)TC;
    const uint8_t* slot = ring + (k % NB) * SLOT;
    read_fragments<KBITS, VBITS, TC, KI8, typename ST::Raw, HD>(slot, r);
    static_assert(TC <= kWave, "one scale per lane");
    const float* sc = reinterpret_cast<const float*>(slot + TC * (KROW + VROW));
    if constexpr (TG == 4 && NB > 1) {
      st.consume_direct(a, nt, r, sc);
      continue;
    }
    const float ksv = sc[lane < (uint32_t)TC ? lane : 0u], vsv = sc[64 + (lane < (uint32_t)TC ? lane : 0u)];
    r.ks[0] = lane < nt ? ksv : 0.0f;  // rows past nt: the request's range check left the slot's old bytes there
    r.vs[0] = lane < nt ? vsv : 0.0f;
    if constexpr (NB == 1) {
      // every fragment is in registers once the LDS reads have returned: the slot is free for tile k + 1, whose
      // requests then have the whole reduction of tile k to land in
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r.ks[0]), "+v"(r.vs[0]) : : "memory");
      request(k + 1u);
      asm volatile("" ::: "memory");
    }
    st.consume(a, nt, r, none, s_ks, s_vs, s_al);
  }
  wait_vmcnt<0>();  // the empty tail requests retire before the wave's LDS is released
  if constexpr (TG == 4 && NB > 1) st.reduce_l();
  // ---- workspace: (m, l) per head, acc[heads][D] — the layout the merge kernels read ---------------------------
  const bool fold = KVQ_AB && a.arrive != nullptr;  // (A-B builds) uniform: the merge runs in THIS launch (stores write-through, ticket below)
  const __amdgpu_buffer_rsrc_t wsr = __builtin_amdgcn_make_buffer_rsrc(a.ws, 0, (int)a.ws_bytes, 0x00020000);
  if (g == 0u && x < a.nq) {
    const int64_t oi = (((int64_t)b * a.Hq + hk * a.nq + x) * a.nsplit + split) * 2;
    const float mv = st.m * 0.693147180559945309f;
    if (fold) {
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(mv), __float_as_uint(st.l)}, wsr, (uint32_t)oi * 4u, 0, 16);
    } else {
      a.ws[oi] = mv;
      a.ws[oi + 1] = st.l;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t h = 4 * g + q;
    if (h < a.nq) {
      float o8[DVN];
#pragma unroll
      for (int c = 0; c < DVN; ++c) {
        int e = c;
        if constexpr (VBITS == 4 && HD == 128) e = c < 4 ? 2 * c : 2 * (c - 4) + 1;
        if constexpr (VBITS == 4 && HD == 64) e = c == 1 ? 2 : (c == 2 ? 1 : c);
        o8[e] = st.acc[c][q] * st.svref;
      }
      const int64_t di = a.acc_off + (((int64_t)b * a.Hq + hk * a.nq + h) * a.nsplit + split) * a.D + DVN * x;
#pragma unroll
      for (int c = 0; c < DVN; c += 4) {
        if (fold)
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(o8[c]), __float_as_uint(o8[c + 1]), __float_as_uint(o8[c + 2]), __float_as_uint(o8[c + 3])},
                                                 wsr, (uint32_t)(di + c) * 4u, 0, 16);
        else
          *reinterpret_cast<f32x4*>(a.ws + di + c) = f32x4{o8[c], o8[c + 1], o8[c + 2], o8[c + 3]};
      }
    }
  }
#if KVQ_AB
  if (fold) {
#ifdef KVQ_FOLD_NOMERGE  // calibration (`make calib_fold`, wrong output): write-through stores + ticket only, what the merge itself costs on top
    (void)arrive_is_last(a.arrive + b * a.Hkv + hk, a.nsplit);
#else
    if (arrive_is_last(a.arrive + b * a.Hkv + hk, a.nsplit)) merge_group_one_wave<4>(a, a.fold_has_new != 0, hk, b);
#endif
  }
#endif
}

#if KVQ_AB
// ---------------------------------------------------------------------------- coalesced streaming variant
// The same whole-line requests as the LDS-staged kernel, but into REGISTERS (a 1 KiB request per instruction, 16 B per
// lane: `buffer_load_dwordx4`, a few cycles to issue where an LDS-DMA piece costs 60-185 on the issuing wave,
// MI355X_MICROARCH.md latency table), and through ONE LDS image per wave on the way to the operand layout: when a tile
// is due, its registers are written to the (swizzled) image with ds_write_b128, re-requested at once for the tile DEPTH
// tiles ahead, and the fragments are read back as the LDS-staged kernel reads them. DEPTH register sets are in flight
// while a tile is reduced (48 + 2 VGPRs each for INT8 keys + INT4 values).
// Measured (batch 8, 16 K tokens, per layer call incl. merge; profiles/r03d_attn_lds_vs_coalesced.txt): 42.5-45.6 us
// against 40.8 for the LDS-DMA ring and 44.7 for the fragment-shaped streaming kernel: A-B builds only.
template <int KBITS, int VBITS, bool KI8, int DEPTH>
__global__ __launch_bounds__(kWave, 2) void decode_attn_coal_mfma_k(const AttnArgs a, const uint32_t tpw) {
  constexpr int HD = 128, TC = 64;
  typedef AttnStream<KBITS, VBITS, TC, HD, KI8, false> ST;
  constexpr int DVN = ST::DVN;
  constexpr int KROW = HD * KBITS / 8, VROW = HD * VBITS / 8;
  constexpr int KCPR = KROW / 16, VCPR = VROW / 16;
  constexpr int KOPS = TC * KROW / 1024, VOPS = TC * VROW / 1024;
  __shared__ __attribute__((aligned(16))) uint8_t img[TC * (KROW + VROW)];
  __shared__ __attribute__((aligned(16))) float s_ks[TC];
  __shared__ __attribute__((aligned(16))) float s_vs[TC];
  __shared__ __attribute__((aligned(16))) float s_al[16];
  const uint32_t lane = threadIdx.x, x = lane & 15u, g = lane >> 4;
  const uint32_t split = blockIdx.x, hk = blockIdx.y, b = blockIdx.z;
  const uint32_t ntiles = (a.T + (uint32_t)TC - 1u) / (uint32_t)TC;
  const uint32_t first = split * tpw;
  const uint32_t last = first + tpw < ntiles ? first + tpw : ntiles;
  const uint32_t n = last - first;
  const uint8_t* k_row0 = a.k + (int64_t)b * a.k_sb + (int64_t)hk * a.k_sh;
  const uint8_t* v_row0 = a.v + (int64_t)b * a.v_sb + (int64_t)hk * a.v_sh;
  const uint32_t k_off = (lane / KCPR) * (uint32_t)a.k_st + ((lane % KCPR) << 4);
  const uint32_t v_off = (lane / VCPR) * (uint32_t)a.v_st + ((lane % VCPR) << 4);
  // where this lane's pieces go in the image: request j covers rows (1024 / KROW) j ...; slot = chunk ^ f(row)
  uint32_t k_dst[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const uint32_t row = (uint32_t)par * (1024u / KROW) + lane / KCPR;
    k_dst[par] = (lane / KCPR) * KROW + (((lane % KCPR) ^ k_swizzle<KCPR>(row)) << 4);
  }
  struct Flight {  // one tile as requested: whole lines, lane-linear
    u32x4 k[KOPS], v[VOPS];
    float ks, vs;
  };
  auto request = [&](const uint32_t k, Flight& f) {
    const uint32_t tt = first + (k < n ? k : 0u);
    const uint32_t t0 = tt * (uint32_t)TC;
    const uint32_t cnt = k < n ? (a.T - t0 < (uint32_t)TC ? a.T - t0 : (uint32_t)TC) : 0u;  // 0: empty descriptors, no traffic
    const __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(k_row0 + (int64_t)t0 * a.k_st), 0, (int)(cnt * (uint32_t)a.k_st), 0x00020000);
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(v_row0 + (int64_t)t0 * a.v_st), 0, (int)(cnt * (uint32_t)a.v_st), 0x00020000);
#pragma unroll
    for (int j = 0; j < KOPS; ++j) f.k[j] = __builtin_amdgcn_raw_buffer_load_b128(kr, k_off, (uint32_t)(j * (1024 / KROW)) * (uint32_t)a.k_st, KVQ_ATTN_KV_AUX);
#pragma unroll
    for (int j = 0; j < VOPS; ++j) f.v[j] = __builtin_amdgcn_raw_buffer_load_b128(vr, v_off, (uint32_t)(j * (1024 / VROW)) * (uint32_t)a.v_st, KVQ_ATTN_KV_AUX);
    const uint32_t ic = lane < cnt ? lane : (cnt ? cnt - 1u : 0u);
    f.ks = cnt ? a.k_scale[t0 + ic] : 0.0f;
    f.vs = lane < cnt ? a.v_scale[t0 + ic] : 0.0f;
  };
  ST st;
  Flight fl[DEPTH];
#pragma unroll
  for (int dpt = 0; dpt < DEPTH; ++dpt) request((uint32_t)dpt, fl[dpt]);
  st.init(a, b, hk);
  typename ST::Raw r;
  const typename ST::Src none = st.src(a, b, hk, 0u, 0u);
  auto step = [&](const uint32_t k, Flight& f) {
    // the previous tile's fragment reads are complete (their values fed its MFMAs); keep the compiler from moving
    // this tile's image writes above them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int j = 0; j < KOPS; ++j) *reinterpret_cast<u32x4*>(img + (uint32_t)(j * 1024) + k_dst[j & 1]) = f.k[j];
#pragma unroll
    for (int j = 0; j < VOPS; ++j) *reinterpret_cast<u32x4*>(img + TC * KROW + (uint32_t)(j * 1024) + lane * 16u) = f.v[j];
    r.ks[0] = f.ks;
    r.vs[0] = f.vs;
    request(k + (uint32_t)DEPTH, f);  // the registers are free again: the tile DEPTH ahead takes them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    read_fragments<KBITS, VBITS, TC, KI8>(img, r);
    const uint32_t t0 = (first + k) * (uint32_t)TC;
    const uint32_t nt = a.T - t0 < (uint32_t)TC ? a.T - t0 : (uint32_t)TC;
    st.consume(a, nt, r, none, s_ks, s_vs, s_al);
  };
  for (uint32_t k = 0; k < n; k += (uint32_t)DEPTH) {
#pragma unroll
    for (int dpt = 0; dpt < DEPTH; ++dpt)
      if (k + (uint32_t)dpt < n) step(k + (uint32_t)dpt, fl[dpt]);
  }
  // ---- workspace: (m, l) per head, acc[heads][D] — the layout the merge kernels read ---------------------------
  if (g == 0u && x < a.nq) {
    float* o = a.ws + (((int64_t)b * a.Hq + hk * a.nq + x) * a.nsplit + split) * 2;
    o[0] = st.m * 0.693147180559945309f;
    o[1] = st.l;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t h = 4 * g + q;
    if (h < a.nq) {
      float o8[DVN];
#pragma unroll
      for (int c = 0; c < DVN; ++c) {
        int e = c;
        if constexpr (VBITS == 4) e = c < 4 ? 2 * c : 2 * (c - 4) + 1;
        o8[e] = st.acc[c][q] * st.svref;
      }
      float* dst = a.ws + a.acc_off + (((int64_t)b * a.Hq + hk * a.nq + h) * a.nsplit + split) * a.D + DVN * x;
#pragma unroll
      for (int c = 0; c < DVN; c += 4) *reinterpret_cast<f32x4*>(dst + c) = f32x4{o8[c], o8[c + 1], o8[c + 2], o8[c + 3]};
    }
  }
}

#endif  // KVQ_AB (coalesced streaming variant)

#if KVQ_AB
// ---------------------------------------------------------------------------- fused single launch
// One launch per layer call: NW waves per workgroup, one TC-token tile per wave, then
//   1. the NW waves' (m, l, acc) are merged through LDS -> ONE partial per workgroup and head
//      (a quarter / an eighth of the two-launch path's split partials, written once, read once);
//   2. the partial goes to the workspace with write-through (sc1) stores; every storing wave drains
//      (s_waitcnt vmcnt(0)), the workgroup's barrier, then ONE lane takes a ticket on the (batch row,
//      kv head)'s arrival word with an agent-scope fetch_add;
//   3. the workgroup that draws the last ticket reads every partial back with sc1 loads (they bypass
//      this CU's L1; no other kind of load touches those bytes), does the log-sum-exp merge with the
//      exact new token and writes the output rows. No fence, no spin, no residency requirement: a
//      workgroup never waits for another one, so any grid size and any dispatch order is fine.
// Arrival word = (epoch << 20) | arrivals, epoch = a process-wide monotone call number. The caller's
// workspace is never zeroed: at kernel entry one lane reads the word (sc1) and, if it carries another
// epoch (garbage, or an earlier call), tries ONE compare-and-swap to (epoch << 20). If that CAS fails
// the word was changed by this launch (an earlier call on the stream has finished), and the first
// change of a launch can only be a successful CAS of this kind, so the word carries the epoch either
// way before any workgroup's fetch_add (issued after its own CAS has returned: vmcnt(0) in between).
// Two extra workgroup columns (blockIdx.x = nwg, nwg + 1; kv head 0 of batch row 0 only) quantise the new
// token's K / V into slot T of the stores: nothing in this launch reads slot T (the tiles' buffer
// descriptors end at row T and the scale loads are clamped below T).
struct FusedArgs {
  unsigned long long* cnt;  // [B * Hkv] arrival words
  float* part_ml;           // [B * Hkv][nwg][16][2]: m (log2 domain), l per head
  float* part_acc;          // [B * Hkv][nwg][nq][D]
  unsigned long long epoch;
  uint32_t nwg;             // workgroup splits per (batch row, kv head)
  int32_t fuse_quant;
};

__device__ inline float f32_lo(unsigned long long v) { return __uint_as_float((uint32_t)v); }
__device__ inline float f32_hi(unsigned long long v) { return __uint_as_float((uint32_t)(v >> 32)); }

// quant_new_token_block for a workgroup of `nthreads` threads (kvq_common.h's version is 256-wide)
template <int IDT>
__device__ inline void quant_new_token_block_n(const NewTokenArgs& a, uint32_t w, float* s_red, const uint32_t nthreads) {
  const uint32_t tid = threadIdx.x;
  const void* x = a.x[w];
  const uint32_t n = a.B * a.H * a.D;
  float m = 0.0f;
  for (uint32_t i = tid; i < n; i += nthreads) {
    const uint32_t d = i % a.D, r = i / a.D;
    m = fmaxf(m, fabsf(load1<IDT>(x, (int64_t)(r / a.H) * a.xs_b[w] + (int64_t)(r % a.H) * a.xs_h[w] + d)));
  }
  m = wave_fmax(m);
  if ((tid & 63u) == 0u) s_red[tid >> 6] = m;
  __syncthreads();
  m = s_red[0];
  for (uint32_t i = 1; i < nthreads / kWave; ++i) m = fmaxf(m, s_red[i]);
  if (a.bits[w] == 8) {
    const float s32 = fmaxf(m / QRange<8>::qmax, a.eps);
    if (tid == 0) *a.scale[w] = Elem<IDT>::round_trip(s32);
    for (uint32_t i = tid; i < n; i += nthreads) {
      const uint32_t d = i % a.D, r = i / a.D;
      const float v = load1<IDT>(x, (int64_t)(r / a.H) * a.xs_b[w] + (int64_t)(r % a.H) * a.xs_h[w] + d);
      a.q[w][(int64_t)(r / a.H) * a.qs_b[w] + (int64_t)(r % a.H) * a.qs_h[w] + d] = (uint8_t)(int8_t)quant1<8>(v, s32);
    }
  } else {
    const float s32 = fmaxf(m / QRange<4>::qmax, a.eps);
    if (tid == 0) *a.scale[w] = Elem<IDT>::round_trip(s32);
    const uint32_t Dq = (a.D + 1) / 2;
    for (uint32_t i = tid; i < a.B * a.H * Dq; i += nthreads) {
      const uint32_t j = i % Dq, r = i / Dq;
      const int64_t xo = (int64_t)(r / a.H) * a.xs_b[w] + (int64_t)(r % a.H) * a.xs_h[w];
      const int hi = quant1<4>(load1<IDT>(x, xo + 2 * j), s32) + 8;
      const int lo = 2 * j + 1 < a.D ? quant1<4>(load1<IDT>(x, xo + 2 * j + 1), s32) + 8 : 8;
      a.q[w][(int64_t)(r / a.H) * a.qs_b[w] + (int64_t)(r % a.H) * a.qs_h[w] + j] = (uint8_t)(((hi & 0xF) << 4) | (lo & 0xF));
    }
  }
}

// floats of dynamic LDS the fused kernel needs (host + device agree through this one function)
__host__ __device__ inline uint32_t fused_lds_floats(uint32_t tc, uint32_t nw, uint32_t hd, uint32_t nq, uint32_t nwg) {
  const uint32_t tile_phase = nw * 2u * tc + 32u * nw + nw * nq * hd;
  const uint32_t nthreads = nw * 64u, nvec = nq * hd / 4u;
  const uint32_t cw = nvec < nthreads ? nvec : nthreads;
  const uint32_t groups = nthreads / cw;
  const uint32_t final_phase = 2u * nq * nwg + 68u + groups * nvec * 4u;  // + 4: s_red is aligned up to 16 bytes
  return tile_phase > final_phase ? tile_phase : final_phase;
}

template <int KBITS, int VBITS, int TC, int HD, int NW>
__global__ __launch_bounds__(NW* kWave) void decode_attn_fused_mfma_k(const AttnArgs a, const FusedArgs f, const NewTokenArgs ntok) {
  typedef AttnTile<KBITS, VBITS, TC, HD> Tile;
  constexpr int DVN = Tile::DVN;
  constexpr uint32_t NTH = NW * kWave;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
  const uint32_t x = lane & 15u, g = lane >> 4;
  if (blockIdx.x >= f.nwg) {  // the new token's K (column nwg) / V (column nwg + 1)
    if (f.fuse_quant && blockIdx.y == 0u && blockIdx.z == 0u) {
      if (a.dtype == KVQ_F16) quant_new_token_block_n<KVQ_F16>(ntok, blockIdx.x - f.nwg, smem, NTH);
      else quant_new_token_block_n<KVQ_BF16>(ntok, blockIdx.x - f.nwg, smem, NTH);
    }
    return;
  }
  const uint32_t split = blockIdx.x, hk = blockIdx.y, b = blockIdx.z;
  const uint32_t bh = b * a.Hkv + hk;
  const uint32_t nq = a.nq;
  unsigned long long old = 0ull;
  if (tid == 0u) old = __hip_atomic_load(&f.cnt[bh], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  // ---- this wave's tile -----------------------------------------------------------------------------
  const uint32_t t0 = (split * NW + wave) * (uint32_t)TC;
  const uint32_t nt = t0 < a.T ? (a.T - t0 < (uint32_t)TC ? a.T - t0 : (uint32_t)TC) : 0u;  // wave-uniform
  Tile tile;
  if (nt > 0u) {
    tile.run(a, b, hk, t0, nt, smem + wave * 2 * TC, smem + wave * 2 * TC + TC);
  } else {  // past the end of the context (last workgroup only): contributes nothing
    tile.m = -INFINITY;
    tile.l = 0.0f;
    tile.svmax = 0.0f;
#pragma unroll
    for (int n = 0; n < DVN; ++n) tile.acc[n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }
  unsigned long long seen = old;
  if (tid == 0u && (old >> 20) != f.epoch)
    __hip_atomic_compare_exchange_strong(&f.cnt[bh], &seen, f.epoch << 20, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);

  // ---- 1. merge the NW waves through LDS ----------------------------------------------------------------
  float* s_m = smem + NW * 2 * TC;  // [NW][16]
  float* s_l = s_m + NW * 16;       // [NW][16]
  float* s_acc = s_l + NW * 16;     // [NW][nq][HD]
  if (g == 0u) s_m[wave * 16 + x] = tile.m;
  __syncthreads();
  {
    float Mx = s_m[x];
#pragma unroll
    for (int w = 1; w < NW; ++w) Mx = fmaxf(Mx, s_m[w * 16 + x]);  // wave 0 always holds >= 1 token: finite
    if (g == 0u) s_l[wave * 16 + x] = tile.l * __builtin_amdgcn_exp2f(tile.m - Mx);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t h = 4 * g + r;
      if (h < nq) {
        float Mh = s_m[h];
#pragma unroll
        for (int w = 1; w < NW; ++w) Mh = fmaxf(Mh, s_m[w * 16 + h]);
        float o8[DVN];
        tile.ordered(r, __builtin_amdgcn_exp2f(s_m[wave * 16 + h] - Mh) * tile.svmax, o8);
        float* dst = s_acc + (wave * nq + h) * HD + DVN * x;
#pragma unroll
        for (int c = 0; c < DVN; c += 4) *reinterpret_cast<f32x4*>(dst + c) = f32x4{o8[c], o8[c + 1], o8[c + 2], o8[c + 3]};
      }
    }
  }
  __syncthreads();

  // ---- 2. the workgroup's partial -> workspace, write-through ---------------------------------------------
  const uint32_t nvec = nq * HD / 4u;
  float* my_ml = f.part_ml + ((int64_t)bh * f.nwg + split) * 32;
  float* my_acc = f.part_acc + ((int64_t)bh * f.nwg + split) * (int64_t)(nq * HD);
  {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(my_acc, 0, (int)(nq * HD * 4u), 0x00020000);
    for (uint32_t v = tid; v < nvec; v += NTH) {
      f32x4 sum = *reinterpret_cast<const f32x4*>(s_acc + v * 4u);
#pragma unroll
      for (int w = 1; w < NW; ++w) sum += *reinterpret_cast<const f32x4*>(s_acc + (uint32_t)w * nq * HD + v * 4u);
      u32x4 bits;
      __builtin_memcpy(&bits, &sum, 16);
      __builtin_amdgcn_raw_buffer_store_b128(bits, rs, v * 16u, 0, 16);  // aux 16 = sc1
    }
    if (tid < 16u) {  // head tid: (M, L) as ONE 8-byte sc1 store
      float M = s_m[tid], L = s_l[tid];
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        M = fmaxf(M, s_m[w * 16 + tid]);
        L += s_l[w * 16 + tid];
      }
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(my_ml) + tid,
                         ((unsigned long long)__float_as_uint(L) << 32) | __float_as_uint(M), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY storing wave drains (also covers lane 0's CAS)
  __syncthreads();
  uint32_t* s_flag = reinterpret_cast<uint32_t*>(smem);  // the scale area is free by now
  if (tid == 0u) {
    asm volatile("" ::"v"((uint32_t)seen));  // the CAS above is the returning form and has returned
    const unsigned long long r = __hip_atomic_fetch_add(&f.cnt[bh], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_flag = (r - (f.epoch << 20)) == (unsigned long long)(f.nwg - 1u) ? 1u : 0u;
  }
  __syncthreads();
  if (*s_flag == 0u) return;
  __syncthreads();  // everyone has read the flag before the area is reused

  // ---- 3. last arriver: log-sum-exp merge of the nwg partials + the exact new token ------------------------
  const uint32_t nwg = f.nwg;
  float* s_w = smem;                // [nq][nwg]: m, then the split weights
  float* s_pl = s_w + nq * nwg;     // [nq][nwg]: l
  float* s_hL = s_pl + nq * nwg;    // [16] 1 / L
  float* s_hw = s_hL + 16;          // [16] weight of the new token
  float* s_red = s_hw + 48;         // [groups][nvec] f32x4 (16-byte aligned: 2 nq nwg + 64 floats; nq nwg even or padded below)
  s_red = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(s_red) + 15u) & ~(uintptr_t)15u);
  const bool has_new = a.kn != nullptr;
  const unsigned long long* all_ml = reinterpret_cast<const unsigned long long*>(f.part_ml + (int64_t)bh * nwg * 32);
  for (uint32_t i = tid; i < nq * nwg; i += NTH) {
    const uint32_t sp = i / nq, h = i - sp * nq;
    const unsigned long long v = __hip_atomic_load(all_ml + sp * 16u + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_w[h * nwg + sp] = f32_lo(v);
    s_pl[h * nwg + sp] = f32_hi(v);
  }
  __syncthreads();
  for (uint32_t h = wave; h < nq; h += NW) {  // one wave per head
    float s_tok = -INFINITY;
    if (has_new) {
      float part = 0.0f;
      for (uint32_t d = lane; d < (uint32_t)HD; d += kWave)
        part += load_elem(a.q, (int64_t)b * a.q_sb + (int64_t)(hk * nq + h) * a.q_sh + d, a.dtype) *
                load_elem(a.kn, (int64_t)b * a.kn_sb + (int64_t)hk * a.kn_sh + d, a.dtype);
      s_tok = wave_fsum(part) * (a.sm_scale * 1.44269504088896341f);
    }
    float mx = s_tok;
    for (uint32_t sp = lane; sp < nwg; sp += kWave) mx = fmaxf(mx, s_w[h * nwg + sp]);
    const float M = wave_fmax(mx);
    float lsum = 0.0f;
    for (uint32_t sp = lane; sp < nwg; sp += kWave) {
      const float wgt = __builtin_amdgcn_exp2f(s_w[h * nwg + sp] - M);
      s_w[h * nwg + sp] = wgt;
      lsum += s_pl[h * nwg + sp] * wgt;
    }
    const float w_new = has_new ? __builtin_amdgcn_exp2f(s_tok - M) : 0.0f;
    const float L = wave_fsum(lsum) + w_new;
    if (lane == 0u) {
      s_hL[h] = 1.0f / L;
      s_hw[h] = w_new;
    }
  }
  __syncthreads();
  {
    const uint32_t cw = nvec < NTH ? nvec : NTH;  // columns (16-byte vectors of the [nq][HD] row) handled side by side
    const uint32_t groups = NTH / cw;
    const uint32_t grp = tid / cw, c0 = tid - grp * cw;
    const float* all_acc = f.part_acc + (int64_t)bh * nwg * (int64_t)(nq * HD);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(all_acc), 0, (int)(nwg * nq * HD * 4u), 0x00020000);
    if (grp < groups) {
      for (uint32_t col = c0; col < nvec; col += cw) {
        const uint32_t h = col / (HD / 4u);
        const float* wrow = s_w + h * nwg;
        f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
        uint32_t sp = grp;
        for (; sp + 3u * groups < nwg; sp += 4u * groups) {  // 4 independent sc1 loads in flight
          u32x4 r0 = __builtin_amdgcn_raw_buffer_load_b128(rs, (sp * nvec + col) * 16u, 0, 16);
          u32x4 r1 = __builtin_amdgcn_raw_buffer_load_b128(rs, ((sp + groups) * nvec + col) * 16u, 0, 16);
          u32x4 r2 = __builtin_amdgcn_raw_buffer_load_b128(rs, ((sp + 2u * groups) * nvec + col) * 16u, 0, 16);
          u32x4 r3 = __builtin_amdgcn_raw_buffer_load_b128(rs, ((sp + 3u * groups) * nvec + col) * 16u, 0, 16);
          f32x4 x0, x1, x2, x3;
          __builtin_memcpy(&x0, &r0, 16);
          __builtin_memcpy(&x1, &r1, 16);
          __builtin_memcpy(&x2, &r2, 16);
          __builtin_memcpy(&x3, &r3, 16);
          o += x0 * wrow[sp] + x1 * wrow[sp + groups] + x2 * wrow[sp + 2u * groups] + x3 * wrow[sp + 3u * groups];
        }
        for (; sp < nwg; sp += groups) {
          u32x4 r0 = __builtin_amdgcn_raw_buffer_load_b128(rs, (sp * nvec + col) * 16u, 0, 16);
          f32x4 x0;
          __builtin_memcpy(&x0, &r0, 16);
          o += x0 * wrow[sp];
        }
        *reinterpret_cast<f32x4*>(s_red + (grp * nvec + col) * 4u) = o;
      }
    }
    __syncthreads();
    for (uint32_t i = tid; i < nq * (uint32_t)HD; i += NTH) {
      const uint32_t h = i / (uint32_t)HD, d = i - h * (uint32_t)HD;
      float t = 0.0f;
      for (uint32_t k = 0; k < groups; ++k) t += s_red[k * nvec * 4u + i];
      if (has_new) t = fmaf(s_hw[h], load_elem(a.vn, (int64_t)b * a.vn_sb + (int64_t)hk * a.vn_sh + d, a.dtype), t);
      t *= s_hL[h];
      const int64_t oi = (int64_t)b * a.o_sb + (int64_t)(hk * nq + h) * a.o_sh + d;
      if (a.dtype == KVQ_F16) reinterpret_cast<f16*>(a.out)[oi] = (f16)t;
      else reinterpret_cast<__bf16*>(a.out)[oi] = (__bf16)t;
    }
  }
}

#endif  // KVQ_AB (fused single launch)

constexpr int kAttnMfmaTC = 128;

constexpr int kAttnMaxSplit = 4096;  // merge: LDS weights + 16 (m, l) pairs per thread; T <= 512 Ki tokens at 128 per split

// One workgroup per (query head, batch row): log-sum-exp merge of the splits and of the exact new
// token. Split weights are computed once (one split per thread) and kept in LDS; the weighted sum
// runs 256 / D split groups wide with independent loads.
// With fuse_quant, two extra workgroups (blockIdx.x = Hq, Hq + 1 of batch row 0) quantise the new
// token's K / V into slot T of the stores (quant_new_token_block): nothing in this launch reads that
// slot, and the partial kernel that read [0, T) has finished, so a decode step is two launches.
__global__ __launch_bounds__(kAttnBlock) void decode_attn_merge_k(const AttnArgs a, const NewTokenArgs nt, const int fuse_quant) {
  __shared__ float s_red[kAttnBlock / kWave];
  if (blockIdx.x >= a.Hq) {
    if (fuse_quant && blockIdx.y == 0u) {
      NewTokenArgs slot = nt;
      if (a.t_dev) {  // the host passed slot 0; the real slot is the device-side T
        const int64_t T = (int64_t)live_tokens(a);
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          slot.q[w] += T * nt.qs_t[w];
          slot.scale[w] += T;
        }
      }
      if (a.dtype == KVQ_F16) quant_new_token_block<KVQ_F16>(slot, blockIdx.x - a.Hq, s_red);
      else quant_new_token_block<KVQ_BF16>(slot, blockIdx.x - a.Hq, s_red);
    }
    return;
  }
  __shared__ float s_wt[kAttnMaxSplit];
  __shared__ __attribute__((aligned(16))) float s_out[kAttnBlock * 4];
  const uint32_t tid = threadIdx.x;
  const uint32_t hq = blockIdx.x, b = blockIdx.y;
  const uint32_t hk = hq / a.nq;
  const bool has_new = a.kn != nullptr;
  const uint32_t wave = tid >> 6, lane = tid & 63u;
  auto block_reduce = [&](float v, bool is_max) -> float {
    v = is_max ? wave_fmax(v) : wave_fsum(v);
    __syncthreads();  // s_red free again
    if (lane == 0u) s_red[wave] = v;
    __syncthreads();
    return is_max ? fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3])) : (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  };
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  float s_tok = -INFINITY;
  if (has_new) {
    float part = 0.0f;
    for (uint32_t d = tid; d < a.D; d += kAttnBlock)
      part += load_elem(a.q, (int64_t)b * a.q_sb + (int64_t)hq * a.q_sh + d, a.dtype) *
              load_elem(a.kn, (int64_t)b * a.kn_sb + (int64_t)hk * a.kn_sh + d, a.dtype);
    s_tok = block_reduce(part, false) * a.sm_scale;
  }
  const float* ml = a.ws + ((int64_t)b * a.Hq + hq) * a.nsplit * 2;
  const float* accb = a.ws + a.acc_off + ((int64_t)b * a.Hq + hq) * a.nsplit * a.D;
  // a.nsplit is the row stride of the partials; with a device-side T only the first ceil(T / TS) are live
  const uint32_t ns = a.t_dev ? (live_tokens(a) + a.TS - 1u) / a.TS : a.nsplit;
  // this thread's splits: tid, tid + 256, ...; (m, l) is read twice (second time from cache) rather
  // than kept in a register array sized for kAttnMaxSplit
  float m_max = s_tok;
  for (uint32_t s = tid; s < ns; s += kAttnBlock) m_max = fmaxf(m_max, ml[2 * s]);
  const float M = block_reduce(m_max, true);
  float lsum = 0.0f;
  for (uint32_t s = tid; s < ns; s += kAttnBlock) {
    const float w = __expf(ml[2 * s] - M);
    s_wt[s] = w;
    lsum += ml[2 * s + 1] * w;
  }
  const float w_new = has_new ? __expf(s_tok - M) : 0.0f;
  const float L = block_reduce(lsum, false) + w_new;  // the barriers inside also publish s_wt
  const float inv = 1.0f / L;
  // weighted sum, 16 bytes per lane: thread = (split group g, 4 elements at d4); 1024 / D groups.
  // (Requesting the first 8 rows + the new-token operands before the reductions measured slower:
  // 7.6 vs 5.1 us at 128 splits.)
  const uint32_t dv = a.D >> 2;
  const uint32_t groups = kAttnBlock / dv;  // 32, 16, 8 or 4
  const uint32_t g = tid / dv, d4 = tid - g * dv;
  f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(accb) + d4;
    uint32_t s = g;
    for (; s + 7u * groups < ns; s += 8u * groups) {  // 8 independent loads in flight: at batch 1 the merge is pure latency
      f32x4 x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = src[(int64_t)(s + (uint32_t)u * groups) * dv];
#pragma unroll
      for (int u = 0; u < 8; ++u) o += x[u] * s_wt[s + (uint32_t)u * groups];
    }
    for (; s + 3u * groups < ns; s += 4u * groups) {  // 4 independent loads in flight
      const f32x4 x0 = src[(int64_t)s * dv];
      const f32x4 x1 = src[(int64_t)(s + groups) * dv];
      const f32x4 x2 = src[(int64_t)(s + 2u * groups) * dv];
      const f32x4 x3 = src[(int64_t)(s + 3u * groups) * dv];
      o += x0 * s_wt[s] + x1 * s_wt[s + groups] + x2 * s_wt[s + 2u * groups] + x3 * s_wt[s + 3u * groups];
    }
    for (; s < ns; s += groups) o += src[(int64_t)s * dv] * s_wt[s];
  }
  *reinterpret_cast<f32x4*>(&s_out[tid * 4]) = o;  // [g][d]
  __syncthreads();
  if (tid < a.D) {
    float t = 0.0f;
    for (uint32_t k = 0; k < groups; ++k) t += s_out[k * a.D + tid];
    if (has_new) t = fmaf(w_new, load_elem(a.vn, (int64_t)b * a.vn_sb + (int64_t)hk * a.vn_sh + tid, a.dtype), t);
    t *= inv;
    const int64_t oi = (int64_t)b * a.o_sb + (int64_t)hq * a.o_sh + tid;
    if (a.dtype == KVQ_F16) reinterpret_cast<f16*>(a.out)[oi] = (f16)t;
    else reinterpret_cast<__bf16*>(a.out)[oi] = (__bf16)t;
  }
}

// ---------------------------------------------------------------------------- merge, one round trip
// The merge above is a chain of dependent round trips (new-token dot -> (m, l) -> (m, l) again -> rows in two
// batches -> new-token V): at batch 1 nothing hides them and a call costs 5 us for 2 MB (9 us inside a decode
// step, where every operand was written by the kernel before and is cold in this XCD's L2). This variant, for up
// to kAttnBlock splits, REQUESTS everything the workgroup reads before it waits for anything — the device-side
// token count, its (m, l) pair, the new token's q / k / v elements and the first kMergePF partial rows of each
// thread (rows past the live splits are read too: the workspace holds them, their values are never used) — and
// synchronises its waves with LDS-only barriers (`__syncthreads()` would drain the outstanding global loads).
// Same arithmetic in the same order as decode_attn_merge_k: bit-identical results.
constexpr int kMergePF = 16;
// A word another kernel wrote, read through the CONSTANT address space: with a uniform address that is an s_load the
// compiler tracks itself (issued where it stands, waited for at the first use), where a global-space load becomes a
// vector load that is waited for on the spot. Only for memory no store of THIS kernel touches.
__device__ __forceinline__ int scalar_load_i32(const int* p) {
  return *reinterpret_cast<const __attribute__((address_space(4))) int*>(reinterpret_cast<uintptr_t>(p));
}
__device__ __forceinline__ void lds_barrier() {  // this workgroup's LDS traffic only; global loads stay in flight
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// The new token's K (w = 0) / V (w = 1) slice quantised into slot T of its store by ONE workgroup, as
// quant_new_token_block does, but in one round trip: every element pair is requested up front (with the device-side
// slot number), stays in registers between the abs-max and the stores, and the workgroup meets at ONE LDS-only
// barrier. Same expressions per element (IEEE divide, round-half-even), so the stored bytes and scale are the
// generic routine's. Uniform precondition (checked by the caller): even D, 2-element-aligned rows, at most
// 2 * kAttnBlock * kNewPairs elements.
constexpr int kNewPairs = 16;
__device__ inline bool new_token_fits_registers(const NewTokenArgs& a, uint32_t w) {
  return (a.D & 1u) == 0u && (uint64_t)a.B * a.H * a.D <= 2ull * kAttnBlock * kNewPairs && ((a.xs_b[w] | a.xs_h[w]) & 1) == 0 &&
         (reinterpret_cast<uintptr_t>(a.x[w]) & 3u) == 0u;
}
template <int IDT>
__device__ inline void quant_new_token_regs(const NewTokenArgs& a, const uint32_t w, float* s_red, const int* t_dev, const uint32_t t_bound,
                                            const void* valid) {
  const uint32_t tid = threadIdx.x;
  const uint32_t np = a.B * a.H * a.D / 2u;  // element pairs
  // the device-side slot number: a SCALAR load (uniform address), issued here and waited for by the compiler at
  // its first use below the fence
  const uint32_t t_raw = (uint32_t)scalar_load_i32(t_dev ? t_dev : reinterpret_cast<const int*>(valid));
  const uint16_t* x = reinterpret_cast<const uint16_t*>(a.x[w]);
  uint32_t raw[kNewPairs];
  uint32_t row[kNewPairs], col[kNewPairs];  // (b, h) row and element offset of the pair
#pragma unroll
  for (int k = 0; k < kNewPairs; ++k) {
    const uint32_t p = tid + (uint32_t)k * kAttnBlock;
    const uint32_t e = 2u * (p < np ? p : np - 1u);
    row[k] = e / a.D;
    col[k] = e - row[k] * a.D;
    const int64_t off = (int64_t)(row[k] / a.H) * a.xs_b[w] + (int64_t)(row[k] % a.H) * a.xs_h[w] + col[k];
    raw[k] = *reinterpret_cast<const uint32_t*>(x + off);
  }
  __builtin_amdgcn_sched_barrier(0);
  const int64_t slot = t_dev ? (int64_t)(t_raw < t_bound ? t_raw : t_bound) : 0;  // host-side T: a.q / a.scale point at the slot already
  float lo[kNewPairs], hi[kNewPairs];
  float m = 0.0f;
#pragma unroll
  for (int k = 0; k < kNewPairs; ++k) {
    lo[k] = Elem<IDT>::widen((uint16_t)(raw[k] & 0xFFFFu));
    hi[k] = Elem<IDT>::widen((uint16_t)(raw[k] >> 16));
    if (tid + (uint32_t)k * kAttnBlock < np) m = fmaxf(m, fmaxf(fabsf(lo[k]), fabsf(hi[k])));
  }
  m = wave_fmax(m);
  if ((tid & 63u) == 0u) s_red[tid >> 6] = m;
  lds_barrier();
  m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  uint8_t* qbase = a.q[w] + slot * a.qs_t[w];
  if (a.bits[w] == 8) {
    const float s32 = fmaxf(m / QRange<8>::qmax, a.eps);
    if (tid == 0u) a.scale[w][slot] = Elem<IDT>::round_trip(s32);
#pragma unroll
    for (int k = 0; k < kNewPairs; ++k) {
      if (tid + (uint32_t)k * kAttnBlock < np) {
        uint8_t* dst = qbase + (int64_t)(row[k] / a.H) * a.qs_b[w] + (int64_t)(row[k] % a.H) * a.qs_h[w] + col[k];
        dst[0] = (uint8_t)(int8_t)quant1<8>(lo[k], s32);
        dst[1] = (uint8_t)(int8_t)quant1<8>(hi[k], s32);
      }
    }
  } else {
    const float s32 = fmaxf(m / QRange<4>::qmax, a.eps);
    if (tid == 0u) a.scale[w][slot] = Elem<IDT>::round_trip(s32);
#pragma unroll
    for (int k = 0; k < kNewPairs; ++k) {
      if (tid + (uint32_t)k * kAttnBlock < np) {
        const int h4 = quant1<4>(lo[k], s32) + 8, l4 = quant1<4>(hi[k], s32) + 8;  // even index -> high nibble
        qbase[(int64_t)(row[k] / a.H) * a.qs_b[w] + (int64_t)(row[k] % a.H) * a.qs_h[w] + (col[k] >> 1)] = (uint8_t)(((h4 & 0xF) << 4) | (l4 & 0xF));
      }
    }
  }
}

// ONE WAVE per query head (head_dim 128, at most 16 splits: what the LDS-staged kernel leaves): the same merge without the
// workgroup — lane s holds split s's (m, l), lane l holds d = l and d = 64 + l of every split's partial (32 loads, all
// requested before anything waits); the weights travel by v_readlane instead of an LDS table, and there is no barrier.
// The arithmetic, operand by operand and in the same order, is decode_attn_merge_fast_k's (its thread (g, d4) sums splits
// g, g + 8, the final row sums the 8 groups in order; its dot product sums d < 64 in wave 0 and d >= 64 in wave 1): equal bits.
__device__ __forceinline__ void merge_one_wave(const AttnArgs& a, const bool has_new, const uint32_t hq, const uint32_t b) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr uint32_t D = 128, G = 8, NSU = 16;  // head_dim, the block kernel's split groups, splits held in registers
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t hk = hq / a.nq;
  const uint32_t nb = a.nsplit;  // 1 ... 16 (host)
  const float* ml = a.ws + ((int64_t)b * a.Hq + hq) * nb * 2;
  const float* src = a.ws + a.acc_off + ((int64_t)b * a.Hq + hq) * nb * D + lane;
  const uint32_t t_raw = (uint32_t)scalar_load_i32(a.t_dev ? a.t_dev : reinterpret_cast<const int*>(a.ws));
  const f32x2 ml_raw = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(ml) + (lane < nb ? lane : nb - 1u));
  uint16_t qraw[2], kraw[2], vraw[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t di = 64u * h + lane;
    qraw[h] = reinterpret_cast<const uint16_t*>(a.q)[(int64_t)b * a.q_sb + (int64_t)hq * a.q_sh + di];
    kraw[h] = reinterpret_cast<const uint16_t*>(a.kn)[(int64_t)b * a.kn_sb + (int64_t)hk * a.kn_sh + di];
    vraw[h] = reinterpret_cast<const uint16_t*>(a.vn)[(int64_t)b * a.vn_sb + (int64_t)hk * a.vn_sh + di];
  }
  float x[2][NSU];
#pragma unroll
  for (int sp = 0; sp < (int)NSU; ++sp) {
    const uint32_t sc = (uint32_t)sp < nb ? (uint32_t)sp : nb - 1u;
#pragma unroll
    for (int h = 0; h < 2; ++h) x[h][sp] = __builtin_nontemporal_load(src + (int64_t)sc * D + 64 * h);
  }
  __builtin_amdgcn_sched_barrier(0);
  const uint32_t T = a.t_dev ? (t_raw < a.T ? t_raw : a.T) : a.T;
  uint32_t ns = a.t_dev ? (T + a.TS - 1u) / a.TS : nb;
  ns = ns < nb ? ns : nb;
  f32x2 mlv = ml_raw;
  if (lane >= ns) mlv = f32x2{-INFINITY, 0.0f};
  auto widen = [&](uint16_t h) { return a.dtype == KVQ_F16 ? Elem<KVQ_F16>::widen(h) : Elem<KVQ_BF16>::widen(h); };
  const float p0 = wave_fsum(widen(has_new ? qraw[0] : (uint16_t)0) * widen(kraw[0]));
  const float p1 = wave_fsum(widen(has_new ? qraw[1] : (uint16_t)0) * widen(kraw[1]));
  const float s_tok = has_new ? ((p0 + p1) + (0.0f + 0.0f)) * a.sm_scale : -INFINITY;
  const float M = fmaxf(wave_fmax(mlv[0]), s_tok);
  const float w = lane < ns ? __expf(mlv[0] - M) : 0.0f;
  const float lw = wave_fsum(lane < ns ? mlv[1] * w : 0.0f);
  const float w_new = has_new ? __expf(s_tok - M) : 0.0f;
  const float L = ((lw + 0.0f) + (0.0f + 0.0f)) + w_new;
  const float inv = 1.0f / L;
  float wt[NSU];
#pragma unroll
  for (int sp = 0; sp < (int)NSU; ++sp) wt[sp] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(w), sp));
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < (int)G; ++k) {
      float o = 0.0f;
#pragma unroll
      for (int u = 0; u < (int)(NSU / G); ++u) {
        const int sp = k + u * (int)G;
        if ((uint32_t)sp < ns) o += x[h][sp] * wt[sp];
      }
      t += o;
    }
    if (has_new) t = fmaf(w_new, widen(vraw[h]), t);
    t *= inv;
    const int64_t oi = (int64_t)b * a.o_sb + (int64_t)hq * a.o_sh + 64 * h + lane;
    if (a.dtype == KVQ_F16) reinterpret_cast<f16*>(a.out)[oi] = (f16)t;
    else reinterpret_cast<__bf16*>(a.out)[oi] = (__bf16)t;
  }
}

// (has_new: without a new token the host points kn / vn at the query, so that the three loads need no branch)
// PF: partial rows each thread requests up front (and holds): ceil(nsplit / (256 / (D/4))) rounded up to 2 / 4 / 16 by the
// host — the 16 splits per head of the LDS-staged kernel need 2, not 16 (14 clamped re-reads of the last row per thread)
template <int PF>
__global__ __launch_bounds__(kAttnBlock) void decode_attn_merge_fast_k(const AttnArgs a, const NewTokenArgs nt, const int fuse_quant_i,
                                                                       const int has_new_i) {
  __shared__ float s_red[3][kAttnBlock / kWave];
  // fuse_quant_i bit 0: the launch's last two workgroups quantise-append the new token; bit 1 (PF == 2, head_dim 128, <= 16
  // splits): ONE WAVE per head, four heads per workgroup (merge_one_wave)
  const int fuse_quant = fuse_quant_i & 1;
  const bool by_wave = PF == 2 && (fuse_quant_i & 2) != 0;
  const uint32_t n_head_wgs = by_wave ? (a.Hq + 3u) / 4u : a.Hq;
  if (blockIdx.x >= n_head_wgs) {
    const uint32_t w_new_tok = blockIdx.x - n_head_wgs;
    if (fuse_quant && blockIdx.y == 0u && new_token_fits_registers(nt, w_new_tok)) {
      if (a.dtype == KVQ_F16) quant_new_token_regs<KVQ_F16>(nt, w_new_tok, s_red[0], a.t_dev, a.T, a.q);
      else quant_new_token_regs<KVQ_BF16>(nt, w_new_tok, s_red[0], a.t_dev, a.T, a.q);
    } else if (fuse_quant && blockIdx.y == 0u) {
      NewTokenArgs slot = nt;
      if (a.t_dev) {
        const int64_t T = (int64_t)live_tokens(a);
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          slot.q[w] += T * nt.qs_t[w];
          slot.scale[w] += T;
        }
      }
      if (a.dtype == KVQ_F16) quant_new_token_block<KVQ_F16>(slot, w_new_tok, s_red[0]);
      else quant_new_token_block<KVQ_BF16>(slot, w_new_tok, s_red[0]);
    }
    return;
  }
  if constexpr (PF == 2) {
    if (by_wave) {
      const uint32_t hq_w = blockIdx.x * 4u + (threadIdx.x >> 6);
      if (hq_w < a.Hq) merge_one_wave(a, has_new_i != 0, hq_w, blockIdx.y);
      return;
    }
  }
  __shared__ float s_wt[kAttnBlock];
  __shared__ __attribute__((aligned(16))) float s_out[kAttnBlock * 4];
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const uint32_t tid = threadIdx.x;
  const uint32_t hq = blockIdx.x, b = blockIdx.y;
  const uint32_t hk = hq / a.nq;
  const bool has_new = has_new_i != 0;
  const uint32_t wave = tid >> 6, lane = tid & 63u;
  const uint32_t nb = a.nsplit;  // 1 ... kAttnBlock (host); the row stride of the partials
  const uint32_t dv = a.D >> 2;
  const uint32_t groups = kAttnBlock / dv;  // 32, 16, 8 or 4
  const uint32_t g = tid / dv, d4 = tid - g * dv;
  const float* ml = a.ws + ((int64_t)b * a.Hq + hq) * nb * 2;
  const f32x4* src = reinterpret_cast<const f32x4*>(a.ws + a.acc_off + ((int64_t)b * a.Hq + hq) * nb * a.D) + d4;

  // ---- requests: straight-line code (clamped indices, selected addresses), nothing waits before the fence -------
  // the device-side token count: a SCALAR buffer load (uniform address; a plain `*a.t_dev` becomes a vector load that is
  // waited for on the spot), issued here and waited for by the compiler at its first use below the fence
  const uint32_t t_raw = (uint32_t)scalar_load_i32(a.t_dev ? a.t_dev : reinterpret_cast<const int*>(a.ws));
  const f32x2 ml_raw = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(ml) + (tid < nb ? tid : nb - 1u));
  const uint32_t di = tid < a.D ? tid : a.D - 1u;
  const uint16_t qb_raw = reinterpret_cast<const uint16_t*>(a.q)[(int64_t)b * a.q_sb + (int64_t)hq * a.q_sh + di];
  const uint16_t kb = reinterpret_cast<const uint16_t*>(a.kn)[(int64_t)b * a.kn_sb + (int64_t)hk * a.kn_sh + di];
  const uint16_t vb = reinterpret_cast<const uint16_t*>(a.vn)[(int64_t)b * a.vn_sb + (int64_t)hk * a.vn_sh + di];
  f32x4 x[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const uint32_t s = g + (uint32_t)u * groups;
    x[u] = __builtin_nontemporal_load(src + (int64_t)(s < nb ? s : nb - 1u) * dv);
  }
  __builtin_amdgcn_sched_barrier(0);
  const uint32_t T = a.t_dev ? (t_raw < a.T ? t_raw : a.T) : a.T;
  const uint16_t qb = (has_new && tid < a.D) ? qb_raw : (uint16_t)0;  // the dot product runs over d < D
  f32x2 mlv = ml_raw;

  // ---- weights ---------------------------------------------------------------------------------------------
  uint32_t ns = a.t_dev ? (T + a.TS - 1u) / a.TS : nb;
  ns = ns < nb ? ns : nb;
  if (tid >= ns) mlv = f32x2{-INFINITY, 0.0f};
  auto widen = [&](uint16_t h) { return a.dtype == KVQ_F16 ? Elem<KVQ_F16>::widen(h) : Elem<KVQ_BF16>::widen(h); };
  {
    const float part = wave_fsum(widen(qb) * widen(kb));  // zero beyond D and without a new token
    const float mw = wave_fmax(mlv[0]);
    if (lane == 0u) {
      s_red[0][wave] = part;
      s_red[1][wave] = mw;
    }
  }
  lds_barrier();
  const float s_tok = has_new ? ((s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3])) * a.sm_scale : -INFINITY;
  const float M = fmaxf(fmaxf(fmaxf(s_red[1][0], s_red[1][1]), fmaxf(s_red[1][2], s_red[1][3])), s_tok);
  {
    const float w = tid < ns ? __expf(mlv[0] - M) : 0.0f;
    s_wt[tid] = w;
    const float lw = wave_fsum(tid < ns ? mlv[1] * w : 0.0f);
    if (lane == 0u) s_red[2][wave] = lw;
  }
  lds_barrier();  // also publishes s_wt
  const float w_new = has_new ? __expf(s_tok - M) : 0.0f;
  const float L = ((s_red[2][0] + s_red[2][1]) + (s_red[2][2] + s_red[2][3])) + w_new;
  const float inv = 1.0f / L;

  // ---- weighted sum: thread = (split group g, 4 elements at d4) ---------------------------------------------
  f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const uint32_t s = g + (uint32_t)u * groups;
    if (s < ns) o += x[u] * s_wt[s];
  }
  for (uint32_t s = g + (uint32_t)PF * groups; s < ns; s += groups) o += src[(int64_t)s * dv] * s_wt[s];
  *reinterpret_cast<f32x4*>(&s_out[tid * 4]) = o;  // [g][d]
  lds_barrier();
  if (tid < a.D) {
    float t = 0.0f;
    for (uint32_t k = 0; k < groups; ++k) t += s_out[k * a.D + tid];
    if (has_new) t = fmaf(w_new, widen(vb), t);
    t *= inv;
    const int64_t oi = (int64_t)b * a.o_sb + (int64_t)hq * a.o_sh + tid;
    if (a.dtype == KVQ_F16) reinterpret_cast<f16*>(a.out)[oi] = (f16)t;
    else reinterpret_cast<__bf16*>(a.out)[oi] = (__bf16)t;
  }
}

constexpr int64_t kLdsSlots = 256 * 4;  // one-wave workgroups of the LDS-staged kernel resident at once (4 per CU)

// tokens per workgroup: one loop iteration (D/16 lanes per token, kAttnUnroll tokens per lane) while
// the grid stays below ~4096 workgroups, whole multiples of it beyond; never more than
// kAttnMaxSplit splits
static bool use_mfma(const kvq_attn_dims_t* d) {
  const int64_t nq = d->Hkv > 0 ? d->Hq / d->Hkv : 0;
  const int64_t min_nq = tunables().attn_mfma_min_nq > 0 ? tunables().attn_mfma_min_nq : 3;
  return (d->D == 128 || d->D == 64) && nq >= min_nq && nq <= 16 && !tunables().attn_force_valu;
}
// streaming kernel: tiles per wave (0 = one-tile kernel). Chosen so that every wave of the launch is resident
// at once (one round) when the batch offers more 64-token tiles than the chip has wave slots.
static int stream_tc() { return KVQ_AB && tunables().attn_stream_tc == 32 ? 32 : 64; }
// LDS-staged streaming kernel (contiguous 1 KiB requests into an LDS ring): head_dim 128 grouped-query shapes.
// attn_lds: -1 (default) / 1 = the LDS-DMA ring kernel serves every streaming plan; 0 = never (A-B builds: the register-
// staged streaming kernel instead; the default library then has no streaming kernel and takes one-tile splits);
// 2 (A-B builds) = the coalesced register-staged kernel
// (head_dim 64, round 4: the ring kernel's HD = 64 instantiation takes INT8 keys only — k_bits; sizing calls pass 8 and cover the
// one-tile plan beside it)
static bool use_lds(const kvq_attn_dims_t* d, int k_bits = 8) {
  return tunables().attn_lds != 0 && use_mfma(d) && (d->D == 128 || (d->D == 64 && k_bits == 8 && !(KVQ_AB && (tunables().attn_lds == 2 || tunables().attn_lds == 3))));
}
static int lds_tc() { return KVQ_AB && tunables().attn_lds_tc == 32 ? 32 : 64; }
// tokens per tile of whichever streaming kernel serves these dims
static int tile_tc(const kvq_attn_dims_t* d, int k_bits = 8) { return use_lds(d, k_bits) ? (d->D == 64 ? 64 : lds_tc()) : stream_tc(); }
static uint32_t stream_tpw(const kvq_attn_dims_t* d, int k_bits = 8) {
  if (!use_mfma(d) || !(d->D == 128 || (d->D == 64 && use_lds(d, k_bits))) || d->T <= 0) return 0;
  const int64_t forced = tunables().attn_stream_tpw;  // -1 = never, 0 = by size, > 0 = that many
  if (forced < 0) return 0;
  const int kStreamTC = tile_tc(d, k_bits);
  const int64_t ntiles = (d->T + kStreamTC - 1) / kStreamTC;
  // wave slots of one round: the LDS-staged kernel holds its tiles in LDS, 4 one-wave workgroups per CU; (A-B) register-
  // staged kernels: 2 waves per SIMD at 64-token tiles, 3 at 32-token tiles
  const bool ring = use_lds(d, k_bits) && !(KVQ_AB && tunables().attn_lds == 2);
  if (!ring && !KVQ_AB) return 0;  // the default library's only streaming kernel is the ring kernel
  int64_t slots = ring ? kLdsSlots : 256 * 4 * (kStreamTC == 64 ? 2 : 3);  // (head_dim 64 too: 2048 slots measured 24.9 us per call against 23.7 at 1024, Llama-3.2-1B batch 8)
  if (KVQ_AB && tunables().attn_stream_slots > 0) slots = tunables().attn_stream_slots;
  int64_t tpw = forced > 0 ? forced : (ntiles * d->B * d->Hkv + slots - 1) / slots;
  // by size: the ring pays from three tiles per wave on (batch 1 at 16 K tokens = two tiles per wave: 14.9 us per layer
  // call against 13.7 for one-tile splits); the register-staged kernels from two
  if (forced <= 0 && tpw < (ring ? 3 : 2)) return 0;
  if (tpw > ntiles) tpw = ntiles;
  return (uint32_t)tpw;
}
static bool plan_onetile(const kvq_attn_dims_t* d, uint32_t* ts, uint32_t* nsplit);
// tpw_out (optional): tiles per wave of the streaming layout when THAT is what the plan chose, 0 when it fell back to
// one-tile splits (the launcher must size its kernel choice from this, never from stream_tpw() on its own)
static bool plan(const kvq_attn_dims_t* d, uint32_t* ts, uint32_t* nsplit, uint32_t* tpw_out = nullptr, int k_bits = 8) {
  if (tpw_out) *tpw_out = 0;
  if (const uint32_t tpw = stream_tpw(d, k_bits)) {  // one wave per tpw tiles of 64 (32) tokens
    const int64_t per = (int64_t)tpw * tile_tc(d, k_bits);
    *ts = (uint32_t)per;
    *nsplit = (uint32_t)((d->T + per - 1) / per);
    if (*nsplit <= (uint32_t)kAttnMaxSplit) {
      if (tpw_out) *tpw_out = tpw;
      return true;
    }
  }
  return plan_onetile(d, ts, nsplit);
}
// one tile (MFMA kernel) / one workgroup pass (VALU kernel) per split: also the only shapes a device-side T takes
static bool plan_onetile(const kvq_attn_dims_t* d, uint32_t* ts, uint32_t* nsplit) {
  if (use_mfma(d)) {  // one wave per split of TC tokens
    int64_t tc = KVQ_AB && tunables().attn_mfma_tc == 64 && d->D == 128 ? 64 : kAttnMfmaTC;
    if ((d->T + tc - 1) / tc > kAttnMaxSplit) tc = kAttnMfmaTC;
    *ts = (uint32_t)tc;
    *nsplit = (uint32_t)((d->T + tc - 1) / tc);
    return *nsplit <= (uint32_t)kAttnMaxSplit;
  }
  const int64_t step = (int64_t)(kAttnBlock / (d->D / 16)) * kAttnUnroll;
  const int64_t bh = d->B * d->Hkv > 0 ? d->B * d->Hkv : 1;
  int64_t m = (d->T * bh + step * 4096 - 1) / (step * 4096);
  if (m < 1) m = 1;
  int64_t per = m * step;
  if (per > kAttnMaxTS) per = kAttnMaxTS / step * step;
  if ((d->T + per - 1) / per > kAttnMaxSplit) per = kAttnMaxTS / step * step;
  *ts = (uint32_t)per;
  *nsplit = (uint32_t)((d->T + per - 1) / per);
  return *nsplit <= (uint32_t)kAttnMaxSplit;
}

// ---- in-launch merge (round 4): the arrival words live in FRONT of the partials, one per (batch row, kv head), padded to
// 16 bytes: their place does not move with T, and a memset of them starts at the caller's workspace pointer
static int64_t arrive_floats(const kvq_attn_dims_t* d) { return (d->B * d->Hkv + 3) / 4 * 4; }
// Does the LDS-staged kernel merge inside its own launch for these dims? (attention over the store, with or without the
// exact new token; not the append step, not a device-side token count.) Needs: the ring plan, head_dim 128, at most 4 query
// heads per kv head and 16 splits (merge_group_one_wave), a workspace the descriptor can range-check.
// A-B key attn_fold (measured slower, see merge_group_one_wave): 0 (default, and always in the default library) = never,
// 1 = where the host call covers several layers (kvq_decode_step_layers: ONE memset of the words per call), 2 = in
// kvq_decode_attn too (a memset per call).
static bool attn_fold_plan(const kvq_attn_dims_t* d) {
  if (!KVQ_AB || tunables().attn_fold <= 0 || !use_lds(d) || d->D != 128 || d->T <= 0 || d->Hq / d->Hkv > 4) return false;
  if (KVQ_AB && (tunables().attn_lds == 2 || tunables().attn_lds == 3 || tunables().attn_lds_nb != 0 || lds_tc() != 64 || tunables().attn_fused)) return false;
  uint32_t ts, ns, tpw;
  if (!plan(d, &ts, &ns, &tpw) || tpw == 0u || ns == 0u || ns > 16u) return false;
  const int64_t rows = d->B * d->Hq * (int64_t)ns;
  return ((rows * 2 + 3) / 4 * 4 + rows * d->D) * 4 < (int64_t(1) << 31);
}

#if KVQ_AB
// ---- fused single launch: which (tokens per wave, waves per workgroup) and how many workgroup splits
struct FusedPlan {
  uint32_t tc, nw, nwg, lds_bytes;
};
constexpr uint32_t kFusedMaxWg = 256;  // workgroup splits per (batch row, kv head): LDS weights of the final merge
static bool plan_fused(const kvq_attn_dims_t* d, FusedPlan* p) {
  if (!use_mfma(d) || d->T <= 0 || !tunables().attn_fused) return false;  // opt-in: measured slower than partial + merge (see header)
  const int64_t nq = d->Hq / d->Hkv;
  int64_t tc = 128, nw = 4;
  if (d->D == 128) {
    const int64_t waves128 = d->B * d->Hkv * ((d->T + 127) / 128);
    if (waves128 < 2048) {  // small batches: more, shorter waves (2+ per SIMD) and still >= 1 workgroup per CU
      tc = 64;
      nw = 8;
    }
    const int64_t ttc = tunables().attn_fused_tc, tnw = tunables().attn_fused_nw;
    if ((ttc == 128 && tnw == 4) || (ttc == 64 && tnw == 8) || (ttc == 32 && tnw == 16) || (ttc == 128 && tnw == 8)) {
      tc = ttc;
      nw = tnw;
    }
  }
  int64_t nwg = (d->T + tc * nw - 1) / (tc * nw);
  uint32_t lds = 4u * fused_lds_floats((uint32_t)tc, (uint32_t)nw, (uint32_t)d->D, (uint32_t)nq, (uint32_t)nwg);
  if (lds > 60u * 1024u && !(tc == 128 && nw == 4)) {  // many query heads per kv head: the 4-wave shape needs the least LDS
    tc = 128;
    nw = 4;
    nwg = (d->T + 511) / 512;
    lds = 4u * fused_lds_floats(128u, 4u, (uint32_t)d->D, (uint32_t)nq, (uint32_t)nwg);
  }
  if (nwg > (int64_t)kFusedMaxWg || lds > 60u * 1024u) return false;
  // partial rows are addressed with 32-bit buffer offsets
  if (nwg * nq * d->D * 4 >= (int64_t(1) << 31)) return false;
  p->tc = (uint32_t)tc;
  p->nw = (uint32_t)nw;
  p->nwg = (uint32_t)nwg;
  p->lds_bytes = lds;
  return true;
}
// workspace floats of the fused path for `nwg` workgroup splits: arrival words, (m, l) rows, acc rows
static int64_t fused_ws_floats(const kvq_attn_dims_t* d, int64_t nwg) {
  const int64_t bh = d->B * d->Hkv, nq = d->Hq / d->Hkv;
  return (2 * bh + 3) / 4 * 4 + bh * nwg * (32 + nq * d->D);
}
static std::atomic<unsigned long long> g_attn_epoch{1};

template <int KBITS, int VBITS>
static void launch_fused(const AttnArgs& a, const FusedPlan& p, const NewTokenArgs* nt, hipStream_t st) {
  FusedArgs f;
  const int64_t bh = (int64_t)a.B * a.Hkv;
  f.cnt = reinterpret_cast<unsigned long long*>(a.ws);
  f.part_ml = a.ws + (2 * bh + 3) / 4 * 4;
  f.part_acc = f.part_ml + bh * p.nwg * 32;
  f.epoch = g_attn_epoch.fetch_add(1, std::memory_order_relaxed) & ((1ull << 44) - 1ull);
  f.nwg = p.nwg;
  f.fuse_quant = nt ? 1 : 0;
  NewTokenArgs none = {};
  const NewTokenArgs& nta = nt ? *nt : none;
  const dim3 grid(p.nwg + (nt ? 2u : 0u), a.Hkv, a.B);
#define KVQ_FUSED(TC_, HD_, NW_) \
  KVQ_LAUNCH((decode_attn_fused_mfma_k<KBITS, VBITS, TC_, HD_, NW_>), grid, dim3(NW_ * kWave), p.lds_bytes, st, a, f, nta)
  if (a.D == 64u) KVQ_FUSED(128, 64, 4);
  else if (p.tc == 128u && p.nw == 4u) KVQ_FUSED(128, 128, 4);
  else if (p.tc == 128u) KVQ_FUSED(128, 128, 8);
  else if (p.tc == 64u) KVQ_FUSED(64, 128, 8);
  else KVQ_FUSED(32, 128, 16);
#undef KVQ_FUSED
}

#endif  // KVQ_AB (fused single launch)

template <int KBITS, int VBITS>
static void launch_partial(const AttnArgs& a, hipStream_t st) {
  const dim3 grid(a.nsplit, a.Hkv, a.B);
  // INT8 keys at head_dim 128 in the streaming kernel: the stored bytes straight into the int8 MFMA, the query as two
  // int8 planes whose cost is amortised over the wave's tiles (50.6 vs 52.3 us per call at batch 8). A-B builds can
  // force that operand form on (attn_k_i8 = 1) or off (0) everywhere.
  constexpr bool kI8 = KBITS == 8;
#if KVQ_AB
  if (a.mfma && a.stream_tpw && a.lds == 2u) {  // coalesced requests into registers, one LDS image per wave
    if (tunables().attn_lds_nb == 1) { KVQ_LAUNCH((decode_attn_coal_mfma_k<KBITS, VBITS, kI8, 1>), grid, dim3(kWave), 0, st, a, a.stream_tpw); return; }
    KVQ_LAUNCH((decode_attn_coal_mfma_k<KBITS, VBITS, kI8, 2>), grid, dim3(kWave), 0, st, a, a.stream_tpw);
    return;
  }
#endif
  if constexpr (kI8) {
    if (a.mfma && a.stream_tpw && a.lds == 1u && a.D == 64u) {  // head_dim 64 (INT8 keys): ring depth 2, eight one-wave workgroups per CU
      constexpr int kSlot64 = 64 * (64 * KBITS / 8 + 64 * VBITS / 8) + 512;
#if KVQ_AB  // ring depth A-B at head_dim 64 (attn_lds_nb = 3 | 4; TG = 4 instantiation)
      if (a.nq <= 4u && tunables().attn_lds_nb == 3) { KVQ_LAUNCH((decode_attn_lds_mfma_k<KBITS, VBITS, 64, kI8, 3, 4, 64>), grid, dim3(kWave), (size_t)(3 * kSlot64), st, a, a.stream_tpw); return; }
      if (a.nq <= 4u && tunables().attn_lds_nb == 4) { KVQ_LAUNCH((decode_attn_lds_mfma_k<KBITS, VBITS, 64, kI8, 4, 4, 64>), grid, dim3(kWave), (size_t)(4 * kSlot64), st, a, a.stream_tpw); return; }
#endif
      if (a.nq <= 4u) KVQ_LAUNCH((decode_attn_lds_mfma_k<KBITS, VBITS, 64, kI8, 2, 4, 64>), grid, dim3(kWave), (size_t)(2 * kSlot64), st, a, a.stream_tpw);
      else KVQ_LAUNCH((decode_attn_lds_mfma_k<KBITS, VBITS, 64, kI8, 2, 1, 64>), grid, dim3(kWave), (size_t)(2 * kSlot64), st, a, a.stream_tpw);
      return;
    }
  }
  if (a.mfma && a.stream_tpw && a.lds && a.D == 128u) {  // (a.lds == 3, A-B: the same kernel with strided tile ownership)
    // ring depth 2: one tile in flight behind the one being reduced, four one-wave workgroups per CU (25.6 KiB each for
    // INT8 keys + INT4 values). Measured at batch 8, 16 K tokens (per layer call incl. merge, profiles/r03d_*): depth 2
    // 40.8 us, depth 3 44.7 us, depth 4 (three workgroups per CU) 86 us; 32-token tiles 42.1-51.6 us.
    constexpr int kSlot = 64 * (128 * KBITS / 8 + 128 * VBITS / 8) + 512;
    int nb = 2;
#if KVQ_AB
    if (tunables().attn_lds_nb >= 1 && tunables().attn_lds_nb <= 4) nb = (int)tunables().attn_lds_nb;
    if (nb == 1) { KVQ_LAUNCH((decode_attn_lds_mfma_k<KBITS, VBITS, 64, kI8, 1>), grid, dim3(kWave), (size_t)kSlot, st, a, a.stream_tpw); return; }
    if (lds_tc() == 32) {
      constexpr int kSlot32 = 32 * (128 * KBITS / 8 + 128 * VBITS / 8) + 512;
      if (nb == 2) KVQ_LAUNCH((decode_attn_lds_mfma_k<KBITS, VBITS, 32, kI8, 2>), grid, dim3(kWave), (size_t)(2 * kSlot32), st, a, a.stream_tpw);
      else if (nb == 3) KVQ_LAUNCH((decode_attn_lds_mfma_k<KBITS, VBITS, 32, kI8, 3>), grid, dim3(kWave), (size_t)(3 * kSlot32), st, a, a.stream_tpw);
      else KVQ_LAUNCH((decode_attn_lds_mfma_k<KBITS, VBITS, 32, kI8, 4>), grid, dim3(kWave), (size_t)(4 * kSlot32), st, a, a.stream_tpw);
      return;
    }
    if (nb == 4) { KVQ_LAUNCH((decode_attn_lds_mfma_k<KBITS, VBITS, 64, kI8, 4>), grid, dim3(kWave), (size_t)(4 * kSlot), st, a, a.stream_tpw); return; }
    if (nb == 3) {
      if constexpr (kI8) {  // (with the one-output score product / consume_direct too)
        if (a.nq <= 4u && tunables().attn_tg != 1) { KVQ_LAUNCH((decode_attn_lds_mfma_k<KBITS, VBITS, 64, kI8, 3, 4>), grid, dim3(kWave), (size_t)(3 * kSlot), st, a, a.stream_tpw); return; }
      }
      KVQ_LAUNCH((decode_attn_lds_mfma_k<KBITS, VBITS, 64, kI8, 3>), grid, dim3(kWave), (size_t)(3 * kSlot), st, a, a.stream_tpw);
      return;
    }
#endif
    (void)nb;
    if constexpr (kI8) {  // at most 4 query heads per kv head: one score output per tile (AttnStream TG = 4)
      if (a.nq <= 4u && !(KVQ_AB && tunables().attn_tg == 1)) {
        KVQ_LAUNCH((decode_attn_lds_mfma_k<KBITS, VBITS, 64, kI8, 2, 4>), grid, dim3(kWave), (size_t)(2 * kSlot), st, a, a.stream_tpw);
        return;
      }
    }
    KVQ_LAUNCH((decode_attn_lds_mfma_k<KBITS, VBITS, 64, kI8, 2>), grid, dim3(kWave), (size_t)(2 * kSlot), st, a, a.stream_tpw);
    return;
  }
#if KVQ_AB
  if (a.mfma && a.stream_tpw) {
    const int64_t ki8 = tunables().attn_k_i8;
    const bool i8 = kI8 && ki8 != 0;
    if constexpr (kI8) {
      if (i8 && stream_tc() == 64 && !tunables().attn_stream_roll) { KVQ_LAUNCH((decode_attn_stream_mfma_k<KBITS, VBITS, 64, 128, kI8, false>), grid, dim3(kWave), 0, st, a, a.stream_tpw); return; }
      if (i8 && stream_tc() == 32) { KVQ_LAUNCH((decode_attn_stream_mfma_k<KBITS, VBITS, 32, 128, kI8, false>), grid, dim3(kWave), 0, st, a, a.stream_tpw); return; }
    }
    if (!i8 && stream_tc() == 64 && tunables().attn_stream_roll) { KVQ_LAUNCH((decode_attn_stream_mfma_k<KBITS, VBITS, 64, 128, false, true>), grid, dim3(kWave), 0, st, a, a.stream_tpw); return; }
    if (!i8 && stream_tc() == 64) { KVQ_LAUNCH((decode_attn_stream_mfma_k<KBITS, VBITS, 64, 128>), grid, dim3(kWave), 0, st, a, a.stream_tpw); return; }
    if (!i8) { KVQ_LAUNCH((decode_attn_stream_mfma_k<KBITS, VBITS, 32, 128>), grid, dim3(kWave), 0, st, a, a.stream_tpw); return; }
    KVQ_LAUNCH((decode_attn_stream_mfma_k<KBITS, VBITS, 64, 128, kI8, true>), grid, dim3(kWave), 0, st, a, a.stream_tpw);
    return;
  }
#endif
  if (a.mfma) {
#if KVQ_AB
    if constexpr (kI8) {
      if (a.D == 128u && tunables().attn_k_i8 > 0) {
        if (a.TS == 64u) KVQ_LAUNCH((decode_attn_partial_mfma_k<KBITS, VBITS, 64, 128, kI8>), grid, dim3(kWave), 0, st, a);
        else KVQ_LAUNCH((decode_attn_partial_mfma_k<KBITS, VBITS, kAttnMfmaTC, 128, kI8>), grid, dim3(kWave), 0, st, a);
        return;
      }
    }
    if (a.D == 128u && a.TS == 64u) { KVQ_LAUNCH((decode_attn_partial_mfma_k<KBITS, VBITS, 64, 128>), grid, dim3(kWave), 0, st, a); return; }
#endif
    if (a.D == 64u) KVQ_LAUNCH((decode_attn_partial_mfma_k<KBITS, VBITS, kAttnMfmaTC, 64>), grid, dim3(kWave), 0, st, a);
    else KVQ_LAUNCH((decode_attn_partial_mfma_k<KBITS, VBITS, kAttnMfmaTC, 128>), grid, dim3(kWave), 0, st, a);
    return;
  }
  if (a.nq == 1) KVQ_LAUNCH((decode_attn_partial_k<KBITS, VBITS, 1>), grid, dim3(kAttnBlock), 0, st, a);
  else if (a.nq == 2) KVQ_LAUNCH((decode_attn_partial_k<KBITS, VBITS, 2>), grid, dim3(kAttnBlock), 0, st, a);
  else if (a.nq <= 4) KVQ_LAUNCH((decode_attn_partial_k<KBITS, VBITS, 4>), grid, dim3(kAttnBlock), 0, st, a);
  else KVQ_LAUNCH((decode_attn_partial_k<KBITS, VBITS, 8>), grid, dim3(kAttnBlock), 0, st, a);
}

}  // namespace kvq

using namespace kvq;

extern "C" {

int64_t kvq_decode_attn_workspace(const kvq_attn_dims_t* d) {
  if (!d || d->B <= 0 || d->Hq <= 0 || d->Hkv <= 0 || d->T < 0 || d->D <= 0) return -1;
  if (d->D != 32 && d->D != 64 && d->D != 128 && d->D != 256) return -1;
  uint32_t ts, ns, ns1;
  if (!plan(d, &ts, &ns)) return -1;
  // kvq_decode_step_dev always takes one-tile splits, which can be more than the streaming layout's: cover both
  if (plan_onetile(d, &ts, &ns1) && ns1 > ns) ns = ns1;
  const int64_t rows = d->B * d->Hq * (int64_t)(ns > 0 ? ns : 1);
  const int64_t legacy = (rows * 2 + 3) / 4 * 4 + rows * d->D;
  // the fused single-launch path (whatever shape the tunables pick: at most one split per 512 tokens)
#if KVQ_AB
  const int64_t fused = use_mfma(d) ? fused_ws_floats(d, (d->T + 511) / 512 > 0 ? (d->T + 511) / 512 : 1) : 0;
  return arrive_floats(d) + (legacy > fused ? legacy : fused);
#else
  return arrive_floats(d) + legacy;
#endif
}

}  // extern "C"

// shared body of kvq_decode_attn / kvq_decode_step; nt != nullptr: quantise the new token in the merge launch
static int decode_attn_impl(const char* name, const void* q, int64_t q_sb, int64_t q_sh, const uint8_t* k_store,
                            const kvq_strides_t* k_st, const float* k_scales, int k_bits, const uint8_t* v_store,
                            const kvq_strides_t* v_st, const float* v_scales, int v_bits, const void* k_new,
                            int64_t kn_sb, int64_t kn_sh, const void* v_new, int64_t vn_sb, int64_t vn_sh, void* out,
                            int64_t o_sb, int64_t o_sh, int dtype, float sm_scale, float* workspace,
                            int64_t workspace_floats, const kvq_attn_dims_t* d, void* stream, const NewTokenArgs* nt,
                            const int32_t* t_dev = nullptr, int fold = 0) {  // fold: 0 = two launches, 1 = merge in the launch (the caller zeroed the arrival words), 2 = zero them here
  if (!d || !q || !out) {
    set_error("%s: NULL q / out / dims", name);
    return KVQ_E_NULL;
  }
  if (const int rcd = check_device(out, name)) return rcd;
  if (d->B <= 0 || d->Hq <= 0 || d->Hkv <= 0 || d->T < 0 || d->B >= (1 << 16) || d->Hkv >= (1 << 16) ||
      d->T >= (int64_t(1) << 31) || d->Hq % d->Hkv != 0 || d->Hq / d->Hkv > (use_mfma(d) ? 16 : 8)) {
    set_error("%s: bad dims B=%lld Hq=%lld Hkv=%lld T=%lld (need Hq %% Hkv == 0, Hq / Hkv <= 8, or <= 16 at head_dim 64 / 128)", name, (long long)d->B,
              (long long)d->Hq, (long long)d->Hkv, (long long)d->T);
    return KVQ_E_DIMS;
  }
  if (d->D != 32 && d->D != 64 && d->D != 128 && d->D != 256) {
    set_error("%s: head_dim %lld not supported (32, 64, 128, 256)", name, (long long)d->D);
    return KVQ_E_DIMS;
  }
  if (dtype != KVQ_F16 && dtype != KVQ_BF16) {
    set_error("%s: dtype %d not supported (KVQ_F16, KVQ_BF16)", name, dtype);
    return KVQ_E_DTYPE;
  }
  if ((k_bits != 8 && k_bits != 4) || (v_bits != 8 && v_bits != 4)) {
    set_error("%s: k_bits / v_bits must be 8 or 4", name);
    return KVQ_E_DIMS;
  }
  if ((k_new == nullptr) != (v_new == nullptr)) {
    set_error("%s: k_new and v_new must be given together", name);
    return KVQ_E_NULL;
  }
  if (d->T == 0 && !k_new) {
    set_error("%s: nothing to attend to (T == 0 and no new token)", name);
    return KVQ_E_DIMS;
  }
  if (d->T > 0 && (!k_store || !v_store || !k_scales || !v_scales || !k_st || !v_st || !workspace)) {
    set_error("%s: NULL store / scales / strides / workspace", name);
    return KVQ_E_NULL;
  }
  const int64_t need = kvq_decode_attn_workspace(d);
  if (d->T > 0 && workspace_floats < need) {
    set_error("%s: workspace of %lld floats, %lld needed", name, (long long)workspace_floats, (long long)need);
    return KVQ_E_DIMS;
  }
  const int kal = k_bits == 8 ? 16 : 8, val = v_bits == 8 ? 16 : 8;
  if (d->T > 0 && (!aligned(k_store, kal) || k_st->b % kal || k_st->h % kal || k_st->t % kal || !aligned(v_store, val) ||
                   v_st->b % val || v_st->h % val || v_st->t % val)) {
    set_error("%s: store rows must be %d / %d byte aligned", name, kal, val);
    return KVQ_E_DIMS;
  }
  if (d->T > 0 && (d->T * k_st->t >= (int64_t(1) << 31) || d->T * v_st->t >= (int64_t(1) << 31))) {
    set_error("%s: one (batch, kv head) row of the store must stay below 2 GiB", name);
    return KVQ_E_DIMS;
  }
  if (!aligned(q, 16) || q_sb % 8 || q_sh % 8) {
    set_error("%s: q must be 16-byte aligned with strides that are multiples of 8 elements", name);
    return KVQ_E_DIMS;
  }
  AttnArgs a;
  a.q = q; a.q_sb = q_sb; a.q_sh = q_sh;
  a.k = k_store; a.v = v_store;
  if (d->T > 0) {
    a.k_sb = k_st->b; a.k_sh = k_st->h; a.k_st = k_st->t;
    a.v_sb = v_st->b; a.v_sh = v_st->h; a.v_st = v_st->t;
  } else {
    a.k_sb = a.k_sh = a.k_st = a.v_sb = a.v_sh = a.v_st = 0;
  }
  a.k_scale = k_scales; a.v_scale = v_scales;
  a.kn = k_new; a.kn_sb = kn_sb; a.kn_sh = kn_sh;
  a.vn = v_new; a.vn_sb = vn_sb; a.vn_sh = vn_sh;
  a.out = out; a.o_sb = o_sb; a.o_sh = o_sh;
  const int64_t arrive_n = arrive_floats(d);  // the arrival words of the in-launch merge sit in front of the partials
  a.ws = workspace ? workspace + arrive_n : workspace;
  workspace_floats -= arrive_n;
  a.arrive = nullptr;
  a.ws_bytes = 0u;
  a.fold_has_new = 0;
  a.sm_scale = sm_scale;
  a.B = (uint32_t)d->B; a.Hq = (uint32_t)d->Hq; a.Hkv = (uint32_t)d->Hkv; a.T = (uint32_t)d->T; a.D = (uint32_t)d->D;
  a.nq = (uint32_t)(d->Hq / d->Hkv);
  a.lpt_shift = ilog2_exact(d->D / 16);
  a.dtype = dtype;
  a.mfma = use_mfma(d) ? 1 : 0;
  a.stream_tpw = 0u;
  a.lds = (!t_dev && use_lds(d, k_bits)) ? (KVQ_AB && tunables().attn_lds == 2 ? 2u : (KVQ_AB && tunables().attn_lds == 3 ? 3u : 1u)) : 0u;
  a.t_dev = t_dev;
  if (!(t_dev ? plan_onetile(d, &a.TS, &a.nsplit) : plan(d, &a.TS, &a.nsplit, &a.stream_tpw, k_bits))) {
    set_error("%s: T=%lld needs more than %d splits of %d tokens", name, (long long)d->T, kAttnMaxSplit, kAttnMaxTS);
    return KVQ_E_DIMS;
  }
  a.acc_off = ((int64_t)a.B * a.Hq * a.nsplit * 2 + 3) / 4 * 4;
  // the split layout actually chosen (a device-side T always takes one-tile splits) must fit too
  if (d->T > 0 && workspace_floats < a.acc_off + (int64_t)a.B * a.Hq * a.nsplit * a.D) {
    set_error("%s: workspace of %lld floats, %lld needed for %u splits", name, (long long)workspace_floats,
              (long long)(a.acc_off + (int64_t)a.B * a.Hq * a.nsplit * a.D), a.nsplit);
    return KVQ_E_DIMS;
  }
  if (d->T > 0 && !aligned(workspace, 16)) {
    set_error("%s: workspace must be 16-byte aligned", name);
    return KVQ_E_DIMS;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (fold && !nt && !t_dev && a.lds == 1u && a.stream_tpw && attn_fold_plan(d)) {  // ONE launch: partials, ticket, merge by the last wave
    if (fold == 2) {
      const hipError_t e = hipMemsetAsync(workspace, 0, (size_t)arrive_n * 4, st);
      if (e != hipSuccess) {
        set_error("%s: hipMemsetAsync of the arrival words: %s", name, hipGetErrorString(e));
        return (int)e;
      }
    }
    a.arrive = reinterpret_cast<uint32_t*>(workspace);
    a.ws_bytes = (uint32_t)((a.acc_off + (int64_t)a.B * a.Hq * a.nsplit * a.D) * 4);
    a.fold_has_new = a.kn ? 1 : 0;
    if (!a.kn) {  // no new token: valid addresses for the merge's unconditional loads (values unused)
      a.kn = a.vn = a.q;
      a.kn_sb = a.vn_sb = a.q_sb;
      a.kn_sh = a.vn_sh = 0;
    }
    if (k_bits == 8 && v_bits == 8) launch_partial<8, 8>(a, st);
    else if (k_bits == 8) launch_partial<8, 4>(a, st);
    else if (v_bits == 8) launch_partial<4, 8>(a, st);
    else launch_partial<4, 4>(a, st);
    return check_launch(name);
  }
#if KVQ_AB
  FusedPlan fp;
  // a captured HIP graph would replay the launch's host-side epoch: the arrival word then already holds
  // epoch | nwg, nobody draws the last ticket and `out` stays stale — the fused plan is refused while capturing
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
  if (!t_dev && cap == hipStreamCaptureStatusNone && plan_fused(d, &fp)) {  // ONE launch: tiles, in-workgroup merge, ticketed final merge, new-token quantise
    if (k_bits == 8 && v_bits == 8) launch_fused<8, 8>(a, fp, nt, st);
    else if (k_bits == 8) launch_fused<8, 4>(a, fp, nt, st);
    else if (v_bits == 8) launch_fused<4, 8>(a, fp, nt, st);
    else launch_fused<4, 4>(a, fp, nt, st);
    return check_launch(name);
  }
#endif
  if (a.nsplit > 0) {
    if (k_bits == 8 && v_bits == 8) launch_partial<8, 8>(a, st);
    else if (k_bits == 8) launch_partial<8, 4>(a, st);
    else if (v_bits == 8) launch_partial<4, 8>(a, st);
    else launch_partial<4, 4>(a, st);
    const int rc = check_launch(name);
    if (rc) return rc;
  }
  NewTokenArgs none = {};
  if (a.nsplit >= 1u && a.nsplit <= (uint32_t)kAttnBlock && a.D <= (uint32_t)kAttnBlock && (!KVQ_AB || tunables().attn_merge_fast))
  {
    AttnArgs af = a;
    if (!a.kn) {  // no new token: valid addresses for the kernel's unconditional loads (values unused)
      af.kn = af.vn = a.q;
      af.kn_sb = af.vn_sb = a.q_sb;
      af.kn_sh = af.vn_sh = 0;
    }
    const uint32_t groups = (uint32_t)kAttnBlock / (a.D / 4u), need = (a.nsplit + groups - 1u) / groups;
    const dim3 mgrid(a.Hq + (nt ? 2u : 0u), a.B);
    if (need <= 2u && a.D == 128u && a.nsplit <= 16u && tunables().attn_merge_wave != 0) {  // one wave per head, no workgroup
      const dim3 wgrid((a.Hq + 3u) / 4u + (nt ? 2u : 0u), a.B);
      KVQ_LAUNCH((decode_attn_merge_fast_k<2>), wgrid, dim3(kAttnBlock), 0, st, af, nt ? *nt : none, (nt ? 1 : 0) | 2, a.kn ? 1 : 0);
    } else if (need <= 2u) KVQ_LAUNCH((decode_attn_merge_fast_k<2>), mgrid, dim3(kAttnBlock), 0, st, af, nt ? *nt : none, nt ? 1 : 0, a.kn ? 1 : 0);
#if KVQ_AB  // (17-32 splits: depth 4 measured within 0.1 us of depth 16; one instantiation less in the default library)
    else if (need <= 4u) KVQ_LAUNCH((decode_attn_merge_fast_k<4>), mgrid, dim3(kAttnBlock), 0, st, af, nt ? *nt : none, nt ? 1 : 0, a.kn ? 1 : 0);
#endif
    else KVQ_LAUNCH((decode_attn_merge_fast_k<kMergePF>), mgrid, dim3(kAttnBlock), 0, st, af, nt ? *nt : none, nt ? 1 : 0, a.kn ? 1 : 0);
  } else
    KVQ_LAUNCH(decode_attn_merge_k, dim3(a.Hq + (nt ? 2u : 0u), a.B), dim3(kAttnBlock), 0, st, a, nt ? *nt : none,
                       nt ? 1 : 0);
  return check_launch(name);
}

extern "C" {

// The split count is not monotone in T (tokens per split grow with T), so a decode loop that sizes
// its workspace ONCE for a capacity asks for the maximum over every T' <= T.
int64_t kvq_decode_attn_workspace_cap(const kvq_attn_dims_t* d) {
  if (!d || d->B <= 0 || d->Hq <= 0 || d->Hkv <= 0 || d->T < 0 || d->D <= 0) return -1;
  if (d->D != 32 && d->D != 64 && d->D != 128 && d->D != 256) return -1;
  kvq_attn_dims_t t = *d;
  uint32_t ts, ns, ns_max = 1;
  if (!plan(&t, &ts, &ns)) return -1;
  ns_max = ns > ns_max ? ns : ns_max;
  // kvq_decode_step_dev (device-side token count) always takes the one-tile plan, which can split finer than the
  // streaming plan a large batch gets: the capacity covers both
  if (plan_onetile(&t, &ts, &ns) && ns > ns_max) ns_max = ns;
  for (int64_t tt = 32; tt < d->T; tt += 32) {  // + 1 below covers a change of split size between two probes
    t.T = tt;
    if (plan(&t, &ts, &ns) && ns > ns_max) ns_max = ns;
    if (plan_onetile(&t, &ts, &ns) && ns > ns_max) ns_max = ns;
  }
  const int64_t rows = d->B * d->Hq * (int64_t)(ns_max + 1);
  const int64_t legacy = (rows * 2 + 3) / 4 * 4 + rows * d->D;
#if KVQ_AB
  const int64_t fused = use_mfma(d) ? fused_ws_floats(d, (d->T + 511) / 512 + 1) : 0;  // monotone in T
  return arrive_floats(d) + (legacy > fused ? legacy : fused);
#else
  return arrive_floats(d) + legacy;
#endif
}

int kvq_decode_attn(const void* q, int64_t q_sb, int64_t q_sh, const uint8_t* k_store, const kvq_strides_t* k_st,
                    const float* k_scales, int k_bits, const uint8_t* v_store, const kvq_strides_t* v_st,
                    const float* v_scales, int v_bits, const void* k_new, int64_t kn_sb, int64_t kn_sh,
                    const void* v_new, int64_t vn_sb, int64_t vn_sh, void* out, int64_t o_sb, int64_t o_sh, int dtype,
                    float sm_scale, float* workspace, int64_t workspace_floats, const kvq_attn_dims_t* d,
                    void* stream) {
  // (A-B key attn_fold = 2: the in-launch merge here too, behind its own memset of the arrival words)
  return decode_attn_impl("kvq_decode_attn", q, q_sb, q_sh, k_store, k_st, k_scales, k_bits, v_store, v_st, v_scales, v_bits,
                          k_new, kn_sb, kn_sh, v_new, vn_sb, vn_sh, out, o_sb, o_sh, dtype, sm_scale, workspace,
                          workspace_floats, d, stream, nullptr, nullptr, KVQ_AB && tunables().attn_fold == 2 ? 2 : 0);
}


int kvq_decode_step(const void* q, int64_t q_sb, int64_t q_sh, const void* k_new, int64_t kn_sb, int64_t kn_sh,
                    const void* v_new, int64_t vn_sb, int64_t vn_sh, uint8_t* k_store, const kvq_strides_t* k_st,
                    float* k_scales, int k_bits, uint8_t* v_store, const kvq_strides_t* v_st, float* v_scales,
                    int v_bits, void* out, int64_t o_sb, int64_t o_sh, int dtype, float sm_scale, float eps,
                    float* workspace, int64_t workspace_floats, const kvq_attn_dims_t* d, void* stream) {
  const char* name = "kvq_decode_step";
  if (!d || !k_new || !v_new || !k_store || !v_store || !k_scales || !v_scales || !k_st || !v_st || !workspace) {
    set_error("%s: NULL argument", name);
    return KVQ_E_NULL;
  }
  // The new token's K / V go into slot T (append_from_past, ops.py:323-330). Small slices (the usual
  // case) are quantised by two extra workgroups of the attention's merge launch: a decode step is two
  // launches. Nothing reads slot T in that launch, and the partial kernel has read [0, T) before it.
  if (d->B * d->Hkv * d->D <= 65536 && (dtype == KVQ_F16 || dtype == KVQ_BF16) && d->T >= 0) {
    NewTokenArgs nt;
    nt.x[0] = k_new; nt.xs_b[0] = kn_sb; nt.xs_h[0] = kn_sh;
    nt.x[1] = v_new; nt.xs_b[1] = vn_sb; nt.xs_h[1] = vn_sh;
    nt.q[0] = k_store + d->T * k_st->t; nt.qs_b[0] = k_st->b; nt.qs_h[0] = k_st->h; nt.scale[0] = k_scales + d->T; nt.bits[0] = k_bits;
    nt.q[1] = v_store + d->T * v_st->t; nt.qs_b[1] = v_st->b; nt.qs_h[1] = v_st->h; nt.scale[1] = v_scales + d->T; nt.bits[1] = v_bits;
    nt.qs_t[0] = k_st->t; nt.qs_t[1] = v_st->t;
    nt.B = (uint32_t)d->B; nt.H = (uint32_t)d->Hkv; nt.D = (uint32_t)d->D; nt.eps = eps;
    return decode_attn_impl(name, q, q_sb, q_sh, k_store, k_st, k_scales, k_bits, v_store, v_st, v_scales, v_bits, k_new, kn_sb,
                            kn_sh, v_new, vn_sb, vn_sh, out, o_sb, o_sh, dtype, sm_scale, workspace, workspace_floats, d,
                            stream, &nt);
  }
  // large slices: attention first, then the regular quantise calls (stream order keeps the
  // attention's reads of [0, T) ahead of the writes to slot T)
  int rc = kvq_decode_attn(q, q_sb, q_sh, k_store, k_st, k_scales, k_bits, v_store, v_st, v_scales, v_bits, k_new,
                           kn_sb, kn_sh, v_new, vn_sb, vn_sh, out, o_sb, o_sh, dtype, sm_scale, workspace,
                           workspace_floats, d, stream);
  if (rc) return rc;
  const kvq_dims_t qd = {1, d->B, d->Hkv, 1, d->D};
  const kvq_strides_t kin = {0, kn_sb, kn_sh, d->D}, vin = {0, vn_sb, vn_sh, d->D};
  float* absmax_ws = workspace + arrive_floats(d);  // only the generic two-pass path uses it (1 float, behind the arrival words); the attention is already enqueued
  rc = k_bits == 8
           ? kvq_quant_i8_tokens(k_new, nullptr, &kin, dtype, reinterpret_cast<int8_t*>(k_store + d->T * k_st->t), k_st,
                                 k_scales + d->T, 0, absmax_ws, eps, &qd, stream)
           : kvq_quant_i4_tokens(k_new, nullptr, &kin, dtype, k_store + d->T * k_st->t, k_st, k_scales + d->T, 0,
                                 absmax_ws, eps, &qd, stream);
  if (rc) return rc;
  return v_bits == 8
             ? kvq_quant_i8_tokens(v_new, nullptr, &vin, dtype, reinterpret_cast<int8_t*>(v_store + d->T * v_st->t), v_st,
                                   v_scales + d->T, 0, absmax_ws, eps, &qd, stream)
             : kvq_quant_i4_tokens(v_new, nullptr, &vin, dtype, v_store + d->T * v_st->t, v_st, v_scales + d->T, 0,
                                   absmax_ws, eps, &qd, stream);
}

int kvq_decode_step_dev(const void* q, int64_t q_sb, int64_t q_sh, const void* k_new, int64_t kn_sb, int64_t kn_sh,
                        const void* v_new, int64_t vn_sb, int64_t vn_sh, uint8_t* k_store, const kvq_strides_t* k_st,
                        float* k_scales, int k_bits, uint8_t* v_store, const kvq_strides_t* v_st, float* v_scales,
                        int v_bits, void* out, int64_t o_sb, int64_t o_sh, int dtype, float sm_scale, float eps,
                        float* workspace, int64_t workspace_floats, const kvq_attn_dims_t* d, const int32_t* t_dev,
                        void* stream) {
  const char* name = "kvq_decode_step_dev";
  if (!d || !t_dev || !k_new || !v_new || !k_store || !v_store || !k_scales || !v_scales || !k_st || !v_st || !workspace) {
    set_error("%s: NULL argument", name);
    return KVQ_E_NULL;
  }
  if (d->T < 1 || d->B * d->Hkv * d->D > 65536 || (dtype != KVQ_F16 && dtype != KVQ_BF16)) {
    set_error("%s: needs an upper bound T >= 1, B * Hkv * D <= 65536 and fp16 / bf16 tensors", name);
    return KVQ_E_DIMS;
  }
  NewTokenArgs nt;  // slot 0: the merge launch adds the device-side T
  nt.x[0] = k_new; nt.xs_b[0] = kn_sb; nt.xs_h[0] = kn_sh;
  nt.x[1] = v_new; nt.xs_b[1] = vn_sb; nt.xs_h[1] = vn_sh;
  nt.q[0] = k_store; nt.qs_b[0] = k_st->b; nt.qs_h[0] = k_st->h; nt.qs_t[0] = k_st->t; nt.scale[0] = k_scales; nt.bits[0] = k_bits;
  nt.q[1] = v_store; nt.qs_b[1] = v_st->b; nt.qs_h[1] = v_st->h; nt.qs_t[1] = v_st->t; nt.scale[1] = v_scales; nt.bits[1] = v_bits;
  nt.B = (uint32_t)d->B; nt.H = (uint32_t)d->Hkv; nt.D = (uint32_t)d->D; nt.eps = eps;
  return decode_attn_impl(name, q, q_sb, q_sh, k_store, k_st, k_scales, k_bits, v_store, v_st, v_scales, v_bits, k_new, kn_sb,
                          kn_sh, v_new, vn_sb, vn_sh, out, o_sb, o_sh, dtype, sm_scale, workspace, workspace_floats, d,
                          stream, &nt, t_dev);
}

int kvq_decode_step_layers(int64_t n_layers, int append, const void* const* q, int64_t q_sb, int64_t q_sh,
                           const void* const* k_new, int64_t kn_sb, int64_t kn_sh, const void* const* v_new,
                           int64_t vn_sb, int64_t vn_sh, uint8_t* const* k_store, const kvq_strides_t* k_st,
                           float* const* k_scales, int k_bits, uint8_t* const* v_store, const kvq_strides_t* v_st,
                           float* const* v_scales, int v_bits, void* const* out, int64_t o_sb, int64_t o_sh, int dtype,
                           float sm_scale, float eps, float* workspace, int64_t workspace_floats,
                           const kvq_attn_dims_t* d, void* stream) {
  const char* name = "kvq_decode_step_layers";
  if (n_layers < 0 || !q || !out || !k_store || !v_store || !k_scales || !v_scales || (append && (!k_new || !v_new)) ||
      ((k_new == nullptr) != (v_new == nullptr))) {
    set_error("%s: NULL pointer table (or n_layers < 0)", name);
    return KVQ_E_NULL;
  }
  // attention only (no append), on the LDS-staged kernel: every layer's merge runs inside its partial launch. ONE memset of
  // the arrival words covers the whole call (each launch leaves them zero for the next: stream order)
  int fold = 0;
  if (!append && n_layers > 0 && d && workspace && d->B > 0 && d->Hkv > 0 && d->Hq > 0 && d->Hq % d->Hkv == 0 && attn_fold_plan(d) &&
      workspace_floats >= kvq_decode_attn_workspace(d)) {
    const hipError_t e = hipMemsetAsync(workspace, 0, (size_t)arrive_floats(d) * 4, reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
      set_error("%s: hipMemsetAsync of the arrival words: %s", name, hipGetErrorString(e));
      return (int)e;
    }
    fold = 1;
  }
  for (int64_t i = 0; i < n_layers; ++i) {
    const void* kn = k_new ? k_new[i] : nullptr;
    const void* vn = v_new ? v_new[i] : nullptr;
    if (fold) {
      const int rc = decode_attn_impl(name, q[i], q_sb, q_sh, k_store[i], k_st, k_scales[i], k_bits, v_store[i], v_st, v_scales[i], v_bits, kn,
                                      kn_sb, kn_sh, vn, vn_sb, vn_sh, out[i], o_sb, o_sh, dtype, sm_scale, workspace, workspace_floats, d,
                                      stream, nullptr, nullptr, 1);
      if (rc) return rc;
      continue;
    }
    const int rc = append ? kvq_decode_step(q[i], q_sb, q_sh, kn, kn_sb, kn_sh, vn, vn_sb, vn_sh, k_store[i], k_st, k_scales[i], k_bits,
                                            v_store[i], v_st, v_scales[i], v_bits, out[i], o_sb, o_sh, dtype, sm_scale, eps,
                                            workspace, workspace_floats, d, stream)
                          : kvq_decode_attn(q[i], q_sb, q_sh, k_store[i], k_st, k_scales[i], k_bits, v_store[i], v_st,
                                            v_scales[i], v_bits, kn, kn_sb, kn_sh, vn, vn_sb, vn_sh, out[i], o_sb, o_sh, dtype,
                                            sm_scale, workspace, workspace_floats, d, stream);
    if (rc) return rc;
  }
  return 0;
}

}  // extern "C"
