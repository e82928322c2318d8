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
// Split-T flash decoding, HBM-bound byte streaming on VALU (no MFMA: one query row per head):
//   decode_attn_partial_k  grid (split, kv head, batch), 256 threads, TS tokens per workgroup,
//                          all Hq/Hkv query heads of the kv head in one pass (K/V read once);
//                          writes (m, l, acc[D]) per (b, hq, split) to the workspace
//   decode_attn_merge_k    grid (hq, batch): log-sum-exp merge of the splits and of the new token
// 16 elements per lane per token: 16-byte (INT8) / 8-byte (INT4) loads, D/16 lanes per token.
#include "kvq_common.h"

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
  float* ws;
  float sm_scale;
  uint32_t B, Hq, Hkv, T, D, TS, nsplit, nq;
  int32_t lpt_shift;  // log2(D / 16): lanes per token
  int32_t dtype;      // KVQ_F16 | KVQ_BF16 (q, k_new, v_new, out)
};

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
__device__ inline void load_q16(const char* p, int dtype, f16x2 (&qv)[8]) {
  const u32x4 a = *reinterpret_cast<const u32x4*>(p);
  const u32x4 b = *(reinterpret_cast<const u32x4*>(p) + 1);
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
  __device__ inline void zero() { w = u32x4{0u, 0u, 0u, 0u}; }
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
  __device__ inline void zero() { w = u32x2{0x88888888u, 0x88888888u}; }  // nibble 8 = value 0
  // nibble = q + 8, even element in the HIGH nibble (ops.py:61-63); pair order per 8 elements:
  // (e0,e2) (e4,e6) (e1,e3) (e5,e7) — load_q16<true> arranges the query the same way
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

__device__ inline float wave_fmax(float v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v = fmaxf(v, __shfl_xor(v, s));
  return v;
}
__device__ inline float wave_fsum(float v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
  return v;
}

template <int KBITS, int VBITS, int NQ>
__global__ __launch_bounds__(kAttnBlock) void decode_attn_partial_k(const AttnArgs a) {
  __shared__ float s_p[NQ][kAttnMaxTS];        // scores, then p[t] * sv[t]
  __shared__ float s_acc[kAttnBlock / kWave][NQ][256];
  __shared__ float s_w[NQ];                    // sum_t p[t] * sv[t]  (bias fold)
  const uint32_t tid = threadIdx.x;
  const uint32_t split = blockIdx.x, hk = blockIdx.y, b = blockIdx.z;
  const uint32_t t0 = split * a.TS;
  const uint32_t nt = a.T - t0 < a.TS ? a.T - t0 : a.TS;
  const uint32_t lpt = 1u << a.lpt_shift;
  const uint32_t ld = tid & (lpt - 1u);        // which 16-element slice of D
  const uint32_t tl = tid >> a.lpt_shift;      // token lane
  const uint32_t TL = kAttnBlock >> a.lpt_shift;

  // ---- phase A: scores ------------------------------------------------------------------
  {
    f16x2 qv[NQ][8];
#pragma unroll
    for (int h = 0; h < NQ; ++h) {
      if ((uint32_t)h < a.nq) {
        const char* qp = reinterpret_cast<const char*>(a.q) +
                         ((int64_t)b * a.q_sb + (int64_t)(hk * a.nq + h) * a.q_sh + (int64_t)ld * 16) * 2;
        load_q16<KBITS == 4>(qp, a.dtype, qv[h]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[h][j] = f16x2{(f16)0.0f, (f16)0.0f};
      }
    }
    const uint8_t* kb = a.k + (int64_t)b * a.k_sb + (int64_t)hk * a.k_sh + (int64_t)t0 * a.k_st +
                        (int64_t)ld * Raw16<KBITS>::kBytes;
    for (uint32_t base = 0; base < nt; base += TL * kAttnUnroll) {  // uniform trip count
      Raw16<KBITS> raw[kAttnUnroll];
#pragma unroll
      for (int u = 0; u < kAttnUnroll; ++u) {
        const uint32_t i = base + u * TL + tl;
        if (i < nt) raw[u].load(kb + (int64_t)i * a.k_st);
        else raw[u].zero();
      }
#pragma unroll
      for (int u = 0; u < kAttnUnroll; ++u) {
        const uint32_t i = base + u * TL + tl;
        f16x2 kp[8];
        raw[u].to_h2(kp);
        const float sc = i < nt ? a.k_scale[t0 + i] * a.sm_scale : 0.0f;
#pragma unroll
        for (int h = 0; h < NQ; ++h) {
          float s = 0.0f;
#pragma unroll
          for (int j = 0; j < 8; ++j) s = __builtin_amdgcn_fdot2(kp[j], qv[h][j], s, false);
          s = group_fadd(s, a.lpt_shift);
          if (ld == 0u && i < nt) s_p[h][i] = s * sc;
        }
      }
    }
  }
  __syncthreads();

  // ---- phase B: per-head softmax over this split (one wave per head, round robin) -----------
  {
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    for (uint32_t h = wave; h < a.nq; h += kAttnBlock / kWave) {
      float m = -INFINITY;
      for (uint32_t i = lane; i < nt; i += kWave) m = fmaxf(m, s_p[h][i]);
      m = wave_fmax(m);
      float l = 0.0f, wsum = 0.0f;
      for (uint32_t i = lane; i < nt; i += kWave) {
        const float p = __expf(s_p[h][i] - m);
        const float pv = p * a.v_scale[t0 + i];
        l += p;
        wsum += pv;
        s_p[h][i] = pv;
      }
      l = wave_fsum(l);
      wsum = wave_fsum(wsum);
      if (lane == 0u) {
        s_w[h] = wsum;
        float* o = a.ws + (((int64_t)b * a.Hq + hk * a.nq + h) * a.nsplit + split) * (a.D + 2);
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
  {
    const uint8_t* vb = a.v + (int64_t)b * a.v_sb + (int64_t)hk * a.v_sh + (int64_t)t0 * a.v_st +
                        (int64_t)ld * Raw16<VBITS>::kBytes;
    for (uint32_t base = 0; base < nt; base += TL * kAttnUnroll) {
      Raw16<VBITS> raw[kAttnUnroll];
#pragma unroll
      for (int u = 0; u < kAttnUnroll; ++u) {
        const uint32_t i = base + u * TL + tl;
        if (i < nt) raw[u].load(vb + (int64_t)i * a.v_st);
        else raw[u].zero();
      }
#pragma unroll
      for (int u = 0; u < kAttnUnroll; ++u) {
        const uint32_t i = base + u * TL + tl;
        if (i < nt) {
          float uf[16];
          raw[u].to_f32_biased(uf);
#pragma unroll
          for (int h = 0; h < NQ; ++h) {
            const float p = s_p[h][i];
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[h][j] = fmaf(uf[j], p, acc[h][j]);
          }
        }
      }
    }
  }
  // token lanes of one wave -> lane tl == 0 of the wave, then the 4 waves through LDS
#pragma unroll
  for (int h = 0; h < NQ; ++h)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float v = acc[h][j];
      for (uint32_t s = lpt; s < (uint32_t)kWave; s <<= 1) v += __shfl_xor(v, (int)s);
      acc[h][j] = v;
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
    a.ws[(((int64_t)b * a.Hq + hk * a.nq + h) * a.nsplit + split) * (a.D + 2) + 2 + d] = v;
  }
}

// One workgroup per (query head, batch row): merge the splits and the exact new token.
__global__ __launch_bounds__(kAttnBlock) void decode_attn_merge_k(const AttnArgs a) {
  __shared__ float s_red[kAttnBlock / kWave];
  __shared__ float s_new;
  const uint32_t tid = threadIdx.x;
  const uint32_t hq = blockIdx.x, b = blockIdx.y;
  const uint32_t hk = hq / a.nq;
  const bool has_new = a.kn != nullptr;
  float s_tok = -INFINITY;
  if (has_new) {
    float part = 0.0f;
    for (uint32_t d = tid; d < a.D; d += kAttnBlock)
      part += load_elem(a.q, (int64_t)b * a.q_sb + (int64_t)hq * a.q_sh + d, a.dtype) *
              load_elem(a.kn, (int64_t)b * a.kn_sb + (int64_t)hk * a.kn_sh + d, a.dtype);
    part = wave_fsum(part);
    if ((tid & 63u) == 0u) s_red[tid >> 6] = part;
    __syncthreads();
    if (tid == 0u) s_new = (s_red[0] + s_red[1] + s_red[2] + s_red[3]) * a.sm_scale;
    __syncthreads();
    s_tok = s_new;
  }
  const float* base = a.ws + ((int64_t)b * a.Hq + hq) * a.nsplit * (a.D + 2);
  float M = s_tok;
  for (uint32_t s = 0; s < a.nsplit; ++s) M = fmaxf(M, base[(int64_t)s * (a.D + 2)]);
  float L = has_new ? __expf(s_tok - M) : 0.0f;
  for (uint32_t s = 0; s < a.nsplit; ++s) {
    const float* p = base + (int64_t)s * (a.D + 2);
    L += p[1] * __expf(p[0] - M);
  }
  const float inv = 1.0f / L;
  for (uint32_t d = tid; d < a.D; d += kAttnBlock) {
    float o = has_new ? __expf(s_tok - M) * load_elem(a.vn, (int64_t)b * a.vn_sb + (int64_t)hk * a.vn_sh + d, a.dtype)
                      : 0.0f;
    for (uint32_t s = 0; s < a.nsplit; ++s) {
      const float* p = base + (int64_t)s * (a.D + 2);
      o = fmaf(__expf(p[0] - M), p[2 + d], o);
    }
    o *= inv;
    const int64_t oi = (int64_t)b * a.o_sb + (int64_t)hq * a.o_sh + d;
    if (a.dtype == KVQ_F16) reinterpret_cast<f16*>(a.out)[oi] = (f16)o;
    else reinterpret_cast<__bf16*>(a.out)[oi] = (__bf16)o;
  }
}

static void plan(const kvq_attn_dims_t* d, uint32_t* ts, uint32_t* nsplit) {
  // >= 2 workgroups per CU when the context allows it; TS a multiple of 128 (the widest token-lane
  // stride) so that every split but the last is full
  const int64_t want = 512;
  const int64_t bh = d->B * d->Hkv > 0 ? d->B * d->Hkv : 1;
  int64_t per = (d->T * bh + want - 1) / want;
  per = (per + 127) / 128 * 128;
  if (per < 128) per = 128;
  if (per > kAttnMaxTS) per = kAttnMaxTS;
  *ts = (uint32_t)per;
  *nsplit = (uint32_t)((d->T + per - 1) / per);
}

template <int KBITS, int VBITS>
static void launch_partial(const AttnArgs& a, hipStream_t st) {
  const dim3 grid(a.nsplit, a.Hkv, a.B);
  if (a.nq == 1) hipLaunchKernelGGL((decode_attn_partial_k<KBITS, VBITS, 1>), grid, dim3(kAttnBlock), 0, st, a);
  else if (a.nq == 2) hipLaunchKernelGGL((decode_attn_partial_k<KBITS, VBITS, 2>), grid, dim3(kAttnBlock), 0, st, a);
  else if (a.nq <= 4) hipLaunchKernelGGL((decode_attn_partial_k<KBITS, VBITS, 4>), grid, dim3(kAttnBlock), 0, st, a);
  else hipLaunchKernelGGL((decode_attn_partial_k<KBITS, VBITS, 8>), grid, dim3(kAttnBlock), 0, st, a);
}

}  // namespace kvq

using namespace kvq;

extern "C" {

int64_t kvq_decode_attn_workspace(const kvq_attn_dims_t* d) {
  if (!d || d->B <= 0 || d->Hq <= 0 || d->Hkv <= 0 || d->T < 0 || d->D <= 0) return -1;
  uint32_t ts, ns;
  plan(d, &ts, &ns);
  return d->B * d->Hq * (int64_t)(ns > 0 ? ns : 1) * (d->D + 2);
}

int kvq_decode_attn(const void* q, int64_t q_sb, int64_t q_sh, const uint8_t* k_store, const kvq_strides_t* k_st,
                    const float* k_scales, int k_bits, const uint8_t* v_store, const kvq_strides_t* v_st,
                    const float* v_scales, int v_bits, const void* k_new, int64_t kn_sb, int64_t kn_sh,
                    const void* v_new, int64_t vn_sb, int64_t vn_sh, void* out, int64_t o_sb, int64_t o_sh, int dtype,
                    float sm_scale, float* workspace, int64_t workspace_floats, const kvq_attn_dims_t* d,
                    void* stream) {
  const char* name = "kvq_decode_attn";
  if (!d || !q || !out) {
    set_error("%s: NULL q / out / dims", name);
    return KVQ_E_NULL;
  }
  if (d->B <= 0 || d->Hq <= 0 || d->Hkv <= 0 || d->T < 0 || d->B >= (1 << 16) || d->Hkv >= (1 << 16) ||
      d->T >= (int64_t(1) << 31) || d->Hq % d->Hkv != 0 || d->Hq / d->Hkv > 8) {
    set_error("%s: bad dims B=%lld Hq=%lld Hkv=%lld T=%lld (need Hq %% Hkv == 0, Hq / Hkv <= 8)", name, (long long)d->B,
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
  a.ws = workspace;
  a.sm_scale = sm_scale;
  a.B = (uint32_t)d->B; a.Hq = (uint32_t)d->Hq; a.Hkv = (uint32_t)d->Hkv; a.T = (uint32_t)d->T; a.D = (uint32_t)d->D;
  a.nq = (uint32_t)(d->Hq / d->Hkv);
  a.lpt_shift = ilog2_exact(d->D / 16);
  a.dtype = dtype;
  plan(d, &a.TS, &a.nsplit);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a.nsplit > 0) {
    if (k_bits == 8 && v_bits == 8) launch_partial<8, 8>(a, st);
    else if (k_bits == 8) launch_partial<8, 4>(a, st);
    else if (v_bits == 8) launch_partial<4, 8>(a, st);
    else launch_partial<4, 4>(a, st);
    const int rc = check_launch(name);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(decode_attn_merge_k, dim3(a.Hq, a.B), dim3(kAttnBlock), 0, st, a);
  return check_launch(name);
}

}  // extern "C"
