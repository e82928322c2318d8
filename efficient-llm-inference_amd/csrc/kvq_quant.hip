// kvq_quant.hip — per-token symmetric INT8 / packed-INT4 quantise kernels for gfx950.
//
// Replaces quantize_int8_per_tensor / quantize_int4_per_tensor_packed
// (src/quantization/ops.py:10-65) as they are applied slice by slice ([B,H,1,D]) by
// QuantizedLayerKV.append (ops.py:174-210): ONE scale per (group, token) over all B*H*D values.
//
//   s32    = max( max|x| / QMAX, eps )            fp32, IEEE division      (ops.py:28, :49)
//   q      = clamp( rint(x32 / s32), QMIN, QMAX )  half-to-even, true division (ops.py:29, :50)
//   stored = RN_in_dtype(s32), kept widened to fp32 in the scale table     (ops.py:30, :65)
//   INT4: nibble = q + 8, even d -> high nibble, odd D padded with nibble 8 (ops.py:52-63)
//
// Roofline: HBM stream, single pass: 2 B in + 1 B (INT8) or 0.5 B (INT4) out per fp16 element.
// Fused kernel: a workgroup owns a tile of TT tokens x all B*H rows of one group. Rows are T*D
// apart in memory, so each row contributes one contiguous TT*D run (coalesced 16 B/lane
// loads). The tile stays PACKED in registers between the abs-max pass and the quantise pass.
//   abs-max   packed u16 max of sign-masked halves -> DPP over the D/8 lanes of a (row, token) ->
//             register max across rows (one-wave tiles, REGMAX) or LDS ds_max_u32 (larger tiles)
//   quotient  x * RN(1/s32) with a distance-to-half-integer guard and an IEEE-divide fallback:
//             bit-exact rint(x / s32) at a third of the cost of dividing every element
//   stores    staged through LDS in output order, 16 B/lane, non-temporal
// Shipped launch shape: ONE WAVE per workgroup (64 threads, 4096-element tile): no barrier costs
// anything and waves of different tiles run decoupled. Larger B*H*D -> quant_tokens_sweep_k;
// odd / unaligned shapes -> the generic two-pass pair.
#include "kvq_common.h"

namespace kvq {

constexpr int kNVMax = 8;                         // 16-byte (8-element) vectors per thread per tile
constexpr int kTileElems = kBlock * kNVMax * 8;   // 16384 elements
constexpr int kMaxTT = 64;

struct QuantArgs {
  PtrTable in;  // per-group input base pointers (this launch's groups)
  Strides is;   // input strides in elements (g unused)
  uint8_t* q;   // store base for this launch's first group
  Strides qs;   // store strides in bytes
  float* scales;
  int64_t ssg;
  float* absmax_ws;  // [G, T] floats (two-pass path)
  float eps;
  uint32_t G, B, H, T, D;
  // fused path
  uint32_t R;        // B*H
  uint32_t TT;       // tokens per tile (power of two)
  int32_t dvshift;   // log2(D/8)
  int32_t vshift;    // log2(TT * D/8)
  uint32_t nvec;     // R * TT * D/8 vectors per full tile
  uint32_t rpc;      // sweep kernel: rows per sweep step
  uint32_t dv;       // sweep kernel, D/8 not a power of two (or > 64): D/8
  uint32_t vpr;      // sweep kernel, same case: TT * D/8 vectors per row run
  uint32_t t_begin;  // first token of this launch's first tile
  int32_t nt_loads;  // non-temporal input loads
  int32_t nt_stores; // non-temporal output stores (tunable quant_nt_stores; -1 = as nt_loads)
  int32_t blk;       // workgroup size of the fused kernel (256, or 64 = one wave per tile)
  int32_t nv;        // vectors per lane per tile (8, or 4 for the small one-wave tile)
  int32_t bh_contig; // rows addressable as r * stride_h on both sides
  uint32_t xcd_group;  // consecutive tiles per XCD for the fused one-tile kernel (0 / 1 = round robin)
  int32_t acc;         // abs-max phase: keep what the table already holds (max with it) instead of overwriting
};

// ---------------------------------------------------------------------------- fused single pass

// DPP lane exchange inside a 16-lane row (single VALU op, no LDS traffic)
template <int CTRL>
__device__ inline uint32_t dpp_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
// unsigned max over aligned groups of 2^LOGW consecutive lanes; every lane gets the group max.
// Order-preserving for the bit patterns of non-negative floats / halves.
__device__ inline uint32_t group_umax(uint32_t v, int logw) {
  if (logw > 0) v = max(v, dpp_u32<0xB1>(v));   // quad_perm(1,0,3,2): lane ^ 1
  if (logw > 1) v = max(v, dpp_u32<0x4E>(v));   // quad_perm(2,3,0,1): lane ^ 2
  if (logw > 2) v = max(v, dpp_u32<0x141>(v));  // row_half_mirror: quads of an 8-lane half
  if (logw > 3) v = max(v, dpp_u32<0x140>(v));  // row_mirror: halves of a 16-lane row
  if (logw > 4) v = max(v, (uint32_t)__shfl_xor((int)v, 16));
  if (logw > 5) v = max(v, (uint32_t)__shfl_xor((int)v, 32));
  return v;
}

// rint(x / s32) + BIAS for 8 elements, returned in the low mantissa bits of
// (1.5 * 2^23 + BIAS + q): bit-exact with IEEE division at a fraction of its cost.
// Fast path: p = x * r with r = RN(1/s32). |p - x/s32| <= 2^-23 |p| (at most two roundings) and
// the correctly rounded quotient RN(x/s32) is within 2^-24 |p| of x/s32, with |p| <= QMAX (1 + 2^-22)
// because |x| <= amax and s32 >= RN(amax/QMAX). So if p is farther than
// m = 1.5 * 2^-22 * QMAX (twice the error bound) from every half-integer, then p, x/s32 and
// RN(x/s32) all round to the same integer; "within m of a half-integer" is |d| >= 0.5 - m with
// d = p - rint(p). Otherwise (INT8: ~1 element in 11,000; INT4: ~1 in
// 200,000) the 8-element vector is redone with the IEEE divide.
// Rounding: adding 1.5 * 2^23 (even) rounds p to the nearest integer, ties to even — the same
// integer as rint(p) — and leaves it in two's complement in the low mantissa bits.
// No clamp is needed for finite inputs: the quotient is bounded by QMAX (1 + 2^-23), which
// rounds to QMAX (the reference's clamp never fires either).
// KVQ_QUANT_CALIB (calibration builds only, `make calib`; results are NOT exact): 1 = fast path
// without the guard / exact redo, 2 = no arithmetic at all (raw input bits are packed) — they bound
// what the guard and the whole quotient cost on top of the tile's memory traffic. 3 = 2 without the
// stores, 4 = 2 without the loads (f16 / bf16 inputs only): the tile pattern's read-only and write-only rates.
// 5 = 2 without the per-token scale store, 6 = 5 without the abs-max result and the two divides (one-wave REGMAX tiles).
#ifndef KVQ_QUANT_CALIB
#define KVQ_QUANT_CALIB 0
#endif
template <int BITS, class V>
__device__ inline void quotient_bits8(const V& v, float s32, float r, uint32_t (&qb)[8]) {
#if KVQ_QUANT_CALIB >= 2
#pragma unroll
  for (int j = 0; j < 8; ++j) qb[j] = __float_as_uint(v.get(j)) >> 16;
  return;
#endif
  constexpr float kBias = BITS == 8 ? 0.0f : 8.0f;
  constexpr float kMagic = 12582912.0f + kBias;  // 1.5 * 2^23 + BIAS
  constexpr float kM = 1.5f * QRange<BITS>::qmax * 0x1p-22f;
  constexpr float kThr = 0.5f - kM;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 r2 = {r, r}, magic2 = {kMagic, kMagic};
  float worst = 0.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {  // two elements per v_pk_*_f32
    const f32x2 x2 = {v.get(2 * k), v.get(2 * k + 1)};
    // sum = RN(x*r + magic) = magic + rint(x*r): the fma rounds the EXACT product, which is
    // within 2^-24 |p| of x/s32 (only r is rounded), inside the same margin
    const f32x2 sum = __builtin_elementwise_fma(x2, r2, magic2);
#if KVQ_QUANT_CALIB == 0
    const f32x2 d = __builtin_elementwise_fma(x2, r2, magic2 - sum);  // x*r - rint(x*r), one rounding
    worst = fmaxf(worst, fmaxf(fabsf(d[0]), fabsf(d[1])));              // one v_max3_f32 with |.| modifiers
#endif
    qb[2 * k] = __float_as_uint(sum[0]);
    qb[2 * k + 1] = __float_as_uint(sum[1]);
  }
  if (__builtin_expect(worst >= kThr, 0)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) qb[j] = __float_as_uint(rintf(v.get(j) / s32) + kMagic);
  }
}

// low bytes of 4 words -> one word {b0, b1, b2, b3} (3 ops per 4 elements)
__device__ inline uint32_t gather_low_bytes(uint32_t q0, uint32_t q1, uint32_t q2, uint32_t q3) {
  const uint32_t lo = __builtin_amdgcn_perm(q1, q0, 0x0C0C0400u);  // {q0.b0, q1.b0, 0, 0}
  const uint32_t hi = __builtin_amdgcn_perm(q3, q2, 0x04000C0Cu);  // {0, 0, q2.b0, q3.b0}
  return lo | hi;
}
// INT8: the two's complement low byte of the magic sum is q
__device__ inline u32x2 pack_i8(const uint32_t (&qb)[8]) {
  u32x2 w;
  w[0] = gather_low_bytes(qb[0], qb[1], qb[2], qb[3]);
  w[1] = gather_low_bytes(qb[4], qb[5], qb[6], qb[7]);
  return w;
}
// INT4: low byte = q + 8 in [1, 15] (clean high nibble); even element -> HIGH nibble (ops.py:61-63)
__device__ inline uint32_t pack_i4(const uint32_t (&qb)[8]) {
  const uint32_t a = gather_low_bytes(qb[0], qb[1], qb[2], qb[3]);  // bytes n0..n3
  const uint32_t b = gather_low_bytes(qb[4], qb[5], qb[6], qb[7]);  // bytes n4..n7
  const uint32_t ta = (a << 4) | (a >> 8);  // byte0 = n0<<4|n1, byte2 = n2<<4|n3
  const uint32_t tb = (b << 4) | (b >> 8);
  return __builtin_amdgcn_perm(tb, ta, 0x06040200u);  // {ta.b0, ta.b2, tb.b0, tb.b2}
}

// ROWU: the tile has >= 256 vectors per row, so in round i every lane of the block works on the
// same row (row index and its offsets are scalar) and (t, d) offsets collapse to wv * 8.
// FULL: every tile of the launch is complete (kBlock * kNVMax vectors, TT tokens inside T), so no
// lane is ever predicated off: no validity masks, no exec juggling.
// BLK: workgroup size. 64 = one wave per workgroup: the three barriers cost nothing and waves of
// different tiles run fully decoupled (tile = 4096 elements).
// REGMAX (one-wave tiles whose row run is at most 64 vectors): a lane keeps the same
// (token, d-vector) in every round and only the row changes, so the abs-max across rows is a
// register max (plus one lane exchange per halving of the row run below 64 vectors), the scale is
// computed per lane, and the kernel needs no LDS, atomics or barriers for it.
// NV: 16-byte vectors per lane per tile (tile = BLK * NV * 8 elements).
// GEO128: the tile geometry of the Llama prefill shape as compile-time constants (8 rows of head_dim 128, 4 tokens per
// one-wave tile: 64 vectors per row run, 16 lanes per token) instead of kernel arguments — the same code with its
// shifts, masks and lane-exchange loops folded.
// NTL / NTS: non-temporal input loads / output stores, COMPILE-TIME (a run-time `if (flag) nt_load else load` is merged
// by the compiler into one plain load: the kernel ran without non-temporal accesses until the end of round 2).
template <int IDT, int BITS, bool ROWU, bool LDS_OUT, bool FULL, int BLK = kBlock, int REGMAX = 0, int NV = kNVMax, bool GEO128 = false,
          bool NTL = true, bool NTS = true>
__global__ __launch_bounds__(BLK) void quant_tokens_fused_k(const QuantArgs a) {
  static_assert(!GEO128 || (BLK == 64 && NV == 8 && REGMAX == 1 && ROWU && FULL), "GEO128: the one-wave register-max tile");
  const uint32_t g_vshift = GEO128 ? 6u : a.vshift, g_dvshift = GEO128 ? 4u : a.dvshift, g_TT = GEO128 ? 4u : a.TT;
  const uint32_t g_R = GEO128 ? 8u : a.R, g_D = GEO128 ? 128u : a.D, g_nvec = GEO128 ? 512u : a.nvec;
  __shared__ __attribute__((aligned(16))) uint32_t s_out[LDS_OUT ? BLK * NV * 8 * BITS / 32 : 4];
  __shared__ uint32_t s_amax[kMaxTT];
  __shared__ float s_scale[kMaxTT], s_rcp[kMaxTT];
  const uint32_t tid = threadIdx.x;
  const uint32_t g = blockIdx.y;
  const uint32_t t0 = a.t_begin + (GEO128 ? blockIdx.x : xcd_grouped_item(blockIdx.x, a.xcd_group, gridDim.x)) * g_TT;
  const uint32_t DV = g_D >> 3;
  const uint32_t wmask = (1u << g_vshift) - 1u;
  // is.t == D and qs.t == Dq (or a single token per tile), so within a row the tile is one
  // contiguous run: element offset = wv * 8
  const char* in = reinterpret_cast<const char*>(a.in.p[g]) + (int64_t)t0 * (GEO128 ? 128 : a.is.t) * Elem<IDT>::size;
  uint8_t* qbase = a.q + (int64_t)g * a.qs.g + (int64_t)t0 * (GEO128 ? (int64_t)(128 * BITS / 8) : a.qs.t);
  constexpr int QV = BITS;  // bytes stored per 8-element vector: 8 (INT8) or 4 (INT4)

  if constexpr (REGMAX == 0) {
    for (uint32_t i = tid; i < (uint32_t)kMaxTT; i += BLK) s_amax[i] = 0u;
    __syncthreads();
  }

  // pass 1: load the tile (stays in registers), per-(row, token) abs-max -> LDS max across rows
  Vec8<IDT> x[NV];
  bool valid[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    uint32_t r, wv;
    if constexpr (ROWU) {
      r = (uint32_t)(i * BLK) >> g_vshift;
      wv = ((uint32_t)(i * BLK) & wmask) + tid;
    } else {
      const uint32_t v = i * BLK + tid;
      r = v >> g_vshift;
      wv = v & wmask;
    }
    valid[i] = FULL || (((uint32_t)(i * BLK) + tid < g_nvec) && (t0 + (wv >> g_dvshift) < a.T));
    if (valid[i]) {
      const char* src = in + ((int64_t)r * a.is.h + (int64_t)wv * 8) * Elem<IDT>::size;
#if KVQ_QUANT_CALIB == 4  // write-only pattern: no loads (the branch on eps is never taken)
      if (a.eps < 0.0f) x[i].load(src);
      else if constexpr (IDT != KVQ_F32) x[i].w = u32x4{tid, (uint32_t)i, r, wv};
#else
      if constexpr (NTL) x[i].load_nt(src);
      else x[i].load(src);
#endif
    }
  }
  // REGMAX = number of lane identities: 1 = the row run is one round wide (or narrower), 2 = two
  // rounds wide (rounds alternate between the run's two halves)
  constexpr int RM = REGMAX > 0 ? REGMAX : 1;
  float reg_s32[RM], reg_rcp[RM];
  if constexpr (REGMAX > 0) {
    uint32_t m[RM];
#pragma unroll
    for (int p = 0; p < RM; ++p) m[p] = 0u;
#pragma unroll
    for (int i = 0; i < NV; ++i) m[i % RM] = max(m[i % RM], x[i].absmax_bits());  // across the tile's rows
#pragma unroll
    for (int p = 0; p < RM; ++p) {
      uint32_t mp = group_umax(m[p], g_dvshift);  // across the D/8 lanes of the token
      if constexpr (RM == 1)
        for (int sh = g_vshift; sh < 6; ++sh) mp = max(mp, (uint32_t)__shfl_xor((int)mp, 1 << sh));  // row run < one wave
#if KVQ_QUANT_CALIB == 6  // calibration: no abs-max reduction result, no divides
      reg_s32[p] = 1.0f + (float)(mp & 1u);
      reg_rcp[p] = reg_s32[p];
#else
      reg_s32[p] = fmaxf(Vec8<IDT>::bits_to_f32(mp) / QRange<BITS>::qmax, a.eps);
      reg_rcp[p] = 1.0f / reg_s32[p];
#endif
      const uint32_t wv = (uint32_t)(p * BLK) + tid;  // this lane's vector in the row run, rounds i % RM == p
#if KVQ_QUANT_CALIB == 5 || KVQ_QUANT_CALIB == 6  // calibration: the per-token scale store never executes
      if ((tid & (DV - 1u)) == 0u && (RM > 1 || (tid >> g_vshift) == 0u) && a.eps < 0.0f)
#else
      if ((tid & (DV - 1u)) == 0u && (RM > 1 || (tid >> g_vshift) == 0u))
#endif
        a.scales[(int64_t)g * a.ssg + t0 + ((wv & wmask) >> g_dvshift)] = Elem<IDT>::round_trip(reg_s32[p]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (FULL || (uint32_t)(i * BLK) < g_nvec) {  // uniform: whole waves reach the lane exchanges
        const uint32_t wv = ((uint32_t)(i * BLK) + tid) & wmask;
        uint32_t m = valid[i] ? x[i].absmax_bits() : 0u;
        m = group_umax(m, g_dvshift);  // the D/8 lanes of one (row, token)
        if (valid[i] && (wv & (DV - 1u)) == 0u) atomicMax(&s_amax[wv >> g_dvshift], m);
      }
    }
    __syncthreads();

    // per-token scale, its reciprocal (both IEEE divides, once per token) and the stored scale
    if (tid < g_TT && (FULL || t0 + tid < a.T)) {
      const float amax = Vec8<IDT>::bits_to_f32(s_amax[tid]);
      const float s32 = fmaxf(amax / QRange<BITS>::qmax, a.eps);
      s_scale[tid] = s32;
      s_rcp[tid] = 1.0f / s32;
      a.scales[(int64_t)g * a.ssg + t0 + tid] = Elem<IDT>::round_trip(s32);
    }
    __syncthreads();
  }

  // pass 2: scale, round, pack, store
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (!FULL && !valid[i]) continue;
    uint32_t r, wv;
    if constexpr (ROWU) {
      r = (uint32_t)(i * BLK) >> g_vshift;
      wv = ((uint32_t)(i * BLK) & wmask) + tid;
    } else {
      const uint32_t v = i * BLK + tid;
      r = v >> g_vshift;
      wv = v & wmask;
    }
    const uint32_t tl = wv >> g_dvshift;
    uint32_t qb[8];
    if constexpr (REGMAX > 0) quotient_bits8<BITS>(x[i], reg_s32[i % RM], reg_rcp[i % RM], qb);
    else quotient_bits8<BITS>(x[i], s_scale[tl], s_rcp[tl], qb);
    if constexpr (LDS_OUT) {
      // stage the packed bytes in LDS in output order (row-major, wv * QV within the row run)
      const uint32_t widx = ((r << g_vshift) + wv) * (QV / 4);
      if constexpr (BITS == 8) {
        const u32x2 w = pack_i8(qb);
        s_out[widx] = w[0];
        s_out[widx + 1] = w[1];
      } else {
        s_out[widx] = pack_i4(qb);
      }
    } else {
      uint8_t* qp = qbase + (int64_t)r * a.qs.h + (int64_t)wv * QV;
      if constexpr (BITS == 8) *reinterpret_cast<u32x2*>(qp) = pack_i8(qb);
      else *reinterpret_cast<uint32_t*>(qp) = pack_i4(qb);
    }
  }

  if constexpr (LDS_OUT) {
    // 16 B per lane, 1 KiB contiguous per wave store: each row of the tile is one run of
    // TT * Dq bytes in the store (ablation: the direct 4 B/lane stores cost ~100 us of 354)
    __syncthreads();
    const uint32_t row_shift = g_vshift + (BITS == 8 ? 3 : 2);  // log2(bytes per full row run)
    const uint32_t row_bytes = 1u << row_shift;
    uint32_t nt = FULL ? g_TT : a.T - t0;
    if (nt > g_TT) nt = g_TT;
    const uint32_t valid_bytes = nt * (g_D * BITS / 8);  // ragged last tile: shorter runs
    const uint32_t total = g_R << row_shift;
    for (uint32_t k = tid * 16u; k < total; k += BLK * 16u) {
      const uint32_t r = k >> row_shift;
      const uint32_t off = k & (row_bytes - 1u);
#if KVQ_QUANT_CALIB == 3  // read-only pattern: the stores stay in the code but never execute
      if (off < valid_bytes && a.eps < 0.0f) {
#else
      if (off < valid_bytes) {
#endif
        const u32x4 w = *reinterpret_cast<const u32x4*>(&s_out[k >> 2]);
        if constexpr (NTS) __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(qbase + (int64_t)r * a.qs.h + off));
        else *reinterpret_cast<u32x4*>(qbase + (int64_t)r * a.qs.h + off) = w;
      }
    }
  }
}

// ---------------------------------------------------------------------------- compile-time tile (what ships for the BASELINE shapes)
// The shapes the benchmarks are quoted on — R = B*H head rows of head_dim D with f16 / bf16 input: 8 x 128 (Llama-3-8B),
// 16 x 64 (gpt2-medium), 12 x 64 (gpt2), 8 x 64 — as ONE WAVE per tile of TT tokens with the whole geometry in the
// template: lane = (row-in-instruction, token, 8-element d-vector); every load instruction of the wave reads RPI = 64 /
// (TT * D/8) rows' contiguous TT*D-element runs (1 KiB, or 2 x 512 B); the abs-max across the rows is a register max
// (+ one lane exchange when two rows share an instruction), across a token's D/8 lanes DPP; the scale is per lane.
// Addressing is by BUFFER instructions: a per-wave descriptor of the group's rows (scalar), a lane offset that never
// changes, the row as the instruction's scalar offset — no 64-bit vector address arithmetic, no pointer per row. A
// ragged last tile costs nothing: lanes of tokens past T carry an out-of-range offset (loads return 0, which cannot
// raise an abs-max; stores are dropped), so ONE instantiation serves complete and ragged tiles.
// Same arithmetic, same order of operations per element as quant_tokens_fused_k (quotient_bits8 / pack_*): bit-identical.
// cache-policy bits of the tile kernel's row loads / output stores (buffer aux immediate: 1 = sc0, 2 = nt, 16 = sc1).
// What ships is nt for both; `make calib_aux` builds the other combinations for A-B runs (profiles/r03f_*).
#ifndef KVQ_TILE_LD_AUX
#define KVQ_TILE_LD_AUX 2
#endif
#ifndef KVQ_TILE_ST_AUX
#define KVQ_TILE_ST_AUX 2
#endif
#ifndef KVQ_TILE_DIRECT
#define KVQ_TILE_DIRECT 0
#endif
#ifndef KVQ_TILE_PRIO  // calibration (`make calib_prio`): 1 = raise the wave's priority once its loads are issued, 2 = high while issuing the loads, low after
#define KVQ_TILE_PRIO 0
#endif
struct QuantTileArgs {
  PtrTable in;      // per-group input base pointers
  uint8_t* q;       // store base of this launch's first group
  float* scales;
  float* absmax;    // split phases: [G, T] fp32 table of this launch's first group
  int64_t qs_g;     // bytes
  int64_t ssg;
  uint32_t is_h;    // input row stride in BYTES
  uint32_t qs_h;    // store row stride in bytes
  uint32_t T;
  uint32_t rows;    // B*H rows of a group; blockIdx.z walks them R at a time (fused: rows == R, gridDim.z == 1)
  float eps;
};

// PHASE 0: the fused quantise (abs-max, scale, quantise: one pass over the tile).
// Split phases of a batch-sharded slice (the scale spans the WHOLE batch, ops.py:27,48, so the abs-max table
// crosses ranks between the two): the same tile walk over R-row groups of a group's B_local*H rows (blockIdx.z),
//   PHASE 1: abs-max of the R rows -> atomicMax into absmax[g, t] (the host zeroes the table first; non-negative floats
//            order like their bit patterns); plain loads, so the rows stay in the Infinity Cache for phase 2
//   PHASE 2: quantise the R rows with scale max(absmax[g, t] / QMAX, eps) (the table completed by all_reduce(MAX)),
//            non-temporal loads (last use); row group 0 stores the scales.
template <int IDT, int BITS, int R, int DV, int TT, int PHASE = 0>
__global__ __launch_bounds__(kWave) void quant_tile_k(const QuantTileArgs a) {
  static_assert(IDT != KVQ_F32, "two-byte inputs");
  constexpr int LPR = TT * DV;              // lanes per row run
  static_assert(LPR == 64 || LPR == 32, "a row run is one wave or half a wave wide");
  constexpr int RPI = 64 / LPR;             // rows per load instruction
  static_assert(R % RPI == 0, "rows pair up");
  constexpr int NV = R / RPI;               // 16-byte vectors per lane
  constexpr int D = DV * 8;
  constexpr int QV = BITS;                  // bytes stored per 8-element vector
  constexpr int ROWB = LPR * QV;            // bytes of one row's run in the store
  constexpr int OUTB = R * ROWB;            // bytes of the tile in the store
  constexpr int NST = (OUTB + 1023) / 1024; // 16 B per lane store instructions
  constexpr uint32_t kOut = 0x80000000u;    // offset no descriptor of ours covers
  __shared__ __attribute__((aligned(16))) uint32_t s_out[PHASE == 1 ? 4 : NST * 256];
  const uint32_t lane = threadIdx.x;
  const uint32_t g = blockIdx.y;
  const uint32_t t0 = blockIdx.x * TT;
  const uint32_t nt = a.T - t0 < (uint32_t)TT ? a.T - t0 : (uint32_t)TT;  // uniform
  const uint32_t r0 = PHASE == 0 ? 0u : blockIdx.z * (uint32_t)R;
  const uint32_t nr = PHASE == 0 ? (uint32_t)R : (a.rows - r0 < (uint32_t)R ? a.rows - r0 : (uint32_t)R);  // rows of this group (uniform)
  const uint32_t sub = RPI == 1 ? 0u : lane / LPR, wv = lane % LPR, tok = wv / DV;

  // rows past nr and tokens past nt are outside the descriptor: loads return 0 (no effect on an abs-max), stores are dropped
  const char* ibase = reinterpret_cast<const char*>(a.in.p[g]) + (int64_t)r0 * a.is_h + (int64_t)t0 * (D * 2);
  const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ibase), 0,
                                                                        (int)((nr - 1u) * a.is_h + nt * (D * 2)), 0x00020000);
  const uint32_t ioff = tok < nt ? sub * a.is_h + wv * 16u : kOut;
  Vec8<IDT> x[NV];
#if KVQ_TILE_PRIO == 2
  __builtin_amdgcn_s_setprio(3);
#endif
#pragma unroll
  for (int i = 0; i < NV; ++i)
    x[i].w = __builtin_amdgcn_raw_buffer_load_b128(irs, ioff, (uint32_t)(i * RPI) * a.is_h, PHASE == 1 ? 0 : KVQ_TILE_LD_AUX /* non-temporal */);
#if KVQ_TILE_PRIO == 1
  __builtin_amdgcn_s_setprio(3);
#elif KVQ_TILE_PRIO == 2
  __builtin_amdgcn_s_setprio(0);
#endif

  constexpr int DVSH = DV == 16 ? 4 : 3;
  float s32;
  if constexpr (PHASE == 2) {
    s32 = fmaxf((tok < nt ? a.absmax[(int64_t)g * a.T + t0 + tok] : 0.0f) / QRange<BITS>::qmax, a.eps);
  } else {
    uint32_t m = 0u;
#pragma unroll
    for (int i = 0; i < NV; ++i) m = max(m, x[i].absmax_bits());  // across the lane's rows
    if constexpr (RPI == 2) m = max(m, (uint32_t)__shfl_xor((int)m, 32));
    m = group_umax(m, DVSH);  // across the D/8 lanes of the token
    if constexpr (PHASE == 1) {
      if ((lane % DV) == 0u && sub == 0u && tok < nt)
        atomicMax(reinterpret_cast<uint32_t*>(a.absmax) + (int64_t)g * a.T + t0 + tok, __float_as_uint(Vec8<IDT>::bits_to_f32(m)));
      return;
    }
    s32 = fmaxf(Vec8<IDT>::bits_to_f32(m) / QRange<BITS>::qmax, a.eps);
  }
  const float rcp = 1.0f / s32;
  if ((lane % DV) == 0u && sub == 0u && tok < nt && r0 == 0u) a.scales[(int64_t)g * a.ssg + t0 + tok] = Elem<IDT>::round_trip(s32);

#if KVQ_TILE_DIRECT  // calibration build (`make calib_direct`): every row's run stored straight from registers, 8 / 4 B per lane
  if constexpr (RPI == 1) {
    uint8_t* ob = a.q + (int64_t)g * a.qs_g + (int64_t)r0 * a.qs_h + (int64_t)t0 * (D * BITS / 8);
    const __amdgpu_buffer_rsrc_t os = __builtin_amdgcn_make_buffer_rsrc(ob, 0, (int)((nr - 1u) * a.qs_h + nt * (D * BITS / 8)), 0x00020000);
    const uint32_t doff = tok < nt ? wv * (uint32_t)QV : kOut;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      uint32_t qb[8];
      quotient_bits8<BITS>(x[i], s32, rcp, qb);
      if constexpr (BITS == 8) __builtin_amdgcn_raw_buffer_store_b64(pack_i8(qb), os, doff, (uint32_t)i * a.qs_h, KVQ_TILE_ST_AUX);
      else __builtin_amdgcn_raw_buffer_store_b32(pack_i4(qb), os, doff, (uint32_t)i * a.qs_h, KVQ_TILE_ST_AUX);
    }
    return;
  }
#endif
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    uint32_t qb[8];
    quotient_bits8<BITS>(x[i], s32, rcp, qb);
    const uint32_t widx = (((uint32_t)(i * RPI) + sub) * LPR + wv) * (QV / 4);  // output order: row-major, wv * QV inside the run
    if constexpr (BITS == 8) {
      const u32x2 w = pack_i8(qb);
      *reinterpret_cast<u32x2*>(&s_out[widx]) = w;
    } else {
      s_out[widx] = pack_i4(qb);
    }
  }
  // one wave: its LDS queue is in order; only the compiler must not move the reads above the writes
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  uint8_t* obase = a.q + (int64_t)g * a.qs_g + (int64_t)r0 * a.qs_h + (int64_t)t0 * (D * BITS / 8);
  const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(obase, 0, (int)((nr - 1u) * a.qs_h + nt * (D * BITS / 8)), 0x00020000);
  // 16 B piece k = 1024 j + 16 lane of the staged tile: row k / ROWB, byte k % ROWB of the row's run
  const uint32_t lrow = lane * 16u / ROWB, loff = lane * 16u % ROWB;
  const uint32_t ooff = loff < nt * (uint32_t)(D * BITS / 8) ? lrow * a.qs_h + loff : kOut;
#pragma unroll
  for (int j = 0; j < NST; ++j) {
    const u32x4 w = *reinterpret_cast<const u32x4*>(&s_out[j * 256 + lane * 4]);
    const uint32_t off = (OUTB % 1024 == 0 || (uint32_t)j * 1024u + lane * 16u < (uint32_t)OUTB) ? ooff : kOut;
    __builtin_amdgcn_raw_buffer_store_b128(w, ors, off, (uint32_t)(j * (1024 / ROWB)) * a.qs_h, KVQ_TILE_ST_AUX /* non-temporal */);
  }
}

// ---------------------------------------------------------------------------- wide tile (single pass, 16384 < B*H*D <= 131072)
// A batched slice too large for the one-wave tile — an un-sharded batch of 32 / 64 rows of Llama-3-8B: R = B*H = 256 / 512
// head rows of 128 — still read ONCE: a 1024-thread workgroup owns TT tokens x ALL R rows (TT * R * D = up to 131072
// elements = 256 KiB of fp16) and keeps them in REGISTERS (NV = 16 vectors of 8 elements per lane, 64 of the 128 VGPRs a
// lane of such a workgroup has), one workgroup per CU. Lane = (row-in-round, token, d-vector): in round i it holds row
// i * RPR + lane_row, so its (token, d-vector) never changes and the abs-max across rows is a register max, then DPP across
// the token's D/8 lanes, a lane exchange across the rows that share a wave, ONE LDS atomic per token per wave. Every load is
// issued before the first is used; the row of a round is the lane's first row plus a uniform stride per round (the host
// admits only layouts where that holds: B*H rows evenly spaced, or rows-per-round a multiple of H).
// Same arithmetic per element as every other quantise kernel (quotient_bits8 / pack_*): bit-identical.
// Against the alternatives for these shapes (Llama-3-8B batch 64 x 512 tokens, profiles/r03w_*): two passes of the one-wave
// tile kernels 3.6 TB/s of single-pass bytes, the swept tile 2.5.
struct QuantWideArgs {
  PtrTable in;
  uint8_t* q;
  float* scales;
  int64_t qs_g, ssg;
  int64_t is_b, is_h, qs_b, qs_h;  // BYTES
  int64_t is_round, qs_round;      // bytes between a lane's rows of consecutive rounds
  uint32_t H, R, T, D;
  uint32_t vshift, dvshift;        // log2(vectors per row run = TT * D/8), log2(D/8)
  float eps;
};

template <int IDT, int BITS, int BLK, int NV>
__global__ __launch_bounds__(BLK) void quant_wide_k(const QuantWideArgs a) {
  static_assert(IDT != KVQ_F32, "two-byte inputs (an fp32 tile of this size does not fit the register file)");
  __shared__ uint32_t s_amax[kMaxTT];
  __shared__ float s_scale[kMaxTT], s_rcp[kMaxTT];
  const uint32_t tid = threadIdx.x, g = blockIdx.y;
  const uint32_t TT = 1u << (a.vshift - a.dvshift), DV = 1u << a.dvshift;
  const uint32_t t0 = blockIdx.x * TT;
  const uint32_t wv = tid & ((1u << a.vshift) - 1u), lr = tid >> a.vshift, tl = wv >> a.dvshift;
  const uint32_t RPR = (uint32_t)BLK >> a.vshift;  // rows per round
  const bool tok_ok = t0 + tl < a.T;
  const uint32_t lb = lr / a.H, lh = lr - lb * a.H;
  const char* in = reinterpret_cast<const char*>(a.in.p[g]) + (int64_t)lb * a.is_b + (int64_t)lh * a.is_h + ((int64_t)t0 * a.D + (int64_t)wv * 8) * 2;
  uint8_t* qp = a.q + (int64_t)g * a.qs_g + (int64_t)lb * a.qs_b + (int64_t)lh * a.qs_h + (int64_t)t0 * (a.D * BITS / 8) + (int64_t)wv * BITS;
  if (tid < (uint32_t)kMaxTT) s_amax[tid] = 0u;
  __syncthreads();

  Vec8<IDT> x[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    x[i].w = u32x4{0u, 0u, 0u, 0u};  // rows past R, tokens past T: cannot raise an abs-max, never stored
    if (tok_ok && (uint32_t)i * RPR + lr < a.R) x[i].load_nt(in + (int64_t)i * a.is_round);
  }
  uint32_t m = 0u;
#pragma unroll
  for (int i = 0; i < NV; ++i) m = max(m, x[i].absmax_bits());
  m = group_umax(m, (int)a.dvshift);                                                            // the token's D/8 lanes
  for (uint32_t sh = a.vshift; sh < 6u; ++sh) m = max(m, (uint32_t)__shfl_xor((int)m, 1 << sh));  // rows sharing the wave
  if ((wv & (DV - 1u)) == 0u && ((tid & 63u) >> a.vshift) == 0u && tok_ok) atomicMax(&s_amax[tl], m);
  __syncthreads();
  if (tid < TT && t0 + tid < a.T) {
    const float s32 = fmaxf(Vec8<IDT>::bits_to_f32(s_amax[tid]) / QRange<BITS>::qmax, a.eps);
    s_scale[tid] = s32;
    s_rcp[tid] = 1.0f / s32;
    a.scales[(int64_t)g * a.ssg + t0 + tid] = Elem<IDT>::round_trip(s32);
  }
  __syncthreads();
  const float s32 = s_scale[tl], rcp = s_rcp[tl];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (tok_ok && (uint32_t)i * RPR + lr < a.R) {
      uint32_t qb[8];
      quotient_bits8<BITS>(x[i], s32, rcp, qb);
      uint8_t* dst = qp + (int64_t)i * a.qs_round;
      if constexpr (BITS == 8) __builtin_nontemporal_store(pack_i8(qb), reinterpret_cast<u32x2*>(dst));
      else __builtin_nontemporal_store(pack_i4(qb), reinterpret_cast<uint32_t*>(dst));
    }
  }
}

#if KVQ_AB
// ---------------------------------------------------------------------------- tile kernel, merged stores
// quant_tile_k's INT4 output leaves a wave as 256-byte pieces per head row (4 tokens x 64 B); the calibration builds of
// round 2 put that write phase at 3.2 TB/s where 512-byte pieces run at 6.2 (profiles/NOTES.md §3.2). Here ONE wave
// quantises TPW consecutive tiles (TPW x TT tokens) — every tile's R row loads issued up front, each tile reduced as
// its own loads arrive (counted vmcnt) — and stages ALL their output in LDS, so that a row's run leaves as one
// TPW x 256-byte (INT4) / TPW x 512-byte (INT8) piece. Same arithmetic per element as quant_tile_k; a row run is one
// wave wide (TT * D/8 == 64). Measured SLOWER than one tile per wave (INT4 242-245 us against 231, INT8 294-309 against
// 272: profiles/r03m_quant_tile_merged_stores.txt) — twice the loads per wave in flight and half the waves cost more than
// the larger pieces return: A-B builds only.
template <int IDT, int BITS, int R, int DV, int TT, int TPW>
__global__ __launch_bounds__(kWave) void quant_tile_merged_k(const QuantTileArgs a) {
  static_assert(IDT != KVQ_F32 && TT * DV == 64, "two-byte inputs, one row run per load instruction");
  constexpr int D = DV * 8;
  constexpr int QV = BITS;
  constexpr int ROWB1 = 64 * QV;            // one tile's bytes of a row run in the store
  constexpr int ROWB = TPW * ROWB1;         // the wave's bytes of a row run
  constexpr int OUTB = R * ROWB;
  constexpr int NST = OUTB / 1024;
  static_assert(OUTB % 1024 == 0 && (1024 % ROWB == 0 || ROWB % 1024 == 0), "whole 1 KiB store instructions");
  constexpr uint32_t kOut = 0x80000000u;
  __shared__ __attribute__((aligned(16))) uint32_t s_out[OUTB / 4];
  const uint32_t lane = threadIdx.x;
  const uint32_t g = blockIdx.y;
  const uint32_t t0 = blockIdx.x * (uint32_t)(TT * TPW);
  const uint32_t nt = a.T - t0 < (uint32_t)(TT * TPW) ? a.T - t0 : (uint32_t)(TT * TPW);  // uniform: tokens of this wave
  const uint32_t tok = lane / DV;
  const char* ibase = reinterpret_cast<const char*>(a.in.p[g]) + (int64_t)t0 * (D * 2);
  const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ibase), 0, (int)((R - 1) * a.is_h + nt * (D * 2)), 0x00020000);
  Vec8<IDT> x[TPW][R];
#pragma unroll
  for (int p = 0; p < TPW; ++p) {
    const uint32_t ioff = (uint32_t)(p * TT) + tok < nt ? (uint32_t)p * 1024u + lane * 16u : kOut;
#pragma unroll
    for (int i = 0; i < R; ++i) x[p][i].w = __builtin_amdgcn_raw_buffer_load_b128(irs, ioff, (uint32_t)i * a.is_h, KVQ_TILE_LD_AUX);
  }
  constexpr int DVSH = DV == 16 ? 4 : 3;
#pragma unroll
  for (int p = 0; p < TPW; ++p) {
    uint32_t m = 0u;
#pragma unroll
    for (int i = 0; i < R; ++i) m = max(m, x[p][i].absmax_bits());
    m = group_umax(m, DVSH);
    const float s32 = fmaxf(Vec8<IDT>::bits_to_f32(m) / QRange<BITS>::qmax, a.eps);
    const float rcp = 1.0f / s32;
    if ((lane % DV) == 0u && (uint32_t)(p * TT) + tok < nt) a.scales[(int64_t)g * a.ssg + t0 + (uint32_t)(p * TT) + tok] = Elem<IDT>::round_trip(s32);
#pragma unroll
    for (int i = 0; i < R; ++i) {
      uint32_t qb[8];
      quotient_bits8<BITS>(x[p][i], s32, rcp, qb);
      const uint32_t widx = ((uint32_t)i * (TPW * 64) + (uint32_t)p * 64u + lane) * (QV / 4);  // [row][tile][vector]
      if constexpr (BITS == 8) *reinterpret_cast<u32x2*>(&s_out[widx]) = pack_i8(qb);
      else s_out[widx] = pack_i4(qb);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  uint8_t* obase = a.q + (int64_t)g * a.qs_g + (int64_t)t0 * (D * BITS / 8);
  const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(obase, 0, (int)((R - 1) * a.qs_h + nt * (D * BITS / 8)), 0x00020000);
#pragma unroll
  for (int j = 0; j < NST; ++j) {
    const uint32_t k = (uint32_t)j * 1024u + lane * 16u;  // byte of the staged tile: row k / ROWB, byte k % ROWB of the row's run
    const uint32_t row = k / ROWB, off = k % ROWB;
    const u32x4 w = *reinterpret_cast<const u32x4*>(&s_out[k / 4]);
    __builtin_amdgcn_raw_buffer_store_b128(w, ors, off < nt * (uint32_t)(D * BITS / 8) ? row * a.qs_h + off : kOut, 0, KVQ_TILE_ST_AUX);
  }
}

#endif  // KVQ_AB (merged stores)

#if KVQ_AB
// ---------------------------------------------------------------------------- pipelined one-wave tiles
// The shipped prefill shape (R rows x 64 vectors per tile, one wave per tile: REGMAX above) as a
// software pipeline: ONE wave walks TPW consecutive tiles and requests tile n + 1's R non-temporal
// loads BEFORE it quantises and stores tile n, so that inside a wave reads are in flight while the
// writes of the previous tile drain (the load-all -> store-all schedule of a single-tile wave alternates
// the two), and the per-wave start-up (kernel-argument loads, address set-up) is paid once per TPW
// tiles. Same arithmetic, same bytes, same order of operations per element as quant_tokens_fused_k.
//   lane = vector `lane` of every row's 64-vector run: token lane >> dvshift, 8 elements at d = 8 (lane & (DV-1))
//   abs-max across the R rows in registers, across the D/8 lanes of a token by DPP; scale per lane
//   packed bytes staged in LDS in output order, stored 16 B per lane (R * 64 * BITS / 128 lanes active)
template <int IDT, int BITS, int R, int TPW>
__global__ __launch_bounds__(kWave) void quant_tokens_pipe_k(const QuantArgs a) {
  static_assert(IDT != KVQ_F32, "two-byte inputs only (the fp32 tile does not fit twice)");
  constexpr int QV = BITS;                       // bytes stored per 8-element vector
  constexpr uint32_t kRowBytes = 64u * QV;       // one row's run of a tile in the store
  __shared__ __attribute__((aligned(16))) uint32_t s_out[R * 64 * QV / 4];
  const uint32_t lane = threadIdx.x;
  const uint32_t g = blockIdx.y;
  const uint32_t DV = a.D >> 3;
  const uint32_t tile0 = blockIdx.x * TPW;
  const char* in_g = reinterpret_cast<const char*>(a.in.p[g]);
  uint8_t* q_g = a.q + (int64_t)g * a.qs.g;
  float* sc_g = a.scales + (int64_t)g * a.ssg;

  auto load_tile = [&](uint32_t tile, Vec8<IDT> (&x)[R]) {
    const char* src = in_g + ((int64_t)(a.t_begin + tile * a.TT) * a.is.t + (int64_t)lane * 8) * Elem<IDT>::size;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      x[r].load_nt(src + (int64_t)r * a.is.h * Elem<IDT>::size);
    }
  };
  auto process_tile = [&](uint32_t tile, const Vec8<IDT> (&x)[R]) {
    const uint32_t t0 = a.t_begin + tile * a.TT;
    uint32_t m = 0u;
#pragma unroll
    for (int r = 0; r < R; ++r) m = max(m, x[r].absmax_bits());  // across the tile's rows
    m = group_umax(m, a.dvshift);                                // across the D/8 lanes of the token
    const float s32 = fmaxf(Vec8<IDT>::bits_to_f32(m) / QRange<BITS>::qmax, a.eps);
    const float rcp = 1.0f / s32;
    if ((lane & (DV - 1u)) == 0u) sc_g[t0 + (lane >> a.dvshift)] = Elem<IDT>::round_trip(s32);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      uint32_t qb[8];
      quotient_bits8<BITS>(x[r], s32, rcp, qb);
      const uint32_t widx = ((uint32_t)r * 64u + lane) * (QV / 4);
      if constexpr (BITS == 8) {
        const u32x2 w = pack_i8(qb);
        s_out[widx] = w[0];
        s_out[widx + 1] = w[1];
      } else {
        s_out[widx] = pack_i4(qb);
      }
    }
    // one wave: its LDS queue is in order; only the compiler must not move the reads above the writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint8_t* qbase = q_g + (int64_t)t0 * a.qs.t;
#pragma unroll
    for (uint32_t k = lane * 16u; k < (uint32_t)R * kRowBytes; k += 64u * 16u) {
      const uint32_t r = k / kRowBytes, off = k % kRowBytes;
      const u32x4 w = *reinterpret_cast<const u32x4*>(&s_out[k >> 2]);
      __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(qbase + (int64_t)r * a.qs.h + off));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the next tile's staging writes stay below these reads
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  Vec8<IDT> xa[R], xb[R];
  load_tile(tile0, xa);
#pragma unroll
  for (int k = 0; k < TPW; k += 2) {
    if (k + 1 < TPW) load_tile(tile0 + k + 1, xb);
    process_tile(tile0 + k, xa);
    if (k + 1 < TPW) {
      if (k + 2 < TPW) load_tile(tile0 + k + 2, xa);
      process_tile(tile0 + k + 1, xb);
    }
  }
}

#endif  // KVQ_AB

// ---------------------------------------------------------------------------- swept tile (any R)
// Batched slices whose B*H*D exceeds the register tile (e.g. B = 64): the workgroup still owns
// TT tokens x all R rows of one group, but sweeps the rows twice in steps of `rpc` rows — once for
// the per-token abs-max (LDS), once to quantise. The second sweep re-reads what the workgroup has
// just read (tile sized to stay cache-resident), no global atomics, no workspace.
// POW2 = false: D/8 is not a power of two (D = 80, 96, 160, 192 ...) or exceeds one wave: index
// decomposition by 32-bit division and one LDS atomic per lane instead of the DPP group max.
template <int IDT, int BITS, bool POW2>
__global__ __launch_bounds__(kBlock) void quant_tokens_sweep_k(const QuantArgs a) {
  __shared__ uint32_t s_amax[kMaxTT];
  __shared__ float s_scale[kMaxTT], s_rcp[kMaxTT];
  const uint32_t tid = threadIdx.x;
  const uint32_t g = blockIdx.y;
  const uint32_t t0 = a.t_begin + blockIdx.x * a.TT;
  const uint32_t DV = a.D >> 3;
  const uint32_t wmask = POW2 ? (1u << a.vshift) - 1u : 0u;
  const char* in = reinterpret_cast<const char*>(a.in.p[g]) + (int64_t)t0 * a.is.t * Elem<IDT>::size;
  uint8_t* qbase = a.q + (int64_t)g * a.qs.g + (int64_t)t0 * a.qs.t;
  constexpr int QV = BITS;
  // vector index in a sweep step -> (row in step, vector in the row's TT-token run, token in tile)
  auto split = [&](uint32_t v, uint32_t& rl, uint32_t& wv, uint32_t& tl) {
    if constexpr (POW2) {
      rl = v >> a.vshift;
      wv = v & wmask;
      tl = wv >> a.dvshift;
    } else {
      rl = v / a.vpr;
      wv = v - rl * a.vpr;
      tl = wv / a.dv;
    }
  };
  auto in_row = [&](uint32_t r) -> int64_t {
    return a.bh_contig ? (int64_t)r * a.is.h : (int64_t)(r / a.H) * a.is.b + (int64_t)(r % a.H) * a.is.h;
  };
  auto q_row = [&](uint32_t r) -> int64_t {
    return a.bh_contig ? (int64_t)r * a.qs.h : (int64_t)(r / a.H) * a.qs.b + (int64_t)(r % a.H) * a.qs.h;
  };

  if (tid < kMaxTT) s_amax[tid] = 0u;
  __syncthreads();

  for (uint32_t r0 = 0; r0 < a.R; r0 += a.rpc) {  // sweep 1: abs-max
    const uint32_t rows = a.R - r0 < a.rpc ? a.R - r0 : a.rpc;
    const uint32_t nv = POW2 ? rows << a.vshift : rows * a.vpr;
#pragma unroll
    for (int i = 0; i < kNVMax; ++i) {
      if ((uint32_t)(i * kBlock) < nv) {  // uniform
        const uint32_t v = i * kBlock + tid;
        uint32_t rl, wv, tl;
        split(v, rl, wv, tl);
        const bool ok = v < nv && t0 + tl < a.T;
        uint32_t m = 0u;
        if (ok) {
          Vec8<IDT> x;
          x.load(in + (in_row(r0 + rl) + (int64_t)wv * 8) * Elem<IDT>::size);
          m = x.absmax_bits();
        }
        if constexpr (POW2) {
          m = group_umax(m, a.dvshift);
          if (ok && (wv & (DV - 1u)) == 0u) atomicMax(&s_amax[tl], m);
        } else {
          if (ok) atomicMax(&s_amax[tl], m);
        }
      }
    }
  }
  __syncthreads();
  if (tid < a.TT && t0 + tid < a.T) {
    const float amax = Vec8<IDT>::bits_to_f32(s_amax[tid]);
    const float s32 = fmaxf(amax / QRange<BITS>::qmax, a.eps);
    s_scale[tid] = s32;
    s_rcp[tid] = 1.0f / s32;
    a.scales[(int64_t)g * a.ssg + t0 + tid] = Elem<IDT>::round_trip(s32);
  }
  __syncthreads();

  for (uint32_t r0 = 0; r0 < a.R; r0 += a.rpc) {  // sweep 2: quantise
    const uint32_t rows = a.R - r0 < a.rpc ? a.R - r0 : a.rpc;
    const uint32_t nv = POW2 ? rows << a.vshift : rows * a.vpr;
#pragma unroll
    for (int i = 0; i < kNVMax; ++i) {
      const uint32_t v = i * kBlock + tid;
      uint32_t rl, wv, tl;
      split(v, rl, wv, tl);
      if (v < nv && t0 + tl < a.T) {
        const uint32_t r = r0 + rl;
        Vec8<IDT> x;
        x.load(in + (in_row(r) + (int64_t)wv * 8) * Elem<IDT>::size);
        uint32_t qb[8];
        quotient_bits8<BITS>(x, s_scale[tl], s_rcp[tl], qb);
        uint8_t* qp = qbase + q_row(r) + (int64_t)wv * QV;
        if constexpr (BITS == 8) *reinterpret_cast<u32x2*>(qp) = pack_i8(qb);
        else *reinterpret_cast<uint32_t*>(qp) = pack_i4(qb);
      }
    }
  }
}

// ---------------------------------------------------------------------------- generic two-pass
// Any D (odd included), any strides / alignment, any B*H*D size.
template <int IDT>
__global__ __launch_bounds__(kBlock) void absmax_tokens_generic_k(const QuantArgs a, uint32_t chunks_per_tok,
                                                                  int64_t chunk_elems) {
  __shared__ float s_red[kBlock / kWave];
  const uint32_t g = blockIdx.y;
  const uint32_t t = blockIdx.x / chunks_per_tok;
  const uint32_t c = blockIdx.x - t * chunks_per_tok;
  const int64_t RD = (int64_t)a.R * a.D;
  const int64_t i0 = (int64_t)c * chunk_elems;
  int64_t i1 = i0 + chunk_elems;
  if (i1 > RD) i1 = RD;
  const void* in = a.in.p[g];
  float m = 0.0f;
#pragma unroll 1  // generic fallback: four 64-bit divisions per element; unrolled it is 5,600 instructions per instantiation
  for (int64_t i = i0 + threadIdx.x; i < i1; i += kBlock) {
    const int64_t d = i % a.D;
    const int64_t r = i / a.D;
    const int64_t h = r % a.H;
    const int64_t b = r / a.H;
    m = fmaxf(m, fabsf(load1<IDT>(in, b * a.is.b + h * a.is.h + (int64_t)t * a.is.t + d)));
  }
  m = block_max_nonneg(m, s_red);
  if (threadIdx.x == 0)
    atomicMax(reinterpret_cast<uint32_t*>(a.absmax_ws) + (int64_t)g * a.T + t, __float_as_uint(m));
}

// One thread per output byte: INT8 one element, INT4 two elements (second may be the odd-D pad).
template <int IDT, int BITS>
__global__ __launch_bounds__(kBlock) void quant_tokens_generic_k(const QuantArgs a, int64_t total_bytes) {
  const int64_t Dq = BITS == 8 ? a.D : (a.D + 1) / 2;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total_bytes; i += (int64_t)gridDim.x * kBlock) {
    const int64_t j = i % Dq;
    int64_t r = i / Dq;
    const int64_t t = r % a.T;
    r /= a.T;
    const int64_t h = r % a.H;
    r /= a.H;
    const int64_t b = r % a.B;
    const int64_t g = r / a.B;
    const float amax = a.absmax_ws[g * a.T + t];
    const float s32 = fmaxf(amax / QRange<BITS>::qmax, a.eps);
    const void* in = a.in.p[g];
    const int64_t ioff = b * a.is.b + h * a.is.h + t * a.is.t;
    uint8_t* qp = a.q + g * a.qs.g + b * a.qs.b + h * a.qs.h + t * a.qs.t;
    if constexpr (BITS == 8) {
      qp[j] = (uint8_t)(int8_t)quant1<8>(load1<IDT>(in, ioff + j), s32);
    } else {
      const int hi = quant1<4>(load1<IDT>(in, ioff + 2 * j), s32) + 8;
      const int lo = (2 * j + 1 < (int64_t)a.D) ? quant1<4>(load1<IDT>(in, ioff + 2 * j + 1), s32) + 8 : 8;
      qp[j] = (uint8_t)(((hi & 0xF) << 4) | (lo & 0xF));
    }
    if (b == 0 && h == 0 && j == 0) a.scales[g * a.ssg + t] = Elem<IDT>::round_trip(s32);
  }
}

// ---------------------------------------------------------------------------- split phases (sharded batch)
// A [B>1,H,1,D] slice whose batch rows are split over ranks still has ONE scale per token (ops.py:27,48:
// abs().max() of the whole slice), so the fused kernels above cannot be used: the abs-max table must
// cross ranks between the reduction and the quantisation (one all_reduce(MAX) of [G,T] fp32).
//   absmax_tokens_tile_k        local rows -> absmax[g,t]
//   quant_tokens_scaled_tile_k  quantise with the GIVEN abs-max values; same quotient / pack / scale code as
//                               the fused kernels, so a 1-rank run is bit-identical with them
// 16-byte vectors; shapes that do not qualify (D % 8, D/8 not a power of two or > 64, token rows not contiguous,
// unaligned) take the generic pair (same arithmetic, scalar accesses; its abs-max table is zeroed first).
// Tile ownership as in the fused kernels: a workgroup owns TT tokens of one group (TT * D/8 = tvec vectors per row run,
// a power of two <= 256) and walks the R = B*H rows, nrs = 256 / tvec of them side by side. The abs-max kernel needs
// no global atomics (the tile's tokens are nobody else's: register max over the rows, DPP over the D/8 lanes of a
// token, LDS max across the side-by-side rows); the quantise kernel computes scale and reciprocal once per lane.
template <int IDT>
__global__ __launch_bounds__(kBlock) void absmax_tokens_tile_k(const QuantArgs a) {
  __shared__ uint32_t s_amax[kBlock];
  const uint32_t tid = threadIdx.x, g = blockIdx.y;
  const uint32_t tvec = 1u << a.vshift, nrs = kBlock >> a.vshift;
  const uint32_t rs = tid >> a.vshift, wv = tid & (tvec - 1u), tl = wv >> a.dvshift;
  const uint32_t t0 = blockIdx.x * a.TT;
  const bool ok = t0 + tl < a.T;
  const char* in = reinterpret_cast<const char*>(a.in.p[g]) + ((int64_t)t0 * a.is.t + (int64_t)wv * 8) * Elem<IDT>::size;
  if (tid < a.TT) s_amax[tid] = 0u;
  __syncthreads();
  uint32_t m = 0u;
  if (ok) {
    for (uint32_t r = rs; r < a.R; r += nrs) {
      Vec8<IDT> x;
      x.load(in + ((int64_t)(r / a.H) * a.is.b + (int64_t)(r % a.H) * a.is.h) * Elem<IDT>::size);
      m = max(m, x.absmax_bits());
    }
  }
  m = group_umax(m, a.dvshift);
  if (ok && (wv & ((1u << a.dvshift) - 1u)) == 0u) atomicMax(&s_amax[tl], m);
  __syncthreads();
  if (tid < a.TT && t0 + tid < a.T) {
    float* dst = a.absmax_ws + (int64_t)g * a.T + t0 + tid;
    const float v = Vec8<IDT>::bits_to_f32(s_amax[tid]);
    *dst = a.acc ? fmaxf(*dst, v) : v;  // the tile's tokens are nobody else's: no atomic needed
  }
}

template <int IDT, int BITS>
__global__ __launch_bounds__(kBlock) void quant_tokens_scaled_tile_k(const QuantArgs a) {
  const uint32_t tid = threadIdx.x, g = blockIdx.y;
  const uint32_t tvec = 1u << a.vshift, nrs = kBlock >> a.vshift;
  const uint32_t rs = tid >> a.vshift, wv = tid & (tvec - 1u), tl = wv >> a.dvshift;
  const uint32_t t0 = blockIdx.x * a.TT;
  if (t0 + tl >= a.T) return;
  constexpr int QV = BITS;
  const char* in = reinterpret_cast<const char*>(a.in.p[g]) + ((int64_t)t0 * a.is.t + (int64_t)wv * 8) * Elem<IDT>::size;
  uint8_t* q = a.q + (int64_t)g * a.qs.g + (int64_t)(t0 + tl) * a.qs.t + (int64_t)(wv & ((1u << a.dvshift) - 1u)) * QV;
  const float s32 = fmaxf(a.absmax_ws[(int64_t)g * a.T + t0 + tl] / QRange<BITS>::qmax, a.eps);
  const float rcp = 1.0f / s32;
  if (rs == 0u && (wv & ((1u << a.dvshift) - 1u)) == 0u) a.scales[(int64_t)g * a.ssg + t0 + tl] = Elem<IDT>::round_trip(s32);
  for (uint32_t r = rs; r < a.R; r += nrs) {
    const uint32_t b = r / a.H, h = r % a.H;
    Vec8<IDT> x;
    x.load(in + ((int64_t)b * a.is.b + (int64_t)h * a.is.h) * Elem<IDT>::size);
    uint32_t qb[8];
    quotient_bits8<BITS>(x, s32, rcp, qb);
    uint8_t* qp = q + (int64_t)b * a.qs.b + (int64_t)h * a.qs.h;
    if constexpr (BITS == 8) *reinterpret_cast<u32x2*>(qp) = pack_i8(qb);
    else *reinterpret_cast<uint32_t*>(qp) = pack_i4(qb);
  }
}

// ---------------------------------------------------------------------------- host side

static uint32_t pow2_floor(uint64_t v) {
  uint32_t p = 1;
  while ((uint64_t)p * 2 <= v) p *= 2;
  return p;
}

template <int IDT, int BITS>
static void launch_quant(const QuantArgs& a, bool fused, hipStream_t st) {
  if (fused && a.rpc) {  // swept tile: B*H*D larger than the register tile
    if constexpr (KVQ_AB || IDT != KVQ_F32) {  // (fp32 batched slices take the generic pair in the default library: host)
      const unsigned tiles = (a.T + a.TT - 1) / a.TT;
#if KVQ_AB
      if (!a.vpr) {
        KVQ_LAUNCH((quant_tokens_sweep_k<IDT, BITS, true>), dim3(tiles, a.G), dim3(kBlock), 0, st, a);
        return;
      }
#endif
      KVQ_LAUNCH((quant_tokens_sweep_k<IDT, BITS, false>), dim3(tiles, a.G), dim3(kBlock), 0, st, a);
    }
  } else if (fused) {
    const unsigned tiles = (a.T + a.TT - 1) / a.TT;
    // LDS-staged 16 B stores need 16-byte aligned row runs in the store
    const int64_t dq = (int64_t)a.D * BITS / 8;
    const bool lds_out = !tunables().quant_direct_stores && dq % 16 == 0 && a.qs.g % 16 == 0 && a.qs.h % 16 == 0 &&
                         a.qs.t % 16 == 0 && aligned(a.q, 16);
    const bool rowu = a.vshift >= 8;
    // complete tiles go to the predicate-free FULL kernel, a ragged last tile to the general one
#if KVQ_AB
    if (a.blk == 64 && a.nv == 4) {  // smallest tile: 2048 elements per wave, row runs <= one wave
      const unsigned n_small = a.nvec == 64u * 4u ? a.T / a.TT : 0u;
      if (n_small) {
        QuantArgs f = a;
        f.t_begin = 0;
        KVQ_LAUNCH((quant_tokens_fused_k<IDT, BITS, false, true, true, 64, true, 4>), dim3(n_small, a.G), dim3(64), 0, st, f);
      }
      if (tiles - n_small) {
        QuantArgs t = a;
        t.t_begin = n_small * a.TT;
        KVQ_LAUNCH((quant_tokens_fused_k<IDT, BITS, false, true, false, 64, false, 4>), dim3(tiles - n_small, a.G), dim3(64), 0, st, t);
      }
      return;
    }
    if (a.blk == 64 && a.nv == 16) {  // 8192-element one-wave tile: twice the row run (512 B INT4 store pieces)
      const unsigned n_small = a.nvec == 64u * 16u ? a.T / a.TT : 0u;
      const bool regmax2 = (1 << a.vshift) == 128 && !tunables().quant_no_regmax;
      if (n_small) {
        QuantArgs f = a;
        f.t_begin = 0;
        if (regmax2)
          KVQ_LAUNCH((quant_tokens_fused_k<IDT, BITS, true, true, true, 64, 2, 16>), dim3(n_small, a.G), dim3(64), 0, st, f);
        else
          KVQ_LAUNCH((quant_tokens_fused_k<IDT, BITS, true, true, true, 64, 0, 16>), dim3(n_small, a.G), dim3(64), 0, st, f);
      }
      if (tiles - n_small) {
        QuantArgs t = a;
        t.t_begin = n_small * a.TT;
        KVQ_LAUNCH((quant_tokens_fused_k<IDT, BITS, true, true, false, 64, 0, 16>), dim3(tiles - n_small, a.G), dim3(64), 0, st, t);
      }
      return;
    }
#endif
    if constexpr (KVQ_AB || IDT != KVQ_F32)
    if (a.blk == 64 || (KVQ_AB && a.blk == 128)) {  // small workgroups (host guarantees ROWU + LDS_OUT eligibility)
      unsigned n_small = a.nvec == (uint32_t)(a.blk * kNVMax) ? a.T / a.TT : 0u;
      const bool regmax = (1 << a.vshift) == a.blk && !tunables().quant_no_regmax;  // one round == one row
      unsigned piped = 0;  // tiles taken by the pipelined kernel (multiples of its tiles-per-wave)
#if KVQ_AB
      if constexpr (IDT != KVQ_F32) {
        const int64_t tpw = tunables().quant_tpw;
        if (a.blk == 64 && regmax && a.R == 8u && (tpw == 2 || tpw == 4 || tpw == 8) && n_small >= (unsigned)tpw && a.bh_contig) {
          piped = n_small / (unsigned)tpw * (unsigned)tpw;
          QuantArgs f = a;
          f.t_begin = 0;
          const dim3 grid(piped / (unsigned)tpw, a.G);
          if (tpw == 2) KVQ_LAUNCH((quant_tokens_pipe_k<IDT, BITS, 8, 2>), grid, dim3(64), 0, st, f);
          else if (tpw == 4) KVQ_LAUNCH((quant_tokens_pipe_k<IDT, BITS, 8, 4>), grid, dim3(64), 0, st, f);
          else KVQ_LAUNCH((quant_tokens_pipe_k<IDT, BITS, 8, 8>), grid, dim3(64), 0, st, f);
        }
      }
#endif
      n_small -= piped;
      if (n_small) {
        QuantArgs f = a;
        f.t_begin = piped * a.TT;
        const dim3 grid(n_small, a.G);
#if KVQ_AB
        if (a.blk == 64 && regmax && IDT != KVQ_F32 && a.R == 8u && a.D == 128u && a.TT == 4u && a.vshift == 6 && a.dvshift == 4 &&
            a.is.t == 128 && a.qs.t == 128 * BITS / 8 && !a.xcd_group && !tunables().quant_lds_pad && tunables().quant_geo128)
        {  // round 2's shipped shape; nt_loads / quant_nt_stores pick the instantiation
          constexpr bool kGeo = IDT != KVQ_F32;
          if (a.nt_loads && a.nt_stores)
            KVQ_LAUNCH((quant_tokens_fused_k<IDT, BITS, true, true, true, 64, 1, kNVMax, kGeo, true, true>), grid, dim3(64), 0, st, f);
          else if (a.nt_loads)
            KVQ_LAUNCH((quant_tokens_fused_k<IDT, BITS, true, true, true, 64, 1, kNVMax, kGeo, true, false>), grid, dim3(64), 0, st, f);
          else if (a.nt_stores)
            KVQ_LAUNCH((quant_tokens_fused_k<IDT, BITS, true, true, true, 64, 1, kNVMax, kGeo, false, true>), grid, dim3(64), 0, st, f);
          else
            KVQ_LAUNCH((quant_tokens_fused_k<IDT, BITS, true, true, true, 64, 1, kNVMax, kGeo, false, false>), grid, dim3(64), 0, st, f);
        }
        else if (a.blk == 128 && regmax)
          KVQ_LAUNCH((quant_tokens_fused_k<IDT, BITS, true, true, true, 128, true>), grid, dim3(128), 0, st, f);
        else if (a.blk == 128)
          KVQ_LAUNCH((quant_tokens_fused_k<IDT, BITS, true, true, true, 128>), grid, dim3(128), 0, st, f);
        else
#endif
        if (regmax)  // quant_lds_pad (A-B): dynamic LDS the kernel never touches, an occupancy cap
          KVQ_LAUNCH((quant_tokens_fused_k<IDT, BITS, true, true, true, 64, true>), grid, dim3(64),
                             (size_t)tunables().quant_lds_pad, st, f);
        else
          KVQ_LAUNCH((quant_tokens_fused_k<IDT, BITS, true, true, true, 64>), grid, dim3(64), 0, st, f);
      }
      if (tiles - n_small - piped) {
        QuantArgs t = a;
        t.t_begin = (n_small + piped) * a.TT;
        const dim3 grid(tiles - n_small - piped, a.G);
#if KVQ_AB
        if (a.blk == 128)
          KVQ_LAUNCH((quant_tokens_fused_k<IDT, BITS, true, true, false, 128>), grid, dim3(128), 0, st, t);
        else
#endif
          KVQ_LAUNCH((quant_tokens_fused_k<IDT, BITS, true, true, false, 64>), grid, dim3(64), 0, st, t);
      }
      return;
    }
    const unsigned n_full = (a.nvec == (uint32_t)(kBlock * kNVMax) && rowu && lds_out) ? a.T / a.TT : 0u;
    if (n_full) {
      QuantArgs f = a;
      f.t_begin = 0;
      KVQ_LAUNCH((quant_tokens_fused_k<IDT, BITS, true, true, true>), dim3(n_full, a.G), dim3(kBlock), 0, st, f);
    }
    const unsigned rest = tiles - n_full;
    if (rest) {
      QuantArgs t = a;
      t.t_begin = n_full * a.TT;
#if KVQ_AB  // the scalar-row instantiations of the ragged tile: the general index split below covers them
      if (rowu && lds_out)
        KVQ_LAUNCH((quant_tokens_fused_k<IDT, BITS, true, true, false>), dim3(rest, a.G), dim3(kBlock), 0, st, t);
      else if (rowu)
        KVQ_LAUNCH((quant_tokens_fused_k<IDT, BITS, true, false, false>), dim3(rest, a.G), dim3(kBlock), 0, st, t);
      else
#endif
      if (lds_out)
        KVQ_LAUNCH((quant_tokens_fused_k<IDT, BITS, false, true, false>), dim3(rest, a.G), dim3(kBlock), 0, st, t);
      else
        KVQ_LAUNCH((quant_tokens_fused_k<IDT, BITS, false, false, false>), dim3(rest, a.G), dim3(kBlock), 0, st, t);
    }
  } else {
    const int64_t RD = (int64_t)a.R * a.D;
    const int64_t chunk_elems = (int64_t)kBlock * 16;
    const uint32_t cpt = (uint32_t)((RD + chunk_elems - 1) / chunk_elems);
    if (hipMemsetAsync(a.absmax_ws, 0, sizeof(float) * (size_t)a.G * a.T, st) != hipSuccess) return;  // surfaced by check_launch
    KVQ_LAUNCH((absmax_tokens_generic_k<IDT>), dim3(a.T * cpt, a.G), dim3(kBlock), 0, st, a, cpt, chunk_elems);
    const int64_t Dq = BITS == 8 ? a.D : (a.D + 1) / 2;
    const int64_t total = (int64_t)a.G * a.B * a.H * a.T * Dq;
    int64_t blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 256 * 32) blocks = 256 * 32;
    KVQ_LAUNCH((quant_tokens_generic_k<IDT, BITS>), dim3((unsigned)blocks), dim3(kBlock), 0, st, a, total);
  }
}

// PHASE 1 for a slice of few tokens (a batch-sharded DECODE APPEND: T = 1; up to tunables().quant_few_tokens = 16). The tile walk gives every 8-row group its own
// one-wave workgroup, and all rows / 8 of them end in an atomicMax on the SAME word of the table: device-scope atomics from
// eight XCDs on one address are served one after the other — 21.5 us for the 8.4 MB of a batch-64 K + V append of 64 tensors
// (rocprofv3: profiles/r04z_rocprofv3_kernel_stats_shardq_append.csv, quant_tile_k<..., 1>). Here ONE 256-thread workgroup
// owns a (group, token): 256 / DV rows per pass, 8 independent 16-byte loads per thread in flight, LDS across its 4 waves,
// then ONE plain store — or a plain max with the word when the caller accumulates (no other workgroup of the launch touches
// it) — no atomics, and no memset launch in front. The maximum does not depend on the order: same table, bit for bit.
// Against the tile walk on [64, 64, 8, T, 128] fp16 (profiles/r04aa_absmax_few_tokens_sweep.jsonl): T = 1 4.3 vs 21.2 us, 2: 4.3 vs
// 11.6, 4: 7.9 vs 10.3, 8: 10.7 vs 13.0, 16: 20.9 vs 21.1, 32: 47.2 vs 48.9, 64: 94.4 vs 90.1 — the default threshold is 16.
template <int IDT, int DV>
__global__ __launch_bounds__(kBlock) void absmax_fewtokens_k(const QuantTileArgs a, const int accumulate) {
  static_assert(IDT != KVQ_F32, "two-byte inputs");
  constexpr int D = DV * 8;
  constexpr int RPP = kBlock / DV;  // rows per pass
  constexpr int UN = 8;
  __shared__ uint32_t s_m[kBlock / kWave];
  const uint32_t tid = threadIdx.x, t = blockIdx.x, g = blockIdx.y;
  const uint32_t piece = tid % DV, slot = tid / DV;
  const char* base = reinterpret_cast<const char*>(a.in.p[g]) + (int64_t)t * (D * 2) + piece * 16u;
  uint32_t m = 0u;
  for (uint32_t r0 = slot; r0 < a.rows; r0 += (uint32_t)(RPP * UN)) {
    Vec8<IDT> x[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const uint32_t r = r0 + (uint32_t)(u * RPP);
      x[u].w = u32x4{0u, 0u, 0u, 0u};
      if (r < a.rows) x[u].w = *reinterpret_cast<const u32x4*>(base + (int64_t)r * a.is_h);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) m = max(m, x[u].absmax_bits());
  }
  m = group_umax(m, 6);  // the whole wave
  if ((tid & 63u) == 0u) s_m[tid >> 6] = m;
  __syncthreads();
  if (tid == 0u) {
    m = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
    float* dst = a.absmax + (int64_t)g * a.T + t;
    float v = Vec8<IDT>::bits_to_f32(m);
    if (accumulate) v = fmaxf(v, *dst);
    *dst = v;
  }
}

// Can the one-wave tile kernels over 8-row groups (quant_tile_k PHASE 1 / 2) serve this call? bits = 0: the abs-max phase
// alone (no store side to check).
static bool split_tile_ok(int bits, const void* in_base, const void* const* in_ptrs, const kvq_strides_t* in_st, int in_dtype,
                          const uint8_t* q, const kvq_strides_t* q_st, const kvq_dims_t* d) {
  const int64_t Rt = d->B * d->H;
  const int esz = in_dtype == KVQ_F32 ? 4 : 2;
  bool tile = tunables().quant_tile && in_dtype != KVQ_F32 && (d->D == 128 || (KVQ_AB && d->D == 64)) &&  // (head_dim 64: A-B builds; the default library takes the 256-thread pair)
              (d->T == 1 || in_st->t == d->D) &&
              (d->B == 1 || in_st->b == d->H * in_st->h) && (in_st->h * 2) % 16 == 0 && in_st->h >= 0 &&
              (Rt - 1) * in_st->h * 2 + 8 * d->D * 2 < (int64_t(1) << 31) && (Rt + 7) / 8 < 65536;
  if (bits)
    tile = tile && q && q_st && (d->T == 1 || q_st->t == d->D * bits / 8) && (d->B == 1 || q_st->b == d->H * q_st->h) && q_st->h % 16 == 0 &&
           q_st->g % 16 == 0 && q_st->h >= 0 && (d->D * bits / 8) % 16 == 0 && aligned(q, 16) && (Rt - 1) * q_st->h + 8 * d->D < (int64_t(1) << 31);
  for (int64_t i = 0; i < d->G && tile; ++i)
    tile = aligned(in_ptrs ? in_ptrs[i] : static_cast<const char*>(in_base) + i * in_st->g * (int64_t)esz, 16) && (in_ptrs ? in_ptrs[i] : in_base) != nullptr;
  return tile;
}

template <int PHASE>
static int split_phase(const char* name, int bits, const void* in_base, const void* const* in_ptrs, const kvq_strides_t* in_st,
                       int in_dtype, uint8_t* q, const kvq_strides_t* q_st, float* scales, int64_t ssg, float* absmax,
                       float eps, const kvq_dims_t* d, void* stream, bool accumulate = false);

template <int BITS>
static int quant_tokens(const void* in_base, const void* const* in_ptrs, const kvq_strides_t* in_st, int in_dtype,
                        uint8_t* q, const kvq_strides_t* q_st, float* scales, int64_t ssg, float* absmax_ws,
                        float eps, const kvq_dims_t* d, void* stream, const char* name) {
  if (!in_st || !q_st || !d || !q || !scales || (!in_base && !in_ptrs)) {
    set_error("%s: NULL argument", name);
    return KVQ_E_NULL;
  }
  if (const int rcd = check_device(q, name)) return rcd;
  if (in_base && in_ptrs) {
    set_error("%s: pass in_base or in_ptrs, not both", name);
    return KVQ_E_DIMS;
  }
  if (d->G < 0 || d->B < 0 || d->H < 0 || d->T < 0 || d->D < 0) {
    set_error("%s: negative dim", name);
    return KVQ_E_DIMS;
  }
  if (in_dtype != KVQ_F16 && in_dtype != KVQ_BF16 && in_dtype != KVQ_F32) {
    set_error("%s: unknown in_dtype %d", name, in_dtype);
    return KVQ_E_DTYPE;
  }
  if (d->G * d->B * d->H * d->T * d->D == 0) return 0;
  if (d->B >= (int64_t(1) << 31) || d->H >= (int64_t(1) << 31) || d->T >= (int64_t(1) << 31) ||
      d->D >= (int64_t(1) << 31) || d->B * d->H >= (int64_t(1) << 31)) {
    set_error("%s: dims too large", name);
    return KVQ_E_DIMS;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int esz = in_dtype == KVQ_F32 ? 4 : 2;
  const int64_t R = d->B * d->H;
  const int64_t Dq = BITS == 8 ? d->D : (d->D + 1) / 2;

  QuantArgs a;
  a.is = to_strides(in_st);
  a.qs = to_strides(q_st);
  a.ssg = ssg;
  a.eps = eps;
  a.B = (uint32_t)d->B;
  a.H = (uint32_t)d->H;
  a.T = (uint32_t)d->T;
  a.D = (uint32_t)d->D;
  a.R = (uint32_t)R;

  // ---- compile-time tile kernel: the BASELINE shapes (quant_tile_k) ---------------------------------------------
  if (tunables().quant_tile && !tunables().quant_force_two_pass && in_dtype != KVQ_F32 && (d->B == 1 || (a.is.b == d->H * a.is.h && a.qs.b == d->H * a.qs.h)) &&
      (d->T == 1 || (a.is.t == d->D && a.qs.t == Dq)) && (a.is.h * 2) % 16 == 0 && a.qs.h % 16 == 0 && a.qs.g % 16 == 0 &&
      ((int64_t)d->D * BITS / 8) % 16 == 0 && aligned(q, 16) && a.is.h >= 0 && a.qs.h >= 0) {
    int tt = 0;  // tokens per tile of the instantiation that serves (R, D); 0 = none
    if (d->D == 128 && R == 8) tt = 4;
    else if (d->D == 64 && (R == 8 || ((R == 12 || R == 16) && (KVQ_AB || in_dtype == KVQ_F16)))) tt = tunables().quant_tile_tt == 4 && KVQ_AB ? 4 : 8;  // 12 / 16 rows: the gpt2 family (fp16)
    const int64_t span_in = (R - 1) * a.is.h * 2 + (int64_t)tt * d->D * 2, span_q = (R - 1) * a.qs.h + (int64_t)tt * Dq;
    if (tt && span_in < (int64_t(1) << 31) && span_q < (int64_t(1) << 31)) {
      bool all_aligned = true;
      for (int64_t i = 0; i < d->G; ++i)
        all_aligned = all_aligned && aligned(in_ptrs ? in_ptrs[i] : static_cast<const char*>(in_base) + i * a.is.g * (int64_t)esz, 16) &&
                      (in_ptrs ? in_ptrs[i] : in_base) != nullptr;
      if (all_aligned) {
        QuantTileArgs ta;
        ta.qs_g = a.qs.g;
        ta.ssg = ssg;
        ta.is_h = (uint32_t)(a.is.h * 2);
        ta.qs_h = (uint32_t)a.qs.h;
        ta.T = (uint32_t)d->T;
        ta.rows = (uint32_t)R;
        ta.absmax = nullptr;
        ta.eps = eps;
        for (int64_t g0 = 0; g0 < d->G; g0 += kPtrsPerLaunch) {
          const int64_t gn = d->G - g0 < kPtrsPerLaunch ? d->G - g0 : kPtrsPerLaunch;
          for (int64_t i = 0; i < gn; ++i)
            ta.in.p[i] = in_ptrs ? in_ptrs[g0 + i] : static_cast<const char*>(in_base) + (g0 + i) * a.is.g * (int64_t)esz;
          ta.q = q + g0 * a.qs.g;
          ta.scales = scales + g0 * ssg;
          const dim3 grid((unsigned)((d->T + tt - 1) / tt), (unsigned)gn);
#if KVQ_AB  // merged stores: TPW tiles per wave (measured slower: 242 / 294 us against 231 / 272, profiles/r03m_*)
          const int64_t tpw = tunables().quant_tile_tpw;
          if ((tpw == 2 || tpw == 4) && R == 8 && in_dtype == KVQ_F16) {
            const int ttw = (int)(tt * tpw);
            const dim3 grid2((unsigned)((d->T + ttw - 1) / ttw), (unsigned)gn);
            if (d->D == 128 && tpw == 2) KVQ_LAUNCH((quant_tile_merged_k<KVQ_F16, BITS, 8, 16, 4, 2>), grid2, dim3(kWave), 0, st, ta);
            else if (d->D == 128) KVQ_LAUNCH((quant_tile_merged_k<KVQ_F16, BITS, 8, 16, 4, 4>), grid2, dim3(kWave), 0, st, ta);
            else if (tpw == 2) KVQ_LAUNCH((quant_tile_merged_k<KVQ_F16, BITS, 8, 8, 8, 2>), grid2, dim3(kWave), 0, st, ta);
            else KVQ_LAUNCH((quant_tile_merged_k<KVQ_F16, BITS, 8, 8, 8, 4>), grid2, dim3(kWave), 0, st, ta);
            const int rc2 = check_launch(name);
            if (rc2) return rc2;
            continue;
          }
#endif
#define KVQ_TILE(IDT_, R_, DV_, TT_) KVQ_LAUNCH((quant_tile_k<IDT_, BITS, R_, DV_, TT_>), grid, dim3(kWave), 0, st, ta)
#if KVQ_AB
#define KVQ_TILE64(IDT_, R_) do { if (tt == 4) KVQ_TILE(IDT_, R_, 8, 4); else KVQ_TILE(IDT_, R_, 8, 8); } while (0)
#else
#define KVQ_TILE64(IDT_, R_) KVQ_TILE(IDT_, R_, 8, 8)
#endif
#define KVQ_TILE_BY_SHAPE(IDT_)                         \
  do {                                                  \
    if (d->D == 128) KVQ_TILE(IDT_, 8, 16, 4);          \
    else if (R == 8) KVQ_TILE64(IDT_, 8);               \
    else if (R == 12) KVQ_TILE64(IDT_, 12);             \
    else KVQ_TILE64(IDT_, 16);                          \
  } while (0)
          if (in_dtype == KVQ_F16) KVQ_TILE_BY_SHAPE(KVQ_F16);
#if KVQ_AB
          else KVQ_TILE_BY_SHAPE(KVQ_BF16);
#else
          else if (d->D == 128) KVQ_TILE(KVQ_BF16, 8, 16, 4);
          else KVQ_TILE64(KVQ_BF16, 8);
#endif
#undef KVQ_TILE_BY_SHAPE
#undef KVQ_TILE64
#undef KVQ_TILE
          const int rc = check_launch(name);
          if (rc) return rc;
        }
        return 0;
      }
    }
  }

  // fused single-pass eligibility (layout); pointer alignment is checked per launch chunk below
  const int dvshift = d->D % 8 == 0 ? ilog2_exact(d->D / 8) : -1;
  const int64_t qvec = BITS == 8 ? 8 : 4;  // bytes stored per 8-element vector
  const bool bh_contig = d->B == 1 || (a.is.b == d->H * a.is.h && a.qs.b == d->H * a.qs.h);
  // D/8 not a power of two, or wider than one wave: swept tile with division-based indexing
  const bool anydv = d->D % 8 == 0 && (dvshift < 0 || d->D / 8 > kWave);
  const bool big = anydv || R * d->D > kTileElems;  // swept-tile kernel instead of the register tile
  // Batched slices larger than the register tile (B*H*D > 16384, e.g. an un-sharded batch of 64): two passes of the
  // one-wave tile kernels over 8-row groups — abs-max into the caller's [G,T] workspace, then quantise — run at 6.1-6.3
  // TB/s of their own traffic each (3.6 TB/s of single-pass bytes) where the swept tile reaches 2.5
  // ---- wide tile: one pass with the slice's TT tokens x all rows in the registers of a 1024-thread workgroup ------
  if (big && !anydv && in_dtype != KVQ_F32 && tunables().quant_wide != 0 && !tunables().quant_force_two_pass &&
      (d->T == 1 || (a.is.t == d->D && a.qs.t == Dq)) && (a.is.b * esz) % 16 == 0 && (a.is.h * esz) % 16 == 0 &&
      a.qs.b % qvec == 0 && a.qs.h % qvec == 0 && a.qs.g % qvec == 0 && aligned(q, qvec)) {
    constexpr int kWideNV = 16;
    const int kWideBlk = KVQ_AB && tunables().quant_wide_blk == 512 ? 512 : 1024;  // (A-B: two 512-thread workgroups per CU)
    const int64_t cap = (int64_t)kWideBlk * kWideNV * 8;  // elements a workgroup holds
    if (R * d->D <= cap) {
      uint32_t tt = pow2_floor((uint64_t)(cap / (R * d->D)));
      if (tt > (uint32_t)kMaxTT) tt = kMaxTT;
      while (tt > 1 && tt / 2 >= (uint64_t)d->T) tt /= 2;
      QuantWideArgs wa;
      wa.dvshift = (uint32_t)dvshift;
      wa.vshift = (uint32_t)(dvshift + ilog2_exact((int64_t)tt));
      const int64_t rpr = kWideBlk >> wa.vshift;  // >= 2: tt * D/8 <= cap / (8 R) < 512 since R * D > 16384, D/8 <= 64
      const bool in_contig = d->B == 1 || a.is.b == d->H * a.is.h, q_contig = d->B == 1 || a.qs.b == d->H * a.qs.h;
      bool ok = rpr >= 1 && (in_contig || rpr % d->H == 0) && (q_contig || rpr % d->H == 0);
      for (int64_t i = 0; ok && i < d->G; ++i) {
        const void* p = in_ptrs ? in_ptrs[i] : static_cast<const char*>(in_base) + i * a.is.g * (int64_t)esz;
        ok = p != nullptr && aligned(p, 16);
      }
      if (ok) {
        wa.qs_g = a.qs.g;
        wa.ssg = ssg;
        wa.is_h = a.is.h * esz;
        wa.is_b = in_contig ? d->H * wa.is_h : a.is.b * esz;
        wa.qs_h = a.qs.h;
        wa.qs_b = q_contig ? d->H * wa.qs_h : a.qs.b;
        wa.is_round = in_contig ? rpr * wa.is_h : rpr / d->H * wa.is_b;
        wa.qs_round = q_contig ? rpr * wa.qs_h : rpr / d->H * wa.qs_b;
        wa.H = (uint32_t)d->H;
        wa.R = (uint32_t)R;
        wa.T = (uint32_t)d->T;
        wa.D = (uint32_t)d->D;
        wa.eps = eps;
        for (int64_t g0 = 0; g0 < d->G; g0 += kPtrsPerLaunch) {
          const int64_t gn = d->G - g0 < kPtrsPerLaunch ? d->G - g0 : kPtrsPerLaunch;
          for (int64_t i = 0; i < gn; ++i)
            wa.in.p[i] = in_ptrs ? in_ptrs[g0 + i] : static_cast<const char*>(in_base) + (g0 + i) * a.is.g * (int64_t)esz;
          wa.q = q + g0 * a.qs.g;
          wa.scales = scales + g0 * ssg;
          const dim3 grid((unsigned)((d->T + tt - 1) / tt), (unsigned)gn);
#if KVQ_AB
          if (kWideBlk == 512) {
            if (in_dtype == KVQ_F16) KVQ_LAUNCH((quant_wide_k<KVQ_F16, BITS, 512, kWideNV>), grid, dim3(512), 0, st, wa);
            else KVQ_LAUNCH((quant_wide_k<KVQ_BF16, BITS, 512, kWideNV>), grid, dim3(512), 0, st, wa);
          } else
#endif
          if (in_dtype == KVQ_F16) KVQ_LAUNCH((quant_wide_k<KVQ_F16, BITS, 1024, kWideNV>), grid, dim3(1024), 0, st, wa);
          else KVQ_LAUNCH((quant_wide_k<KVQ_BF16, BITS, 1024, kWideNV>), grid, dim3(1024), 0, st, wa);
          const int rc = check_launch(name);
          if (rc) return rc;
        }
        return 0;
      }
    }
  }
  if (big && !anydv && absmax_ws && !tunables().quant_force_two_pass && split_tile_ok(BITS, in_base, in_ptrs, in_st, in_dtype, q, q_st, d)) {
    int rc = split_phase<1>(name, BITS, in_base, in_ptrs, in_st, in_dtype, nullptr, nullptr, nullptr, 0, absmax_ws, eps, d, stream, false);
    if (rc) return rc;
    return split_phase<2>(name, BITS, in_base, in_ptrs, in_st, in_dtype, q, q_st, scales, ssg, absmax_ws, eps, d, stream, false);
  }
  bool fused = !tunables().quant_force_two_pass && d->D % 8 == 0 && d->D <= kTileElems &&
               (d->T == 1 || (a.is.t == d->D && a.qs.t == Dq)) && (big ? (KVQ_AB || in_dtype != KVQ_F32) : bh_contig) &&
               (a.is.b * esz) % 16 == 0 && a.qs.b % qvec == 0 &&
               (a.is.h * esz) % 16 == 0 && (a.is.t * esz) % 16 == 0 && a.qs.h % qvec == 0 &&
               a.qs.t % qvec == 0 && a.qs.g % qvec == 0 && aligned(q, qvec);
  a.rpc = 0;
  a.dv = a.vpr = 0;
  a.t_begin = 0;
  a.blk = kBlock;
  a.nv = kNVMax;
  a.nt_loads = (int32_t)tunables().nt_loads;
  // -1 (default): non-temporal like the loads. (A first measurement, in a loop that rewrote the SAME 268 MB store, had
  // write-back stores ahead for INT4: the store was sitting in the 256 MB Infinity Cache. With rotating buffers:
  // INT4 0.231-0.232 ms non-temporal vs 0.247-0.248 write-back, INT8 0.270-0.271 vs 0.300-0.302.)
  a.nt_stores = tunables().quant_nt_stores < 0 ? a.nt_loads : (int32_t)(tunables().quant_nt_stores != 0);
  a.bh_contig = bh_contig ? 1 : 0;
  a.xcd_group = (uint32_t)(tunables().quant_xcd_group > 1 ? tunables().quant_xcd_group : 0);
  // (default library: ONE swept-tile kernel, the division-indexed one; the shift-indexed instantiation is an A-B variant —
  // power-of-two shapes this large take the split-phase tile kernels above when the caller passed a workspace)
  if (fused && big && (anydv || !KVQ_AB)) {
    const int64_t dv = d->D / 8;
    int64_t tt = (256 * 1024) / (R * d->D * esz);
    if (tt < 1) tt = 1;
    if (tt > kMaxTT) tt = kMaxTT;
    if (tt * dv > kBlock * kNVMax) tt = (kBlock * kNVMax) / dv;  // >= 1: D <= kTileElems
    if (tt > d->T) tt = d->T;
    a.TT = (uint32_t)tt;
    a.dvshift = a.vshift = 0;
    a.nvec = 0;
    a.dv = (uint32_t)dv;
    a.vpr = (uint32_t)(tt * dv);
    a.rpc = (uint32_t)((kBlock * kNVMax) / a.vpr);
  } else if (fused && big) {
    // tile = R x TT tokens, kept around 256 KiB so the second sweep is served from cache
    uint64_t tt = pow2_floor((uint64_t)((256 * 1024) / (R * d->D * esz) > 0 ? (256 * 1024) / (R * d->D * esz) : 1));
    if (tt > (uint64_t)kMaxTT) tt = kMaxTT;
    while (tt > 1 && tt * d->D > (uint64_t)kTileElems) tt /= 2;
    while (tt > 1 && tt / 2 >= (uint64_t)d->T) tt /= 2;
    a.TT = (uint32_t)tt;
    a.dvshift = dvshift;
    a.vshift = dvshift + ilog2_exact((int64_t)tt);
    a.nvec = 0;
    a.rpc = (uint32_t)((kBlock * kNVMax) >> a.vshift);  // rows covered by one sweep step (>= 1)
  } else if (fused) {
    uint32_t tt = pow2_floor((uint64_t)(kTileElems / (R * d->D)));
    if (tt > kMaxTT) tt = kMaxTT;
    while (tt > 1 && tt / 2 >= (uint64_t)d->T) tt /= 2;  // do not over-tile short appends
    a.TT = tt;
    a.dvshift = dvshift;
    a.vshift = dvshift + ilog2_exact(tt);
    a.nvec = (uint32_t)(R * tt * (d->D / 8));
    // one-wave workgroups: tile of 64 * kNVMax vectors, every row run >= 64 vectors (ROWU), 16-byte
    // aligned row runs in the store (LDS_OUT)
    const int64_t sblk = KVQ_AB && tunables().quant_block == 128 ? 128 : 64;
    const int64_t snv = !KVQ_AB ? kNVMax : (sblk == 64 && tunables().quant_nv == 4) ? 4 : (sblk == 64 && tunables().quant_nv == 16) ? 16 : kNVMax;
    const int64_t tile64 = sblk * snv * 8;
    const int64_t dq16 = (int64_t)d->D * BITS / 8;
    // (fp32 input — the reference's CPU dtype, never what a GPU model holds — keeps to the 256-thread kernels in the
    // default library)
    if ((KVQ_AB || in_dtype != KVQ_F32) && (tunables().quant_block == 64 || tunables().quant_block == 128) && R * d->D <= tile64 && dq16 % 16 == 0 &&
        a.qs.g % 16 == 0 && a.qs.h % 16 == 0 && a.qs.t % 16 == 0 && aligned(q, 16)) {
      uint32_t t64 = pow2_floor((uint64_t)(tile64 / (R * d->D)));
      if (t64 > kMaxTT) t64 = kMaxTT;
      const bool row_ok = snv == 4 ? (int64_t)t64 * (d->D / 8) <= 64  // register abs-max handles short row runs
                                   : (int64_t)t64 * (d->D / 8) >= sblk;
      if (row_ok && (int64_t)t64 <= d->T) {
        a.blk = (int32_t)sblk;
        a.nv = (int32_t)snv;
        a.TT = t64;
        a.vshift = dvshift + ilog2_exact(t64);
        a.nvec = (uint32_t)(R * t64 * (d->D / 8));
      }
    }
  } else {
    a.TT = 1;
    a.dvshift = a.vshift = 0;
    a.nvec = 0;
    a.rpc = 0;
    if (!absmax_ws) {
      set_error("%s: absmax_ws workspace required for this shape (two-pass path)", name);
      return KVQ_E_NULL;
    }
  }

  for (int64_t g0 = 0; g0 < d->G; g0 += kPtrsPerLaunch) {
    const int64_t gn = d->G - g0 < kPtrsPerLaunch ? d->G - g0 : kPtrsPerLaunch;
    bool fused_here = fused;
    for (int64_t i = 0; i < gn; ++i) {
      const void* p = in_ptrs ? in_ptrs[g0 + i]
                              : static_cast<const char*>(in_base) + (g0 + i) * a.is.g * (int64_t)esz;
      if (!p) {
        set_error("%s: in_ptrs[%lld] is NULL", name, (long long)(g0 + i));
        return KVQ_E_NULL;
      }
      if (!aligned(p, 16)) fused_here = false;
      a.in.p[i] = p;
    }
    if (!fused_here && !absmax_ws) {
      set_error("%s: absmax_ws workspace required (unaligned input falls back to two-pass)", name);
      return KVQ_E_NULL;
    }
    a.G = (uint32_t)gn;
    a.q = q + g0 * a.qs.g;
    a.scales = scales + g0 * ssg;
    a.absmax_ws = absmax_ws ? absmax_ws + g0 * d->T : nullptr;
    QuantArgs b = a;
    if (!fused_here && fused) {  // re-derive generic fields
      b.TT = 1;
      b.dvshift = b.vshift = 0;
      b.nvec = 0;
      b.rpc = 0;
      b.dv = b.vpr = 0;
    }
    switch (in_dtype) {
      case KVQ_F16: launch_quant<KVQ_F16, BITS>(b, fused_here, st); break;
      case KVQ_BF16: launch_quant<KVQ_BF16, BITS>(b, fused_here, st); break;
      case KVQ_F32: launch_quant<KVQ_F32, BITS>(b, fused_here, st); break;
    }
    const int rc = check_launch(name);
    if (rc) return rc;
  }
  (void)Dq;
  return 0;
}

// shared argument checks + QuantArgs of the split-phase entry points; phase 1: q == nullptr
template <int PHASE>
static int split_phase(const char* name, int bits, const void* in_base, const void* const* in_ptrs, const kvq_strides_t* in_st,
                       int in_dtype, uint8_t* q, const kvq_strides_t* q_st, float* scales, int64_t ssg, float* absmax,
                       float eps, const kvq_dims_t* d, void* stream, bool accumulate) {
  if (!in_st || !d || !absmax || (!in_base && !in_ptrs) || (PHASE == 2 && (!q || !q_st || !scales))) {
    set_error("%s: NULL argument", name);
    return KVQ_E_NULL;
  }
  if (const int rcd = check_device(absmax, name)) return rcd;
  if (in_base && in_ptrs) {
    set_error("%s: pass in_base or in_ptrs, not both", name);
    return KVQ_E_DIMS;
  }
  if (d->G < 0 || d->B < 0 || d->H < 0 || d->T < 0 || d->D < 0 || d->B >= (int64_t(1) << 31) || d->H >= (int64_t(1) << 31) ||
      d->T >= (int64_t(1) << 31) || d->D >= (int64_t(1) << 31) || d->B * d->H >= (int64_t(1) << 31)) {
    set_error("%s: bad dims", name);
    return KVQ_E_DIMS;
  }
  if (in_dtype != KVQ_F16 && in_dtype != KVQ_BF16 && in_dtype != KVQ_F32) {
    set_error("%s: unknown in_dtype %d", name, in_dtype);
    return KVQ_E_DTYPE;
  }
  if (PHASE == 2 && bits != 8 && bits != 4) {
    set_error("%s: bits must be 8 or 4", name);
    return KVQ_E_DIMS;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->G * d->B * d->H * d->T * d->D == 0) return 0;
  const int esz = in_dtype == KVQ_F32 ? 4 : 2;
  QuantArgs a = {};
  a.acc = accumulate ? 1 : 0;
  a.is = to_strides(in_st);
  if (PHASE == 2) a.qs = to_strides(q_st);
  a.ssg = ssg;
  a.eps = eps;
  a.B = (uint32_t)d->B; a.H = (uint32_t)d->H; a.T = (uint32_t)d->T; a.D = (uint32_t)d->D;
  a.R = (uint32_t)(d->B * d->H);
  const int64_t qvec = bits == 8 ? 8 : 4;
  const int dvshift = d->D % 8 == 0 ? ilog2_exact(d->D / 8) : -1;
  bool vec = dvshift >= 0 && dvshift <= 6 && (d->T == 1 || a.is.t == d->D) && (a.is.b * esz) % 16 == 0 &&
             (a.is.h * esz) % 16 == 0 && (a.is.t * esz) % 16 == 0;
  if (PHASE == 2)
    vec = vec && a.qs.g % qvec == 0 && a.qs.b % qvec == 0 && a.qs.h % qvec == 0 && a.qs.t % qvec == 0 && aligned(q, qvec);
  // ---- one-wave tile kernels over 8-row groups (quant_tile_k PHASE 1 / 2): what a batch-sharded prefill chunk takes ----
  const int64_t Rt = d->B * d->H;
  const bool tile = split_tile_ok(PHASE == 2 ? bits : 0, in_base, in_ptrs, in_st, in_dtype, q, q_st, d);
  if (tile) {
    const int tt = d->D == 128 ? 4 : 8;
    QuantTileArgs ta;
    ta.qs_g = PHASE == 2 ? a.qs.g : 0;
    ta.ssg = ssg;
    ta.is_h = (uint32_t)(a.is.h * 2);
    ta.qs_h = PHASE == 2 ? (uint32_t)a.qs.h : 0u;
    ta.T = (uint32_t)d->T;
    ta.rows = (uint32_t)Rt;
    ta.eps = eps;
    const bool few = PHASE == 1 && d->T <= tunables().quant_few_tokens && Rt > 8;  // one workgroup per (group, token), no atomics
    if (PHASE == 1 && !few && !accumulate && hipMemsetAsync(absmax, 0, sizeof(float) * (size_t)(d->G * d->T), st) != hipSuccess) return check_launch(name);
    for (int64_t g0 = 0; g0 < d->G; g0 += kPtrsPerLaunch) {
      const int64_t gn = d->G - g0 < kPtrsPerLaunch ? d->G - g0 : kPtrsPerLaunch;
      for (int64_t i = 0; i < gn; ++i)
        ta.in.p[i] = in_ptrs ? in_ptrs[g0 + i] : static_cast<const char*>(in_base) + (g0 + i) * a.is.g * (int64_t)esz;
      ta.absmax = absmax + g0 * d->T;
      ta.q = PHASE == 2 ? q + g0 * a.qs.g : nullptr;
      ta.scales = PHASE == 2 ? scales + g0 * ssg : nullptr;
      if constexpr (PHASE == 1) {
        if (few) {
          const dim3 fgrid((unsigned)d->T, (unsigned)gn);
          if (d->D == 128) {
            if (in_dtype == KVQ_F16) KVQ_LAUNCH((absmax_fewtokens_k<KVQ_F16, 16>), fgrid, dim3(kBlock), 0, st, ta, accumulate ? 1 : 0);
            else KVQ_LAUNCH((absmax_fewtokens_k<KVQ_BF16, 16>), fgrid, dim3(kBlock), 0, st, ta, accumulate ? 1 : 0);
          } else {
            if (in_dtype == KVQ_F16) KVQ_LAUNCH((absmax_fewtokens_k<KVQ_F16, 8>), fgrid, dim3(kBlock), 0, st, ta, accumulate ? 1 : 0);
            else KVQ_LAUNCH((absmax_fewtokens_k<KVQ_BF16, 8>), fgrid, dim3(kBlock), 0, st, ta, accumulate ? 1 : 0);
          }
          const int rcf = check_launch(name);
          if (rcf) return rcf;
          continue;
        }
      }
      const dim3 grid((unsigned)((d->T + tt - 1) / tt), (unsigned)gn, (unsigned)((Rt + 7) / 8));
#define KVQ_PH(IDT_, BITS_, DV_, TT_) KVQ_LAUNCH((quant_tile_k<IDT_, BITS_, 8, DV_, TT_, PHASE>), grid, dim3(kWave), 0, st, ta)
#if KVQ_AB
#define KVQ_PH_SHAPE(IDT_, BITS_) do { if (d->D == 128) KVQ_PH(IDT_, BITS_, 16, 4); else KVQ_PH(IDT_, BITS_, 8, 8); } while (0)
#else
#define KVQ_PH_SHAPE(IDT_, BITS_) KVQ_PH(IDT_, BITS_, 16, 4)
#endif
      if (PHASE == 1 || bits == 8) {  // the abs-max phase never looks at BITS: one instantiation (8)
        if (in_dtype == KVQ_F16) KVQ_PH_SHAPE(KVQ_F16, 8);
        else KVQ_PH_SHAPE(KVQ_BF16, 8);
      } else {
        if constexpr (PHASE == 2) {
          if (in_dtype == KVQ_F16) KVQ_PH_SHAPE(KVQ_F16, 4);
          else KVQ_PH_SHAPE(KVQ_BF16, 4);
        }
      }
#undef KVQ_PH_SHAPE
#undef KVQ_PH
      const int rc = check_launch(name);
      if (rc) return rc;
    }
    return 0;
  }
  if (vec) {  // tile = TT tokens (power of two) x D/8 vectors <= 256 vectors per row run
    uint32_t tt = pow2_floor((uint64_t)(kBlock >> dvshift));
    while (tt > 1 && tt / 2 >= (uint64_t)d->T) tt /= 2;
    a.TT = tt;
    a.dvshift = dvshift;
    a.vshift = dvshift + ilog2_exact(tt);
  }
  for (int64_t g0 = 0; g0 < d->G; g0 += kPtrsPerLaunch) {
    const int64_t gn = d->G - g0 < kPtrsPerLaunch ? d->G - g0 : kPtrsPerLaunch;
    bool vec_here = vec;
    for (int64_t i = 0; i < gn; ++i) {
      const void* p = in_ptrs ? in_ptrs[g0 + i] : static_cast<const char*>(in_base) + (g0 + i) * a.is.g * (int64_t)esz;
      if (!p) {
        set_error("%s: in_ptrs[%lld] is NULL", name, (long long)(g0 + i));
        return KVQ_E_NULL;
      }
      if (!aligned(p, 16)) vec_here = false;
      a.in.p[i] = p;
    }
    a.G = (uint32_t)gn;
    a.absmax_ws = absmax + g0 * d->T;
    if (PHASE == 2) {
      a.q = q + g0 * a.qs.g;
      a.scales = scales + g0 * ssg;
    }
    const dim3 grid(vec_here ? (unsigned)((d->T + a.TT - 1) / a.TT) : 1u, (unsigned)gn);
#define KVQ_BY_DTYPE(EXPR_F16, EXPR_BF16, EXPR_F32) \
  switch (in_dtype) { case KVQ_F16: EXPR_F16; break; case KVQ_BF16: EXPR_BF16; break; default: EXPR_F32; break; }
    if (PHASE == 1) {
      if (vec_here) {
        KVQ_BY_DTYPE(KVQ_LAUNCH((absmax_tokens_tile_k<KVQ_F16>), grid, dim3(kBlock), 0, st, a),
                     KVQ_LAUNCH((absmax_tokens_tile_k<KVQ_BF16>), grid, dim3(kBlock), 0, st, a),
                     KVQ_LAUNCH((absmax_tokens_tile_k<KVQ_F32>), grid, dim3(kBlock), 0, st, a))
      } else {
        if (!accumulate && hipMemsetAsync(a.absmax_ws, 0, sizeof(float) * (size_t)(gn * d->T), st) != hipSuccess) return check_launch(name);
        const int64_t RD = (int64_t)a.R * a.D, chunk_elems = (int64_t)kBlock * 16;
        const uint32_t cpt = (uint32_t)((RD + chunk_elems - 1) / chunk_elems);
        const dim3 ggrid((unsigned)(a.T * cpt), (unsigned)gn);
        KVQ_BY_DTYPE(KVQ_LAUNCH((absmax_tokens_generic_k<KVQ_F16>), ggrid, dim3(kBlock), 0, st, a, cpt, chunk_elems),
                     KVQ_LAUNCH((absmax_tokens_generic_k<KVQ_BF16>), ggrid, dim3(kBlock), 0, st, a, cpt, chunk_elems),
                     KVQ_LAUNCH((absmax_tokens_generic_k<KVQ_F32>), ggrid, dim3(kBlock), 0, st, a, cpt, chunk_elems))
      }
    } else if (vec_here) {
      if (bits == 8) {
        KVQ_BY_DTYPE(KVQ_LAUNCH((quant_tokens_scaled_tile_k<KVQ_F16, 8>), grid, dim3(kBlock), 0, st, a),
                     KVQ_LAUNCH((quant_tokens_scaled_tile_k<KVQ_BF16, 8>), grid, dim3(kBlock), 0, st, a),
                     KVQ_LAUNCH((quant_tokens_scaled_tile_k<KVQ_F32, 8>), grid, dim3(kBlock), 0, st, a))
      } else {
        KVQ_BY_DTYPE(KVQ_LAUNCH((quant_tokens_scaled_tile_k<KVQ_F16, 4>), grid, dim3(kBlock), 0, st, a),
                     KVQ_LAUNCH((quant_tokens_scaled_tile_k<KVQ_BF16, 4>), grid, dim3(kBlock), 0, st, a),
                     KVQ_LAUNCH((quant_tokens_scaled_tile_k<KVQ_F32, 4>), grid, dim3(kBlock), 0, st, a))
      }
    } else {
      const int64_t Dq = bits == 8 ? d->D : (d->D + 1) / 2;
      const int64_t total = gn * d->B * d->H * d->T * Dq;
      int64_t gb = (total + kBlock - 1) / kBlock;
      if (gb > 256 * 32) gb = 256 * 32;
      if (bits == 8) {
        KVQ_BY_DTYPE(KVQ_LAUNCH((quant_tokens_generic_k<KVQ_F16, 8>), dim3((unsigned)gb), dim3(kBlock), 0, st, a, total),
                     KVQ_LAUNCH((quant_tokens_generic_k<KVQ_BF16, 8>), dim3((unsigned)gb), dim3(kBlock), 0, st, a, total),
                     KVQ_LAUNCH((quant_tokens_generic_k<KVQ_F32, 8>), dim3((unsigned)gb), dim3(kBlock), 0, st, a, total))
      } else {
        KVQ_BY_DTYPE(KVQ_LAUNCH((quant_tokens_generic_k<KVQ_F16, 4>), dim3((unsigned)gb), dim3(kBlock), 0, st, a, total),
                     KVQ_LAUNCH((quant_tokens_generic_k<KVQ_BF16, 4>), dim3((unsigned)gb), dim3(kBlock), 0, st, a, total),
                     KVQ_LAUNCH((quant_tokens_generic_k<KVQ_F32, 4>), dim3((unsigned)gb), dim3(kBlock), 0, st, a, total))
      }
    }
#undef KVQ_BY_DTYPE
    const int rc = check_launch(name);
    if (rc) return rc;
  }
  return 0;
}

}  // namespace kvq

using namespace kvq;

extern "C" {

int kvq_quant_i8_tokens(const void* in_base, const void* const* in_ptrs, const kvq_strides_t* in_st, int in_dtype,
                        int8_t* q, const kvq_strides_t* q_st, float* scales, int64_t scale_stride_g,
                        float* absmax_ws, float eps, const kvq_dims_t* dims, void* stream) {
  return quant_tokens<8>(in_base, in_ptrs, in_st, in_dtype, reinterpret_cast<uint8_t*>(q), q_st, scales,
                         scale_stride_g, absmax_ws, eps, dims, stream, "kvq_quant_i8_tokens");
}

int kvq_quant_i4_tokens(const void* in_base, const void* const* in_ptrs, const kvq_strides_t* in_st, int in_dtype,
                        uint8_t* packed, const kvq_strides_t* p_st, float* scales, int64_t scale_stride_g,
                        float* absmax_ws, float eps, const kvq_dims_t* dims, void* stream) {
  return quant_tokens<4>(in_base, in_ptrs, in_st, in_dtype, packed, p_st, scales, scale_stride_g, absmax_ws, eps,
                         dims, stream, "kvq_quant_i4_tokens");
}

int kvq_absmax_tokens(const void* in_base, const void* const* in_ptrs, const kvq_strides_t* in_st, int in_dtype,
                      float* absmax, const kvq_dims_t* dims, void* stream) {
  return split_phase<1>("kvq_absmax_tokens", 8, in_base, in_ptrs, in_st, in_dtype, nullptr, nullptr, nullptr, 0, absmax, 0.0f,
                        dims, stream);
}

int kvq_absmax_tokens_acc(const void* in_base, const void* const* in_ptrs, const kvq_strides_t* in_st, int in_dtype,
                          float* absmax, const kvq_dims_t* dims, void* stream) {
  return split_phase<1>("kvq_absmax_tokens_acc", 8, in_base, in_ptrs, in_st, in_dtype, nullptr, nullptr, nullptr, 0, absmax, 0.0f,
                        dims, stream, true);
}

int kvq_quant_tokens_from_absmax(int bits, const void* in_base, const void* const* in_ptrs, const kvq_strides_t* in_st,
                                 int in_dtype, uint8_t* q, const kvq_strides_t* q_st, float* scales,
                                 int64_t scale_stride_g, const float* absmax, float eps, const kvq_dims_t* dims,
                                 void* stream) {
  return split_phase<2>("kvq_quant_tokens_from_absmax", bits, in_base, in_ptrs, in_st, in_dtype, q, q_st, scales, scale_stride_g,
                        const_cast<float*>(absmax), eps, dims, stream);
}

}  // extern "C"
