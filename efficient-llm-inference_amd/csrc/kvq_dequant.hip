// kvq_dequant.hip — INT8 / packed-INT4 -> fp16|bf16|fp32 dequantise kernels for gfx950.
//
// Replaces the reference's two CUDA kernels (src/cuda/extensions.py:37-68) and the per-slice
// Python loop + T-way torch.cat that drives them (src/quantization/ops.py:213-269).
//
// Roofline: pure HBM stream, no reuse, no MFMA. Algorithmic bytes per element: INT8 1 in + 2 out,
// INT4 0.5 in + 2 out (fp16 out). The kernel is 80 % stores, so the design goal is that every
// wave-level store instruction writes whole, contiguous 128-byte lines (16 B per lane).
//
// Arithmetic (bit-exact with the reference): out = RN_out( float(q) * scale_f32 ) with
// q = int8, or (nibble - 8) where the EVEN element of a byte pair is the HIGH nibble
// (ops.py:61-63, extensions.py:61). The product is formed as (float)(int) * s — not as an
// fma against -8*s — so that the sign of a zero result matches (q<0, stored fp16 scale 0).
#include "kvq_common.h"

namespace kvq {

struct DequantArgs {
  const uint8_t* q;     // int8 or packed nibbles
  Strides qs;           // strides in q bytes
  const float* scales;  // [G, >=T] stored scales widened to fp32
  int64_t ssg;          // scale_stride_g
  void* out;
  Strides os;  // strides in out elements
  uint32_t BH, H;
  uint32_t T, D;     // D = unpacked head_dim
  uint32_t row_len;  // T*D
  int32_t dshift;    // log2(D) or -1
  uint32_t cpr;      // chunks per (g,b,h) row
  uint32_t total_items;
  uint32_t xcd_group;  // consecutive chunks per XCD (0 / 1 = round robin); only with one chunk per workgroup
};

template <int NW, bool NTL = false>
__device__ inline void load_words(const uint8_t* p, uint32_t (&w)[NW]) {
  if constexpr (NW == 1) {
    w[0] = NTL ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(p)) : *reinterpret_cast<const uint32_t*>(p);
  } else if constexpr (NW == 2) {
    const u32x2 v = NTL ? __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p)) : *reinterpret_cast<const u32x2*>(p);
    w[0] = v[0];
    w[1] = v[1];
  } else {
#pragma unroll
    for (int i = 0; i < NW / 4; ++i) {
      const u32x4 v = *(reinterpret_cast<const u32x4*>(p) + i);
      w[4 * i] = v[0];
      w[4 * i + 1] = v[1];
      w[4 * i + 2] = v[2];
      w[4 * i + 3] = v[3];
    }
  }
}

// 8 consecutive elements from the quantised words of one lane. `wq` points at the first word
// of this 8-element group (INT8: 2 words, INT4: 1 word).
template <int BITS>
__device__ inline void expand8(const uint32_t* wq, float s, float (&x)[8]) {
  if constexpr (BITS == 8) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int v = (int)(int8_t)((wq[i >> 2] >> (8 * (i & 3))) & 0xFFu);
      x[i] = mul_exact((float)v, s);
    }
  } else {
    const uint32_t w = wq[0];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int sh = 8 * (i >> 1) + ((i & 1) ? 0 : 4);  // even element = high nibble
      const int v = (int)((w >> sh) & 0xFu) - 8;
      x[i] = mul_exact((float)v, s);
    }
  }
}

// ---------------------------------------------------------------------------- fast path
// One work item = one CHUNK of a (g,b,h) row; T*D elements of a row are contiguous on both
// sides (q.t == Dq, out.t == D), D % 8 == 0. Lane l of step u owns LE consecutive elements:
// LE = 8 makes every store instruction of a wave one contiguous 1 KiB run.
// Per-token scales of the chunk are staged through LDS once (CHUNK/D tokens), then broadcast.
// BLK = 64 (shipped): one wave per workgroup, 2048-element chunk = 4 KiB of output — the
// granularity at which plain copies are fastest on this chip; the LDS barrier is then free.
template <int ODT, int BITS, int LE, int UNROLL, bool NT, bool LDS_SC, bool NTL = false, int BLK = kBlock>
__global__ __launch_bounds__(BLK) void dequant_tokens_fast_k(const DequantArgs a) {
  constexpr int CHUNK = BLK * LE * UNROLL;
  constexpr int NW = LE * BITS / 32;
  constexpr int MAXTOK = CHUNK / 8 + 2;
  __shared__ float s_scale[LDS_SC ? MAXTOK : 1];
  const uint32_t tid = threadIdx.x;
  const uint32_t D = a.D;

  for (uint32_t wg = blockIdx.x; wg < a.total_items; wg += gridDim.x) {
    const uint32_t item = xcd_grouped_item(wg, a.xcd_group, a.total_items);
    const uint32_t row = item / a.cpr;
    const uint32_t chunk = item - row * a.cpr;
    const uint32_t g = row / a.BH;
    const uint32_t bh = row - g * a.BH;
    const uint32_t b = bh / a.H;
    const uint32_t h = bh - b * a.H;
    const uint32_t e0 = chunk * (uint32_t)CHUNK;
    const uint32_t tok0 = a.dshift >= 0 ? (e0 >> a.dshift) : (e0 / D);
    const uint32_t rem0 = e0 - tok0 * D;

    const uint8_t* qrow = a.q + (int64_t)g * a.qs.g + (int64_t)b * a.qs.b + (int64_t)h * a.qs.h;
    char* orow = reinterpret_cast<char*>(a.out) +
                 ((int64_t)g * a.os.g + (int64_t)b * a.os.b + (int64_t)h * a.os.h) * Elem<ODT>::size;
    const float* srow = a.scales + (int64_t)g * a.ssg;

    // 1. issue all quantised loads of this item (independent, UNROLL in flight per lane)
    uint32_t w[UNROLL][NW];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint32_t e = e0 + (u * BLK + tid) * LE;
      if (e < a.row_len) {
        load_words<NW, NTL>(qrow + ((int64_t)e * BITS) / 8, w[u]);
      } else {
#pragma unroll
        for (int i = 0; i < NW; ++i) w[u][i] = 0;
      }
    }

    // 2. stage this chunk's per-token scales in LDS
    if constexpr (LDS_SC) {
      uint32_t ntok = (rem0 + CHUNK - 1) / D + 1;
      if (ntok > a.T - tok0) ntok = a.T - tok0;
      for (uint32_t i = tid; i < ntok; i += BLK) s_scale[i] = srow[tok0 + i];
      __syncthreads();
    }

    // 3. expand, scale, round, store (16 B per lane per store)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint32_t off = (u * BLK + tid) * LE;
      const uint32_t e = e0 + off;
      if (e < a.row_len) {
#pragma unroll
        for (int k = 0; k < LE / 8; ++k) {
          const uint32_t r = rem0 + off + k * 8;
          const uint32_t tl = a.dshift >= 0 ? (r >> a.dshift) : (r / D);
          float s;
          if constexpr (LDS_SC) s = s_scale[tl];
          else s = srow[tok0 + tl];
          float x[8];
          expand8<BITS>(&w[u][k * BITS / 4], s, x);
          store8<ODT, NT>(orow + (int64_t)(e + k * 8) * Elem<ODT>::size, x);
        }
      }
    }
    if constexpr (LDS_SC) __syncthreads();
  }
}

// ---------------------------------------------------------------------------- generic path
// Any D (odd included), any strides, any alignment: one thread per output element.
template <int ODT, int BITS>
__global__ __launch_bounds__(kBlock) void dequant_tokens_generic_k(const DequantArgs a, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int64_t d = i % a.D;
    int64_t r = i / a.D;
    const int64_t t = r % a.T;
    r /= a.T;
    const int64_t h = r % a.H;
    r /= a.H;
    const int64_t B = a.BH / a.H;
    const int64_t b = r % B;
    const int64_t g = r / B;
    const float s = a.scales[g * a.ssg + t];
    const uint8_t* qp = a.q + g * a.qs.g + b * a.qs.b + h * a.qs.h + t * a.qs.t;
    int v;
    if constexpr (BITS == 8) {
      v = (int)(int8_t)qp[d];
    } else {
      const uint8_t byte = qp[d >> 1];
      v = (int)((d & 1) ? (byte & 0x0F) : (byte >> 4)) - 8;
    }
    store1<ODT>(a.out, g * a.os.g + b * a.os.b + h * a.os.h + t * a.os.t + d, mul_exact((float)v, s));
  }
}

// ---------------------------------------------------------------------------- flat kernels
// The reference's own entry points: one scalar scale per launch, contiguous buffers.
// (extensions.py:37-48 and :50-68; fixes its 32-bit index overflow for n >= 2^31.)
template <int BITS>
__global__ __launch_bounds__(kBlock) void dequant_flat_vec_k(const uint8_t* __restrict__ q, float s,
                                                             uint16_t* __restrict__ out, int64_t n_groups8) {
  // n_groups8 groups of 8 output elements, everything 16-byte aligned
  for (int64_t gi = (int64_t)blockIdx.x * kBlock + threadIdx.x; gi < n_groups8; gi += (int64_t)gridDim.x * kBlock) {
    uint32_t w[BITS / 4];
    load_words<BITS / 4>(q + gi * BITS, w);
    float x[8];
    expand8<BITS>(w, s, x);
    store8<KVQ_F16, false>(out + gi * 8, x);
  }
}

// The same entry points for LARGE buffers: the token-table kernel's launch shape (one-wave workgroups, one contiguous
// chunk each — 4 KiB of output for INT4, 2 KiB for INT8 —, all loads in flight, non-temporal stores) with the one scalar
// scale. The grid-stride kernel above stays for small calls (the reference's per-slice use: 768-1024 elements).
#ifndef KVQ_FLAT_CHUNK_MIN
#define KVQ_FLAT_CHUNK_MIN (1 << 17)
#endif
constexpr int64_t kFlatChunkMin = KVQ_FLAT_CHUNK_MIN;  // groups of 8 elements (1 Mi elements) from which the chunk kernel is used
template <int BITS>
__global__ __launch_bounds__(64) void dequant_flat_chunk_k(const uint8_t* __restrict__ q, float s, uint16_t* __restrict__ out,
                                                           int64_t n_groups8) {
  constexpr int UNROLL = BITS == 4 ? 4 : 2;
  const int64_t g0 = (int64_t)blockIdx.x * (64 * UNROLL) + threadIdx.x;
  uint32_t w[UNROLL][BITS / 4];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) {
    const int64_t gi = g0 + u * 64;
    if (gi < n_groups8) load_words<BITS / 4, BITS == 8>(q + gi * BITS, w[u]);
  }
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) {
    const int64_t gi = g0 + u * 64;
    if (gi < n_groups8) {
      float x[8];
      expand8<BITS>(w[u], s, x);
      store8<KVQ_F16, true>(out + gi * 8, x);
    }
  }
}

__global__ __launch_bounds__(kBlock) void dequant_i8_flat_scalar_k(const int8_t* __restrict__ q, float s,
                                                                   uint16_t* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    store1<KVQ_F16>(out, i, mul_exact((float)(int)q[i], s));
}

__global__ __launch_bounds__(kBlock) void dequant_i4_flat_scalar_k(const uint8_t* __restrict__ p, float s,
                                                                   uint16_t* __restrict__ out, int64_t out_n,
                                                                   int64_t orig_last, int64_t total_last) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < out_n; i += (int64_t)gridDim.x * kBlock) {
    const uint8_t byte = p[i >> 1];
    const int v = (int)((i & 1) ? (byte & 0x0F) : (byte >> 4)) - 8;
    const int64_t last = i % total_last;
    store1<KVQ_F16>(out, i, last < orig_last ? mul_exact((float)v, s) : 0.0f);
  }
}

// ---------------------------------------------------------------------------- host side

static inline unsigned grid_for(int64_t work_blocks, int64_t cap) {
  if (work_blocks < 1) work_blocks = 1;
  return (unsigned)(work_blocks < cap ? work_blocks : cap);
}

struct Variant {
  int le, unroll;
  bool nt, lds;
  int blk = kBlock;
};
// Index = tunable "dequant_variant" (A-B builds). Shipped (profiles/r01_microbench.txt, r02ag_*): one-wave workgroups,
// contiguous 1 KiB stores per wave instruction, non-temporal stores; INT4 4 vectors per lane (4 KiB of output per
// wave), INT8 2 vectors per lane + non-temporal loads.
constexpr int kDefaultVariantI4 = 21;
constexpr int kDefaultVariantI8 = 23;
#if KVQ_AB
static const Variant kVariants[] = {
    {8, 4, false, true},   // 0
    {8, 4, true, true},    // 1
    {16, 2, false, true},  // 2
    {16, 2, true, true},   // 3
    {32, 1, false, true},  // 4
    {32, 1, true, true},   // 5
    {8, 8, false, true},   // 6
    {8, 4, false, false},  // 7
    {8, 2, false, true},   // 8
    {8, 8, true, true},    // 9
    {16, 4, false, true},  // 10
    {32, 2, false, true},  // 11
    {8, 2, true, true},    // 12
    {8, 1, true, true},    // 13
    {8, 4, true, false},   // 14
    {8, 1, false, true},   // 15
    {8, 4, true, true},    // 16: as 1 + non-temporal loads
    {8, 2, true, true},    // 17: as 12 + non-temporal loads
    {8, 1, true, false},   // 18: one vector per thread, scales straight from global (no LDS, no barrier)
    {8, 2, true, false},   // 19
    {8, 1, true, false},   // 20: 18 + non-temporal loads
    {8, 4, true, true, 64},   // 21: as 1, one-wave workgroups (INT4 default)
    {8, 8, true, true, 64},   // 22
    {8, 2, true, true, 64},   // 23: as 17 (NT loads), one-wave workgroups (INT8 default)
    {8, 4, true, true, 64},   // 24: as 21 + NT loads
    {8, 16, true, true, 64},  // 25
    {8, 2, true, true, 64},   // 26
    {8, 1, true, true, 64},   // 27
    {8, 1, true, true, 64},   // 28: 27 + NT loads
    {8, 4, true, true, 128},  // 29
    {8, 2, true, true, 128},  // 30: NT loads
    {8, 4, false, true, 64},  // 31: as 21 with write-back (not non-temporal) stores
    {8, 4, false, true, 64},  // 32: 31 + NT loads
    {8, 2, false, true, 64},  // 33: as 23 with write-back stores (NT loads)
    {8, 8, false, true, 64},  // 34: 8 KiB of output per wave, write-back stores, NT loads
    {8, 2, false, true, 64},  // 35: 33 without NT loads
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
static Variant variant_of(int v) { return kVariants[v]; }
#else
constexpr int kNumVariants = 0;  // "dequant_variant" is an A-B key: the default library holds the two shipped shapes
static Variant variant_of(int v) { return v == kDefaultVariantI4 ? Variant{8, 4, true, true, 64} : Variant{8, 2, true, true, 64}; }
#endif

template <int ODT, int BITS, int LE, int UNROLL, bool NT, bool LDS_SC, bool NTL = false, int BLK = kBlock>
static void launch_fast(const DequantArgs& a, unsigned grid, hipStream_t st) {
  KVQ_LAUNCH((dequant_tokens_fast_k<ODT, BITS, LE, UNROLL, NT, LDS_SC, NTL, BLK>), dim3(grid), dim3(BLK), 0, st, a);
}

template <int ODT, int BITS>
static bool launch_fast_variant(int v, const DequantArgs& a, unsigned grid, hipStream_t st) {
  switch (v) {
    case 21: launch_fast<ODT, BITS, 8, 4, true, true, false, 64>(a, grid, st); return true;
    case 23: launch_fast<ODT, BITS, 8, 2, true, true, true, 64>(a, grid, st); return true;
#if KVQ_AB
    case 0: launch_fast<ODT, BITS, 8, 4, false, true>(a, grid, st); return true;
    case 1: launch_fast<ODT, BITS, 8, 4, true, true>(a, grid, st); return true;
    case 2: launch_fast<ODT, BITS, 16, 2, false, true>(a, grid, st); return true;
    case 3: launch_fast<ODT, BITS, 16, 2, true, true>(a, grid, st); return true;
    case 4: launch_fast<ODT, BITS, 32, 1, false, true>(a, grid, st); return true;
    case 5: launch_fast<ODT, BITS, 32, 1, true, true>(a, grid, st); return true;
    case 6: launch_fast<ODT, BITS, 8, 8, false, true>(a, grid, st); return true;
    case 7: launch_fast<ODT, BITS, 8, 4, false, false>(a, grid, st); return true;
    case 8: launch_fast<ODT, BITS, 8, 2, false, true>(a, grid, st); return true;
    case 9: launch_fast<ODT, BITS, 8, 8, true, true>(a, grid, st); return true;
    case 10: launch_fast<ODT, BITS, 16, 4, false, true>(a, grid, st); return true;
    case 11: launch_fast<ODT, BITS, 32, 2, false, true>(a, grid, st); return true;
    case 12: launch_fast<ODT, BITS, 8, 2, true, true>(a, grid, st); return true;
    case 13: launch_fast<ODT, BITS, 8, 1, true, true>(a, grid, st); return true;
    case 14: launch_fast<ODT, BITS, 8, 4, true, false>(a, grid, st); return true;
    case 15: launch_fast<ODT, BITS, 8, 1, false, true>(a, grid, st); return true;
    case 16: launch_fast<ODT, BITS, 8, 4, true, true, true>(a, grid, st); return true;
    case 17: launch_fast<ODT, BITS, 8, 2, true, true, true>(a, grid, st); return true;
    case 18: launch_fast<ODT, BITS, 8, 1, true, false>(a, grid, st); return true;
    case 19: launch_fast<ODT, BITS, 8, 2, true, false>(a, grid, st); return true;
    case 20: launch_fast<ODT, BITS, 8, 1, true, false, true>(a, grid, st); return true;
    case 22: launch_fast<ODT, BITS, 8, 8, true, true, false, 64>(a, grid, st); return true;
    case 24: launch_fast<ODT, BITS, 8, 4, true, true, true, 64>(a, grid, st); return true;
    case 25: launch_fast<ODT, BITS, 8, 16, true, true, false, 64>(a, grid, st); return true;
    case 26: launch_fast<ODT, BITS, 8, 2, true, true, false, 64>(a, grid, st); return true;
    case 27: launch_fast<ODT, BITS, 8, 1, true, true, false, 64>(a, grid, st); return true;
    case 28: launch_fast<ODT, BITS, 8, 1, true, true, true, 64>(a, grid, st); return true;
    case 29: launch_fast<ODT, BITS, 8, 4, true, true, false, 128>(a, grid, st); return true;
    case 30: launch_fast<ODT, BITS, 8, 2, true, true, true, 128>(a, grid, st); return true;
    case 31: launch_fast<ODT, BITS, 8, 4, false, true, false, 64>(a, grid, st); return true;
    case 32: launch_fast<ODT, BITS, 8, 4, false, true, true, 64>(a, grid, st); return true;
    case 33: launch_fast<ODT, BITS, 8, 2, false, true, true, 64>(a, grid, st); return true;
    case 34: launch_fast<ODT, BITS, 8, 8, false, true, true, 64>(a, grid, st); return true;
    case 35: launch_fast<ODT, BITS, 8, 2, false, true, false, 64>(a, grid, st); return true;
#endif
  }
  return false;
}

template <int BITS>
static int dequant_tokens(const uint8_t* q, const kvq_strides_t* q_st, const float* scales, int64_t ssg,
                          void* out, const kvq_strides_t* out_st, int out_dtype, const kvq_dims_t* d,
                          void* stream, const char* name) {
  if (!q || !q_st || !scales || !out || !out_st || !d) {
    set_error("%s: NULL argument", name);
    return KVQ_E_NULL;
  }
  if (const int rcd = check_device(out, name)) return rcd;
  if (d->G < 0 || d->B < 0 || d->H < 0 || d->T < 0 || d->D < 0) {
    set_error("%s: negative dim", name);
    return KVQ_E_DIMS;
  }
  if (out_dtype != KVQ_F16 && out_dtype != KVQ_BF16 && out_dtype != KVQ_F32) {
    set_error("%s: unknown out_dtype %d", name, out_dtype);
    return KVQ_E_DTYPE;
  }
  const int64_t rows = d->G * d->B * d->H;
  const int64_t total = rows * d->T * d->D;
  if (total == 0) return 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);

  DequantArgs a;
  a.q = q;
  a.qs = to_strides(q_st);
  a.scales = scales;
  a.ssg = ssg;
  a.out = out;
  a.os = to_strides(out_st);
  const int64_t Dq = BITS == 8 ? d->D : (d->D + 1) / 2;
  const int64_t row_len = d->T * d->D;
  const int esz = out_dtype == KVQ_F32 ? 4 : 2;

  int v = (int)tunables().dequant_variant;
  if (v < 0 || v >= kNumVariants) v = BITS == 4 ? kDefaultVariantI4 : kDefaultVariantI8;
  const Variant var = variant_of(v);
  const int64_t chunk = (int64_t)var.blk * var.le * var.unroll;

  // fast path: rows contiguous in (t,d) on both sides, 16-byte vectors everywhere, 32-bit indices
  const int64_t qvec = var.le * BITS / 8;  // bytes per lane load
  bool fast = d->D % 8 == 0 && row_len % var.le == 0 && a.qs.t == Dq && a.os.t == d->D &&
              row_len < (int64_t(1) << 31) && rows < (int64_t(1) << 31) && d->D <= (1 << 20) &&
              aligned(q, qvec < 16 ? qvec : 16) && aligned(out, 16) &&
              a.qs.g % qvec == 0 && a.qs.b % qvec == 0 && a.qs.h % qvec == 0 &&
              (a.os.g * esz) % 16 == 0 && (a.os.b * esz) % 16 == 0 && (a.os.h * esz) % 16 == 0;
  const int64_t cpr = (row_len + chunk - 1) / chunk;
  if (fast && rows * cpr >= (int64_t(1) << 32)) fast = false;

  a.BH = (uint32_t)(d->B * d->H);
  a.H = (uint32_t)d->H;
  a.T = (uint32_t)d->T;
  a.D = (uint32_t)d->D;

  if (fast) {
    a.row_len = (uint32_t)row_len;
    a.dshift = ilog2_exact(d->D);
    a.cpr = (uint32_t)cpr;
    a.total_items = (uint32_t)(rows * cpr);
    a.xcd_group = (uint32_t)(tunables().dequant_xcd_group > 1 ? tunables().dequant_xcd_group : 0);
    int64_t cap = tunables().dequant_grid > 0 ? tunables().dequant_grid : (int64_t(1) << 31) - 1;
    const unsigned grid = grid_for(a.total_items, cap);
    bool ok = false;
    switch (out_dtype) {
      case KVQ_F16: ok = launch_fast_variant<KVQ_F16, BITS>(v, a, grid, st); break;
      case KVQ_BF16: ok = launch_fast_variant<KVQ_BF16, BITS>(v, a, grid, st); break;
      case KVQ_F32: ok = launch_fast_variant<KVQ_F32, BITS>(v, a, grid, st); break;
    }
    if (!ok) {
      set_error("%s: no such variant %d", name, v);
      return KVQ_E_DIMS;
    }
    return check_launch(name);
  }

  // generic path (64-bit indexing; T, D, H carried as 32-bit fields must fit)
  if (d->T >= (int64_t(1) << 32) || d->D >= (int64_t(1) << 32) || d->B * d->H >= (int64_t(1) << 32)) {
    set_error("%s: dims too large for the generic path", name);
    return KVQ_E_DIMS;
  }
  a.row_len = 0;
  a.dshift = -1;
  a.cpr = 0;
  a.total_items = 0;
  a.xcd_group = 0;
  const unsigned grid = grid_for((total + kBlock - 1) / kBlock, 256 * 32);
  switch (out_dtype) {
    case KVQ_F16: KVQ_LAUNCH((dequant_tokens_generic_k<KVQ_F16, BITS>), dim3(grid), dim3(kBlock), 0, st, a, total); break;
    case KVQ_BF16: KVQ_LAUNCH((dequant_tokens_generic_k<KVQ_BF16, BITS>), dim3(grid), dim3(kBlock), 0, st, a, total); break;
    case KVQ_F32: KVQ_LAUNCH((dequant_tokens_generic_k<KVQ_F32, BITS>), dim3(grid), dim3(kBlock), 0, st, a, total); break;
  }
  return check_launch(name);
}

}  // namespace kvq

using namespace kvq;

extern "C" {

int kvq_dequant_i8_tokens(const int8_t* q, const kvq_strides_t* q_st, const float* scales,
                          int64_t scale_stride_g, void* out, const kvq_strides_t* out_st, int out_dtype,
                          const kvq_dims_t* dims, void* stream) {
  return dequant_tokens<8>(reinterpret_cast<const uint8_t*>(q), q_st, scales, scale_stride_g, out, out_st,
                           out_dtype, dims, stream, "kvq_dequant_i8_tokens");
}

int kvq_dequant_i4_tokens(const uint8_t* packed, const kvq_strides_t* p_st, const float* scales,
                          int64_t scale_stride_g, void* out, const kvq_strides_t* out_st, int out_dtype,
                          const kvq_dims_t* dims, void* stream) {
  return dequant_tokens<4>(packed, p_st, scales, scale_stride_g, out, out_st, out_dtype, dims, stream,
                           "kvq_dequant_i4_tokens");
}

int kvq_dequant_i8_f16_flat(const int8_t* q, float scale, void* out_f16, int64_t n, void* stream) {
  if (n < 0) {
    set_error("kvq_dequant_i8_f16_flat: n < 0");
    return KVQ_E_DIMS;
  }
  if (n == 0) return 0;
  if (!q || !out_f16) {
    set_error("kvq_dequant_i8_f16_flat: NULL argument");
    return KVQ_E_NULL;
  }
  if (const int rcd = check_device(out_f16, "kvq_dequant_i8_f16_flat")) return rcd;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (n % 8 == 0 && aligned(q, 8) && aligned(out_f16, 16)) {
    const int64_t ng = n / 8;
    if (ng >= kFlatChunkMin && (ng + 127) / 128 < (int64_t(1) << 31))
      KVQ_LAUNCH((dequant_flat_chunk_k<8>), dim3((unsigned)((ng + 127) / 128)), dim3(64), 0, st, reinterpret_cast<const uint8_t*>(q),
                         scale, reinterpret_cast<uint16_t*>(out_f16), ng);
    else
      KVQ_LAUNCH((dequant_flat_vec_k<8>), dim3(grid_for((ng + kBlock - 1) / kBlock, 256 * 16)), dim3(kBlock), 0,
                         st, reinterpret_cast<const uint8_t*>(q), scale, reinterpret_cast<uint16_t*>(out_f16), ng);
  } else {
    KVQ_LAUNCH(dequant_i8_flat_scalar_k, dim3(grid_for((n + kBlock - 1) / kBlock, 256 * 16)), dim3(kBlock), 0,
                       st, q, scale, reinterpret_cast<uint16_t*>(out_f16), n);
  }
  return check_launch("kvq_dequant_i8_f16_flat");
}

int kvq_dequant_i4_f16_flat(const uint8_t* packed, float scale, void* out_f16, int64_t n_packed,
                            int64_t packed_last, int64_t orig_last_dim, void* stream) {
  if (n_packed < 0 || packed_last < 0 || orig_last_dim < 0 || (packed_last > 0 && n_packed % packed_last != 0)) {
    set_error("kvq_dequant_i4_f16_flat: bad sizes n_packed=%lld packed_last=%lld orig_last_dim=%lld",
              (long long)n_packed, (long long)packed_last, (long long)orig_last_dim);
    return KVQ_E_DIMS;
  }
  if (n_packed == 0) return 0;
  if (!packed || !out_f16) {
    set_error("kvq_dequant_i4_f16_flat: NULL argument");
    return KVQ_E_NULL;
  }
  if (const int rcd = check_device(out_f16, "kvq_dequant_i4_f16_flat")) return rcd;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t total_last = packed_last * 2;
  if (orig_last_dim >= total_last && n_packed % 4 == 0 && aligned(packed, 4) && aligned(out_f16, 16)) {
    const int64_t ng = n_packed / 4;  // no pad column to zero: pure vector path
    if (ng >= kFlatChunkMin && (ng + 255) / 256 < (int64_t(1) << 31))
      KVQ_LAUNCH((dequant_flat_chunk_k<4>), dim3((unsigned)((ng + 255) / 256)), dim3(64), 0, st, packed, scale,
                         reinterpret_cast<uint16_t*>(out_f16), ng);
    else
      KVQ_LAUNCH((dequant_flat_vec_k<4>), dim3(grid_for((ng + kBlock - 1) / kBlock, 256 * 16)), dim3(kBlock), 0,
                         st, packed, scale, reinterpret_cast<uint16_t*>(out_f16), ng);
  } else {
    const int64_t out_n = n_packed * 2;
    KVQ_LAUNCH(dequant_i4_flat_scalar_k, dim3(grid_for((out_n + kBlock - 1) / kBlock, 256 * 16)), dim3(kBlock),
                       0, st, packed, scale, reinterpret_cast<uint16_t*>(out_f16), out_n, orig_last_dim, total_last);
  }
  return check_launch("kvq_dequant_i4_f16_flat");
}

}  // extern "C"
