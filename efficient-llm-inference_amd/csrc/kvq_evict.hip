// kvq_evict.hip — KV eviction kernels for gfx950: sliding-window compaction and chunk-summary
// mean-pooling over [G,B,H,T,D] tensors.
//
// Replaces trim_kv_sliding_window (src/cache/implementations.py:124-140) and
// chunk_summarize_kv (implementations.py:295-346).
//
// Roofline: HBM streams. Window compaction moves 2*esz bytes per kept element. Mean-pool reads
// every old token once (esz*B*H*D*old) plus the kept tail, writes n_chunks + keep rows: ~64:1
// read-dominated at chunk_size 64.
#include "kvq_common.h"

namespace kvq {

// ---------------------------------------------------------------------------- window compaction

struct CopyArgs {
  PtrTable in;  // per-group source base, already advanced to the first kept token
  Strides isb;  // source strides in BYTES
  char* out;    // destination base of this launch's first group
  Strides osb;  // destination strides in BYTES
  uint32_t BH, H;
  uint32_t segs_per_bh;  // 1 when (t,d) is contiguous on both sides, else W (one segment per token)
  int64_t seg_bytes;     // bytes per segment
  uint32_t cps;          // chunks per segment
  uint32_t one_base;     // 1: ONE allocation, group g starts at in.p[0] + g * isb.g (no 128-pointer limit per launch: a paged pool of thousands of blocks is one launch)
};

constexpr int kCopyUnroll = 1;  // one 16-byte vector per thread + non-temporal load/store: the fastest copy recipe
                                // on this chip (profiles/r01e_microbench_calibration.txt: 6.58 TB/s; 5.9 at 4 vectors)
constexpr int64_t kCopyChunk = (int64_t)kBlock * 16 * kCopyUnroll;  // 4 KiB per work item

template <bool VEC>
__global__ __launch_bounds__(kBlock) void copy_rows_k(const CopyArgs a) {
  const uint32_t g = blockIdx.y;
  uint32_t item = blockIdx.x;
  const uint32_t chunk = item % a.cps;
  item /= a.cps;
  const uint32_t seg = item % a.segs_per_bh;
  const uint32_t bh = item / a.segs_per_bh;
  const uint32_t b = bh / a.H, h = bh - b * a.H;
  const char* gsrc = a.one_base ? reinterpret_cast<const char*>(a.in.p[0]) + (int64_t)g * a.isb.g : reinterpret_cast<const char*>(a.in.p[g]);
  const char* src = gsrc + (int64_t)b * a.isb.b + (int64_t)h * a.isb.h + (int64_t)seg * a.isb.t;
  char* dst = a.out + (int64_t)g * a.osb.g + (int64_t)b * a.osb.b + (int64_t)h * a.osb.h + (int64_t)seg * a.osb.t;
  const int64_t c0 = (int64_t)chunk * kCopyChunk;
  if constexpr (VEC) {
    u32x4 v[kCopyUnroll];
#pragma unroll
    for (int u = 0; u < kCopyUnroll; ++u) {
      const int64_t o = c0 + ((int64_t)u * kBlock + threadIdx.x) * 16;
      if (o < a.seg_bytes) v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + o));
    }
#pragma unroll
    for (int u = 0; u < kCopyUnroll; ++u) {
      const int64_t o = c0 + ((int64_t)u * kBlock + threadIdx.x) * 16;
      if (o < a.seg_bytes) __builtin_nontemporal_store(v[u], reinterpret_cast<u32x4*>(dst + o));
    }
  } else {  // 2-byte granularity (elem_size is 2 or 4)
    int64_t c1 = c0 + kCopyChunk;
    if (c1 > a.seg_bytes) c1 = a.seg_bytes;
    for (int64_t o = c0 + (int64_t)threadIdx.x * 2; o < c1; o += kBlock * 2)
      *reinterpret_cast<uint16_t*>(dst + o) = *reinterpret_cast<const uint16_t*>(src + o);
  }
}

// ---------------------------------------------------------------------------- token gather

struct GatherArgs {
  PtrTable in;
  Strides isb;  // bytes
  char* out;
  Strides osb;  // bytes
  const int32_t* idx;
  uint32_t H, n_idx;
  uint32_t T;          // source tokens: idx values outside [0, T) never become an address
  uint32_t row_bytes;  // D * elem_size
  uint32_t vecs;       // 16-byte vectors per token row (VEC) or 2-byte units (scalar)
};

// One lane = one 16-byte piece of one output token row; the lanes of a row read one contiguous
// source row (HBM-bound row gather: 2 * elem_size bytes per kept element). The index table is the
// caller's DEVICE memory: an index outside [0, T) (torch's index_select raises there) writes a
// row of zeros instead of reading out of bounds — one compare per row piece.
template <bool VEC>
__global__ __launch_bounds__(kBlock) void gather_tokens_k(const GatherArgs a, uint32_t items_per_g) {
  const uint32_t g = blockIdx.y;
  constexpr uint32_t kUnit = VEC ? 16u : 2u;
  // 64-bit counter: items_per_g may sit within one grid stride of 2^32
  for (uint64_t it = (uint64_t)blockIdx.x * kBlock + threadIdx.x; it < items_per_g; it += (uint64_t)gridDim.x * kBlock) {
    const uint32_t item = (uint32_t)it;
    const uint32_t v = item % a.vecs;
    uint32_t r = item / a.vecs;
    const uint32_t j = r % a.n_idx;
    r /= a.n_idx;
    const uint32_t h = r % a.H, b = r / a.H;
    const int64_t t = a.idx[j];
    if ((uint64_t)t >= (uint64_t)a.T) {  // also catches negative indices
      char* bad = a.out + (int64_t)g * a.osb.g + (int64_t)b * a.osb.b + (int64_t)h * a.osb.h + (int64_t)j * a.osb.t +
                  (int64_t)v * kUnit;
      if constexpr (VEC) *reinterpret_cast<u32x4*>(bad) = u32x4{0u, 0u, 0u, 0u};
      else *reinterpret_cast<uint16_t*>(bad) = 0;
      continue;
    }
    const char* src = reinterpret_cast<const char*>(a.in.p[g]) + (int64_t)b * a.isb.b + (int64_t)h * a.isb.h +
                      t * a.isb.t + (int64_t)v * kUnit;
    char* dst = a.out + (int64_t)g * a.osb.g + (int64_t)b * a.osb.b + (int64_t)h * a.osb.h + (int64_t)j * a.osb.t +
                (int64_t)v * kUnit;
    if constexpr (VEC)
      __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src)), reinterpret_cast<u32x4*>(dst));
    else
      *reinterpret_cast<uint16_t*>(dst) = *reinterpret_cast<const uint16_t*>(src);
  }
}

// The same gather as 4 KiB work items (round 4): one 16-byte piece per thread, NO loop — blockIdx.x walks the output row of
// one (batch row, head) (blockIdx.z) in 256-piece steps, so a workgroup writes 4 KiB of contiguous output (256 / vecs token
// rows) and reads that many source rows: the launch shape of copy_rows_k, the fastest copy recipe measured on this chip
// (6.58 TB/s against 4.6-5.4 for grid-stride loops, profiles/r01e_microbench_calibration.txt). No division by n_idx per piece
// (the (b, h) row is grid dimension y, the tensor z); vecs is a power of two for every head_dim the library's other fast paths take.
__global__ __launch_bounds__(kBlock) void gather_rows_k(const GatherArgs a, const uint32_t vshift) {
  const uint32_t g = blockIdx.z, bh = blockIdx.y;  // dispatch order x, y, z: one tensor's rows are walked together before the next tensor's
  const uint32_t b = bh / a.H, h = bh - b * a.H;
  const uint32_t piece = blockIdx.x * kBlock + threadIdx.x;  // of this (b, h) row's n_idx * vecs pieces
  const uint32_t j = piece >> vshift, v = piece & ((1u << vshift) - 1u);
  if (j >= a.n_idx) return;
  const int64_t t = a.idx[j];
  char* dst = a.out + (int64_t)g * a.osb.g + (int64_t)b * a.osb.b + (int64_t)h * a.osb.h + (int64_t)j * a.osb.t + (int64_t)v * 16;
  u32x4 x = {0u, 0u, 0u, 0u};  // an index outside [0, T): a row of zeros, never an address
  if ((uint64_t)t < (uint64_t)a.T)
    x = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.in.p[g]) + (int64_t)b * a.isb.b +
                                                                  (int64_t)h * a.isb.h + t * a.isb.t + (int64_t)v * 16));
  __builtin_nontemporal_store(x, reinterpret_cast<u32x4*>(dst));
}

// ---------------------------------------------------------------------------- chunk mean-pool

struct PoolArgs {
  PtrTable in;
  Strides is;  // elements
  void* out;   // base of this launch's first group
  Strides os;  // elements
  uint32_t B, H, D;
  uint32_t T, Tout;
  uint32_t old_len, n_chunks, chunk;
  float chunk_f;  // (float)chunk_size: the divisor, also for the ragged last chunk
  int32_t nt_loads;
};

// independent 16-byte loads in flight per lane (raw packed registers: 4 VGPRs each for fp16/bf16)
template <int DT>
struct PoolBatch {
  static constexpr int n = DT == KVQ_F32 ? 8 : 16;
};

// Fast path: D % 8 == 0, 16-byte aligned rows. One lane owns 8 consecutive d of one output
// token; the D/8 lanes of a token read one contiguous row per step. fp32 accumulation is
// sequential in t (bit-reproducible, mirrored by the oracle).
// NTL: non-temporal loads, a COMPILE-TIME choice (a run-time `if (flag) load_nt else load` is merged by the compiler into
// plain loads: until the end of round 2 this kernel never issued a non-temporal load, whatever the tunable said)
template <int DT, bool NTL = true>
__global__ __launch_bounds__(kBlock) void chunk_pool_vec_k(const PoolArgs a, uint32_t items_per_g) {
  constexpr int NB = PoolBatch<DT>::n;
  const uint32_t g = blockIdx.y;
  const uint32_t DV = a.D >> 3;
  const int64_t tstride = a.is.t * Elem<DT>::size;
  // grid-stride only when a benchmark caps the grid; the shipped launch is one item per thread
  for (uint32_t item = blockIdx.x * blockDim.x + threadIdx.x; item < items_per_g; item += gridDim.x * blockDim.x) {
    const uint32_t dv = item % DV;
    uint32_t r = item / DV;
    const uint32_t j = r % a.Tout;
    r /= a.Tout;
    const uint32_t h = r % a.H, b = r / a.H;
    const char* in = reinterpret_cast<const char*>(a.in.p[g]) +
                     ((int64_t)b * a.is.b + (int64_t)h * a.is.h + (int64_t)dv * 8) * Elem<DT>::size;
    char* out = reinterpret_cast<char*>(a.out) +
                ((int64_t)g * a.os.g + (int64_t)b * a.os.b + (int64_t)h * a.os.h + (int64_t)j * a.os.t +
                 (int64_t)dv * 8) * Elem<DT>::size;
    float acc[8];
    if (j >= a.n_chunks) {  // recent tail: exact copy
      load8<DT>(in + (int64_t)(a.old_len + (j - a.n_chunks)) * tstride, acc);
      store8<DT, false>(out, acc);
      continue;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.0f;
    const uint32_t t0 = j * a.chunk;
    uint32_t n = a.old_len - t0;
    if (n > a.chunk) n = a.chunk;
    const char* p = in + (int64_t)t0 * tstride;
    uint32_t i = 0;
    for (; i + NB <= n; i += NB) {
      Vec8<DT> x[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        if constexpr (NTL) x[u].load_nt(p + (int64_t)(i + u) * tstride);
        else x[u].load(p + (int64_t)(i + u) * tstride);
      }
#pragma unroll
      for (int u = 0; u < NB; ++u)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += x[u].get(k);
    }
    for (; i < n; ++i) {
      Vec8<DT> x;
      x.load(p + (int64_t)i * tstride);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += x.get(k);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = acc[k] / a.chunk_f;
    store8<DT, false>(out, acc);
  }
}

// One-wave workgroup = ONE output row (b, h, j): a pooled chunk or a copied tail row. The 64 lanes are G groups of
// DV = D / 8 lanes; group g owns rows RPG g .. RPG g + RPG - 1 of the chunk (chunk = G RPG rows), so the wave's RPG
// load instructions, ALL in flight at once, cover the chunk's chunk * D * 2 contiguous bytes exactly once and the
// wave ends — the access shape this memory system reads fastest (kvq_microbench poolpat: 7.1 TB/s against 5.8 for
// chunk_pool_vec_k's four-chunks-per-wave walk). The fp32 sum stays SEQUENTIAL in t, bit for bit the order of
// chunk_pool_vec_k and the oracle: G phases; in phase p every group adds its RPG rows to the sum it receives from the
// group below it (ds_bpermute), so after phase p group p holds the exact chain over rows 0 .. RPG (p + 1) - 1 (the other
// groups' values of that phase are never used), and group G - 1 ends with the whole chunk. Rows past a ragged last
// chunk's end enter as +0.0, which leaves an fp32 sum that started at +0.0 unchanged (the reference zero-pads too).
// (A variant without the redundant phases — 64 v_permlane16/32_swap instructions transpose the chunk so that every
// lane sums ONE dword of all 64 rows — was bit-identical and no faster: 5.76-5.85 vs 5.63-5.69 ms on the same box.)
#ifndef KVQ_POOL_CALIB
#define KVQ_POOL_CALIB 0
#endif
template <int DT, int G, int RPG>
__global__ __launch_bounds__(64) void chunk_pool_wave_k(const PoolArgs a) {
  static_assert(DT != KVQ_F32, "16-bit element types");
  constexpr uint32_t DV = 64 / G;
  const uint32_t g = blockIdx.y;
  uint32_t r = blockIdx.x;
  const uint32_t j = r % a.Tout;
  r /= a.Tout;
  const uint32_t h = r % a.H, b = r / a.H;
  const uint32_t lane = threadIdx.x, lg = lane / DV, dv = lane % DV;
  const int64_t tstride = a.is.t * 2;
  const char* in = reinterpret_cast<const char*>(a.in.p[g]) + ((int64_t)b * a.is.b + (int64_t)h * a.is.h + (int64_t)dv * 8) * 2;
  char* out = reinterpret_cast<char*>(a.out) +
              ((int64_t)g * a.os.g + (int64_t)b * a.os.b + (int64_t)h * a.os.h + (int64_t)j * a.os.t + (int64_t)dv * 8) * 2;
  if (j >= a.n_chunks) {  // recent tail: exact copy (wave-uniform branch)
    if (lg == 0u)
      *reinterpret_cast<u32x4*>(out) = *reinterpret_cast<const u32x4*>(in + (int64_t)(a.old_len + (j - a.n_chunks)) * tstride);
    return;
  }
  const uint32_t t0 = j * a.chunk;
  uint32_t n = a.old_len - t0;
  if (n > a.chunk) n = a.chunk;  // >= 1
  // the chunk's valid rows as a buffer: rows past a ragged chunk's end are out of range and load as zeros
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(a.in.p[g]) + ((int64_t)b * a.is.b + (int64_t)h * a.is.h) * 2 + (int64_t)t0 * tstride), 0,
      (int)(n * (uint32_t)tstride), 0x00020000);
  u32x4 raw[RPG];
#pragma unroll
  for (int i = 0; i < RPG; ++i)  // all in flight
    raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (lg * RPG + i) * (uint32_t)tstride + dv * 16u, 0, 2);
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 acc2[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) acc2[k] = f32x2{0.0f, 0.0f};
  // widen once, then packed fp32 adds (IEEE per component: the same sums as scalar adds, half the instructions —
  // with scalar adds the kernel measured 5.58-5.74 ms at config 5, packed 5.21 ms; widening inside every phase
  // instead, to save registers: 5.45 ms)
  f32x2 xf[RPG][4];
#pragma unroll
  for (int i = 0; i < RPG; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k)
      xf[i][k] = f32x2{Elem<DT>::widen((uint16_t)(raw[i][k] & 0xFFFFu)), Elem<DT>::widen((uint16_t)(raw[i][k] >> 16))};
#if KVQ_POOL_CALIB == 1  // calibration (`make calib_pool`, NOT the oracle's order): every group sums its own RPG rows once, the G partial
  // sums are then added group by group — what a blocked summation order would cost instead of the sequential one
#pragma unroll
  for (int i = 0; i < RPG; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc2[k] += xf[i][k];
#pragma unroll
  for (int ph = 1; ph < G; ++ph) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x2 up = f32x2{__shfl_up(acc2[k][0], DV), __shfl_up(acc2[k][1], DV)};
      if (lg == (uint32_t)ph) acc2[k] = up + acc2[k];
    }
  }
#else
#pragma unroll
  for (int ph = 0; ph < G; ++ph) {
    if (ph) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc2[k] = f32x2{__shfl_up(acc2[k][0], DV), __shfl_up(acc2[k][1], DV)};
    }
#pragma unroll
    for (int i = 0; i < RPG; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc2[k] += xf[i][k];
  }
#endif
  float acc[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    acc[2 * k] = acc2[k][0];
    acc[2 * k + 1] = acc2[k][1];
  }
  if (lg == (uint32_t)(G - 1)) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = acc[k] / a.chunk_f;
    store8<DT, false>(out, acc);
  }
}

// Generic path: any D / strides / alignment; one thread per output element.
template <int DT>
__global__ __launch_bounds__(kBlock) void chunk_pool_generic_k(const PoolArgs a, int64_t items_per_g) {
  const uint32_t g = blockIdx.y;
  const void* in = a.in.p[g];
  for (int64_t item = (int64_t)blockIdx.x * kBlock + threadIdx.x; item < items_per_g;
       item += (int64_t)gridDim.x * kBlock) {
    const int64_t d = item % a.D;
    int64_t r = item / a.D;
    const int64_t j = r % a.Tout;
    r /= a.Tout;
    const int64_t h = r % a.H, b = r / a.H;
    const int64_t ibase = b * a.is.b + h * a.is.h + d;
    const int64_t o = (int64_t)g * a.os.g + b * a.os.b + h * a.os.h + j * a.os.t + d;
    if (j >= a.n_chunks) {
      store1<DT>(a.out, o, load1<DT>(in, ibase + (int64_t)(a.old_len + (j - a.n_chunks)) * a.is.t));
      continue;
    }
    const int64_t t0 = j * (int64_t)a.chunk;
    int64_t n = (int64_t)a.old_len - t0;
    if (n > a.chunk) n = a.chunk;
    float acc = 0.0f;
    for (int64_t i = 0; i < n; ++i) acc += load1<DT>(in, ibase + (t0 + i) * a.is.t);
    store1<DT>(a.out, o, acc / a.chunk_f);
  }
}

// ---------------------------------------------------------------------------- host side

static int fill_ptrs(PtrTable& tbl, const void* base, const void* const* ptrs, int64_t g0, int64_t gn,
                     int64_t stride_g_bytes, int64_t extra_bytes, const char* name) {
  for (int64_t i = 0; i < gn; ++i) {
    const char* p = ptrs ? static_cast<const char*>(ptrs[g0 + i])
                         : static_cast<const char*>(base) + (g0 + i) * stride_g_bytes;
    if (!p) {
      set_error("%s: in_ptrs[%lld] is NULL", name, (long long)(g0 + i));
      return KVQ_E_NULL;
    }
    tbl.p[i] = p + extra_bytes;
  }
  return 0;
}

static int common_checks(const void* in_base, const void* const* in_ptrs, const kvq_strides_t* in_st, void* out,
                         const kvq_strides_t* out_st, const kvq_dims_t* d, const char* name) {
  if (!in_st || !out_st || !d || !out || (!in_base && !in_ptrs)) {
    set_error("%s: NULL argument", name);
    return KVQ_E_NULL;
  }
  if (const int rcd = check_device(out, name)) return rcd;
  if (in_base && in_ptrs) {
    set_error("%s: pass in_base or in_ptrs, not both", name);
    return KVQ_E_DIMS;
  }
  if (d->G < 0 || d->B < 0 || d->H < 0 || d->T < 0 || d->D < 0) {
    set_error("%s: negative dim", name);
    return KVQ_E_DIMS;
  }
  if (d->B * d->H >= (int64_t(1) << 31) || d->T >= (int64_t(1) << 31) || d->D >= (int64_t(1) << 31)) {
    set_error("%s: dims too large", name);
    return KVQ_E_DIMS;
  }
  return 0;
}

}  // namespace kvq

using namespace kvq;

extern "C" {

int kvq_window_compact(const void* in_base, const void* const* in_ptrs, const kvq_strides_t* in_st, void* out,
                       const kvq_strides_t* out_st, int elem_size, int64_t window, const kvq_dims_t* d,
                       void* stream) {
  const char* name = "kvq_window_compact";
  int rc = common_checks(in_base, in_ptrs, in_st, out, out_st, d, name);
  if (rc) return rc;
  if ((elem_size != 2 && elem_size != 4) || window < 0) {
    set_error("%s: elem_size must be 2 or 4 and window >= 0", name);
    return KVQ_E_DIMS;
  }
  const int64_t W = window < d->T ? window : d->T;
  if (d->G * d->B * d->H * W * d->D == 0) return 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);

  CopyArgs a;
  a.isb = Strides{in_st->g * elem_size, in_st->b * elem_size, in_st->h * elem_size, in_st->t * elem_size};
  a.osb = Strides{out_st->g * elem_size, out_st->b * elem_size, out_st->h * elem_size, out_st->t * elem_size};
  a.BH = (uint32_t)(d->B * d->H);
  a.H = (uint32_t)d->H;
  const bool tcontig = in_st->t == d->D && out_st->t == d->D;
  a.segs_per_bh = tcontig ? 1u : (uint32_t)W;
  a.seg_bytes = (tcontig ? W * d->D : d->D) * elem_size;
  a.cps = (uint32_t)((a.seg_bytes + kCopyChunk - 1) / kCopyChunk);
  const int64_t blocks = (int64_t)a.BH * a.segs_per_bh * a.cps;
  if (blocks >= (int64_t(1) << 31)) {
    set_error("%s: too many segments", name);
    return KVQ_E_DIMS;
  }
  const int64_t first_kept_bytes = (d->T - W) * a.isb.t;
  // more groups than one launch's pointer table holds, all in ONE allocation (a paged pool's blocks): the kernel adds
  // g * stride itself, up to 65,535 groups per launch (PagedKVCache.get_kv at 32 K tokens: 1 launch per pool instead of 4)
  a.one_base = (in_base && d->G > kPtrsPerLaunch) ? 1u : 0u;
  const int64_t per_launch = a.one_base ? 65535 : kPtrsPerLaunch;

  for (int64_t g0 = 0; g0 < d->G; g0 += per_launch) {
    const int64_t gn = d->G - g0 < per_launch ? d->G - g0 : per_launch;
    rc = fill_ptrs(a.in, in_base, in_ptrs, g0, a.one_base ? 1 : gn, a.isb.g, first_kept_bytes, name);
    if (rc) return rc;
    a.out = static_cast<char*>(out) + g0 * a.osb.g;
    bool vec = a.seg_bytes % 16 == 0 && aligned(a.out, 16) && a.isb.b % 16 == 0 && a.isb.h % 16 == 0 &&
               a.isb.t % 16 == 0 && a.osb.g % 16 == 0 && a.osb.b % 16 == 0 && a.osb.h % 16 == 0 && a.osb.t % 16 == 0 &&
               (!a.one_base || a.isb.g % 16 == 0);
    for (int64_t i = 0; i < (a.one_base ? 1 : gn) && vec; ++i) vec = aligned(a.in.p[i], 16);
    if (vec)
      KVQ_LAUNCH((copy_rows_k<true>), dim3((unsigned)blocks, (unsigned)gn), dim3(kBlock), 0, st, a);
    else
      KVQ_LAUNCH((copy_rows_k<false>), dim3((unsigned)blocks, (unsigned)gn), dim3(kBlock), 0, st, a);
    rc = check_launch(name);
    if (rc) return rc;
  }
  return 0;
}

int kvq_gather_tokens(const void* in_base, const void* const* in_ptrs, const kvq_strides_t* in_st, void* out,
                      const kvq_strides_t* out_st, int elem_size, const int32_t* idx, int64_t n_idx,
                      const kvq_dims_t* d, void* stream) {
  const char* name = "kvq_gather_tokens";
  int rc = common_checks(in_base, in_ptrs, in_st, out, out_st, d, name);
  if (rc) return rc;
  if ((elem_size != 2 && elem_size != 4) || n_idx < 0 || n_idx >= (int64_t(1) << 31)) {
    set_error("%s: elem_size must be 2 or 4 and 0 <= n_idx < 2^31", name);
    return KVQ_E_DIMS;
  }
  if (d->G * d->B * d->H * n_idx * d->D == 0) return 0;
  if (!idx) {
    set_error("%s: idx is NULL", name);
    return KVQ_E_NULL;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  GatherArgs a;
  a.isb = Strides{in_st->g * elem_size, in_st->b * elem_size, in_st->h * elem_size, in_st->t * elem_size};
  a.osb = Strides{out_st->g * elem_size, out_st->b * elem_size, out_st->h * elem_size, out_st->t * elem_size};
  a.idx = idx;
  a.H = (uint32_t)d->H;
  a.n_idx = (uint32_t)n_idx;
  a.T = (uint32_t)d->T;
  a.row_bytes = (uint32_t)(d->D * elem_size);
  for (int64_t g0 = 0; g0 < d->G; g0 += kPtrsPerLaunch) {
    const int64_t gn = d->G - g0 < kPtrsPerLaunch ? d->G - g0 : kPtrsPerLaunch;
    rc = fill_ptrs(a.in, in_base, in_ptrs, g0, gn, a.isb.g, 0, name);
    if (rc) return rc;
    a.out = static_cast<char*>(out) + g0 * a.osb.g;
    bool vec = a.row_bytes % 16 == 0 && aligned(a.out, 16) && a.isb.b % 16 == 0 && a.isb.h % 16 == 0 &&
               a.isb.t % 16 == 0 && a.osb.g % 16 == 0 && a.osb.b % 16 == 0 && a.osb.h % 16 == 0 && a.osb.t % 16 == 0;
    for (int64_t i = 0; i < gn && vec; ++i) vec = aligned(a.in.p[i], 16);
    a.vecs = a.row_bytes / (vec ? 16u : 2u);
    const int64_t items = d->B * d->H * n_idx * (int64_t)a.vecs;
    if (items >= (int64_t(1) << 32)) {
      set_error("%s: too many elements per group", name);
      return KVQ_E_DIMS;
    }
    const int vsh = vec ? ilog2_exact((int64_t)a.vecs) : -1;  // -1: not a power of two
    if (vec && vsh >= 0 && d->B * d->H < 65536 && tunables().gather_rows != 0) {  // 4 KiB items, one piece per thread
      const int64_t row_pieces = n_idx * (int64_t)a.vecs;
      KVQ_LAUNCH(gather_rows_k, dim3((unsigned)((row_pieces + kBlock - 1) / kBlock), (unsigned)(d->B * d->H), (unsigned)gn), dim3(kBlock), 0, st, a, (uint32_t)vsh);
      rc = check_launch(name);
      if (rc) return rc;
      continue;
    }
    int64_t blocks = (items + kBlock - 1) / kBlock;
    if (blocks > 256 * 64) blocks = 256 * 64;
    if (vec)
      KVQ_LAUNCH((gather_tokens_k<true>), dim3((unsigned)blocks, (unsigned)gn), dim3(kBlock), 0, st, a, (uint32_t)items);
    else
      KVQ_LAUNCH((gather_tokens_k<false>), dim3((unsigned)blocks, (unsigned)gn), dim3(kBlock), 0, st, a, (uint32_t)items);
    rc = check_launch(name);
    if (rc) return rc;
  }
  return 0;
}

int kvq_chunk_meanpool(const void* in_base, const void* const* in_ptrs, const kvq_strides_t* in_st, void* out,
                       const kvq_strides_t* out_st, int dtype, int64_t chunk_size, int64_t keep_last,
                       const kvq_dims_t* d, void* stream) {
  const char* name = "kvq_chunk_meanpool";
  int rc = common_checks(in_base, in_ptrs, in_st, out, out_st, d, name);
  if (rc) return rc;
  if (dtype != KVQ_F16 && dtype != KVQ_BF16 && dtype != KVQ_F32) {
    set_error("%s: unknown dtype %d", name, dtype);
    return KVQ_E_DTYPE;
  }
  if (chunk_size <= 0 || keep_last < 0 || chunk_size >= (int64_t(1) << 31)) {
    set_error("%s: chunk_size must be > 0 and keep_last >= 0", name);
    return KVQ_E_DIMS;
  }
  if (d->G * d->B * d->H * d->T * d->D == 0) return 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int esz = dtype == KVQ_F32 ? 4 : 2;
  const int64_t keep = keep_last < d->T ? keep_last : d->T;
  const int64_t old = d->T - keep;
  const int64_t n_chunks = old > 0 ? (old + chunk_size - 1) / chunk_size : 0;
  const int64_t Tout = n_chunks + keep;  // == T when old <= 0 (pure copy)

  PoolArgs a;
  a.is = to_strides(in_st);
  a.os = to_strides(out_st);
  a.B = (uint32_t)d->B;
  a.H = (uint32_t)d->H;
  a.D = (uint32_t)d->D;
  a.T = (uint32_t)d->T;
  a.Tout = (uint32_t)Tout;
  a.old_len = (uint32_t)(old > 0 ? old : 0);
  a.n_chunks = (uint32_t)n_chunks;
  a.chunk = (uint32_t)chunk_size;
  a.chunk_f = (float)chunk_size;
  a.nt_loads = (int32_t)tunables().nt_loads;

  const int64_t items_vec = d->B * d->H * Tout * (d->D / 8);
  const int64_t items_gen = d->B * d->H * Tout * d->D;
  for (int64_t g0 = 0; g0 < d->G; g0 += kPtrsPerLaunch) {
    const int64_t gn = d->G - g0 < kPtrsPerLaunch ? d->G - g0 : kPtrsPerLaunch;
    rc = fill_ptrs(a.in, in_base, in_ptrs, g0, gn, a.is.g * esz, 0, name);
    if (rc) return rc;
    a.out = static_cast<char*>(out) + g0 * a.os.g * esz;
    bool vec = d->D % 8 == 0 && items_vec < (int64_t(1) << 31) && aligned(a.out, 16) &&
               (a.is.b * esz) % 16 == 0 && (a.is.h * esz) % 16 == 0 && (a.is.t * esz) % 16 == 0 &&
               (a.os.g * esz) % 16 == 0 && (a.os.b * esz) % 16 == 0 && (a.os.h * esz) % 16 == 0 &&
               (a.os.t * esz) % 16 == 0;
    for (int64_t i = 0; i < gn && vec; ++i) vec = aligned(a.in.p[i], 16);
    // one wave per output row when a chunk is exactly the wave's G x RPG rows (16-bit types, D = 64 / 128 / 256)
    const int64_t lane_groups = d->D == 256 ? 2 : d->D == 128 ? 4 : d->D == 64 ? 8 : 0;
    const int64_t rpg = lane_groups ? chunk_size / lane_groups : 0;
    const bool wave = vec && dtype != KVQ_F32 && tunables().pool_wave && lane_groups && chunk_size % lane_groups == 0 &&
                      (rpg == 8 || rpg == 16) && d->B * d->H * Tout < (int64_t(1) << 31) &&
                      chunk_size * a.is.t * 2 < (int64_t(1) << 31);  // 32-bit offsets inside a chunk
    if (wave) {
      const dim3 grid((unsigned)(d->B * d->H * Tout), (unsigned)gn);
#define KVQ_POOL_WAVE(DT_, G_, R_) KVQ_LAUNCH((chunk_pool_wave_k<DT_, G_, R_>), grid, dim3(64), 0, st, a)
#define KVQ_POOL_WAVE_DT(G_, R_)                  \
  do {                                            \
    if (dtype == KVQ_F16) KVQ_POOL_WAVE(KVQ_F16, G_, R_); \
    else KVQ_POOL_WAVE(KVQ_BF16, G_, R_);         \
  } while (0)
      if (lane_groups == 4 && rpg == 16) KVQ_POOL_WAVE_DT(4, 16);
      else if (lane_groups == 4) KVQ_POOL_WAVE_DT(4, 8);
      else if (lane_groups == 8 && rpg == 16) KVQ_POOL_WAVE_DT(8, 16);
      else if (lane_groups == 8) KVQ_POOL_WAVE_DT(8, 8);
      else if (rpg == 16) KVQ_POOL_WAVE_DT(2, 16);
      else KVQ_POOL_WAVE_DT(2, 8);
#undef KVQ_POOL_WAVE_DT
#undef KVQ_POOL_WAVE
    } else if (vec) {
      // Launch shape (measured on a config-5-shaped launch, 64-thread workgroups): one item per
      // thread 6.04 TB/s, persistent grids of 2048 / 4096 / 16384 workgroups 5.85 / 5.94 / 6.00.
      const int64_t blk = tunables().pool_block == 256 || tunables().pool_block == 128 ? tunables().pool_block : 64;
      int64_t want = (items_vec + blk - 1) / blk;
      if (tunables().pool_grid > 0) {  // benchmarks only: persistent grid-stride launch
        const int64_t cap = tunables().pool_grid * (kBlock / blk);
        const int64_t per_g = (cap + gn - 1) / gn > 8 ? (cap + gn - 1) / gn : 8;
        if (want > per_g) want = per_g;
      }
      const unsigned blocks = (unsigned)want;
      switch (dtype) {
#if KVQ_AB
#define KVQ_POOL_VEC(DT_)                                                                                                             \
  do {                                                                                                                                \
    if (a.nt_loads) KVQ_LAUNCH((chunk_pool_vec_k<DT_, true>), dim3(blocks, (unsigned)gn), dim3((unsigned)blk), 0, st, a, (uint32_t)items_vec); \
    else KVQ_LAUNCH((chunk_pool_vec_k<DT_, false>), dim3(blocks, (unsigned)gn), dim3((unsigned)blk), 0, st, a, (uint32_t)items_vec);  \
  } while (0)
#else  // non-temporal loads ship (the plain-load instantiation is an A-B variant)
#define KVQ_POOL_VEC(DT_) KVQ_LAUNCH((chunk_pool_vec_k<DT_, true>), dim3(blocks, (unsigned)gn), dim3((unsigned)blk), 0, st, a, (uint32_t)items_vec)
#endif
        case KVQ_F16: KVQ_POOL_VEC(KVQ_F16); break;
        case KVQ_BF16: KVQ_POOL_VEC(KVQ_BF16); break;
        case KVQ_F32: KVQ_POOL_VEC(KVQ_F32); break;
#undef KVQ_POOL_VEC
      }
    } else {
      int64_t blocks = (items_gen + kBlock - 1) / kBlock;
      if (blocks > 256 * 32) blocks = 256 * 32;
      switch (dtype) {
        case KVQ_F16: KVQ_LAUNCH((chunk_pool_generic_k<KVQ_F16>), dim3((unsigned)blocks, (unsigned)gn), dim3(kBlock), 0, st, a, items_gen); break;
        case KVQ_BF16: KVQ_LAUNCH((chunk_pool_generic_k<KVQ_BF16>), dim3((unsigned)blocks, (unsigned)gn), dim3(kBlock), 0, st, a, items_gen); break;
        case KVQ_F32: KVQ_LAUNCH((chunk_pool_generic_k<KVQ_F32>), dim3((unsigned)blocks, (unsigned)gn), dim3(kBlock), 0, st, a, items_gen); break;
      }
    }
    rc = check_launch(name);
    if (rc) return rc;
  }
  return 0;
}

}  // extern "C"
