// kvq_common.h — shared device helpers for the gfx950 KV quantize/dequantize/eviction kernels.
// CDNA4 only: wave64, 16-byte vector memory ops, no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/kvq_hip.h"

namespace kvq {

typedef _Float16 f16;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int kBlock = 256;  // 4 waves of 64
constexpr int kWave = 64;

struct Strides {
  int64_t g, b, h, t;
};
static inline Strides to_strides(const kvq_strides_t* s) { return Strides{s->g, s->b, s->h, s->t}; }

// Up to KVQ_PTRS_PER_LAUNCH group base pointers travel by value in the kernarg segment, so a
// legacy tuple of separately allocated [B,H,T,D] tensors needs no host->device table copy.
constexpr int kPtrsPerLaunch = 128;
struct PtrTable {
  const void* p[kPtrsPerLaunch];
};

// ------------------------------------------------------------------ error reporting (host)
void set_error(const char* fmt, ...);
int check_launch(const char* what);
int check_device(const void* p, const char* name);  // 0, or KVQ_E_DEVICE when `p` lives on another device than the thread's current one

// kvq_time_next_launch: HIP events the NEXT kernel launch of the calling thread binds to its own dispatch
// (hipExtLaunchKernelGGL start / stop timestamps): the kernel's duration without queue gaps. Taken (and cleared) by
// the first launch that follows, whichever entry point makes it; {nullptr, nullptr} = plain launch.
struct TimingEvents {
  hipEvent_t start, stop;
};
TimingEvents take_timing_events();

// kvq_kernel_log: every launch notes its kernel's host stub; the log resolves the stubs to the names a profiler
// prints (hipKernelNameRefByPtr + demangling), so a benchmark labels its roofline with the kernel that actually ran.
void note_launch(const void* host_stub);

// The one way this library launches a kernel.
#define KVQ_LAUNCH(KERNEL, GRID, BLOCK, LDS, ST, ...)                                                                \
  do {                                                                                                               \
    ::kvq::note_launch(reinterpret_cast<const void*>(&KERNEL));                                                      \
    const ::kvq::TimingEvents kvq_ev_ = ::kvq::take_timing_events();                                                 \
    if (kvq_ev_.start || kvq_ev_.stop)                                                                               \
      hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, ST, kvq_ev_.start, kvq_ev_.stop, 0, __VA_ARGS__);              \
    else                                                                                                             \
      hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, ST, __VA_ARGS__);                                                 \
  } while (0)

// A-B builds (`make ab`, -DKVQ_AB=1) keep every experiment that lost a measurement next to what ships; the default
// library holds the shipped instantiations and the generic fallbacks only, and kvq_set_tunable refuses A-B keys.
#ifndef KVQ_AB
#define KVQ_AB 0
#endif

// Process-global knobs (kvq_set_tunable). "test knob": routes a call to SHIPPED code it would not take by size or
// shape, settable in every build. Everything else is an A-B key: it selects code that only `make ab` builds contain.
struct Tunables {
  // ---- test knobs
  int64_t quant_force_two_pass;  // 1 = generic two-pass quantise for every shape
  int64_t quant_direct_stores;   // 1 = skip the LDS-staged 16 B stores of the 256-thread fused kernel
  int64_t quant_block;           // general fused quantise kernel: 64 (default, one wave per tile) or 256 threads; 128 in A-B builds (measured: 241 / 261 / 270 us)
  int64_t quant_wide;            // 1 (default) = single-pass 1024-thread register tile for batched slices of 16384 < B*H*D <= 131072 elements; 0 = split phases / swept tile
  int64_t quant_tile;            // 1 (default) = compile-time-geometry one-wave tile kernel where the shape has one; 0 = general kernels
  int64_t gather_rows;           // token gather: 1 (default) = 4 KiB work items, one 16-byte piece per thread (gather_rows_k); 0 = the grid-stride kernel (same bytes)
  int64_t pool_wave;             // chunk mean-pool: one wave per output row when the shape allows (1, default) or the per-lane-group walk (0)
  int64_t attn_force_valu;       // 1 = decode attention never takes an MFMA kernel
  int64_t attn_stream_tpw;       // streaming MFMA kernel: 64-token tiles per wave; 0 = by size (only when tiles exceed wave slots), -1 = never
  int64_t attn_merge_wave;       // merge of <= 16 splits at head_dim 128: 1 (default) = one wave per head, 0 = the 256-thread workgroup per head (same bits)
  int64_t attn_lds;              // LDS-staged MFMA kernel (contiguous row loads): -1 = by shape (default), 0 = never, 1 = wherever it applies
  // ---- A-B keys
  int64_t dequant_variant;       // -1 = shipped default
  int64_t dequant_grid;          // 0 = one chunk per workgroup
  int64_t dequant_xcd_group;     // consecutive chunks per XCD (xcd_grouped_item): 0 / 1 = round robin
  int64_t quant_xcd_group;       // same for the one-wave quantise tiles
  int64_t pool_grid;             // cap the chunk mean-pool grid (256-thread equivalents); 0 = one item per thread
  int64_t pool_block;            // chunk mean-pool workgroup size: 64 (default, +7 %), 128 or 256
  int64_t quant_nv;              // 4 = 2048-element one-wave tiles, 16 = 8192, else 8
  int64_t quant_no_regmax;       // 1 = keep the LDS abs-max in one-wave tiles
  int64_t quant_wide_blk;        // wide quantise tile: 1024 threads (default) | 512 (two workgroups per CU, half the tile)
  int64_t quant_geo128;          // round 2's GEO128 instantiation of the general kernel instead of the tile kernel (needs quant_tile = 0)
  int64_t quant_nt_stores;       // GEO128 kernel: non-temporal output stores (1), write-back stores (0), -1 (default) = as nt_loads
  int64_t quant_tpw;             // tiles per wave of the pipelined one-wave quantise kernel (2 | 4 | 8); 0 = one tile per wave
  int64_t quant_lds_pad;         // bytes of unused dynamic LDS on the one-wave quantise launch (caps waves per CU)
  int64_t quant_tile_tpw;        // merged-store tile kernel: tiles per wave 2 | 4 (8-row shapes, fp16); 0 = the one-tile kernel
  int64_t quant_tile_tt;         // tile kernel at head_dim 64: tokens per tile 8 (one row per load instruction) or 4 (two rows); 0 = shipped choice
  int64_t attn_mfma_min_nq;      // fewest query heads per kv head that take the MFMA kernels at head_dim 64 / 128 (default 1 since round 4; 3 before)
  int64_t attn_mfma_tc;          // tokens per one-wave split of the MFMA kernel: 128 (default) or 64
  int64_t attn_stream_tc;        // tokens per tile of the streaming kernel: 64 (default) or 32
  int64_t attn_stream_slots;     // wave slots the streaming plan fills in one round (default 2048 = 2 per SIMD)
  int64_t attn_tg;               // LDS-staged kernel, <= 4 query heads per kv head: 0 = one score output per tile (default), 1 = one per 16-token group
  int64_t attn_lds_tc;           // LDS-staged streaming kernel: tokens per tile 64 (default) | 32
  int64_t attn_lds_nb;           // LDS-staged streaming kernel: ring depth 2 | 3 | 4; 0 = by kinds (3 where four workgroups fit a CU)
  int64_t attn_stream_roll;      // streaming kernel, 64-token tiles: re-request a tile's registers piece by piece for the tile after next (1) or whole tiles between reductions (0)
  int64_t attn_merge_fast;       // 1 (default) = merge kernel that requests everything up front (<= 256 splits); 0 = the chained one
  int64_t attn_k_i8;             // INT8 keys at head_dim 128: stored bytes straight into v_mfma_i32_16x16x64_i8 (query as two int8 planes): -1 = streaming kernel only (default), 0 = never, 1 = always
  int64_t quant_few_tokens;      // abs-max phase of a slice of at most this many tokens (a sharded decode append: 1): one workgroup per (group, token) and a plain store instead of the tile walk's atomics on one word; 0 = never (tests / A-B)
  int64_t attn_ring_dev;         // kvq_decode_step_dev (device-side token count) on the LDS-staged ring kernel where the host-side call takes it (1, default); 0 = always one-tile splits, as before round 4
  int64_t attn_new_token_parts;  // decode step, new-token slices past 8,192 elements: one workgroup per 8,192 elements, each taking the whole slice's abs-max (1, default); 0 = the generic one-workgroup routine (tests)
  int64_t attn_onepass;          // A-B: up to 2,048 stored tokens in ONE launch, one 8-wave workgroup per (batch row, kv head), merge from LDS (decode_attn_onepass_k): 0 (default) = never; measured slower (r04x)
  int64_t attn_fold;             // merge inside the LDS-staged kernel's launch (arrival ticket, last wave merges; same bits): 0 (default) = never, 1 = in kvq_decode_step_layers (one memset per host call), 2 = kvq_decode_attn too; measured slower (r04b)
  int64_t attn_fused;            // 1 = decode attention as ONE launch (decode_attn_fused_mfma_k) where it applies; default 0: partial + merge measured faster
  int64_t attn_fused_tc;         // fused launch, head_dim 128: tokens per wave 128 | 64 | 32 (with attn_fused_nw 4 or 8 | 8 | 16); 0 = by batch size
  int64_t attn_fused_nw;
  int64_t nt_loads;              // non-temporal input loads in the general quantise / pool kernels (default 1)
};
Tunables& tunables();


// ------------------------------------------------------------------ element conversion

template <int DT>
struct Elem;  // storage type + widen/narrow (RN-even, what torch .float()/.to(dtype) do)

template <>
struct Elem<KVQ_F16> {
  typedef f16 type;
  static constexpr int size = 2;
  __device__ static inline float widen(uint16_t bits) {
    f16 h;
    __builtin_memcpy(&h, &bits, 2);
    return (float)h;
  }
  __device__ static inline uint32_t pack2(float a, float b) {
    f16x2 v = {(f16)a, (f16)b};
    uint32_t u;
    __builtin_memcpy(&u, &v, 4);
    return u;
  }
  __device__ static inline float round_trip(float v) { return (float)(f16)v; }
};

template <>
struct Elem<KVQ_BF16> {
  typedef __bf16 type;
  static constexpr int size = 2;
  __device__ static inline float widen(uint16_t bits) { return __uint_as_float((uint32_t)bits << 16); }
  __device__ static inline uint32_t pack2(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};
    uint32_t u;
    __builtin_memcpy(&u, &v, 4);
    return u;
  }
  __device__ static inline float round_trip(float v) { return (float)(__bf16)v; }
};

template <>
struct Elem<KVQ_F32> {
  typedef float type;
  static constexpr int size = 4;
  __device__ static inline float round_trip(float v) { return v; }
};

// Load 8 consecutive elements (16-byte aligned for 2-byte types, 32 bytes for f32) as fp32.
template <int DT>
__device__ inline void load8(const void* p, float (&x)[8]) {
  if constexpr (DT == KVQ_F32) {
    const u32x4 a = *reinterpret_cast<const u32x4*>(p);
    const u32x4 b = *(reinterpret_cast<const u32x4*>(p) + 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      x[i] = __uint_as_float(a[i]);
      x[4 + i] = __uint_as_float(b[i]);
    }
  } else {
    const u32x4 a = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      x[2 * i] = Elem<DT>::widen((uint16_t)(a[i] & 0xFFFFu));
      x[2 * i + 1] = Elem<DT>::widen((uint16_t)(a[i] >> 16));
    }
  }
}

// Store 8 fp32 values as 8 consecutive elements of DT (RN-even), 16-byte vector stores.
template <int DT, bool NT>
__device__ inline void store8(void* p, const float (&x)[8]) {
  if constexpr (DT == KVQ_F32) {
    u32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a[i] = __float_as_uint(x[i]);
      b[i] = __float_as_uint(x[4 + i]);
    }
    if constexpr (NT) {
      __builtin_nontemporal_store(a, reinterpret_cast<u32x4*>(p));
      __builtin_nontemporal_store(b, reinterpret_cast<u32x4*>(p) + 1);
    } else {
      *reinterpret_cast<u32x4*>(p) = a;
      *(reinterpret_cast<u32x4*>(p) + 1) = b;
    }
  } else {
    u32x4 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = Elem<DT>::pack2(x[2 * i], x[2 * i + 1]);
    if constexpr (NT) {
      __builtin_nontemporal_store(a, reinterpret_cast<u32x4*>(p));
    } else {
      *reinterpret_cast<u32x4*>(p) = a;
    }
  }
}

// Tile element storage: 2-byte dtypes stay PACKED in registers (4 VGPRs per 8 elements) between
// the abs-max pass and the quantise pass; fp32 keeps 8 floats.
template <int IDT>
struct Vec8 {
  u32x4 w;
  __device__ inline void load(const void* p) { w = *reinterpret_cast<const u32x4*>(p); }
  // read-once stream: non-temporal (keeps the line out of the way of the stores' write-back)
  __device__ inline void load_nt(const void* p) { w = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
  // max |x| as an order-preserving bit pattern: sign-masked halves compared as unsigned ints
  __device__ inline uint32_t absmax_bits() const {
    typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
    u16x2 h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t a = w[i] & 0x7FFF7FFFu;
      __builtin_memcpy(&h[i], &a, 4);
    }
    const u16x2 m = __builtin_elementwise_max(__builtin_elementwise_max(h[0], h[1]),
                                              __builtin_elementwise_max(h[2], h[3]));  // v_pk_max_u16
    return max((uint32_t)m[0], (uint32_t)m[1]);
  }
  __device__ static inline float bits_to_f32(uint32_t b) { return Elem<IDT>::widen((uint16_t)b); }
  __device__ inline float get(int j) const {
    return Elem<IDT>::widen((uint16_t)((j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xFFFFu)));
  }
};
template <>
struct Vec8<KVQ_F32> {
  float x[8];
  __device__ inline void load(const void* p) { load8<KVQ_F32>(p, x); }
  __device__ inline void load_nt(const void* p) { load8<KVQ_F32>(p, x); }
  __device__ inline uint32_t absmax_bits() const {
    float m = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(x[j]));
    return __float_as_uint(m);
  }
  __device__ static inline float bits_to_f32(uint32_t b) { return __uint_as_float(b); }
  __device__ inline float get(int j) const { return x[j]; }
};

// scalar element access for the generic (any shape / stride) kernels
template <int DT>
__device__ inline float load1(const void* base, int64_t idx) {
  if constexpr (DT == KVQ_F32) {
    return reinterpret_cast<const float*>(base)[idx];
  } else {
    return Elem<DT>::widen(reinterpret_cast<const uint16_t*>(base)[idx]);
  }
}
template <int DT>
__device__ inline void store1(void* base, int64_t idx, float v) {
  if constexpr (DT == KVQ_F32) {
    reinterpret_cast<float*>(base)[idx] = v;
  } else {
    reinterpret_cast<uint16_t*>(base)[idx] = (uint16_t)(Elem<DT>::pack2(v, 0.0f) & 0xFFFFu);
  }
}

// fp32 product that stays an fp32 product. Without the opaque barrier hipcc folds
// (f16)(a * b) into v_fma_mixlo_f16 a, b, +0.0, and (-0.0) + (+0.0) = +0.0 loses the sign of a
// zero result (q < 0 with a stored fp16 scale of 0): the reference yields -0.0 there.
__device__ inline float mul_exact(float a, float b) {
  float p = a * b;
  asm("" : "+v"(p));
  return p;
}

// ------------------------------------------------------------------ wave64 reductions

// max over aligned groups of `width` consecutive lanes (width = power of two <= 64); every
// lane of a group receives the group's max. DPP within a 16-lane row, bpermute across rows.
__device__ inline float group_max(float v, int width) {
  if (width > 1) v = fmaxf(v, __shfl_xor(v, 1));
  if (width > 2) v = fmaxf(v, __shfl_xor(v, 2));
  if (width > 4) v = fmaxf(v, __shfl_xor(v, 4));
  if (width > 8) v = fmaxf(v, __shfl_xor(v, 8));
  if (width > 16) v = fmaxf(v, __shfl_xor(v, 16));
  if (width > 32) v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}

__device__ inline float wave_max(float v) { return group_max(v, 64); }

// block-wide max of non-negative floats through LDS (s_red: >= kBlock/kWave floats)
__device__ inline float block_max_nonneg(float v, float* s_red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  float r = s_red[0];
#pragma unroll
  for (int i = 1; i < kBlock / kWave; ++i) r = fmaxf(r, s_red[i]);
  __syncthreads();
  return r;
}

// ------------------------------------------------------------------ the reference's quantiser, one element
template <int BITS>
struct QRange;
template <>
struct QRange<8> {
  static constexpr float qmax = 127.0f, qmin = -127.0f;
};
template <>
struct QRange<4> {
  static constexpr float qmax = 7.0f, qmin = -8.0f;
};

template <int BITS>
__device__ inline int quant1(float x, float s32) {
  const float r = rintf(x / s32);  // IEEE fp32 divide + round-half-even, as torch does
  return (int)fminf(fmaxf(r, QRange<BITS>::qmin), QRange<BITS>::qmax);
}

// K and V of ONE new token (a decode step's append, ops.py:323-330): slice w (0 = K, 1 = V) of
// [B,H,1,D] is quantised by ONE workgroup of kBlock threads into slot T of its store. The slices are
// a few KB, so the arithmetic is the reference's own (IEEE divide per element), bit-identical with
// the bulk kernels by construction. Runs as two extra workgroups of the attention merge launch
// (kvq_decode_step) or as its own two-workgroup launch.
struct NewTokenArgs {
  const void* x[2];
  int64_t xs_b[2], xs_h[2];  // elements
  uint8_t* q[2];             // slot T of the store
  int64_t qs_b[2], qs_h[2];  // bytes
  int64_t qs_t[2];           // bytes per token row (device-side slot: kvq_decode_step_dev)
  float* scale[2];           // &scales[T]
  int32_t bits[2];
  uint32_t B, H, D;
  float eps;
  uint32_t nparts;           // > 1: workgroups per slice (quant_new_token_parts, merge.inc: slices past the register path); 0 / 1: one
};

template <int IDT>
__device__ inline void quant_new_token_block(const NewTokenArgs& a, uint32_t w, float* s_red) {
  const uint32_t tid = threadIdx.x;
  const void* x = a.x[w];
  const uint32_t n = a.B * a.H * a.D;
  float m = 0.0f;
  for (uint32_t i = tid; i < n; i += kBlock) {
    const uint32_t d = i % a.D, r = i / a.D;
    m = fmaxf(m, fabsf(load1<IDT>(x, (int64_t)(r / a.H) * a.xs_b[w] + (int64_t)(r % a.H) * a.xs_h[w] + d)));
  }
  m = block_max_nonneg(m, s_red);
  if (a.bits[w] == 8) {
    const float s32 = fmaxf(m / QRange<8>::qmax, a.eps);
    if (tid == 0) *a.scale[w] = Elem<IDT>::round_trip(s32);
    for (uint32_t i = tid; i < n; i += kBlock) {
      const uint32_t d = i % a.D, r = i / a.D;
      const float v = load1<IDT>(x, (int64_t)(r / a.H) * a.xs_b[w] + (int64_t)(r % a.H) * a.xs_h[w] + d);
      a.q[w][(int64_t)(r / a.H) * a.qs_b[w] + (int64_t)(r % a.H) * a.qs_h[w] + d] = (uint8_t)(int8_t)quant1<8>(v, s32);
    }
  } else {
    const float s32 = fmaxf(m / QRange<4>::qmax, a.eps);
    if (tid == 0) *a.scale[w] = Elem<IDT>::round_trip(s32);
    const uint32_t Dq = (a.D + 1) / 2;
    for (uint32_t i = tid; i < a.B * a.H * Dq; i += kBlock) {
      const uint32_t j = i % Dq, r = i / Dq;
      const int64_t xo = (int64_t)(r / a.H) * a.xs_b[w] + (int64_t)(r % a.H) * a.xs_h[w];
      const int hi = quant1<4>(load1<IDT>(x, xo + 2 * j), s32) + 8;
      const int lo = 2 * j + 1 < a.D ? quant1<4>(load1<IDT>(x, xo + 2 * j + 1), s32) + 8 : 8;
      a.q[w][(int64_t)(r / a.H) * a.qs_b[w] + (int64_t)(r % a.H) * a.qs_h[w] + j] = (uint8_t)(((hi & 0xF) << 4) | (lo & 0xF));
    }
  }
}

// Workgroup id -> work item with `k` consecutive items per XCD: the dispatcher deals workgroups round robin
// over the 8 XCDs (MI355X_MICROARCH: blocks b and b + 8 share one), so with the identity map XCD x sees
// every 8th item; this map gives it runs of k consecutive items (k x the item's bytes contiguous per XCD L2)
// while all XCDs stay inside one window of 8 k items. A speed knob only: any k gives the same results.
// Items past the last complete window keep the identity map.
__device__ inline uint32_t xcd_grouped_item(uint32_t wg, uint32_t k, uint32_t total) {
  if (k <= 1u) return wg;
  const uint32_t win = 8u * k;
  const uint32_t base = wg / win * win;
  if (base + win > total) return wg;
  const uint32_t r = wg - base;
  return base + (r & 7u) * k + (r >> 3);
}

static inline int ilog2_exact(int64_t v) {  // log2 if power of two else -1
  if (v <= 0 || (v & (v - 1))) return -1;
  int s = 0;
  while ((int64_t(1) << s) < v) ++s;
  return s;
}

static inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace kvq
