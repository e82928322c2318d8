/*
 * kvq_oracle.c — plain-C CPU restatement of the reference's KV quantize / dequantize /
 * chunk-summary arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY: it is the checker (tests/, __graft_entry__.smoke()) and the
 * `cpu_baseline` leg of bench.py ("kind": "port": scalar code, timed on 1 core and — the `_mt`
 * entry points, the same loops cut into token ranges over pthreads — on every usable host core).
 * The product path (efficient-llm-inference_amd/) never links, loads or calls it.
 *
 * Parity status: PINNED — tests/test_oracle_c.py checks every function here against the numpy
 * oracle (oracle/kvq_oracle.py), which is itself pinned to golden vectors captured from the
 * reference (the npz fixtures under tests/golden), and against those goldens directly.
 *
 * Layout everywhere: contiguous [G, B, H, T, D], one scale per (g, t) over all B*H*D values —
 * the slice QuantizedLayerKV.append sees (reference src/quantization/ops.py:174-210).
 * dtype codes: 0 = fp16 (uint16 bits), 2 = fp32.  (bf16 is covered by the numpy oracle.)
 *
 * Build: make -C oracle   (gcc -O2, no -ffast-math: IEEE division and rint are the point).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <string.h>

/* ---- fp16 <-> fp32, bit exact (round-to-nearest-even), no compiler half type needed -------- */

static float h2f(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1Fu;
  uint32_t man = h & 0x3FFu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else { /* subnormal: normalise */
      int e = -1;
      do {
        e++;
        man <<= 1;
      } while ((man & 0x400u) == 0);
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7F800000u | (man << 13);
  } else {
    bits = sign | ((exp + 112u) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

static uint16_t f2h(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t ax = x & 0x7FFFFFFFu;
  if (ax >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | ((ax > 0x7F800000u) ? 0x200u : 0u));
  if (ax >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u); /* >= 65520 rounds to inf */
  if (ax < 0x33000001u) return (uint16_t)sign;              /* <= 2^-25 rounds to 0 (ties-to-even) */
  int32_t e = (int32_t)(ax >> 23) - 127;
  uint32_t m = (ax & 0x7FFFFFu) | 0x800000u;
  if (e < -14) { /* subnormal half */
    int shift = -14 - e + 13;
    uint32_t r = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1u))) r++;
    return (uint16_t)(sign | r);
  }
  uint32_t r = ((uint32_t)(e + 15) << 10) | ((m >> 13) & 0x3FFu);
  uint32_t rem = m & 0x1FFFu;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;
  return (uint16_t)(sign | r);
}

static inline float load_in(const void* p, int64_t i, int dtype) {
  return dtype == 2 ? ((const float*)p)[i] : h2f(((const uint16_t*)p)[i]);
}
static inline void store_out(void* p, int64_t i, float v, int dtype) {
  if (dtype == 2) ((float*)p)[i] = v;
  else ((uint16_t*)p)[i] = f2h(v);
}
static inline float round_to_dtype(float v, int dtype) { return dtype == 2 ? v : h2f(f2h(v)); }

uint16_t kvq_oracle_f2h(float f) { return f2h(f); }
float kvq_oracle_h2f(uint16_t h) { return h2f(h); }

/* ---- quantise: ops.py:10-65 applied per [B,H,1,D] slice (ops.py:339-342) ------------------- */
/* bits = 8: q int8 [G,B,H,T,D]; bits = 4: packed uint8 [G,B,H,T,ceil(D/2)], even d = HIGH nibble
 * (ops.py:61-63), odd D padded with q = 0 -> nibble 8 (ops.py:54-59).
 * scales_f32[g*T+t] = the stored scale (rounded to the input dtype, ops.py:30,65) widened.
 * Work item = one (g, t) slice; items [i0, i1) of the G*T. */
typedef struct {
  const void* x; int dtype, bits; int64_t G, B, H, T, D; float eps; uint8_t* q; float* scales_f32;
} quant_job_t;

static void quant_range(const quant_job_t* j, int64_t i0, int64_t i1) {
  const void* x = j->x;
  const int dtype = j->dtype, bits = j->bits;
  const int64_t B = j->B, H = j->H, T = j->T, D = j->D;
  const float eps = j->eps;
  const float qmax = bits == 8 ? 127.0f : 7.0f, qmin = bits == 8 ? -127.0f : -8.0f;
  const int64_t Dq = bits == 8 ? D : (D + 1) / 2;
  for (int64_t it = i0; it < i1; ++it) {
    const int64_t g = it / T, t = it % T;
    float amax = 0.0f; /* x.float().abs().max()  (ops.py:26-27, :47-48) */
    for (int64_t b = 0; b < B; ++b)
      for (int64_t h = 0; h < H; ++h) {
        const int64_t base = (((g * B + b) * H + h) * T + t) * D;
        for (int64_t d = 0; d < D; ++d) {
          const float a = fabsf(load_in(x, base + d, dtype));
          if (a > amax) amax = a;
        }
      }
    float s32 = amax / qmax; /* (max_abs / 127).clamp(min=eps)  (ops.py:28, :49) */
    if (s32 < eps) s32 = eps;
    j->scales_f32[g * T + t] = round_to_dtype(s32, dtype);
    for (int64_t b = 0; b < B; ++b)
      for (int64_t h = 0; h < H; ++h) {
        const int64_t base = (((g * B + b) * H + h) * T + t) * D;
        uint8_t* qrow = j->q + (((g * B + b) * H + h) * T + t) * Dq;
        if (bits == 4) memset(qrow, 0, (size_t)Dq);
        for (int64_t d = 0; d < D; ++d) {
          float r = rintf(load_in(x, base + d, dtype) / s32); /* true division, half-to-even */
          if (r < qmin) r = qmin;
          if (r > qmax) r = qmax;
          const int v = (int)r;
          if (bits == 8) qrow[d] = (uint8_t)(int8_t)v;
          else qrow[d >> 1] |= (uint8_t)(((v + 8) & 0xF) << ((d & 1) ? 0 : 4));
        }
        if (bits == 4 && (D & 1)) qrow[D >> 1] |= 8; /* pad element q=0 -> nibble 8, low half */
      }
  }
}

/* ---- dequantise: ops.py:68-133 / extensions.py:37-68 per slice + cat (ops.py:213-269) --------
 * Work item = one (g, b, h, t) row of D values; items [i0, i1) of the G*B*H*T. */
typedef struct {
  const uint8_t* q; const float* scales_f32; int bits; int64_t G, B, H, T, D; void* out; int out_dtype;
} dequant_job_t;

static void dequant_range(const dequant_job_t* j, int64_t i0, int64_t i1) {
  const int bits = j->bits;
  const int64_t B = j->B, H = j->H, T = j->T, D = j->D;
  const int64_t Dq = bits == 8 ? D : (D + 1) / 2;
  for (int64_t it = i0; it < i1; ++it) {
    const int64_t t = it % T, g = it / (B * H * T);
    const float s = j->scales_f32[g * T + t];
    const uint8_t* qrow = j->q + it * Dq;
    const int64_t obase = it * D;
    for (int64_t d = 0; d < D; ++d) {
      int v;
      if (bits == 8) v = (int)(int8_t)qrow[d];
      else v = (int)((d & 1) ? (qrow[d >> 1] & 0x0F) : (qrow[d >> 1] >> 4)) - 8;
      store_out(j->out, obase + d, (float)v * s, j->out_dtype);
    }
  }
}

/* ---- chunk summary: implementations.py:295-346 (fp32 accumulate, sequential in t) -----------
 * x [R, T, D] -> out [R, Tout, D], Tout = ceil(old/chunk) + keep, old = T - min(keep_last, T).
 * Work item = one OUTPUT row (r, j): a summary row (j < n) or a copied recent row; items [i0, i1) of R*Tout. */
typedef struct {
  const void* x; int dtype; int64_t R, T, D, chunk, keep_last; void* out;
} pool_job_t;

static void pool_range(const pool_job_t* p, int64_t i0, int64_t i1) {
  const void* x = p->x;
  const int dtype = p->dtype;
  const int64_t T = p->T, D = p->D, chunk = p->chunk;
  const int64_t keep = p->keep_last < T ? p->keep_last : T;
  const int64_t old = T - keep;
  const int64_t n = old > 0 ? (old + chunk - 1) / chunk : 0;
  const int64_t Tout = n + keep;
  for (int64_t it = i0; it < i1; ++it) {
    const int64_t r = it / Tout, j = it % Tout;
    if (j < n) {
      for (int64_t d = 0; d < D; ++d) {
        float acc = 0.0f;
        for (int64_t i = 0; i < chunk && j * chunk + i < old; ++i) acc += load_in(x, (r * T + j * chunk + i) * D + d, dtype);
        store_out(p->out, (r * Tout + j) * D + d, acc / (float)chunk, dtype); /* divisor = chunk_size always */
      }
    } else {
      const int64_t i = j - n;
      for (int64_t d = 0; d < D; ++d)
        store_out(p->out, (r * Tout + n + i) * D + d, load_in(x, (r * T + old + i) * D + d, dtype), dtype);
    }
  }
}

/* ---- the same loops over n_threads pthreads: contiguous item ranges, no shared writes -------- */
typedef struct { int kind; const void* job; int64_t i0, i1; } slice_t;

static void* run_slice(void* arg) {
  const slice_t* s = (const slice_t*)arg;
  if (s->kind == 0) quant_range((const quant_job_t*)s->job, s->i0, s->i1);
  else if (s->kind == 1) dequant_range((const dequant_job_t*)s->job, s->i0, s->i1);
  else pool_range((const pool_job_t*)s->job, s->i0, s->i1);
  return 0;
}

#define KVQ_ORACLE_MAX_THREADS 512
static void run_items(int kind, const void* job, int64_t n_items, int n_threads) {
  if (n_threads > KVQ_ORACLE_MAX_THREADS) n_threads = KVQ_ORACLE_MAX_THREADS;
  if ((int64_t)n_threads > n_items) n_threads = (int)n_items;
  if (n_threads <= 1) {
    const slice_t all = {kind, job, 0, n_items};
    run_slice((void*)&all);
    return;
  }
  pthread_t th[KVQ_ORACLE_MAX_THREADS];
  slice_t sl[KVQ_ORACLE_MAX_THREADS];
  int started[KVQ_ORACLE_MAX_THREADS];
  for (int k = 0; k < n_threads; ++k) {
    sl[k].kind = kind;
    sl[k].job = job;
    sl[k].i0 = n_items * k / n_threads;
    sl[k].i1 = n_items * (k + 1) / n_threads;
    started[k] = k > 0 && pthread_create(&th[k], 0, run_slice, &sl[k]) == 0;
  }
  run_slice(&sl[0]); /* the calling thread takes the first range */
  for (int k = 1; k < n_threads; ++k) {
    if (started[k]) pthread_join(th[k], 0);
    else run_slice(&sl[k]); /* a thread that could not start: its range is done here */
  }
}

void kvq_oracle_quant_tokens_mt(const void* x, int dtype, int bits, int64_t G, int64_t B, int64_t H, int64_t T,
                                int64_t D, float eps, uint8_t* q, float* scales_f32, int n_threads) {
  const quant_job_t j = {x, dtype, bits, G, B, H, T, D, eps, q, scales_f32};
  run_items(0, &j, G * T, n_threads);
}
void kvq_oracle_quant_tokens(const void* x, int dtype, int bits, int64_t G, int64_t B, int64_t H, int64_t T,
                             int64_t D, float eps, uint8_t* q, float* scales_f32) {
  kvq_oracle_quant_tokens_mt(x, dtype, bits, G, B, H, T, D, eps, q, scales_f32, 1);
}

void kvq_oracle_dequant_tokens_mt(const uint8_t* q, const float* scales_f32, int bits, int64_t G, int64_t B, int64_t H,
                                  int64_t T, int64_t D, void* out, int out_dtype, int n_threads) {
  const dequant_job_t j = {q, scales_f32, bits, G, B, H, T, D, out, out_dtype};
  run_items(1, &j, G * B * H * T, n_threads);
}
void kvq_oracle_dequant_tokens(const uint8_t* q, const float* scales_f32, int bits, int64_t G, int64_t B, int64_t H,
                               int64_t T, int64_t D, void* out, int out_dtype) {
  kvq_oracle_dequant_tokens_mt(q, scales_f32, bits, G, B, H, T, D, out, out_dtype, 1);
}

void kvq_oracle_chunk_summarize_mt(const void* x, int dtype, int64_t R, int64_t T, int64_t D, int64_t chunk,
                                   int64_t keep_last, void* out, int n_threads) {
  const pool_job_t p = {x, dtype, R, T, D, chunk, keep_last, out};
  const int64_t keep = keep_last < T ? keep_last : T;
  const int64_t old = T - keep;
  const int64_t n = old > 0 ? (old + chunk - 1) / chunk : 0;
  run_items(2, &p, R * (n + keep), n_threads);
}
void kvq_oracle_chunk_summarize(const void* x, int dtype, int64_t R, int64_t T, int64_t D, int64_t chunk,
                                int64_t keep_last, void* out) {
  kvq_oracle_chunk_summarize_mt(x, dtype, R, T, D, chunk, keep_last, out, 1);
}
