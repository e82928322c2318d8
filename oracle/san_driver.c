/* Sanitizer driver of the C oracle (TEST INFRASTRUCTURE, like the oracle itself; never part of the product).
 *
 * `make -C oracle san` builds kvq_oracle.c twice around this main — AddressSanitizer + UBSan, and ThreadSanitizer — and
 * tests/test_oracle_c.py runs both: every entry point (1 thread and several), ragged shapes (odd D, T below a thread's
 * range, more threads than items, zero-sized inputs), exact-size heap buffers so that one byte out of range is reported.
 * Also checks that the threaded results equal the single-threaded ones byte for byte. Exit code 0 = clean.
 * (GPU AddressSanitizer is not available on the pool; the HIP kernels are checked by range-limited buffer descriptors
 * and the parity suite instead — DESIGN.md §5.) */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void kvq_oracle_quant_tokens_mt(const void* x, int dtype, int bits, int64_t G, int64_t B, int64_t H, int64_t T, int64_t D,
                                float eps, uint8_t* q, float* scales_f32, int n_threads);
void kvq_oracle_dequant_tokens_mt(const uint8_t* q, const float* scales_f32, int bits, int64_t G, int64_t B, int64_t H,
                                  int64_t T, int64_t D, void* out, int out_dtype, int n_threads);
void kvq_oracle_chunk_summarize_mt(const void* x, int dtype, int64_t R, int64_t T, int64_t D, int64_t chunk,
                                   int64_t keep_last, void* out, int n_threads);

static uint32_t rng = 12345u;
static uint32_t next_u32(void) { rng = rng * 1664525u + 1013904223u; return rng; }

/* dtype codes of the oracle: 0 fp16, 1 bf16, 2 fp32 */
static size_t esize(int dtype) { return dtype == 2 ? 4 : 2; }

static void* exact(size_t n) { return malloc(n ? n : 1); }  /* exact-size blocks: ASan's red zones sit right behind the data */

static void fill(void* p, size_t n_elts, int dtype) {
  for (size_t i = 0; i < n_elts; ++i) {
    if (dtype == 2) {
      float f = ((int32_t)(next_u32() >> 8) - (1 << 23)) / (float)(1 << 21);
      memcpy((char*)p + 4 * i, &f, 4);
    } else {  /* finite half / bfloat16 bit patterns: exponent kept below the maximum */
      uint16_t h = (uint16_t)(next_u32() >> 16);
      if (dtype == 0 && (h & 0x7C00u) == 0x7C00u) h &= (uint16_t)~0x0400u;
      if (dtype == 1 && (h & 0x7F80u) == 0x7F80u) h &= (uint16_t)~0x0080u;
      memcpy((char*)p + 2 * i, &h, 2);
    }
  }
}

static int check_case(int dtype, int bits, int64_t G, int64_t B, int64_t H, int64_t T, int64_t D, int threads) {
  const int64_t n = G * B * H * T * D, Dq = bits == 8 ? D : (D + 1) / 2, nq = G * B * H * T * Dq;
  void* x = exact((size_t)n * esize(dtype));
  fill(x, (size_t)n, dtype);
  uint8_t *q1 = exact((size_t)nq), *qn = exact((size_t)nq);
  float *s1 = exact((size_t)(G * T) * 4), *sn = exact((size_t)(G * T) * 4);
  kvq_oracle_quant_tokens_mt(x, dtype, bits, G, B, H, T, D, 1e-8f, q1, s1, 1);
  kvq_oracle_quant_tokens_mt(x, dtype, bits, G, B, H, T, D, 1e-8f, qn, sn, threads);
  int bad = memcmp(q1, qn, (size_t)nq) != 0 || memcmp(s1, sn, (size_t)(G * T) * 4) != 0;
  for (int od = 0; od < 3; ++od) {
    void *o1 = exact((size_t)n * esize(od)), *on = exact((size_t)n * esize(od));
    kvq_oracle_dequant_tokens_mt(q1, s1, bits, G, B, H, T, D, o1, od, 1);
    kvq_oracle_dequant_tokens_mt(q1, s1, bits, G, B, H, T, D, on, od, threads);
    bad |= memcmp(o1, on, (size_t)n * esize(od)) != 0;
    free(o1);
    free(on);
  }
  free(x); free(q1); free(qn); free(s1); free(sn);
  if (bad) fprintf(stderr, "threaded != single-threaded: dtype %d bits %d [%lld,%lld,%lld,%lld,%lld] x%d\n", dtype, bits,
                   (long long)G, (long long)B, (long long)H, (long long)T, (long long)D, threads);
  return bad;
}

static int check_pool(int dtype, int64_t R, int64_t T, int64_t D, int64_t chunk, int64_t keep_last, int threads) {
  const int64_t keep = keep_last < T ? keep_last : T, old = T - keep, nch = old > 0 ? (old + chunk - 1) / chunk : 0;
  const int64_t Tout = nch + keep;
  void* x = exact((size_t)(R * T * D) * esize(dtype));
  fill(x, (size_t)(R * T * D), dtype);
  void *o1 = exact((size_t)(R * Tout * D) * esize(dtype)), *on = exact((size_t)(R * Tout * D) * esize(dtype));
  kvq_oracle_chunk_summarize_mt(x, dtype, R, T, D, chunk, keep_last, o1, 1);
  kvq_oracle_chunk_summarize_mt(x, dtype, R, T, D, chunk, keep_last, on, threads);
  const int bad = memcmp(o1, on, (size_t)(R * Tout * D) * esize(dtype)) != 0;
  free(x); free(o1); free(on);
  if (bad) fprintf(stderr, "pool threaded != single: dtype %d R %lld T %lld D %lld chunk %lld keep %lld x%d\n", dtype, (long long)R,
                   (long long)T, (long long)D, (long long)chunk, (long long)keep_last, threads);
  return bad;
}

int main(void) {
  int bad = 0, cases = 0;
  static const int64_t shapes[][5] = {{1, 1, 1, 1, 1}, {2, 1, 3, 7, 5}, {3, 2, 4, 9, 64}, {1, 1, 8, 33, 128}, {2, 3, 1, 2, 31},
                                      {4, 1, 2, 1, 16}, {1, 2, 2, 100, 24}, {2, 1, 1, 5, 0}, {0, 1, 1, 5, 8}, {1, 1, 1, 0, 8}};
  static const int threads[] = {2, 3, 8, 37};
  for (size_t s = 0; s < sizeof shapes / sizeof shapes[0]; ++s)
    for (int dtype = 0; dtype < 3; ++dtype)
      for (int bits = 4; bits <= 8; bits += 4)
        for (size_t t = 0; t < sizeof threads / sizeof threads[0]; ++t, ++cases)
          bad += check_case(dtype, bits, shapes[s][0], shapes[s][1], shapes[s][2], shapes[s][3], shapes[s][4], threads[t]);
  static const int64_t pools[][5] = {{1, 40, 8, 8, 8}, {3, 45, 5, 8, 8}, {2, 6, 16, 8, 8}, {4, 33, 3, 4, 0}, {2, 300, 24, 64, 16},
                                     {1, 19, 7, 64, 3}, {5, 1, 1, 1, 0}, {2, 10, 4, 3, 100}};
  for (size_t s = 0; s < sizeof pools / sizeof pools[0]; ++s)
    for (int dtype = 0; dtype < 3; ++dtype)
      for (size_t t = 0; t < sizeof threads / sizeof threads[0]; ++t, ++cases)
        bad += check_pool(dtype, pools[s][0], pools[s][1], pools[s][2], pools[s][3], pools[s][4], threads[t]);
  printf("san_driver: %d cases, %d mismatches\n", cases, bad);
  return bad ? 1 : 0;
}
