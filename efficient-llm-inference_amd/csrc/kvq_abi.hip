// kvq_abi.hip — version, error reporting and tunables of libkvq_hip.so (include/kvq_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "kvq_common.h"

namespace kvq {

static thread_local char g_err[512] = "no error";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// The reference never checks its launches (extensions.py:79,105); we do, every time.
int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s (%d)", what, hipGetErrorString(e), (int)e);
    return (int)e;
  }
  return 0;
}

Tunables& tunables() {
  static Tunables t = {-1, 0, 0, 0, 0, 64, 64, 8, 0, 0, 0, 3, 128, 1};
  return t;
}

}  // namespace kvq

extern "C" {

int kvq_version(void) { return KVQ_VERSION; }

const char* kvq_last_error_string(void) { return kvq::g_err; }

int kvq_set_tunable(const char* key, int64_t value) {
  if (!key) return KVQ_E_NULL;
  kvq::Tunables& t = kvq::tunables();
  if (!strcmp(key, "dequant_variant")) t.dequant_variant = value;
  else if (!strcmp(key, "dequant_grid")) t.dequant_grid = value;
  else if (!strcmp(key, "quant_force_two_pass")) t.quant_force_two_pass = value;
  else if (!strcmp(key, "quant_direct_stores")) t.quant_direct_stores = value;
  else if (!strcmp(key, "pool_grid")) t.pool_grid = value;
  else if (!strcmp(key, "nt_loads")) t.nt_loads = value;
  else if (!strcmp(key, "quant_block")) t.quant_block = value;
  else if (!strcmp(key, "pool_block")) t.pool_block = value;
  else if (!strcmp(key, "quant_no_regmax")) t.quant_no_regmax = value;
  else if (!strcmp(key, "quant_nv")) t.quant_nv = value;
  else if (!strcmp(key, "quant_lds_pad")) t.quant_lds_pad = value;
  else if (!strcmp(key, "attn_force_valu")) t.attn_force_valu = value;
  else if (!strcmp(key, "attn_mfma_min_nq")) t.attn_mfma_min_nq = value;
  else if (!strcmp(key, "attn_mfma_tc")) t.attn_mfma_tc = value;
  else {
    kvq::set_error("kvq_set_tunable: unknown key '%s'", key);
    return KVQ_E_DIMS;
  }
  return 0;
}

int64_t kvq_get_tunable(const char* key) {
  if (!key) return 0;
  kvq::Tunables& t = kvq::tunables();
  if (!strcmp(key, "dequant_variant")) return t.dequant_variant;
  if (!strcmp(key, "dequant_grid")) return t.dequant_grid;
  if (!strcmp(key, "quant_force_two_pass")) return t.quant_force_two_pass;
  if (!strcmp(key, "quant_direct_stores")) return t.quant_direct_stores;
  if (!strcmp(key, "pool_grid")) return t.pool_grid;
  if (!strcmp(key, "nt_loads")) return t.nt_loads;
  if (!strcmp(key, "quant_block")) return t.quant_block;
  if (!strcmp(key, "pool_block")) return t.pool_block;
  if (!strcmp(key, "quant_no_regmax")) return t.quant_no_regmax;
  if (!strcmp(key, "quant_nv")) return t.quant_nv;
  if (!strcmp(key, "quant_lds_pad")) return t.quant_lds_pad;
  if (!strcmp(key, "attn_force_valu")) return t.attn_force_valu;
  if (!strcmp(key, "attn_mfma_min_nq")) return t.attn_mfma_min_nq;
  if (!strcmp(key, "attn_mfma_tc")) return t.attn_mfma_tc;
  return 0;
}

int64_t kvq_chunk_summary_len(int64_t T, int64_t chunk_size, int64_t keep_last) {
  if (T < 0 || chunk_size <= 0 || keep_last < 0) return -1;
  const int64_t keep = keep_last < T ? keep_last : T;
  const int64_t old = T - keep;
  if (old <= 0) return T;
  return (old + chunk_size - 1) / chunk_size + keep;
}

}  // extern "C"
