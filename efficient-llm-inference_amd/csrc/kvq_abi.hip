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

static thread_local TimingEvents g_timing = {nullptr, nullptr};
TimingEvents take_timing_events() {
  const TimingEvents e = g_timing;
  g_timing = TimingEvents{nullptr, nullptr};
  return e;
}

Tunables& tunables() {
  static Tunables t = [] {
    Tunables d = {};  // every knob 0 unless named here
    d.dequant_variant = -1;
    d.pool_block = 64;
    d.pool_wave = 1;
    d.quant_block = 64;
    d.quant_nv = 8;
    d.attn_mfma_min_nq = 3;
    d.attn_mfma_tc = 128;
    d.nt_loads = 1;
    d.quant_nt_stores = -1;
    d.quant_geo128 = 1;
    d.attn_k_i8 = -1;
    d.attn_merge_fast = 1;
    d.attn_stream_roll = 1;
    return d;
  }();
  return t;
}

// name -> field, one table for kvq_set_tunable and kvq_get_tunable
struct TunableKey {
  const char* name;
  int64_t Tunables::*field;
};
static const TunableKey kTunableKeys[] = {
    {"dequant_variant", &Tunables::dequant_variant},
    {"dequant_grid", &Tunables::dequant_grid},
    {"dequant_xcd_group", &Tunables::dequant_xcd_group},
    {"quant_xcd_group", &Tunables::quant_xcd_group},
    {"quant_force_two_pass", &Tunables::quant_force_two_pass},
    {"quant_direct_stores", &Tunables::quant_direct_stores},
    {"pool_grid", &Tunables::pool_grid},
    {"nt_loads", &Tunables::nt_loads},
    {"quant_block", &Tunables::quant_block},
    {"pool_block", &Tunables::pool_block},
    {"pool_wave", &Tunables::pool_wave},
    {"quant_no_regmax", &Tunables::quant_no_regmax},
    {"quant_nv", &Tunables::quant_nv},
    {"quant_lds_pad", &Tunables::quant_lds_pad},
    {"quant_tpw", &Tunables::quant_tpw},
    {"quant_nt_stores", &Tunables::quant_nt_stores},
    {"quant_geo128", &Tunables::quant_geo128},
    {"attn_force_valu", &Tunables::attn_force_valu},
    {"attn_mfma_min_nq", &Tunables::attn_mfma_min_nq},
    {"attn_mfma_tc", &Tunables::attn_mfma_tc},
    {"attn_fused", &Tunables::attn_fused},
    {"attn_k_i8", &Tunables::attn_k_i8},
    {"attn_merge_fast", &Tunables::attn_merge_fast},
    {"attn_stream_roll", &Tunables::attn_stream_roll},
    {"attn_stream_tpw", &Tunables::attn_stream_tpw},
    {"attn_stream_slots", &Tunables::attn_stream_slots},
    {"attn_stream_tc", &Tunables::attn_stream_tc},
    {"attn_fused_tc", &Tunables::attn_fused_tc},
    {"attn_fused_nw", &Tunables::attn_fused_nw},
};
static int64_t* tunable_slot(const char* key) {
  for (const TunableKey& k : kTunableKeys)
    if (!strcmp(key, k.name)) return &(tunables().*(k.field));
  return nullptr;
}

}  // namespace kvq

extern "C" {

int kvq_version(void) { return KVQ_VERSION; }

const char* kvq_last_error_string(void) { return kvq::g_err; }

int kvq_time_next_launch(void* start_event, void* stop_event) {
  kvq::g_timing = kvq::TimingEvents{reinterpret_cast<hipEvent_t>(start_event), reinterpret_cast<hipEvent_t>(stop_event)};
  return 0;
}

int kvq_set_tunable(const char* key, int64_t value) {
  if (!key) return KVQ_E_NULL;
  int64_t* slot = kvq::tunable_slot(key);
  if (!slot) {
    kvq::set_error("kvq_set_tunable: unknown key '%s'", key);
    return KVQ_E_DIMS;
  }
  *slot = value;
  return 0;
}

int64_t kvq_get_tunable(const char* key) {
  if (!key) return 0;
  const int64_t* slot = kvq::tunable_slot(key);
  return slot ? *slot : 0;
}

int64_t kvq_chunk_summary_len(int64_t T, int64_t chunk_size, int64_t keep_last) {
  if (T < 0 || chunk_size <= 0 || keep_last < 0) return -1;
  const int64_t keep = keep_last < T ? keep_last : T;
  const int64_t old = T - keep;
  if (old <= 0) return T;
  return (old + chunk_size - 1) / chunk_size + keep;
}

}  // extern "C"
