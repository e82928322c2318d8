// kvq_abi.hip — version, error reporting and tunables of libkvq_hip.so (include/kvq_hip.h).
#include <cxxabi.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kvq_common.h"

namespace kvq {

static thread_local char g_err[512] = "no error";
static thread_local TimingEvents g_timing = {nullptr, nullptr};

// Every failing return of the library passes through here; it also DISARMS a pending kvq_time_next_launch pair, so that
// a call that failed validation cannot leave its events to be taken by some later launch of the thread.
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  g_timing = TimingEvents{nullptr, nullptr};
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

// Device guard: the library launches on the calling thread's CURRENT device; a buffer that lives on
// another device of the same process (a single-process multi-GPU caller) must be refused, not dereferenced by
// the wrong GPU. One hipPointerGetAttributes per entry point on the buffer the call writes. A pointer the runtime does not
// know (a host test's fake address, no device at all) is left to the later checks / the launch.
int check_device(const void* p, const char* name) {
  if (!p) return 0;
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  if (at.type == hipMemoryTypeDevice && at.device != cur) {
    set_error("%s: the output buffer lives on device %d but the calling thread's current device is %d (hipSetDevice / "
              "torch.cuda.device(...) before the call; the stream argument must belong to that device too)", name, at.device, cur);
    return KVQ_E_DEVICE;
  }
  return 0;
}

TimingEvents take_timing_events() {
  const TimingEvents e = g_timing;
  g_timing = TimingEvents{nullptr, nullptr};
  return e;
}

// kernels launched by the calling thread since kvq_kernel_log_clear(): distinct host stubs, in first-launch order
constexpr int kLogCap = 32;
static thread_local const void* g_log[kLogCap];
static thread_local int g_log_n = 0;
void note_launch(const void* host_stub) {
  for (int i = 0; i < g_log_n; ++i)
    if (g_log[i] == host_stub) return;
  if (g_log_n < kLogCap) g_log[g_log_n++] = host_stub;
}

Tunables& tunables() {
  static Tunables t = [] {
    Tunables d = {};  // every knob 0 unless named here
    d.dequant_variant = -1;
    d.pool_block = 64;
    d.pool_wave = 1;
    d.gather_rows = 1;
    d.attn_ring_dev = 1;
    d.attn_new_token_parts = 1;
    d.quant_few_tokens = 16;  // measured crossover against the tile walk: profiles/r04aa_absmax_few_tokens_sweep.jsonl
    d.quant_block = 64;
    d.quant_nv = 8;
    d.attn_mfma_min_nq = 1;
    d.attn_mfma_tc = 128;
    d.nt_loads = 1;
    d.quant_nt_stores = -1;
    d.quant_tile = 1;
    d.quant_wide = 1;
    d.attn_lds = -1;
    d.attn_merge_wave = 1;
    d.attn_k_i8 = -1;
    d.attn_merge_fast = 1;
    d.attn_stream_roll = 1;
    return d;
  }();
  return t;
}

// name -> field, one table for kvq_set_tunable and kvq_get_tunable. `ab`: the key selects code that exists in
// A-B builds only (`make ab`); the default library refuses to set it.
struct TunableKey {
  const char* name;
  int64_t Tunables::*field;
  bool ab;
};
static const TunableKey kTunableKeys[] = {
    // test knobs of the default library: they route a call to SHIPPED code it would not take by size / shape
    {"quant_force_two_pass", &Tunables::quant_force_two_pass, false},
    {"quant_direct_stores", &Tunables::quant_direct_stores, false},
    {"quant_tile", &Tunables::quant_tile, false},
    {"quant_wide", &Tunables::quant_wide, false},
    {"quant_block", &Tunables::quant_block, false},
    {"pool_wave", &Tunables::pool_wave, false},
    {"gather_rows", &Tunables::gather_rows, false},
    {"attn_ring_dev", &Tunables::attn_ring_dev, false},
    {"attn_new_token_parts", &Tunables::attn_new_token_parts, false},
    {"quant_few_tokens", &Tunables::quant_few_tokens, false},
    {"attn_force_valu", &Tunables::attn_force_valu, false},
    {"attn_stream_tpw", &Tunables::attn_stream_tpw, false},
    {"attn_lds", &Tunables::attn_lds, false},
    {"attn_merge_wave", &Tunables::attn_merge_wave, false},
    // A-B keys
    {"dequant_variant", &Tunables::dequant_variant, true},
    {"dequant_grid", &Tunables::dequant_grid, true},
    {"dequant_xcd_group", &Tunables::dequant_xcd_group, true},
    {"quant_xcd_group", &Tunables::quant_xcd_group, true},
    {"pool_grid", &Tunables::pool_grid, true},
    {"nt_loads", &Tunables::nt_loads, true},
    {"pool_block", &Tunables::pool_block, true},
    {"quant_no_regmax", &Tunables::quant_no_regmax, true},
    {"quant_nv", &Tunables::quant_nv, true},
    {"quant_lds_pad", &Tunables::quant_lds_pad, true},
    {"quant_tpw", &Tunables::quant_tpw, true},
    {"quant_nt_stores", &Tunables::quant_nt_stores, true},
    {"quant_wide_blk", &Tunables::quant_wide_blk, true},
    {"quant_geo128", &Tunables::quant_geo128, true},
    {"quant_tile_tt", &Tunables::quant_tile_tt, true},
    {"quant_tile_tpw", &Tunables::quant_tile_tpw, true},
    {"attn_mfma_min_nq", &Tunables::attn_mfma_min_nq, true},
    {"attn_mfma_tc", &Tunables::attn_mfma_tc, true},
    {"attn_fused", &Tunables::attn_fused, true},
    {"attn_fold", &Tunables::attn_fold, true},
    {"attn_onepass", &Tunables::attn_onepass, true},
    {"attn_k_i8", &Tunables::attn_k_i8, true},
    {"attn_merge_fast", &Tunables::attn_merge_fast, true},
    {"attn_stream_roll", &Tunables::attn_stream_roll, true},
    {"attn_lds_nb", &Tunables::attn_lds_nb, true},
    {"attn_tg", &Tunables::attn_tg, true},
    {"attn_lds_tc", &Tunables::attn_lds_tc, true},
    {"attn_stream_slots", &Tunables::attn_stream_slots, true},
    {"attn_stream_tc", &Tunables::attn_stream_tc, true},
    {"attn_fused_tc", &Tunables::attn_fused_tc, true},
    {"attn_fused_nw", &Tunables::attn_fused_nw, true},
};
static const TunableKey* tunable_key(const char* key) {
  for (const TunableKey& k : kTunableKeys)
    if (!strcmp(key, k.name)) return &k;
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

int kvq_timing_armed(void) { return (kvq::g_timing.start || kvq::g_timing.stop) ? 1 : 0; }

int kvq_set_tunable(const char* key, int64_t value) {
  if (!key) return KVQ_E_NULL;
  const kvq::TunableKey* k = kvq::tunable_key(key);
  if (!k) {
    kvq::set_error("kvq_set_tunable: unknown key '%s'", key);
    return KVQ_E_DIMS;
  }
  if (k->ab && !KVQ_AB) {
    kvq::set_error("kvq_set_tunable: '%s' is an A-B key; this is the default library (build `make ab`)", key);
    return KVQ_E_DIMS;
  }
  kvq::tunables().*(k->field) = value;
  return 0;
}

int64_t kvq_get_tunable(const char* key) {
  if (!key) return 0;
  const kvq::TunableKey* k = kvq::tunable_key(key);
  return k ? kvq::tunables().*(k->field) : 0;
}

int kvq_is_ab_build(void) { return KVQ_AB; }

void kvq_kernel_log_clear(void) { kvq::g_log_n = 0; }

int64_t kvq_kernel_log(char* buf, int64_t n) {
  if (!buf || n <= 0) return kvq::g_log_n;
  int64_t off = 0;
  buf[0] = 0;
  for (int i = 0; i < kvq::g_log_n; ++i) {
    const char* mangled = hipKernelNameRefByPtr(kvq::g_log[i], nullptr);
    int status = -1;
    char* dem = mangled ? abi::__cxa_demangle(mangled, nullptr, nullptr, &status) : nullptr;
    const char* name = (status == 0 && dem) ? dem : (mangled ? mangled : "?");
    const int w = snprintf(buf + off, (size_t)(n - off), "%s%s", i ? "\n" : "", name);
    free(dem);
    if (w < 0 || off + w >= n) break;
    off += w;
  }
  return kvq::g_log_n;
}

int64_t kvq_chunk_summary_len(int64_t T, int64_t chunk_size, int64_t keep_last) {
  if (T < 0 || chunk_size <= 0 || keep_last < 0) return -1;
  const int64_t keep = keep_last < T ? keep_last : T;
  const int64_t old = T - keep;
  if (old <= 0) return T;
  return (old + chunk_size - 1) / chunk_size + keep;
}

}  // extern "C"
