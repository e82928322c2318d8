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
// Split-T flash decoding:
//   decode_attn_partial_k       VALU kernel (any supported head_dim, <= 8 query heads per kv head):
//                               grid (split, kv head, batch), 256 threads, TS tokens per workgroup, all
//                               Hq/Hkv query heads of the kv head in one pass (K/V read once); 16
//                               elements per lane per token: 16-byte (INT8) / 8-byte (INT4) loads
//   decode_attn_partial_mfma_k  head_dim 64 / 128 with 3..16 query heads per kv head: both products on the
//                               matrix cores, one wave per 128-token split (see its header below)
//   decode_attn_stream_mfma_k   the same tile arithmetic for larger batches (more 64-token tiles than wave slots):
//                               one wave walks several tiles, online softmax, ONE partial per wave; the registers
//                               of the tile being reduced are re-requested for the tile after next (ROLL)
//   all write (m, l) and acc[D] per (batch, query head, split) to the caller's workspace
//   decode_attn_merge_fast_k    grid (query head, batch): log-sum-exp merge of the splits and of the exact new
//                               token, every operand requested up front (<= 256 splits); two extra workgroups
//                               quantise the new token into slot T of the stores (a decode step = two launches)
//   decode_attn_merge_k         the same arithmetic as a chain of dependent loads (any number of splits)
//   decode_attn_fused_mfma_k    opt-in single launch (in-workgroup merge + arrival ticket): measured slower
#include <atomic>
#include <type_traits>

#include "kvq_common.h"

// cache policy of the K / V row loads (buffer-load aux immediate: 2 = non-temporal, what ships; `make calib_attn`
// builds may override it for A-B runs)
#ifndef KVQ_ATTN_KV_AUX
#define KVQ_ATTN_KV_AUX 2
#endif

// One translation unit, cut into parts that are read on their own (csrc/attn/*.inc, included in this order; the *_ab parts hold
// only `#if KVQ_AB` code: variants that lost a measurement). host.inc closes the namespace before its extern "C" section.
namespace kvq {
#include "attn/common.inc"
#include "attn/partial_valu.inc"
#include "attn/tile_mfma.inc"
#include "attn/stream.inc"
#include "attn/fold_ab.inc"
#include "attn/ring.inc"
#include "attn/coal_ab.inc"
#include "attn/fused_ab.inc"
#include "attn/onepass_ab.inc"
#include "attn/merge.inc"
#include "attn/host.inc"
