/*
 * kvq_hip.h — C ABI of libkvq_hip.so: MI355X (gfx950) KV-cache quantize / dequantize /
 * eviction kernels.
 *
 * This is the drop-in boundary for the reference's native plugin `kvq_ext`
 * (reference src/cuda/extensions.py:116-119, a pybind11 module JIT-built from an inline
 * CUDA string) and for the torch-op arithmetic of src/quantization/ops.py and
 * src/cache/implementations.py that the MI355X build moves into HIP.
 *
 * Conventions
 *   - plain C: raw device pointers, sizes, strides; no torch / C++ types.
 *   - the library never allocates, frees or retains memory; every buffer is the caller's.
 *   - every entry point returns 0 on success, a negative KVQ_E_* code for argument errors,
 *     or a positive hipError_t for launch errors; kvq_last_error_string() describes the last
 *     failure on the calling thread.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream). All work is
 *     enqueued asynchronously; there is NO host synchronisation anywhere on the path
 *     (the reference syncs once per call through float(scale), ops.py:87,117).
 *   - stateless and re-entrant; safe to call from several host threads.
 *
 * KV views.  A KV tensor set is addressed as [G, B, H, T, D] with the last dim contiguous:
 *   G  independent scale groups (layer x K|V); never share a scale
 *   B,H batch rows and heads; ONE scale spans all B*H*D values of a token (ops.py:27,48:
 *       abs().max() over the whole [B,H,1,D] slice)
 *   T  tokens, D head_dim.
 * Strides are in ELEMENTS of the pointed-to type. The reference's legacy tuple
 * tuple_L[(k, v)] of [B,H,T,D] tensors (ops.py:178-179,217) is G = 2L views that live in
 * separate allocations: pass them as a host array of G device pointers (`*_ptrs`), or pass a
 * single base + strides.g when they are one buffer.
 *
 * Scale tables.  scales[g * scale_stride_g + t] is the STORED scale of token t widened to
 * fp32: the reference computes the scale in fp32, quantises with that fp32 value and stores
 * it rounded to the input dtype (ops.py:26-30,47-50,65); dequantisation widens the stored
 * value (ops.py:87,90,117,133). The quantise kernels reproduce exactly that split.
 */
#ifndef KVQ_HIP_H
#define KVQ_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KVQ_VERSION 100 /* 0.1.0 */

/* floating-point element types of KV tensors */
#define KVQ_F16 0
#define KVQ_BF16 1
#define KVQ_F32 2

/* argument errors (negative); launch errors are positive hipError_t values */
#define KVQ_E_NULL (-1)      /* required pointer is NULL */
#define KVQ_E_DIMS (-2)      /* negative / inconsistent dims or strides */
#define KVQ_E_DTYPE (-3)     /* unknown dtype code */
#define KVQ_E_DEVICE (-4)    /* the output buffer lives on another device than the calling thread's current one */

/* Pointer-list entry points take any number of groups; they launch once per 128 pointers
 * (the pointers travel by value in the kernel-argument segment). */

typedef struct {
  int64_t G, B, H, T, D;
} kvq_dims_t;

typedef struct {
  int64_t g, b, h, t; /* element strides; the D stride is 1 */
} kvq_strides_t;

int kvq_version(void);
const char* kvq_last_error_string(void);

/* ---- reference-equivalent flat entry points (one host scalar scale per launch) ---------- */

/* Replaces kvq_ext.dequant_int8_to_fp16(Tensor q, double scale) — extensions.py:70-86,
 * kernel :37-48: out[i] = half(float(q[i]) * scale), n elements, q and out contiguous. */
int kvq_dequant_i8_f16_flat(const int8_t* q, float scale, void* out_f16, int64_t n, void* stream);

/* Replaces kvq_ext.dequant_int4_packed_to_fp16(Tensor packed, double scale, int64 orig_last_dim)
 * — extensions.py:88-114, kernel :50-68. `packed` holds n_packed bytes, rows of `packed_last`
 * bytes; out holds 2*n_packed halves, rows of 2*packed_last. Even index = HIGH nibble (:61),
 * value (nibble-8)*scale, columns >= orig_last_dim written as 0 (:65-66). */
int kvq_dequant_i4_f16_flat(const uint8_t* packed, float scale, void* out_f16, int64_t n_packed,
                            int64_t packed_last, int64_t orig_last_dim, void* stream);

/* ---- token-table dequantise: a whole layer set per launch -------------------------------- */

/* Replaces the 2*L*T per-slice dequantize_int8_per_tensor calls + T-way torch.cat of
 * QuantizedLayerKV.get_kv (ops.py:213-269) / QuantizedKVCache.to_past_key_values (:345-355).
 * q [G,B,H,T,D] int8 (strides q_st) -> out [G,B,H,T,D] of out_dtype (strides out_st).
 * out = RN_out(float(q) * scales[g,t]). */
int kvq_dequant_i8_tokens(const int8_t* q, const kvq_strides_t* q_st, const float* scales,
                          int64_t scale_stride_g, void* out, const kvq_strides_t* out_st,
                          int out_dtype, const kvq_dims_t* dims, void* stream);

/* Same for packed INT4 (dequantize_int4_per_tensor_packed, ops.py:93-133). dims->D is the
 * UNPACKED head_dim (orig_last_dim); packed rows hold (D+1)/2 bytes; p_st strides are in bytes. */
int kvq_dequant_i4_tokens(const uint8_t* packed, const kvq_strides_t* p_st, const float* scales,
                          int64_t scale_stride_g, void* out, const kvq_strides_t* out_st,
                          int out_dtype, const kvq_dims_t* dims, void* stream);

/* ---- token-table quantise ---------------------------------------------------------------- */

/* Replaces quantize_int8_per_tensor (ops.py:10-30) applied per token slice by
 * QuantizedLayerKV.append (ops.py:174-210) under init_from_prompt_past (:333-342) /
 * append_from_past (:323-330).
 *   in:  [G,B,H,T,D] of in_dtype. Either in_ptrs (HOST array of G device pointers, strides.g
 *        ignored) or in_base (+ in_st->g); exactly one of them non-NULL.
 *   q:   [G,B,H,T,D] int8 store, strides q_st (a window of a larger [.., Tcap, D] store).
 *   scales: stored scale widened to fp32, scales[g*scale_stride_g + t].
 *   absmax_ws: caller workspace of G*T floats (used by the two-pass path; contents undefined
 *        afterwards).
 *   s32 = max(max|x| / 127, eps) in fp32; q = clamp(rint(x / s32), -127, 127);
 *   stored = RN_in_dtype(s32). */
int kvq_quant_i8_tokens(const void* in_base, const void* const* in_ptrs, const kvq_strides_t* in_st,
                        int in_dtype, int8_t* q, const kvq_strides_t* q_st, float* scales,
                        int64_t scale_stride_g, float* absmax_ws, float eps,
                        const kvq_dims_t* dims, void* stream);

/* Same for quantize_int4_per_tensor_packed (ops.py:33-65): s32 = max(max|x|/7, eps);
 * q = clamp(rint(x/s32), -8, 7); nibble = q+8; even d -> high nibble; odd D padded with
 * nibble 8 (ops.py:54-59). packed rows hold (D+1)/2 bytes; p_st in bytes. */
int kvq_quant_i4_tokens(const void* in_base, const void* const* in_ptrs, const kvq_strides_t* in_st,
                        int in_dtype, uint8_t* packed, const kvq_strides_t* p_st, float* scales,
                        int64_t scale_stride_g, float* absmax_ws, float eps,
                        const kvq_dims_t* dims, void* stream);

/* ---- split phases: a batched slice whose batch rows are sharded over ranks -------------------- */

/* The reference's scale spans the WHOLE [B,H,1,D] slice (ops.py:27,48), so when B is split over ranks
 *  the abs-max must cross ranks between the reduction and the quantisation:
 *   kvq_absmax_tokens            local rows -> absmax[g*T + t] = max |x| over this rank's [B_local,H,D]
 *   all_reduce(MAX) of absmax    (the caller's collective: RCCL over xGMI, torch.distributed)
 *   kvq_quant_tokens_from_absmax quantise the local rows with scale max(absmax/QMAX, eps), store the scale
 * On one rank (no all_reduce) the pair is bit-identical with kvq_quant_i8_tokens / kvq_quant_i4_tokens.
 * absmax: contiguous [G,T] fp32, overwritten by phase 1 (no pre-initialisation needed). */
int kvq_absmax_tokens(const void* in_base, const void* const* in_ptrs, const kvq_strides_t* in_st, int in_dtype,
                      float* absmax, const kvq_dims_t* dims, void* stream);

/* kvq_absmax_tokens without the fill: absmax[g*T + t] = max(absmax[g*T + t], max |x| over this rank's [B_local,H,D]).
 * The caller initialises the table (non-negative values; zeros for a plain abs-max): a layer-chunked pass zeroes its
 * whole [G,T] table with ONE fill instead of one per chunk (a fill is a launch: 4.5 us each on MI355X). */
int kvq_absmax_tokens_acc(const void* in_base, const void* const* in_ptrs, const kvq_strides_t* in_st, int in_dtype,
                          float* absmax, const kvq_dims_t* dims, void* stream);

/* bits: 8 (int8 store, strides in bytes = elements) or 4 (packed store, (D+1)/2 bytes per row). */
int kvq_quant_tokens_from_absmax(int bits, const void* in_base, const void* const* in_ptrs, const kvq_strides_t* in_st,
                                 int in_dtype, uint8_t* q, const kvq_strides_t* q_st, float* scales,
                                 int64_t scale_stride_g, const float* absmax, float eps, const kvq_dims_t* dims,
                                 void* stream);

/* ---- eviction ---------------------------------------------------------------------------- */

/* Replaces trim_kv_sliding_window (src/cache/implementations.py:124-140), materialised:
 * out[g,b,h,0:W,:] = in[g,b,h,T-W:T,:] with W = min(window, T). The reference returns views
 * and leaves the byte movement to the next torch.cat; a persistent buffer must compact.
 * in and out must not overlap. elem_size is 2 or 4 bytes. A pointer list goes 128 groups per launch; a single base with
 * in_st->g (one allocation: e.g. a paged pool's blocks) is ONE launch for up to 65,535 groups. */
int kvq_window_compact(const void* in_base, const void* const* in_ptrs, const kvq_strides_t* in_st,
                       void* out, const kvq_strides_t* out_st, int elem_size, int64_t window,
                       const kvq_dims_t* dims, void* stream);

/* Replaces chunk_summarize_kv (implementations.py:295-346) on [G,B,H,T,D]:
 *   keep = min(keep_last, T); old = T - keep; n = ceil(old / chunk_size)
 *   out[.., j, :]     = RN_dtype( (sum_{i<chunk_size, j*chunk_size+i<old} in[.., j*chunk_size+i, :]) / chunk_size )   j < n
 *   out[.., n + i, :] = in[.., old + i, :]                                                  i < keep
 * fp32 accumulation, sequential in t; divisor is chunk_size even for the ragged last chunk
 * (zero padding, :326-333). If old <= 0 the call copies in to out unchanged (T rows).
 * Output has kvq_chunk_summary_len(T, chunk_size, keep_last) tokens. */
int kvq_chunk_meanpool(const void* in_base, const void* const* in_ptrs, const kvq_strides_t* in_st,
                       void* out, const kvq_strides_t* out_st, int dtype, int64_t chunk_size,
                       int64_t keep_last, const kvq_dims_t* dims, void* stream);

int64_t kvq_chunk_summary_len(int64_t T, int64_t chunk_size, int64_t keep_last);

/* Replaces the index_select + torch.cat of the sparse eviction family — trim_kv_prefix_window,
 * trim_kv_strided, trim_kv_block_old, trim_kv_budget_old (src/cache/implementations.py:143-292):
 *   out[g,b,h,j,:] = in[g,b,h,idx[j],:]   j < n_idx
 * idx is a DEVICE array of n_idx int32 token indices in [0, dims->T) (built by the host from the
 * policy). The kernel compares every index with dims->T: an index outside the range yields a row of
 * zeros, never an out-of-bounds read (torch's index_select raises there; the Python wrappers validate
 * host-built index lists before the call). elem_size is 2 or 4 bytes. */
int kvq_gather_tokens(const void* in_base, const void* const* in_ptrs, const kvq_strides_t* in_st,
                      void* out, const kvq_strides_t* out_st, int elem_size, const int32_t* idx,
                      int64_t n_idx, const kvq_dims_t* dims, void* stream);

/* ---- decode attention over the quantised store -------------------------- */

typedef struct {
  int64_t B, Hq, Hkv, T, D; /* T = stored (quantised) tokens attended; Hq % Hkv == 0, Hq/Hkv <= 8 (<= 16 at head_dim 64 / 128) */
} kvq_attn_dims_t;

/* Workspace floats kvq_decode_attn needs for these dims (-1 on bad dims). */
int64_t kvq_decode_attn_workspace(const kvq_attn_dims_t* dims);
/* Largest kvq_decode_attn_workspace over every T' <= dims->T (it is not monotone in T): what a decode
 * loop allocates once for its reserved capacity. */
int64_t kvq_decode_attn_workspace_cap(const kvq_attn_dims_t* dims);

/* One decode step of one layer without materialising the fp16 cache. Replaces, for a single
 * query token, QuantizedKVCache.to_past_key_values (src/quantization/ops.py:345-355: dequantise
 * every stored token) + the model's attention over [dequantised past, new token]
 * (src/benchmarking/benchmarker.py:470-471):
 *   s[t]   = sm_scale * k_scales[t] * sum_d q[d] * k_int[t,d]            t < T
 *   s_new  = sm_scale * sum_d q[d] * k_new[d]                             (if k_new)
 *   p      = softmax over {s[0..T), s_new}
 *   out[d] = sum_t p[t] * v_scales[t] * v_int[t,d] + p_new * v_new[d]
 * q / k_new / v_new / out: [B, H, D] of `dtype` (KVQ_F16 | KVQ_BF16) with element strides
 * (stride_b, stride_h), D contiguous; q heads Hq, k_new / v_new heads Hkv; query head h uses kv
 * head h / (Hq/Hkv). k_store / v_store: one layer's [B, Hkv, Tcap, Dq] rows (int8, or packed INT4
 * with the even element in the high nibble), byte strides k_st / v_st (g unused); *_scales[t] the
 * per-token scale table of that layer. k_bits / v_bits: 8 or 4. D in {32, 64, 128, 256}.
 * fp32 accumulation; the reference's intermediate fp16 rounding of the dequantised values is
 * skipped, so results agree within fp16 tolerance, not bit-exactly. */
int kvq_decode_attn(const void* q, int64_t q_stride_b, int64_t q_stride_h,
                    const uint8_t* k_store, const kvq_strides_t* k_st, const float* k_scales, int k_bits,
                    const uint8_t* v_store, const kvq_strides_t* v_st, const float* v_scales, int v_bits,
                    const void* k_new, int64_t kn_stride_b, int64_t kn_stride_h,
                    const void* v_new, int64_t vn_stride_b, int64_t vn_stride_h,
                    void* out, int64_t out_stride_b, int64_t out_stride_h, int dtype, float sm_scale,
                    float* workspace, int64_t workspace_floats, const kvq_attn_dims_t* dims, void* stream);

/* kvq_decode_attn over the T stored tokens + the new token, THEN the new token's K / V quantised into
 * slot T of the stores (scales[T]): everything one layer does per decode step in the reference
 * (append_from_past ops.py:323-330, to_past_key_values :345-355, attention) behind one call.
 * The caller guarantees capacity for slot T (k_st / v_st describe the whole [B,Hkv,Tcap,Dq] store)
 * and counts the token as stored afterwards. workspace as for kvq_decode_attn (>= 1 float).
 * Two launches: the new token is quantised by two extra workgroups of the merge launch.
 * The workspace needs no initialisation. */
int kvq_decode_step(const void* q, int64_t q_stride_b, int64_t q_stride_h,
                    const void* k_new, int64_t kn_stride_b, int64_t kn_stride_h,
                    const void* v_new, int64_t vn_stride_b, int64_t vn_stride_h,
                    uint8_t* k_store, const kvq_strides_t* k_st, float* k_scales, int k_bits,
                    uint8_t* v_store, const kvq_strides_t* v_st, float* v_scales, int v_bits,
                    void* out, int64_t out_stride_b, int64_t out_stride_h, int dtype, float sm_scale, float eps,
                    float* workspace, int64_t workspace_floats, const kvq_attn_dims_t* dims, void* stream);

/* kvq_decode_step with the stored-token count in DEVICE memory: every argument of the two launches is then the
 * same from one decode step to the next, so a HIP graph that captured the step (together with the model's own
 * kernels) can be replayed for every later token — the host loop of the reference's generate_with_quantized_kv
 * (src/benchmarking/benchmarker.py:465-486) leaves the critical path. `*t_dev` (int32) = tokens stored before
 * the step; the caller increments it on the same stream afterwards. dims->T is the HOST's upper bound on
 * *t_dev (e.g. the store's capacity - 1): it sizes the grid and the workspace (kvq_decode_attn_workspace for
 * that T); workgroups beyond the live count exit at once. The caller guarantees *t_dev <= dims->T and room for
 * slot *t_dev. One-tile kernels only (no streaming / fused variants); B * Hkv * D <= 65536. */
int kvq_decode_step_dev(const void* q, int64_t q_stride_b, int64_t q_stride_h,
                        const void* k_new, int64_t kn_stride_b, int64_t kn_stride_h,
                        const void* v_new, int64_t vn_stride_b, int64_t vn_stride_h,
                        uint8_t* k_store, const kvq_strides_t* k_st, float* k_scales, int k_bits,
                        uint8_t* v_store, const kvq_strides_t* v_st, float* v_scales, int v_bits,
                        void* out, int64_t out_stride_b, int64_t out_stride_h, int dtype, float sm_scale, float eps,
                        float* workspace, int64_t workspace_floats, const kvq_attn_dims_t* dims, const int32_t* t_dev,
                        void* stream);

/* kvq_decode_step (append != 0) or kvq_decode_attn with the new token given (append == 0) for n_layers
 * layers behind ONE host call: the per-layer launches are enqueued back to back on `stream`, so a decode step costs one trip through the binding instead of one per
 * layer — the reference pays 2*T Python-level calls per layer and step here (ops.py:345-355). Every
 * pointer argument is a HOST array of n_layers device pointers; dims, strides, kinds and dtype are
 * shared by all layers; `workspace` is reused by consecutive launches (stream order makes that safe).
 * k_new / v_new may be NULL when append == 0 (attention over the stored tokens only). */
int kvq_decode_step_layers(int64_t n_layers, int append, const void* const* q, int64_t q_stride_b, int64_t q_stride_h,
                           const void* const* k_new, int64_t kn_stride_b, int64_t kn_stride_h,
                           const void* const* v_new, int64_t vn_stride_b, int64_t vn_stride_h,
                           uint8_t* const* k_store, const kvq_strides_t* k_st, float* const* k_scales, int k_bits,
                           uint8_t* const* v_store, const kvq_strides_t* v_st, float* const* v_scales, int v_bits,
                           void* const* out, int64_t out_stride_b, int64_t out_stride_h, int dtype, float sm_scale,
                           float eps, float* workspace, int64_t workspace_floats, const kvq_attn_dims_t* dims,
                           void* stream);

/* Measurement aid: the NEXT kernel launch of the calling thread — whichever entry point makes it — binds these two
 * hipEvent_t (already created; either may be NULL) to its own dispatch (hipExtLaunchKernelGGL): stop - start is then
 * the kernel's duration as a profiler sees it, without the queue gaps two hipEventRecord calls around the launch
 * include. One-shot (the launch takes them). A library call that returns an error disarms the pair (no later launch
 * of the thread can take the events of a call that failed); kvq_time_next_launch(NULL, NULL) disarms a pair no launch
 * took — a call that succeeds WITHOUT launching (empty dims) or fails before it reaches the library leaves it armed, so a
 * careful caller still does that (the Python binding does, in a finally). kvq_timing_armed(): 1 while a pair is pending. */
int kvq_time_next_launch(void* start_event, void* stop_event);
int kvq_timing_armed(void);

/* Measurement aid: which kernels ran. Every launch of the calling thread notes its kernel; kvq_kernel_log writes the
 * distinct kernels launched since kvq_kernel_log_clear(), in first-launch order, one demangled name per line (what
 * rocprofv3 --kernel-trace prints) into buf (n bytes, NUL-terminated, truncated to fit) and returns their count.
 * bench.py labels every `roofline.kernel` with this, never with a name typed by hand. */
void kvq_kernel_log_clear(void);
int64_t kvq_kernel_log(char* buf, int64_t n);

/* ---- tuning knobs ---------------------------------------------------------------------------- */

/* Two kinds of keys.
 * Test knobs (every build) route a call to SHIPPED code it would not take by size or shape; results never change:
 *   "quant_force_two_pass" (0/1: generic two-pass quantise), "quant_direct_stores" (0/1: no LDS-staged stores in the
 *   256-thread quantise kernel), "quant_tile" (1 = compile-time one-wave tile kernel where the shape has one, default;
 *   0 = general kernels), "quant_wide" (1 = single-pass 1024-thread register tile for batched slices of 16384 < B*H*D <=
 *   131072 two-byte elements, default; 0 = split phases / swept tile), "quant_block" (general quantise kernel: 64-thread one-wave tiles, default, or 256; 128 in A-B builds), "pool_wave" (1 = one wave per output row where the shape allows, default; 0 = per-lane-group
 *   walk), "gather_rows" (token gather: 1 = 4 KiB work items, one 16-byte piece per thread, default; 0 = the grid-stride kernel), "attn_ring_dev" (kvq_decode_step_dev on the LDS-staged ring kernel where the host-side call takes it: 1, default; 0 = always one-tile splits), "attn_new_token_parts" (kvq_decode_step with a new-token slice past 8,192 elements: 1 = one workgroup per 8,192 elements, default; 0 = the generic one-workgroup routine), "quant_few_tokens" (kvq_absmax_tokens on slices of at most this many tokens: one workgroup per (group, token) and a plain store; 0 = always the tile walk's atomics), "attn_force_valu" (0/1), "attn_stream_tpw" (tiles per wave of the streaming attention kernel: -1 never,
 *   0 by size, > 0 that many), "attn_lds" (LDS-staged MFMA attention: -1 by shape, 0 never, 1 wherever it applies),
 *   "attn_merge_wave" (merge of <= 16 splits at head_dim 128: 1 = one wave per head, default; 0 = one workgroup per head).
 * A-B keys select variants that lost a measurement and exist only in the A-B library (`make -C csrc ab` ->
 *   lib/ab/libkvq_hip.so, kvq_is_ab_build() == 1); the default library returns KVQ_E_DIMS for them:
 *   dequantise  "dequant_variant" (0..35), "dequant_grid", "dequant_xcd_group"
 *   quantise    "quant_nv" (8|4|16), "quant_lds_pad", "quant_tpw" (0|2|4|8),
 *               "quant_no_regmax", "quant_xcd_group", "quant_geo128", "quant_nt_stores", "quant_tile_tt" (8|4), "nt_loads",
 *               "quant_wide_blk" (1024|512), "quant_tile_tpw" (0|2|4)
 *   eviction    "pool_grid", "pool_block" (64|128|256)
 *   attention   "attn_mfma_min_nq", "attn_mfma_tc" (128|64), "attn_merge_fast" (0 = chained merge by choice),
 *               "attn_stream_tc" (64|32), "attn_stream_slots", "attn_stream_roll", "attn_tg" (1 = one score output per
 *               16-token group in the LDS-staged kernel at <= 4 query heads per kv head too), "attn_k_i8" (-1|0|1; tolerance-level
 *               difference), "attn_fused" (one launch per call; refused while the stream is being captured into a
 *               HIP graph: its arrival epoch is a launch argument), "attn_fused_tc" / "attn_fused_nw", "attn_fold" (the merge
 *               INSIDE the LDS-staged kernel's launch, by arrival ticket: 1 = in kvq_decode_step_layers, 2 = kvq_decode_attn too),
 *               "attn_onepass" (<= 2,048 stored tokens in ONE launch, one 8-wave workgroup per (batch row, kv head), merge from LDS).
 * Returns 0, or KVQ_E_DIMS for an unknown key / an A-B key in the default library. Process-global. */
int kvq_set_tunable(const char* key, int64_t value);
int64_t kvq_get_tunable(const char* key);
int kvq_is_ab_build(void);

#ifdef __cplusplus
}
#endif
#endif /* KVQ_HIP_H */
