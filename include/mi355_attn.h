/*
 * mi355_attn.h — C ABI of libmi355_attn.so: paged-KV attention for MI355X (gfx950).
 *
 * This is the drop-in boundary for the hot path of foundation-model-stack/vllm-triton-backend:
 * chunked/context prefill attention + paged-KV decode over vLLM block tables. Every entry point
 * states the reference interface it replaces (paths relative to the reference repository,
 * LIB/ = ibm-triton-lib/ibm_triton_lib/).
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless stated otherwise;
 *   - every function returns 0 on success or a negative MI355_ERR_* code; the message of the last
 *     failure on the calling thread is available from mi355_last_error();
 *   - no allocation, no host/device synchronisation, no exceptions: kernels are enqueued on the
 *     stream handed in (a hipStream_t), so calls can be captured into a hipGraph;
 *   - all buffers are caller owned; scratch comes from a caller-provided workspace whose size is
 *     given by mi355_attn_workspace_bytes();
 *   - strides are in ELEMENTS of the tensor's dtype.
 */
#ifndef MI355_ATTN_H
#define MI355_ATTN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_ATTN_VERSION 601 /* major*10000 + minor*100 + patch */
/*
 * Version notes (what a caller written against an older header must know)
 *   0.6.1  No change to the structs or the entry points. A prefill over an fp8 flash-layout cache (plain attention, head size
 *          128, >= 2048 keys) is read as fp8 by the fast prefill kernel itself ("prefill_mfma_pw_fp8"): such a call no
 *          longer asks for a 16-bit scratch cache - mi355_attn_workspace_bytes() answers the 256 KiB counter block, or
 *          the key-split partials - and is served whatever the step's mix of prefill and decode rows. Short fp8 prompts
 *          (what the short-prompt kernel serves for 16-bit caches) run on that kernel's fp8 form ("prefill_mfma_lat_fp8").
 *   0.6.0  write_new_kv is accepted for steps with prefill rows too (see the field): mi355_decode_write_fusable() answers 1
 *          for them where the short-prompt kernel or an LDS-DMA kernel serves the prefill rows. Nothing else changes.
 *   0.5.1  No change to the structs or the entry points. The workspace's zero-filled 256 KiB head is now two regions:
 *          [0, 192 KiB) the counters of 0.3.1, [192 KiB, 256 KiB) one byte per (128-row Q block, KV head) of an f16
 *          prefill call - rows whose scores left the fast kernel's range are flagged there and computed again by a
 *          second launch of the same call, which clears them (zero on entry, zero on exit, like the counters). New
 *          kernels behind the same call: a short-prompt prefill kernel, a two-level in-kernel merge for decode steps
 *          planned with many splits (one launch where a graph captured at max_model_len took two). The measurement
 *          switches (environment variables) are read only in a process that also sets MI355_LAB=1.
 *   0.5.0  Multi-token decode steps (speculative decoding / MTP verification) on the split-KV decode kernel, several
 *          query tokens of a sequence per wave. skip_decodes / only_decodes became query-length thresholds (1 keeps
 *          its meaning); reserved2 became decode_rows_hint (0 = as before).
 *   0.4.0  mi355_attn_params grew at its END (slot_mapping, slot_mapping_i32, new_kv_all_rows, reserved2): zero the
 *          struct before filling it and nothing changes. A fused decode write (write_new_kv) now honours
 *          slot_mapping[i] < 0 (padding row: the cache is not written) when a slot mapping is handed in.
 *   0.3.1  WORKSPACE CONTRACT: a prefill call for which mi355_attn_workspace_bytes() answers 256 KiB (the bf16 / f16
 *          D = 128 fast path) treats the head of the workspace as TICKET COUNTERS that must be zero on entry and are
 *          left zero on exit. Calls of that kind used to need no workspace and ignored the pointer; a caller that
 *          hands such calls a non-zeroed scratch must now either zero-fill it once (as the header always asked for
 *          the decode counters) or pass NULL / 0, which selects the static work-item deal. A launch that was aborted
 *          mid-kernel leaves the counters non-zero: zero-fill the workspace again after a device error.
 *   0.3.0  write_new_kv (fused cache write of a decode step), non_causal.
 *   0.2.0  lse output.
 */

#if defined(__GNUC__)
#define MI355_API __attribute__((visibility("default")))
#else
#define MI355_API
#endif

/* error codes */
#define MI355_OK 0
#define MI355_ERR_BAD_ARG (-1)     /* NULL pointer, negative size, inconsistent shapes            */
#define MI355_ERR_UNSUPPORTED (-2) /* valid request this build has no kernel for                  */
#define MI355_ERR_HIP (-3)         /* the HIP runtime refused the launch                          */
#define MI355_ERR_WORKSPACE (-4)   /* workspace missing or smaller than mi355_attn_workspace_bytes */

/* element types */
typedef enum mi355_dtype {
  MI355_F32 = 0,
  MI355_F16 = 1,
  MI355_BF16 = 2,
  MI355_FP8_E4M3 = 3, /* OCP e4m3fn (gfx950 native; NOT MI300's fnuz) */
  MI355_FP8_E5M2 = 4
} mi355_dtype;

/* which kernel family mi355_unified_attention launches (mirrors `force_selection`,
 * LIB/kernels/triton_unified_attention.py:859,:884) */
typedef enum mi355_kernel_select {
  MI355_SELECT_AUTO = 0,
  MI355_SELECT_2D = 2,     /* one pass over the whole KV range per Q block ("2D" kernel, :275-523) */
  MI355_SELECT_3D = 3,     /* split-KV partials + reduce_segments ("3D" path, :526-836)            */
  MI355_SELECT_GENERIC = 9 /* the shape-agnostic correctness kernel (any dtype/head size/layout)   */
} mi355_kernel_select;

/* opaque hipStream_t */
typedef void* mi355_stream_t;

/*
 * Parameters of one unified attention call.
 *
 * Replaces the argument list of `unified_attention(q, k, v, out, cu_seqlens_q, max_seqlen_q,
 * seqused_k, max_seqlen_k, avg_seqlen_q, avg_seqlen_k, softmax_scale, causal, window_size,
 * block_table, softcap, q_descale, k_descale, v_descale, alibi_slopes, force_selection)`
 * (LIB/kernels/triton_unified_attention.py:839-860) plus the strides its launcher derives
 * (:905-923).
 *
 * KV-cache addressing covers both cache layouts of the reference with one formula:
 *   K element (page p, slot o, kv head h, dim d) lives at
 *       k_cache + p*k_stride_page + o*k_stride_slot + h*k_stride_head
 *               + (d / k_x)*k_stride_dx + (d % k_x)*k_stride_d
 *   V element at  v_cache + p*v_stride_page + o*v_stride_slot + h*v_stride_head + d*v_stride_d.
 *   flash layout  [num_pages, page, Hk, D] (triton_unified_attention.py:279-280): k_x = D,
 *       k_stride_dx = 0, k_stride_d = v_stride_d = 1;
 *   legacy v0 layout K [num_pages, Hk, D/x, page, x], V [num_pages, Hk, D, page]
 *       (LIB/kernels/legacy/triton_paged_decode_attention_2d.py:103-104): k_x = x.
 *
 * Optional "new token" source (context_attention_fwd, LIB/kernels/legacy/triton_prefix_prefill.py
 * :589-606): when k_new/v_new are non-NULL, sequences with query_len > 1 read key positions
 * >= context_len from the linear tensors k_new/v_new [num_tokens, Hk, D] (row of position j is
 * cu_seqlens_q[i] + j - context_len) instead of from the cache.
 */
typedef struct mi355_attn_params {
  /* tensors */
  const void* q;               /* [num_tokens, Hq, D], last dim contiguous                      */
  void* out;                   /* [num_tokens, Hq, D], last dim contiguous, written in place    */
  const void* k_cache;         /* see layout note                                                */
  const void* v_cache;
  const int32_t* block_table;  /* [num_seqs, >= ceil(max_seqlen_k / page_size)] physical pages   */
  const int32_t* cu_seqlens_q; /* [num_seqs + 1] exclusive prefix sum of query lengths           */
  const int32_t* seqused_k;    /* [num_seqs] total key length (context + query) per sequence     */
  const float* alibi_slopes;   /* [Hq] or NULL                                                   */
  const float* k_scale;        /* device fp32 scalar (element 0 is read) or NULL = 1.0; fp8 KV   */
  const float* v_scale;
  const void* k_new;           /* optional linear new-token K/V, see above; NULL for the unified */
  const void* v_new;           /*   path                                                          */

  /* element types */
  int32_t q_dtype;             /* mi355_dtype of q, out, k_new, v_new                            */
  int32_t kv_dtype;            /* mi355_dtype of the caches (== q_dtype, or an fp8 type)         */

  /* sizes */
  int32_t num_tokens;          /* T  = q.shape[0]                                                */
  int32_t num_seqs;            /* S  = len(seqused_k)                                            */
  int32_t num_q_heads;         /* Hq                                                             */
  int32_t num_kv_heads;        /* Hk, Hq % Hk == 0                                               */
  int32_t head_size;           /* D                                                              */
  int32_t page_size;           /* tokens per KV page ("block_size" in vLLM)                      */
  int32_t max_seqlen_q;        /* host-known upper bounds; used for dispatch and grid sizing     */
  int32_t max_seqlen_k;

  /* strides, in elements */
  int64_t q_stride_token, q_stride_head;
  int64_t out_stride_token, out_stride_head;
  int64_t k_stride_page, k_stride_slot, k_stride_head, k_stride_dx, k_stride_d;
  int32_t k_x;
  int32_t reserved0;
  int64_t v_stride_page, v_stride_slot, v_stride_head, v_stride_d;
  int64_t block_table_stride;
  int64_t new_stride_token, new_stride_head; /* strides of k_new / v_new                         */

  /* scalars */
  float scale;                 /* softmax scale                                                  */
  float softcap;               /* > 0 enables cap * tanh(s / cap) (:25-29,:914)                   */
  int32_t sliding_window;      /* 0 = off; else keep keys with query_pos - key_pos < window      */
                               /*   (= 1 + window_size[0], :915)                                  */
  int32_t skip_decodes;        /* N >= 1: leave rows of sequences with query_len <= N untouched  */
                               /*   (1: triton_prefix_prefill.py:83-84)                           */
  int32_t only_decodes;        /* N >= 1: process only sequences with query_len <= N             */
                               /*   (1: filter_by_query_len, triton_paged_decode_attention_2d.py:143-148; */
                               /*   N > 1: multi-token decode rows - speculative decoding - of a  */
                               /*   mixed batch, which the split-KV kernel takes N tokens a wave) */
  int32_t kernel_select;       /* mi355_kernel_select                                            */
  int32_t num_segments;        /* split-KV segment count of the decode path / key-split count of */
                               /*   a prefill (1 = every Q block walks its whole key range in    */
                               /*   one pass, the reference's 2D kernel); 0 = library picks       */
  int32_t reserved1;

  /* optional second output (library version >= 0.2.0; not in the reference): the natural-log sum of exponentials of every
   * row's masked, scaled scores, fp32. With it partial results over disjoint key ranges - e.g. one long sequence whose
   * pages are striped over GPUs, SURVEY.md 8e "cross-GPU split-KV" - merge exactly:
   *   lse = log sum_r exp(lse_r),  out = sum_r out_r * exp(lse_r - lse)      (same math as reduce_segments, :804-828)
   * A row that sees no key gets -inf (and out 0). NULL = not wanted. */
  float* lse;                  /* [num_tokens, Hq] or NULL                                        */
  int64_t lse_stride_token;    /* elements between tokens; heads are contiguous                   */

  /* fused paged-cache write of a decode step (library version >= 0.3.0): replaces the separate
   * reshape_and_cache_flash launch in front of the attention (triton_attn.py:393-405) for calls in which every sequence
   * has ONE query token. With write_new_kv != 0, k_new / v_new [num_tokens, Hk, D] hold, for sequence i, the key / value
   * of its LAST position seqused_k[i] - 1 (the token being decoded): the kernel stores the row into the cache page of
   * that position - saturating fp8(x / scale) for an fp8 cache, exactly what mi355_reshape_and_cache_flash stores - and
   * attends over it, whatever the cache held there before. Requires max_seqlen_q == 1, num_tokens == num_seqs, the flash
   * layout and the matrix-core decode kernel (mi355_decode_write_fusable() answers for a parameter block); the caches
   * are written although the struct declares them const.
   * Library version >= 0.6.0: also a step WITH PREFILL ROWS (max_seqlen_q > 1) whose prefill rows the short-prompt kernel or
   * an LDS-DMA kernel serves (plain attention, head size 128, 16-bit cache of the query's type, key ranges below the
   * long-prefill kernel's, no key split; mi355_decode_write_fusable() answers): k_new / v_new [num_tokens, Hk, D] then hold the
   * key / value of EVERY query token (token t of sequence i = position seqused_k[i] - query_len_i + t); the launches attend
   * over them straight from these tensors and store them into their pages - by slot_mapping when one is handed in
   * (negative: not stored), else by position through the block table. One-token rows of such a step ride the decode launch
   * and its fused write. */
  int32_t write_new_kv;
  /* 0 (every op of the reference's backend path): causal - query t of a sequence sees keys j <= t + seqused_k - query_len.
   * 1: every query row sees ALL seqused_k keys of its sequence (prefill_flash_attention(causal=False),
   * triton_flash_attention.py:1326-1484; no sliding window / ALiBi with it). Served by the 64-rows-per-wave matrix-core
   * kernel where it applies (bf16 / f16, head size 128, 16-bit cache or linear k_new / v_new), else by the shape-agnostic
   * kernel. */
  int32_t non_causal;

  /* library version >= 0.4.0 */
  /* With write_new_kv: the step's slot mapping as vLLM hands it to reshape_and_cache_flash (triton_attn.py:396-405),
   * int64 or int32, [num_tokens]; at most one non-NULL. Row i's K/V is stored only if its slot is >= 0: a negative slot
   * marks a padding row of a captured graph (triton_attn.py:149-151) and leaves the cache untouched, whatever
   * seqused_k / block_table hold for that row. The POSITION written is still seqused_k[i] - 1 through the block table -
   * for a live row that is the slot vLLM computed, by construction of a decode step. Both NULL: every row is stored. */
  const int64_t* slot_mapping;
  const int32_t* slot_mapping_i32;
  /* With k_new / v_new: 1 = rows of sequences with query_len == 1 read their key positions >= context_len from the
   * linear tensors as well (self-attention over linear K/V with no cache behind it, prefill_flash_attention); 0 = the
   * legacy ops' rule, such rows read the cache only (chunked_prefill_paged_decode) */
  int32_t new_kv_all_rows;
  /* library version >= 0.5.0. A step that mixes prefills with shorter rows is served by two launches, and sequences of
   * up to N query tokens are the decode launch's. 0: the library picks N = 16 / G (what one column group of the packed
   * decode kernel holds - a one-token row costs nothing extra there). A caller that KNOWS its decode rows carry more
   * tokens (speculative decoding with k drafts: 1 + k) says so here and gets N = that, up to 32 / G (two column groups;
   * ignored where they do not apply: features, head size 256, more). Host-known, so capture-stable. */
  int32_t decode_rows_hint;
} mi355_attn_params;

/*
 * Parameters of the paged-cache write.
 * Replaces `torch.ops._C_cache_ops.reshape_and_cache_flash(key, value, key_cache, value_cache,
 * slot_mapping, kv_cache_dtype, k_scale, v_scale)` as called at LIB/backend/triton_attn.py:396-405
 * (CPU restatement in the reference: scripts/vllm_utils.py:377-401).
 * For token t: slot = slot_mapping[t]; slot < 0 is skipped (triton_attn.py:149-151); otherwise
 * cache[slot / page_size, slot % page_size, :, :] = key[t] (value likewise); with an fp8 cache the
 * stored value is saturating fp8(x / scale).
 */
typedef struct mi355_cache_params {
  const void* key;             /* [num_tokens, Hk, D]                                            */
  const void* value;
  void* k_cache;               /* [num_pages, page_size, Hk, D] (flash layout)                   */
  void* v_cache;
  const int64_t* slot_mapping; /* [num_tokens] int64 (vLLM) ...                                   */
  const int32_t* slot_mapping_i32; /* ... or int32 (reference harness, scripts/benchmark.py:1225); exactly one non-NULL */
  const float* k_scale;        /* device fp32 scalar or NULL = 1.0                               */
  const float* v_scale;
  int32_t src_dtype;           /* mi355_dtype of key/value                                       */
  int32_t cache_dtype;         /* mi355_dtype of the caches                                      */
  int32_t num_tokens, num_kv_heads, head_size, page_size;
  int64_t key_stride_token, key_stride_head;
  int64_t value_stride_token, value_stride_head;
  int64_t k_stride_page, k_stride_slot, k_stride_head;
  int64_t v_stride_page, v_stride_slot, v_stride_head;
} mi355_cache_params;

/* library version (MI355_ATTN_VERSION of the build) */
MI355_API int mi355_attn_version(void);

/* message of the last error on this thread ("" if none); never NULL */
MI355_API const char* mi355_last_error(void);

/* name of the kernel family the last successful mi355_unified_attention call on this thread
 * dispatched to ("decode_splitkv", "prefill_mfma", "generic", ...); for tests and profiling */
MI355_API const char* mi355_last_kernel(void);

/*
 * Bytes of scratch mi355_unified_attention needs for these parameters (host-side arithmetic only;
 * depends on sizes and upper bounds, never on device data, so it is capture-stable).
 * Replaces the three per-call torch.empty scratch tensors at triton_unified_attention.py:950-971
 * (decode partials; here also the partials of a key-split prefill and the scratch cache of the repack path).
 * The workspace must be ZERO-FILLED ONCE by its owner after allocation: its first 256 KiB hold the
 * arrival counters of the in-kernel split merge, the work-item ticket counters of the bf16 / f16 prefill
 * kernel and (last 64 KiB) the fix-up flags of an f16 prefill call, all of which every call leaves at zero again (a call that was aborted mid-kernel does not: zero-fill
 * again after a device error). One workspace serves one stream at a time. A prefill call whose answer is
 * exactly those 256 KiB also runs WITHOUT a workspace (NULL, 0): its work items are then dealt statically.
 */
MI355_API size_t mi355_attn_workspace_bytes(const mi355_attn_params* p);

/*
 * Unified causal paged attention (prefill, chunked prefill, decode, mixed batches).
 * Replaces `unified_attention` and the three kernels it launches
 * (LIB/kernels/triton_unified_attention.py:839-1030: kernel_unified_attention_2d :275-523,
 *  kernel_unified_attention_3d :526-754, reduce_segments :757-836); with k_new/v_new and the legacy
 * strides it also serves `context_attention_fwd` (legacy/triton_prefix_prefill.py:588-765),
 * `paged_attention_triton_2d/3d` (legacy/triton_paged_decode_attention_2d.py:283-398,
 * legacy/triton_paged_decode_attention_3d.py:348-499) and `chunked_prefill_paged_decode`
 * (legacy/triton_chunked_prefill_paged_decode.py:28-117).
 *
 * Which kernel serves a call (AUTO; mi355_last_kernel() names it afterwards):
 *   matrix-core kernels - f16/bf16 queries, flash-layout cache of the same type or fp8 e4m3fn/e5m2 with scalar
 *     scales, page size a power of two >= 16, head size any multiple of 8 (16 with an fp8 cache) up to 256 (run on
 *     the next of 64/128/256; the reference pads to the next power of two, :353,:912; fp8 prefill up to 128):
 *       max_seqlen_q == 1            -> split-KV decode ("decode_splitkv[_fp8]" / "decode_single[_fp8]")
 *       every sequence a prefill     -> Q-block prefill ("prefill_mfma[_fp8][_feat]"; "..._ksplit" when few Q blocks
 *                                       face a long context: key tiles dealt to several workgroups, partials merged
 *                                       by lse through the workspace)
 *       mixed batch                  -> both, prefill rows then decode rows ("<prefill kernel>+<decode kernel>", e.g. "prefill_mfma+decode_splitkv")
 *     the legacy v0 layout (16-bit, k_x == 8, head size 64/128/256) with max_seqlen_q == 1 -> the same split-KV decode
 *     kernel reading that layout directly ("decode_splitkv_v0" / "decode_single_v0")
 *     a 16-bit or fp8 cache in any layout the strides describe and/or linear k_new/v_new, max_seqlen_k a true bound
 *     (context_attention_fwd, chunked_prefill_paged_decode; paged_attention_2d/3d over fp8 or 4-D v0 caches): every sequence's keys are first gathered into a
 *     flash-layout scratch cache in the workspace (num_seqs * ceil(max_seqlen_k / 16) pages of K and V), the kernels
 *     above run on that; query_len == 1 rows of a mixed batch read the caller's cache directly when the v0 decode
 *     kernel covers it ("repack+<prefill kernel>[+<decode kernel>]")
 *     (an fp8 cache is dequantised into the 16-bit scratch: (fp8 -> f32) * scale -> query type; since 0.6.1 a plain fp8
 *     flash-layout prefill at head size 128 does not come this way: "prefill_mfma_pw_fp8" widens the tiles itself)
 *   everything else (f32, head sizes that are not a multiple of 8 or exceed 256, ...) -> "generic".
 */
MI355_API int mi355_unified_attention(const mi355_attn_params* p, void* workspace, size_t workspace_bytes,
                            mi355_stream_t stream);

/*
 * The legacy ops under their own names. Both take the SAME parameter block, workspace rules and return
 * codes as mi355_unified_attention and run the same kernels; they check that the block really describes the op and fix
 * the op's row semantics, so that a caller binding the legacy names cannot get the unified behaviour by accident:
 *
 *   mi355_context_attention_fwd_v0 - `context_attention_fwd` (legacy/triton_prefix_prefill.py:588-765): context keys
 *     from the paged cache in the layout the strides describe (v0: K [nb, Hk, D/x, page, x], V [nb, Hk, D, page]), the
 *     keys of the tokens being prefilled from the linear k_new / v_new (required). Rows of sequences with
 *     query_len == 1 are left untouched (:83-84): skip_decodes is forced on, only_decodes must be 0.
 *   mi355_paged_attention_v0 - `paged_attention_triton_2d/3d` (legacy/triton_paged_decode_attention_2d.py:283-398,
 *     ..._3d.py:348-499): one query token per sequence (max_seqlen_q == 1, num_tokens == num_seqs), every key from the
 *     cache (k_new / v_new must be NULL).
 * mi355_attn_workspace_bytes() answers for these calls when the block is filled the same way (skip_decodes = 1 for the
 * first).
 */
MI355_API int mi355_context_attention_fwd_v0(const mi355_attn_params* p, void* workspace, size_t workspace_bytes,
                                             mi355_stream_t stream);
MI355_API int mi355_paged_attention_v0(const mi355_attn_params* p, void* workspace, size_t workspace_bytes,
                                       mi355_stream_t stream);

/* 1 if mi355_unified_attention serves these parameters with write_new_kv = 1 (host arithmetic only), else 0: the caller
 * then issues mi355_reshape_and_cache_flash followed by the plain call. (The name is 0.3.0's; since 0.6.0 it answers for
 * prefill steps as well.) */
MI355_API int mi355_decode_write_fusable(const mi355_attn_params* p);

/*
 * Merge of partial attention results over DISJOINT key ranges (library version >= 0.4.0): the exchange step of a
 * cross-GPU split-KV / context-parallel call after the ranks' (out, lse) have been gathered, and the
 * arithmetic of reduce_segments (LIB/kernels/triton_unified_attention.py:804-828) on normalised partials:
 *     lse = log sum_r exp(lse_r),   out = sum_r out_r * exp(lse_r - lse).
 * part_out [parts, num_tokens, Hq, D] contiguous, in the query type `dtype` (bf16 / f16: what the kernels wrote, no f32
 * copy); part_lse [parts, num_tokens, Hq] f32 (a range that saw no key: -inf). out [num_tokens, Hq, D] with the given
 * element strides, lse [num_tokens, Hq] or NULL. 1 <= parts <= 8. One launch, no workspace.
 */
MI355_API int mi355_merge_attention_partials(const void* part_out, const float* part_lse, int parts, void* out, float* lse, int dtype,
                                             int num_tokens, int num_q_heads, int head_size, int64_t out_stride_token, int64_t out_stride_head,
                                             int64_t lse_stride_token, mi355_stream_t stream);

/* Paged-cache write, see mi355_cache_params. */
MI355_API int mi355_reshape_and_cache_flash(const mi355_cache_params* p, mi355_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355_ATTN_H */
