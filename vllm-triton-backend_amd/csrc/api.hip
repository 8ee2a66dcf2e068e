// C ABI of libmi355_attn.so (include/mi355_attn.h): argument validation and kernel dispatch.
// Host-side only; mirrors the dispatch of unified_attention
// (LIB/kernels/triton_unified_attention.py:861-884): batches containing a prefill take the
// single-pass Q-block kernel, decode-only batches take split-KV + reduce.
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "common.h"

namespace mi355 {

static thread_local char g_error[512] = "";
static thread_local const char* g_kernel = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}
void set_kernel_name(const char* name) { g_kernel = name; }
const char* mi355_last_kernel_name() { return g_kernel; }

static bool is_q_dtype(int d) { return d == MI355_F32 || d == MI355_F16 || d == MI355_BF16; }
static bool is_fp8(int d) { return d == MI355_FP8_E4M3 || d == MI355_FP8_E5M2; }

static int validate(const mi355_attn_params* p) {
  if (!p) { set_error("params is NULL"); return MI355_ERR_BAD_ARG; }
  if (p->num_tokens < 0 || p->num_seqs < 0) { set_error("negative num_tokens/num_seqs"); return MI355_ERR_BAD_ARG; }
  if (p->num_tokens == 0 || p->num_seqs == 0) return MI355_OK;
  if (!p->q || !p->out || !p->k_cache || !p->v_cache || !p->block_table || !p->cu_seqlens_q || !p->seqused_k) {
    set_error("q/out/k_cache/v_cache/block_table/cu_seqlens_q/seqused_k must be non-NULL");
    return MI355_ERR_BAD_ARG;
  }
  if ((p->k_new == nullptr) != (p->v_new == nullptr)) { set_error("k_new and v_new must be given together"); return MI355_ERR_BAD_ARG; }
  if (!is_q_dtype(p->q_dtype)) { set_error("q dtype %d is not one of f32/f16/bf16", p->q_dtype); return MI355_ERR_UNSUPPORTED; }
  if (p->kv_dtype != p->q_dtype && !is_fp8(p->kv_dtype)) {
    set_error("kv dtype %d must equal the q dtype %d or be an fp8 type", p->kv_dtype, p->q_dtype);
    return MI355_ERR_UNSUPPORTED;
  }
  if (p->num_q_heads <= 0 || p->num_kv_heads <= 0 || p->num_q_heads % p->num_kv_heads != 0) {
    set_error("num_q_heads %d must be a positive multiple of num_kv_heads %d", p->num_q_heads, p->num_kv_heads);
    return MI355_ERR_BAD_ARG;
  }
  if (p->head_size <= 0 || p->page_size <= 0 || p->k_x <= 0 || p->head_size % p->k_x != 0) {
    set_error("bad head_size %d / page_size %d / k_x %d", p->head_size, p->page_size, p->k_x);
    return MI355_ERR_BAD_ARG;
  }
  if (p->sliding_window < 0) { set_error("sliding_window must be >= 0"); return MI355_ERR_BAD_ARG; }
  if (p->max_seqlen_q < 0 || p->max_seqlen_k < 0) {      // bounds, not exact: 0 = none given (the kernels read every row to its own length)
    set_error("max_seqlen_q %d / max_seqlen_k %d must not be negative", p->max_seqlen_q, p->max_seqlen_k);
    return MI355_ERR_BAD_ARG;
  }
  if (p->lse && p->lse_stride_token < p->num_q_heads) { set_error("lse_stride_token %lld is smaller than num_q_heads %d", (long long)p->lse_stride_token, p->num_q_heads); return MI355_ERR_BAD_ARG; }
  if (p->decode_rows_hint < 0) { set_error("decode_rows_hint must not be negative"); return MI355_ERR_BAD_ARG; }
  if (p->skip_decodes < 0 || p->only_decodes < 0) { set_error("skip_decodes / only_decodes are query-length thresholds: not negative"); return MI355_ERR_BAD_ARG; }
  if (p->skip_decodes && p->only_decodes) { set_error("skip_decodes and only_decodes exclude each other"); return MI355_ERR_BAD_ARG; }
  if (p->non_causal && (p->sliding_window > 0 || p->alibi_slopes || p->write_new_kv)) {
    set_error("non_causal attention takes neither a sliding window nor ALiBi slopes nor a fused cache write");
    return MI355_ERR_UNSUPPORTED;
  }
  if (p->slot_mapping && p->slot_mapping_i32) { set_error("at most one of slot_mapping / slot_mapping_i32 may be non-NULL"); return MI355_ERR_BAD_ARG; }
  if (p->new_kv_all_rows && (!p->k_new || p->write_new_kv)) { set_error("new_kv_all_rows needs k_new / v_new and excludes write_new_kv"); return MI355_ERR_BAD_ARG; }
  if (p->write_new_kv) {
    if (!p->k_new || !p->v_new) { set_error("write_new_kv needs k_new / v_new"); return MI355_ERR_BAD_ARG; }
    if (p->skip_decodes || p->only_decodes) { set_error("write_new_kv excludes skip_decodes / only_decodes"); return MI355_ERR_BAD_ARG; }
    const bool decode_step = p->max_seqlen_q == 1 && p->num_tokens == p->num_seqs;
    // a decode step (one query token per sequence: the split-KV kernel), or - library 0.6.0 - a prefill step the short-prompt
    // kernel serves in one launch (prefill_write_fusable); nothing else carries a fused write
    if (decode_step ? !decode_write_fusable(*p) : !prefill_write_fusable(*p)) {
      set_error("write_new_kv: this step is not served with a fused cache write (ask mi355_decode_write_fusable first; max_seqlen_q %d, num_tokens %d, num_seqs %d)",
                p->max_seqlen_q, p->num_tokens, p->num_seqs);
      return MI355_ERR_UNSUPPORTED;
    }
  }
  return MI355_OK;
}

// AUTO policy. The reference picks 3D iff max_seqlen_q == 1 (:884). We do the same for the
// split-KV decode kernel, send everything else the MFMA prefill kernel covers there, and the
// remainder to the generic kernel.
enum class Path { Generic, Decode, Prefill, PrefillPlusDecode, Repacked };

static Path choose(const mi355_attn_params& p) {
  const int sel = p.kernel_select;
  if (p.write_new_kv) {      // validated: the fused decode kernel / the prefill kernels that carry the write take it
    if (p.max_seqlen_q == 1 && p.num_tokens == p.num_seqs) return Path::Decode;
    return (p.num_seqs == 1 || (int64_t)p.num_seqs * p.max_seqlen_q == p.num_tokens) ? Path::Prefill : Path::PrefillPlusDecode;
  }
  // non-causal: prefill_pw_kernel takes it (a context that covers the whole sequence); everything it does not serve
  // (f32, other head sizes, soft-cap ...) runs on the shape-agnostic kernel, which reads linear k_new / v_new itself
  const bool nc_fast = p.non_causal && sel != MI355_SELECT_GENERIC && sel != MI355_SELECT_3D && (p.k_new ? repack_supported(p) && prefill_pw_applicable(repacked_params(p, nullptr, 0))
                                                                               : prefill_supported(p) && prefill_pw_applicable(p));
  if (p.non_causal && !nc_fast) return Path::Generic;
  if (sel == MI355_SELECT_GENERIC) return Path::Generic;
  // legacy ops: cache in the v0 layout and/or new keys in linear tensors - gathered into a flash-layout scratch
  // cache first, then the kernels below run on that (repack.hip)
  if (sel != MI355_SELECT_3D && repack_supported(p)) return Path::Repacked;
  const bool dec_ok = decode_supported(p);
  const bool pre_ok = prefill_supported(p);
  if (sel == MI355_SELECT_3D) return dec_ok ? Path::Decode : Path::Generic;
  if (sel == MI355_SELECT_2D) return pre_ok ? Path::Prefill : Path::Generic;
  if (p.max_seqlen_q <= 1 && dec_ok) return Path::Decode;
  // multi-token decode steps (speculative decoding / MTP verification): a few query tokens per sequence share one
  // stream of the sequence's K/V in the decode kernel's matrix columns. More tokens than one work unit holds would
  // stream it once per unit (64 x 8 tokens x 8192 keys, 4 per unit: 838 us against the prefill kernel's 438).
  if (dec_ok && decode_pack_shift(p) && p.max_seqlen_q <= (1 << decode_pack_shift(p))) return Path::Decode;
  // mixed batch: prefill rows on the MFMA Q-block kernel, query_len == 1 rows on the split-KV kernel
  // (what the reference's legacy glue does with two kernels, chunked_prefill_paged_decode:28-117;
  // its unified 2D kernel instead pads every decode row to a BLOCK_M-row Q block)
  // (a batch whose sequences all carry max_seqlen_q tokens has no such row: host-known, no device read)
  const bool uniform_prefill = (int64_t)p.num_seqs * p.max_seqlen_q == p.num_tokens;
  if (pre_ok && dec_ok && p.num_seqs > 1 && !uniform_prefill && !p.skip_decodes && !p.only_decodes) return Path::PrefillPlusDecode;
  if (pre_ok) return Path::Prefill;
  if (dec_ok) return Path::Decode;
  return Path::Generic;
}

// workspace of a call that needs no repacking: follows the dispatch decision, host-known sizes only
static size_t plain_workspace_bytes(const mi355_attn_params& p) {
  switch (choose(p)) {
    case Path::Decode: return decode_workspace_bytes(p);
    case Path::Prefill: return prefill_workspace_bytes(p);
    case Path::PrefillPlusDecode: {          // the two run one after the other on the stream and share the bytes
      mi355_attn_params pp = p, pd = p;
      pp.skip_decodes = pd.only_decodes = p.write_new_kv ? 1 : decode_rows_max_q(p);
      const size_t a = prefill_workspace_bytes(pp), b = decode_workspace_bytes(pd);
      return a > b ? a : b;
    }
    default: return 0;
  }
}

static int dispatch_plain(const mi355_attn_params& p, void* workspace, size_t workspace_bytes, hipStream_t s) {
  int rc;
  switch (choose(p)) {
    case Path::Decode:
      return launch_decode(p, workspace, workspace_bytes, s);
    case Path::Prefill:
      return launch_prefill_ws(p, workspace, workspace_bytes, s);
    case Path::PrefillPlusDecode: {
      mi355_attn_params pp = p, pd = p;
      pp.skip_decodes = pd.only_decodes = p.write_new_kv ? 1 : decode_rows_max_q(p);    // (1, or what one packed decode unit holds: multi-token decode rows; a fused cache write: one-token rows only)
      rc = launch_prefill_ws(pp, workspace, workspace_bytes, s);
      static thread_local char prefill_name[64];
      snprintf(prefill_name, sizeof(prefill_name), "%s", g_kernel);
      if (rc == MI355_OK) rc = launch_decode(pd, workspace, workspace_bytes, s);
      if (rc == MI355_OK) {                      // "<prefill kernel>+<decode kernel>"
        static thread_local char both[96];
        snprintf(both, sizeof(both), "%s+%s", prefill_name, g_kernel);
        set_kernel_name(both);
      }
      return rc;
    }
    default:
      rc = launch_generic(p, s);
      if (rc == MI355_OK) set_kernel_name("generic");
      return rc;
  }
}

// Repacked call: how its query_len == 1 rows are served, and the bytes the attention kernels want ahead of the scratch.
struct RepackPlan {
  bool direct_decode;        // decode rows read the caller's cache with the split-KV kernel (no copy of their keys)
  mi355_attn_params pd;      // that decode call
  size_t head;
};

static RepackPlan plan_repack(const mi355_attn_params& p) {
  RepackPlan r;
  r.pd = p;
  r.pd.k_new = r.pd.v_new = nullptr;         // decode rows never read the linear source (generic_attn.hip: use_new)
  r.pd.only_decodes = 1;
  const bool uniform_prefill = (int64_t)p.num_seqs * p.max_seqlen_q == p.num_tokens;
  r.direct_decode = !p.skip_decodes && !p.new_kv_all_rows && !uniform_prefill && p.num_seqs > 1 && decode_supported(r.pd);
  mi355_attn_params pr = repacked_params(p, nullptr, 0);
  if (r.direct_decode) pr.skip_decodes = 1;
  const size_t a = plain_workspace_bytes(pr), b = r.direct_decode ? decode_workspace_bytes(r.pd) : 0;
  r.head = a > b ? a : b;
  // the first 256 KiB of every workspace are the split-KV arrival counters, which stay zero between calls
  // (include/mi355_attn.h): the scratch never lands there, also when this call needs no split-KV partials
  const size_t counters = (size_t)256 << 10;
  if (r.head < counters) r.head = counters;
  return r;
}

}  // namespace mi355

using namespace mi355;

extern "C" {

int mi355_attn_version(void) { return MI355_ATTN_VERSION; }

const char* mi355_last_error(void) { return g_error; }

const char* mi355_last_kernel(void) { return g_kernel; }

size_t mi355_attn_workspace_bytes(const mi355_attn_params* p) {
  if (!p || p->num_tokens <= 0 || p->num_seqs <= 0) return 0;
  if (validate(p) != MI355_OK) return 0;
  if (choose(*p) == Path::Repacked) return repack_scratch_bytes(*p, plan_repack(*p).head);
  return plain_workspace_bytes(*p);
}

int mi355_unified_attention(const mi355_attn_params* p, void* workspace, size_t workspace_bytes,
                            mi355_stream_t stream) {
  int rc = validate(p);
  if (rc != MI355_OK) return rc;
  if (p->num_tokens == 0 || p->num_seqs == 0) return MI355_OK;
  hipStream_t s = (hipStream_t)stream;
  if (choose(*p) != Path::Repacked) return dispatch_plain(*p, workspace, workspace_bytes, s);

  const RepackPlan plan = plan_repack(*p);
  const size_t need = repack_scratch_bytes(*p, plan.head);
  if (!workspace || workspace_bytes < need) {
    set_error("workspace of %zu bytes is smaller than the %zu this call needs (mi355_attn_workspace_bytes)", workspace_bytes, need);
    return MI355_ERR_WORKSPACE;
  }
  rc = launch_repack(*p, workspace, plan.head, p->skip_decodes || plan.direct_decode, s);
  if (rc != MI355_OK) return rc;
  mi355_attn_params pr = repacked_params(*p, workspace, plan.head);
  if (plan.direct_decode) pr.skip_decodes = 1;
  rc = dispatch_plain(pr, workspace, plan.head, s);
  static thread_local char name[128];
  snprintf(name, sizeof(name), "repack+%s", g_kernel);
  if (rc == MI355_OK && plan.direct_decode) {
    rc = launch_decode(plan.pd, workspace, plan.head, s);
    const size_t n = strlen(name);
    snprintf(name + n, sizeof(name) - n, "+%s", g_kernel);
  }
  if (rc == MI355_OK) set_kernel_name(name);
  return rc;
}

int mi355_decode_write_fusable(const mi355_attn_params* p) {
  if (!p || p->num_tokens <= 0 || p->num_seqs <= 0) return 0;
  if (p->max_seqlen_q == 1 && p->num_tokens == p->num_seqs) return decode_write_fusable(*p) ? 1 : 0;
  return prefill_write_fusable(*p) ? 1 : 0;      // (library 0.6.0: a prefill step in one launch of the short-prompt kernel)
}

int mi355_context_attention_fwd_v0(const mi355_attn_params* p, void* workspace, size_t workspace_bytes, mi355_stream_t stream) {
  if (!p) { set_error("params is NULL"); return MI355_ERR_BAD_ARG; }
  if (!p->k_new || !p->v_new) { set_error("context_attention_fwd needs the linear k_new / v_new of the tokens being prefilled"); return MI355_ERR_BAD_ARG; }
  if (p->only_decodes) { set_error("context_attention_fwd never computes query_len == 1 rows (only_decodes must be 0)"); return MI355_ERR_BAD_ARG; }
  mi355_attn_params q = *p;
  q.skip_decodes = 1;                          // rows of query_len == 1 sequences stay untouched (triton_prefix_prefill.py:83-84)
  return mi355_unified_attention(&q, workspace, workspace_bytes, stream);
}

int mi355_paged_attention_v0(const mi355_attn_params* p, void* workspace, size_t workspace_bytes, mi355_stream_t stream) {
  if (!p) { set_error("params is NULL"); return MI355_ERR_BAD_ARG; }
  if (p->k_new || p->v_new) { set_error("paged_attention reads every key from the cache (k_new / v_new must be NULL)"); return MI355_ERR_BAD_ARG; }
  if (p->max_seqlen_q != 1 || p->num_tokens != p->num_seqs) {
    set_error("paged_attention is a decode op: one query token per sequence (max_seqlen_q %d, num_tokens %d, num_seqs %d)", p->max_seqlen_q, p->num_tokens, p->num_seqs);
    return MI355_ERR_BAD_ARG;
  }
  return mi355_unified_attention(p, workspace, workspace_bytes, stream);
}

int mi355_merge_attention_partials(const void* part_out, const float* part_lse, int parts, void* out, float* lse, int dtype, int num_tokens,
                                   int num_q_heads, int head_size, int64_t out_stride_token, int64_t out_stride_head, int64_t lse_stride_token,
                                   mi355_stream_t stream) {
  if (num_tokens < 0 || num_q_heads <= 0 || head_size <= 0) { set_error("merge: bad sizes"); return MI355_ERR_BAD_ARG; }
  if (num_tokens == 0) return MI355_OK;
  if (!part_out || !part_lse || !out) { set_error("merge: part_out / part_lse / out must be non-NULL"); return MI355_ERR_BAD_ARG; }
  if (lse && lse_stride_token < num_q_heads) { set_error("merge: lse_stride_token is smaller than num_q_heads"); return MI355_ERR_BAD_ARG; }
  return launch_merge_partials(part_out, part_lse, parts, out, lse, dtype, num_tokens, num_q_heads, head_size, out_stride_token, out_stride_head,
                               lse_stride_token, (hipStream_t)stream);
}

int mi355_reshape_and_cache_flash(const mi355_cache_params* p, mi355_stream_t stream) {
  if (!p) { set_error("params is NULL"); return MI355_ERR_BAD_ARG; }
  if (p->num_tokens < 0) { set_error("negative num_tokens"); return MI355_ERR_BAD_ARG; }
  if (p->num_tokens == 0) return MI355_OK;
  if (!p->key || !p->value || !p->k_cache || !p->v_cache) { set_error("key/value/k_cache/v_cache must be non-NULL"); return MI355_ERR_BAD_ARG; }
  if ((p->slot_mapping == nullptr) == (p->slot_mapping_i32 == nullptr)) {
    set_error("exactly one of slot_mapping / slot_mapping_i32 must be non-NULL");
    return MI355_ERR_BAD_ARG;
  }
  if (p->num_kv_heads <= 0 || p->head_size <= 0 || p->page_size <= 0) { set_error("bad num_kv_heads/head_size/page_size"); return MI355_ERR_BAD_ARG; }
  return launch_cache_write(*p, (hipStream_t)stream);
}

}  // extern "C"
