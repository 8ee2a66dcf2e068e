"""Legacy op surface of the reference (`ibm_triton_lib.kernels.legacy`, LIB/kernels/legacy/__init__.py:18-30):
same names and keyword signatures, old vLLM v0 cache layout
(K [num_blocks, Hk, D/x, block_size, x] or [num_blocks, Hk, D, block_size]; V [num_blocks, Hk, D, block_size]).

All four ops are served by libmi355_attn.so through `mi355_unified_attention`: the C ABI's stride
description covers the v0 layout and its optional linear "new token" K/V source covers
context_attention_fwd, so one launch replaces the reference's kernels (and the two launches of
chunked_prefill_paged_decode). `paged_attention_2d/3d` over a 16-bit 5-D v0 cache (x = 8) with head size
64/128/256 run on the split-KV MFMA decode kernel ("decode_*_v0": a (page, head) tile of that layout is one
contiguous block of MFMA-shaped 16-byte pieces). `context_attention_fwd` and `chunked_prefill_paged_decode` (16-bit
or fp8 cache) gather each sequence's keys (context pages of either v0 form, new rows from the linear tensors) into a
flash-layout scratch cache in the workspace and run the matrix-core prefill kernel on it ("repack+prefill_dma..."),
decode rows of a mixed batch going straight to the v0 decode kernel when it covers the cache; `paged_attention_2d/3d`
over fp8 or 4-D caches take the same gather pass and then the split-KV kernel; fp32 runs on the shape-agnostic HIP kernel.
"""

from __future__ import annotations

import torch

from ... import _lib
from ..unified import fill_attn_params, launch

_arange_cache: dict = {}


def _fp8_view(cache: torch.Tensor, kv_cache_dtype: str) -> torch.Tensor:
    """FP8 caches arrive as uint8 (triton_prefix_prefill.py:620-634; triton_paged_decode_attention_2d.py:305-319)."""
    if "fp8" not in kv_cache_dtype:
        return cache
    assert cache.dtype == torch.uint8
    if kv_cache_dtype in ("fp8", "fp8_e4m3"):
        return cache.view(torch.float8_e4m3fn)
    if kv_cache_dtype == "fp8_e5m2":
        return cache.view(torch.float8_e5m2)
    raise ValueError("Unsupported FP8 dtype:", kv_cache_dtype)


def _decode_cu_seqlens(num_seqs: int, device) -> torch.Tensor:
    key = (device.type, device.index, num_seqs)
    t = _arange_cache.get(key)
    if t is None:
        t = torch.arange(num_seqs + 1, dtype=torch.int32, device=device)
        _arange_cache[key] = t
    return t


# The legacy signatures carry no maximum key length. The block table's width is a bound the host knows without a device
# read, and for the calls the library serves straight from the caller's cache (paged_attention_2d/3d over a 16-bit 5-D
# cache) it only sizes the split plan. The repack path, though, sizes its scratch cache from the bound
# (num_seqs * ceil(bound / 16) pages of K and of V), and a vLLM-sized table (max_model_len / block_size entries) times a
# few hundred sequences asks for tens of GiB. So: the library is asked what the call needs with the table bound
# (mi355_attn_workspace_bytes: host arithmetic); only beyond _SCRATCH_SOFT_LIMIT is a tighter bound needed - the
# caller's `max_seq_len=` (an extension of the reference signatures: no device read, graph-capturable) or, failing
# that, ONE read-back of seq_lens.max() (a host sync; refused with a message while the stream is capturing). A scratch
# that still exceeds _SCRATCH_HARD_LIMIT is refused with a message instead of an allocation failure.
_SCRATCH_SOFT_LIMIT = 256 << 20
_SCRATCH_HARD_LIMIT = 16 << 30


def _bounded_params(build, table_bound: int, extra: int, seq_lens: torch.Tensor, num_seqs: int, max_seq_len, name: str):
    """`build(bound)` -> (params, keepalive). Returns them for the loosest bound whose workspace is acceptable."""
    import ctypes as C

    lib = _lib.load()
    bound = table_bound + extra
    if max_seq_len is not None:
        bound = min(bound, int(max_seq_len))
    p, keep = build(bound)
    need = lib.mi355_attn_workspace_bytes(C.byref(p))
    if need > _SCRATCH_SOFT_LIMIT and max_seq_len is None and num_seqs > 0:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError(f"mi355_attn.{name}: a {need >> 20} MiB scratch follows from the block table's width; pass max_seq_len= "
                               "(reading seq_lens back would synchronise, which a capturing stream cannot)")
        bound = min(bound, int(seq_lens[:num_seqs].max().item()))
        p, keep = build(bound)
        need = lib.mi355_attn_workspace_bytes(C.byref(p))
    if need > _SCRATCH_HARD_LIMIT:
        raise ValueError(f"mi355_attn.{name}: {num_seqs} sequences of up to {bound} keys need a {need >> 20} MiB scratch cache for "
                         "the legacy-layout repack; split the batch or use unified_attention over the flash layout")
    return p, keep


def _require_gpu(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"mi355_attn.{name} needs tensors on an MI355X (cuda/hip) device; there is no CPU path")


@torch.inference_mode()
def context_attention_fwd(
    q, k, v, o, kv_cache_dtype: str, k_cache, v_cache, b_loc, b_start_loc, b_seq_len, max_input_len,
    k_scale: torch.Tensor, v_scale: torch.Tensor, alibi_slopes=None, sliding_window=None, sm_scale=None, max_seq_len=None,
):
    """Chunked prefill: context keys from the paged cache, new keys from the linear k/v; rows of
    sequences with query_len == 1 are left untouched (LIB/kernels/legacy/triton_prefix_prefill.py:588-765, :83-84)."""
    _require_gpu(q, "context_attention_fwd")
    k_cache, v_cache = _fp8_view(k_cache, kv_cache_dtype), _fp8_view(v_cache, kv_cache_dtype)
    if (k_cache.dtype == torch.uint8 or v_cache.dtype == torch.uint8) and kv_cache_dtype == "auto":
        raise ValueError("kv_cache_dtype='auto' unsupported for FP8 KV Cache prefill kernel")
    Lq, Lk, Lv = q.shape[-1], k.shape[-1], v.shape[-1]
    assert Lq == Lk and Lk == Lv
    if sm_scale is None:
        sm_scale = 1.0 / (Lq**0.5)
    assert b_seq_len.shape[0] + 1 == len(b_start_loc)
    if sliding_window is None or sliding_window <= 0:
        sliding_window = 0
    # the signature carries no maximum key length: context fits the block table, new keys number at most max_input_len
    p, keep = _bounded_params(
        lambda bound: fill_attn_params(
            q, k_cache, v_cache, o, b_start_loc, max_input_len, b_seq_len, bound, sm_scale,
            (sliding_window - 1, 0) if sliding_window else (-1, -1), b_loc, 0.0, k_scale, v_scale, alibi_slopes, None,
            k_new=k, v_new=v, skip_decodes=True, legacy_v0_layout=True),
        b_loc.shape[1] * v_cache.shape[3], max_input_len, b_seq_len, b_seq_len.shape[0], max_seq_len, "context_attention_fwd")
    launch(p, q.device, "mi355_context_attention_fwd_v0")
    del keep


def _paged_decode(output, query, key_cache, value_cache, scale, k_scale, v_scale, kv_cache_dtype, block_tables, seq_lens,
                  alibi_slopes, block_size, num_seqs, num_query_heads, num_queries_per_kv, head_size, name, max_seq_len=None):
    _require_gpu(query, name)
    key_cache, value_cache = _fp8_view(key_cache, kv_cache_dtype), _fp8_view(value_cache, kv_cache_dtype)
    assert num_seqs <= 4096  # the reference's static launch grid (triton_paged_decode_attention_2d.py:355)
    assert value_cache.shape[3] == block_size and query.shape[1] == num_query_heads and query.shape[2] == head_size
    cu = _decode_cu_seqlens(num_seqs, query.device)
    # the legacy signature carries no maximum sequence length: the block table's width bounds it (host-known)
    p, keep = _bounded_params(
        lambda bound: fill_attn_params(
            query[:num_seqs], key_cache, value_cache, output[:num_seqs], cu, 1, seq_lens[:num_seqs], bound, scale, (-1, -1),
            block_tables, 0.0, k_scale, v_scale, alibi_slopes, None, legacy_v0_layout=True),
        block_tables.shape[1] * block_size, 0, seq_lens, num_seqs, max_seq_len, name)
    launch(p, query.device, "mi355_paged_attention_v0")
    del keep


def paged_attention_2d(output, query, key_cache, value_cache, scale, k_scale, v_scale, kv_cache_dtype, block_tables, seq_lens,
                       alibi_slopes, block_size, num_seqs, num_query_heads, num_queries_per_kv, head_size, max_seq_len=None):
    """Paged decode over the v0 layout (LIB/kernels/legacy/triton_paged_decode_attention_2d.py:283-398)."""
    _paged_decode(output, query, key_cache, value_cache, scale, k_scale, v_scale, kv_cache_dtype, block_tables, seq_lens,
                  alibi_slopes, block_size, num_seqs, num_query_heads, num_queries_per_kv, head_size, "paged_attention_2d", max_seq_len)


def paged_attention_3d(output, query, key_cache, value_cache, scale, k_scale, v_scale, kv_cache_dtype, block_tables, seq_lens,
                       alibi_slopes, block_size, num_seqs, num_query_heads, num_queries_per_kv, head_size, max_seq_len=None):
    """Split-KV variant in the reference (LIB/kernels/legacy/triton_paged_decode_attention_3d.py:348-499); same result."""
    _paged_decode(output, query, key_cache, value_cache, scale, k_scale, v_scale, kv_cache_dtype, block_tables, seq_lens,
                  alibi_slopes, block_size, num_seqs, num_query_heads, num_queries_per_kv, head_size, "paged_attention_3d", max_seq_len)


def chunked_prefill_paged_decode(query, key, value, output, kv_cache_dtype, key_cache, value_cache, block_table, query_start_loc,
                                 seq_lens, max_query_len, k_scale, v_scale, alibi_slopes, sliding_window, scale, max_seq_len=None):
    """context_attention_fwd for the prefills + paged decode for query_len == 1 rows
    (LIB/kernels/legacy/triton_chunked_prefill_paged_decode.py:28-117) in ONE launch: prefill rows take
    new keys from the linear key/value, decode rows read everything from the cache."""
    _require_gpu(query, "chunked_prefill_paged_decode")
    key_cache, value_cache = _fp8_view(key_cache, kv_cache_dtype), _fp8_view(value_cache, kv_cache_dtype)
    sw = sliding_window if sliding_window is not None and sliding_window > 0 else 0
    p, keep = _bounded_params(
        lambda bound: fill_attn_params(
            query, key_cache, value_cache, output, query_start_loc, max_query_len, seq_lens, bound, scale,
            (sw - 1, 0) if sw else (-1, -1), block_table, 0.0, k_scale, v_scale, alibi_slopes, None,
            k_new=key, v_new=value, legacy_v0_layout=True),
        block_table.shape[1] * value_cache.shape[3], max_query_len, seq_lens, seq_lens.shape[0], max_seq_len, "chunked_prefill_paged_decode")
    launch(p, query.device)
    del keep


__all__ = ["context_attention_fwd", "paged_attention_2d", "paged_attention_3d", "chunked_prefill_paged_decode"]
