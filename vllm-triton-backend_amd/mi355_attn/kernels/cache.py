"""`reshape_and_cache_flash` — same positional signature as
`torch.ops._C_cache_ops.reshape_and_cache_flash(key, value, key_cache, value_cache, slot_mapping,
kv_cache_dtype, k_scale, v_scale)` as called at LIB/backend/triton_attn.py:396-405, served by
libmi355_attn.so (mi355_reshape_and_cache_flash)."""

from __future__ import annotations

import ctypes as C

import torch

from .. import _lib


def _cache_view(cache: torch.Tensor, kv_cache_dtype: str) -> torch.Tensor:
    """vLLM stores fp8 caches as uint8 and names the format in `kv_cache_dtype`
    (triton_attn.py:407-409; legacy/triton_prefix_prefill.py:622-634)."""
    if cache.dtype != torch.uint8:
        return cache
    if kv_cache_dtype in ("fp8", "fp8_e4m3"):
        return cache.view(torch.float8_e4m3fn)
    if kv_cache_dtype == "fp8_e5m2":
        return cache.view(torch.float8_e5m2)
    raise ValueError(f"Unsupported FP8 dtype: {kv_cache_dtype}")


def reshape_and_cache_flash(key, value, key_cache, value_cache, slot_mapping, kv_cache_dtype="auto", k_scale=None, v_scale=None):
    if not key.is_cuda:
        raise RuntimeError("mi355_attn.reshape_and_cache_flash needs tensors on an MI355X (cuda/hip) device; there is no CPU path")
    key_cache = _cache_view(key_cache, kv_cache_dtype)
    value_cache = _cache_view(value_cache, kv_cache_dtype)
    if key.dim() != 3 or key_cache.dim() != 4:
        raise ValueError("key/value must be [T, Hk, D] and the caches [num_blocks, block_size, Hk, D]")
    if key.stride(2) != 1 or value.stride(2) != 1 or key_cache.stride(3) != 1 or value_cache.stride(3) != 1:
        raise ValueError("last dimension of key/value/caches must be contiguous")
    if slot_mapping.dtype not in (torch.int64, torch.int32) or not slot_mapping.is_contiguous():
        raise ValueError("slot_mapping must be a contiguous int64 or int32 tensor")
    p = _lib.CacheParams()
    p.key, p.value = key.data_ptr(), value.data_ptr()
    p.k_cache, p.v_cache = key_cache.data_ptr(), value_cache.data_ptr()
    if slot_mapping.dtype == torch.int64:
        p.slot_mapping = slot_mapping.data_ptr()
    else:
        p.slot_mapping_i32 = slot_mapping.data_ptr()
    keep = []
    for name, s in (("k_scale", k_scale), ("v_scale", v_scale)):
        if s is None:
            continue
        if not isinstance(s, torch.Tensor):
            s = torch.tensor([float(s)], dtype=torch.float32, device=key.device)
        elif s.dtype != torch.float32:
            s = s.to(torch.float32)
        keep.append(s)
        setattr(p, name, s.data_ptr())
    p.src_dtype = _lib.dtype_code(key.dtype)
    p.cache_dtype = _lib.dtype_code(key_cache.dtype)
    # slot_mapping may be longer than key (padding for graph capture) or shorter (num_actual_tokens)
    p.num_tokens = min(slot_mapping.shape[0], key.shape[0])
    p.num_kv_heads, p.head_size = key.shape[1], key.shape[2]
    p.page_size = key_cache.shape[1]
    p.key_stride_token, p.key_stride_head = key.stride(0), key.stride(1)
    p.value_stride_token, p.value_stride_head = value.stride(0), value.stride(1)
    p.k_stride_page, p.k_stride_slot, p.k_stride_head = key_cache.stride()[:3]
    p.v_stride_page, p.v_stride_slot, p.v_stride_head = value_cache.stride()[:3]
    rc = _lib.load().mi355_reshape_and_cache_flash(C.byref(p), _lib.current_stream_handle(key.device))
    _lib.check(rc, "mi355_reshape_and_cache_flash")
    del keep
