"""Registered torch ops over the C ABI: torch.ops.mi355_attn.{unified_attention, reshape_and_cache_flash,
decode_attention_and_cache_write, prefill_attention_and_cache_write}.

The reference's forward() calls a registered op for the cache write (torch.ops._C_cache_ops.reshape_and_cache_flash,
LIB/backend/triton_attn.py:396-405) and a Python function for the attention; vLLM wraps the whole backend call in its own
custom op. Registering both here makes the two calls visible to the dispatcher on their own: they trace as
opaque nodes (fake implementations below: both only mutate their output arguments) instead of breaking the graph at the
ctypes boundary. The implementations are the same host code as mi355_attn.kernels: CUDA/HIP tensors only, no CPU kernel
is registered - a CPU tensor fails in the dispatcher, loudly.
"""

from __future__ import annotations

from typing import Optional

import torch

from .kernels.cache import reshape_and_cache_flash as _reshape_and_cache_flash
from .kernels.unified import decode_attention_and_cache_write as _decode_attention_and_cache_write
from .kernels.unified import prefill_attention_and_cache_write as _prefill_attention_and_cache_write
from .kernels.unified import unified_attention as _unified_attention

_DEF = torch.library.Library("mi355_attn", "DEF")
_DEF.define(
    "unified_attention(Tensor q, Tensor k, Tensor v, Tensor(a!) out, Tensor cu_seqlens_q, int max_seqlen_q, Tensor seqused_k, "
    "int max_seqlen_k, float softmax_scale, int window_left, int window_right, Tensor block_table, float softcap, "
    "Tensor? k_descale, Tensor? v_descale, Tensor? alibi_slopes, str kv_cache_dtype, int decode_rows_hint=0) -> ()"
)
_DEF.define(
    "reshape_and_cache_flash(Tensor key, Tensor value, Tensor(a!) key_cache, Tensor(b!) value_cache, Tensor slot_mapping, "
    "str kv_cache_dtype, Tensor? k_scale, Tensor? v_scale) -> ()"
)

_DEF.define(
    "decode_attention_and_cache_write(Tensor q, Tensor key, Tensor value, Tensor(a!) key_cache, Tensor(b!) value_cache, Tensor(c!) out, "
    "Tensor cu_seqlens_q, Tensor seqused_k, int max_seqlen_k, float softmax_scale, Tensor block_table, Tensor slot_mapping, "
    "Tensor? k_scale, Tensor? v_scale, str kv_cache_dtype) -> ()"
)

_DEF.define(
    "prefill_attention_and_cache_write(Tensor q, Tensor key, Tensor value, Tensor(a!) key_cache, Tensor(b!) value_cache, Tensor(c!) out, "
    "Tensor cu_seqlens_q, int max_seqlen_q, Tensor seqused_k, int max_seqlen_k, float softmax_scale, Tensor block_table, Tensor slot_mapping, "
    "Tensor? k_scale, Tensor? v_scale, str kv_cache_dtype, int decode_rows_hint=0) -> ()"
)

_FP8 = {"fp8": torch.float8_e4m3fn, "fp8_e4m3": torch.float8_e4m3fn, "fp8_e5m2": torch.float8_e5m2}


def _unified_attention_impl(q, k, v, out, cu_seqlens_q, max_seqlen_q, seqused_k, max_seqlen_k, softmax_scale, window_left, window_right,
                            block_table, softcap, k_descale: Optional[torch.Tensor], v_descale: Optional[torch.Tensor],
                            alibi_slopes: Optional[torch.Tensor], kv_cache_dtype: str, decode_rows_hint: int = 0) -> None:
    if kv_cache_dtype in _FP8 and k.dtype == torch.uint8:      # vLLM hands fp8 caches over as uint8
        k, v = k.view(_FP8[kv_cache_dtype]), v.view(_FP8[kv_cache_dtype])
    _unified_attention(q=q, k=k, v=v, out=out, cu_seqlens_q=cu_seqlens_q, max_seqlen_q=max_seqlen_q, seqused_k=seqused_k,
                       max_seqlen_k=max_seqlen_k, avg_seqlen_q=0, avg_seqlen_k=0, softmax_scale=softmax_scale, causal=True,
                       window_size=(window_left, window_right), block_table=block_table, softcap=softcap, q_descale=None,
                       k_descale=k_descale, v_descale=v_descale, alibi_slopes=alibi_slopes, decode_rows_hint=decode_rows_hint)


def _reshape_and_cache_flash_impl(key, value, key_cache, value_cache, slot_mapping, kv_cache_dtype: str,
                                  k_scale: Optional[torch.Tensor], v_scale: Optional[torch.Tensor]) -> None:
    _reshape_and_cache_flash(key, value, key_cache, value_cache, slot_mapping, kv_cache_dtype, k_scale, v_scale)


def _decode_attention_and_cache_write_impl(q, key, value, key_cache, value_cache, out, cu_seqlens_q, seqused_k, max_seqlen_k, softmax_scale,
                                          block_table, slot_mapping, k_scale: Optional[torch.Tensor], v_scale: Optional[torch.Tensor],
                                          kv_cache_dtype: str) -> None:
    """A decode step (one query token per sequence) in ONE launch when the fused kernel serves the configuration, else
    the cache write followed by the attention: the same results either way."""
    kc, vc = key_cache, value_cache
    if kv_cache_dtype in _FP8 and kc.dtype == torch.uint8:
        kc, vc = kc.view(_FP8[kv_cache_dtype]), vc.view(_FP8[kv_cache_dtype])
    n = q.shape[0]
    if _decode_attention_and_cache_write(q, key[:n], value[:n], kc, vc, out, seqused_k, max_seqlen_k, softmax_scale, block_table,
                                         k_scale, v_scale, cu_seqlens_q, slot_mapping if slot_mapping.shape[0] >= n else None):
        return
    _reshape_and_cache_flash(key, value, key_cache, value_cache, slot_mapping, kv_cache_dtype, k_scale, v_scale)
    _unified_attention(q=q, k=kc, v=vc, out=out, cu_seqlens_q=cu_seqlens_q, max_seqlen_q=1, seqused_k=seqused_k, max_seqlen_k=max_seqlen_k,
                       avg_seqlen_q=0, avg_seqlen_k=0, softmax_scale=softmax_scale, causal=True, window_size=(-1, -1), block_table=block_table,
                       softcap=0.0, q_descale=None, k_descale=k_scale, v_descale=v_scale)


def _prefill_attention_and_cache_write_impl(q, key, value, key_cache, value_cache, out, cu_seqlens_q, max_seqlen_q: int, seqused_k, max_seqlen_k: int,
                                           softmax_scale: float, block_table, slot_mapping, k_scale: Optional[torch.Tensor], v_scale: Optional[torch.Tensor],
                                           kv_cache_dtype: str, decode_rows_hint: int = 0) -> None:
    """A plain prefill (or mixed) step: ONE launch when the short-prompt kernel serves it with the cache write inside, else the
    cache write followed by the attention: the same cache bytes and the same output either way."""
    n = q.shape[0]
    if kv_cache_dtype not in _FP8 and _prefill_attention_and_cache_write(
            q, key, value, key_cache, value_cache, out, cu_seqlens_q, max_seqlen_q, seqused_k, max_seqlen_k, softmax_scale, block_table,
            slot_mapping if slot_mapping.shape[0] >= n else None):
        return
    _reshape_and_cache_flash(key, value, key_cache, value_cache, slot_mapping, kv_cache_dtype, k_scale, v_scale)
    _unified_attention_impl(q, key_cache, value_cache, out, cu_seqlens_q, max_seqlen_q, seqused_k, max_seqlen_k, softmax_scale, -1, -1, block_table, 0.0,
                            k_scale, v_scale, None, kv_cache_dtype, decode_rows_hint)


def _nothing(*args, **kwargs) -> None:      # fake / meta: the ops return nothing and only write their (a!)/(b!) arguments
    return None


_IMPL = torch.library.Library("mi355_attn", "IMPL")
_IMPL.impl("unified_attention", _unified_attention_impl, "CUDA")
_IMPL.impl("reshape_and_cache_flash", _reshape_and_cache_flash_impl, "CUDA")
_IMPL.impl("decode_attention_and_cache_write", _decode_attention_and_cache_write_impl, "CUDA")
_IMPL.impl("decode_attention_and_cache_write", _nothing, "Meta")
_IMPL.impl("prefill_attention_and_cache_write", _prefill_attention_and_cache_write_impl, "CUDA")
_IMPL.impl("prefill_attention_and_cache_write", _nothing, "Meta")
_IMPL.impl("unified_attention", _nothing, "Meta")
_IMPL.impl("reshape_and_cache_flash", _nothing, "Meta")

unified_attention = torch.ops.mi355_attn.unified_attention
reshape_and_cache_flash = torch.ops.mi355_attn.reshape_and_cache_flash
decode_attention_and_cache_write = torch.ops.mi355_attn.decode_attention_and_cache_write
prefill_attention_and_cache_write = torch.ops.mi355_attn.prefill_attention_and_cache_write
