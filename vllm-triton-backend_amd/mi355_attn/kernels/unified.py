"""`unified_attention` — same keyword signature as the reference op
(LIB/kernels/triton_unified_attention.py:839-860), served by libmi355_attn.so.

Host logic only: shape/stride extraction, the reference's pre-launch asserts (:861-867) and the
marshalling into `mi355_attn_params`. All arithmetic happens in the HIP kernels.
"""

from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from .. import _lib

_SELECT = {None: _lib.SELECT_AUTO, 0: _lib.SELECT_AUTO, 2: _lib.SELECT_2D, 3: _lib.SELECT_3D, 9: _lib.SELECT_GENERIC}


def _scalar_tensor(x, device) -> Optional[torch.Tensor]:
    """k_descale / v_descale arrive as fp32 tensors (the kernel reads element 0,
    triton_unified_attention.py:438,:453), as None (harness, scripts/callers/unified_triton.py:76-77)
    or as Python floats."""
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        if x.dtype != torch.float32:
            x = x.to(torch.float32)
        return x
    return torch.tensor([float(x)], dtype=torch.float32, device=device)


def fill_attn_params(
    q, k, v, out, cu_seqlens_q, max_seqlen_q, seqused_k, max_seqlen_k, softmax_scale, window_size,
    block_table, softcap, k_descale, v_descale, alibi_slopes, force_selection,
    k_new=None, v_new=None, skip_decodes=False, only_decodes=False, num_segments=0,
    legacy_v0_layout=False, lse=None, write_new_kv=False, non_causal=False, slot_mapping=None, new_kv_all_rows=False,
    decode_rows_hint=0,
):
    """Build the C struct. Returns (params, keepalive) — keepalive holds temporaries whose device
    memory the struct points to."""
    keep = []
    if q.dim() != 3 or out.dim() != 3:
        raise ValueError("q and out must be [num_tokens, num_heads, head_size]")
    if q.stride(2) != 1 or out.stride(2) != 1:
        raise ValueError("last dimension of q and out must be contiguous")
    if cu_seqlens_q.dtype != torch.int32 or seqused_k.dtype != torch.int32 or block_table.dtype != torch.int32:
        raise ValueError("cu_seqlens_q, seqused_k and block_table must be int32")
    if not (cu_seqlens_q.is_contiguous() and seqused_k.is_contiguous()):
        raise ValueError("cu_seqlens_q and seqused_k must be contiguous")
    if block_table.dim() != 2 or block_table.stride(1) != 1:
        raise ValueError("block_table must be [num_seqs, max_blocks] with a contiguous last dimension")

    p = _lib.AttnParams()
    p.q, p.out = q.data_ptr(), out.data_ptr()
    p.k_cache, p.v_cache = k.data_ptr(), v.data_ptr()
    p.block_table = block_table.data_ptr()
    p.cu_seqlens_q = cu_seqlens_q.data_ptr()
    p.seqused_k = seqused_k.data_ptr()
    if alibi_slopes is not None:
        if alibi_slopes.dtype != torch.float32:
            alibi_slopes = alibi_slopes.to(torch.float32)
        if alibi_slopes.device != q.device:
            alibi_slopes = alibi_slopes.to(q.device)
        alibi_slopes = alibi_slopes.contiguous()
        keep.append(alibi_slopes)
        p.alibi_slopes = alibi_slopes.data_ptr()
    ks, vs = _scalar_tensor(k_descale, q.device), _scalar_tensor(v_descale, q.device)
    keep += [ks, vs]
    p.k_scale, p.v_scale = _lib.ptr(ks), _lib.ptr(vs)
    p.q_dtype = _lib.dtype_code(q.dtype)
    p.kv_dtype = _lib.dtype_code(k.dtype)
    p.num_tokens, p.num_q_heads, p.head_size = q.shape
    p.num_seqs = seqused_k.shape[0]
    p.max_seqlen_q, p.max_seqlen_k = int(max_seqlen_q), int(max_seqlen_k)
    p.q_stride_token, p.q_stride_head = q.stride(0), q.stride(1)
    p.out_stride_token, p.out_stride_head = out.stride(0), out.stride(1)
    if legacy_v0_layout:
        # K [num_pages, Hk, D/x, page, x] (or 4-D [num_pages, Hk, D, page] with x = 1), V [num_pages, Hk, D, page]
        # (LIB/kernels/legacy/triton_paged_decode_attention_2d.py:103-104,:385-390)
        p.num_kv_heads = v.shape[1]
        p.page_size = v.shape[3]
        if k.dim() == 5:
            p.k_x = k.shape[4]
            p.k_stride_page, p.k_stride_head, p.k_stride_dx, p.k_stride_slot, p.k_stride_d = k.stride()
        else:
            p.k_x = 1
            p.k_stride_page, p.k_stride_head, p.k_stride_dx, p.k_stride_slot = k.stride()
            p.k_stride_d = 1
        p.v_stride_page, p.v_stride_head, p.v_stride_d, p.v_stride_slot = v.stride()
    else:
        # flash layout [num_pages, page, Hk, D] (triton_unified_attention.py:279-280)
        p.num_kv_heads = k.shape[2]
        p.page_size = v.shape[1]
        p.k_x = p.head_size
        p.k_stride_page, p.k_stride_slot, p.k_stride_head, p.k_stride_d = k.stride()
        p.k_stride_dx = 0
        p.v_stride_page, p.v_stride_slot, p.v_stride_head, p.v_stride_d = v.stride()
    p.block_table_stride = block_table.stride(0)
    if k_new is not None:
        if k_new.stride(2) != 1 or v_new.stride(2) != 1 or k_new.stride() != v_new.stride():
            raise ValueError("linear k/v must share strides and have a contiguous last dimension")
        p.k_new, p.v_new = k_new.data_ptr(), v_new.data_ptr()
        p.new_stride_token, p.new_stride_head = k_new.stride(0), k_new.stride(1)
    p.scale = float(softmax_scale)
    p.softcap = float(softcap) if softcap is not None else 0.0
    p.sliding_window = 1 + int(window_size[0]) if window_size is not None else 0
    p.skip_decodes, p.only_decodes = int(skip_decodes), int(only_decodes)
    try:
        p.kernel_select = _SELECT[force_selection]
    except KeyError:
        raise ValueError(f"force_selection must be None, 2, 3 or 9, got {force_selection}") from None
    p.num_segments = int(num_segments)
    p.decode_rows_hint = int(decode_rows_hint or 0)
    p.write_new_kv = int(bool(write_new_kv))
    p.non_causal = int(bool(non_causal))
    p.new_kv_all_rows = int(bool(new_kv_all_rows))
    if slot_mapping is not None:
        if slot_mapping.dtype not in (torch.int64, torch.int32) or not slot_mapping.is_contiguous() or slot_mapping.shape[0] < q.shape[0]:
            raise ValueError("slot_mapping must be a contiguous int64 or int32 tensor with one entry per query token")
        if slot_mapping.dtype == torch.int64:
            p.slot_mapping = slot_mapping.data_ptr()
        else:
            p.slot_mapping_i32 = slot_mapping.data_ptr()
    if lse is not None:
        if lse.dtype != torch.float32 or lse.dim() != 2 or lse.shape[0] != q.shape[0] or lse.shape[1] != q.shape[1] or lse.stride(1) != 1:
            raise ValueError("softmax_lse must be a float32 [num_tokens, num_heads] tensor with contiguous heads")
        p.lse, p.lse_stride_token = lse.data_ptr(), lse.stride(0)
    return p, keep


def launch(p, device: torch.device, entry: str = "mi355_unified_attention") -> None:
    """One C-ABI call on the current stream: `entry` is mi355_unified_attention or one of the legacy names
    (mi355_context_attention_fwd_v0, mi355_paged_attention_v0: same parameter block and workspace rules)."""
    lib = _lib.load()
    nbytes = lib.mi355_attn_workspace_bytes(C.byref(p))
    ws = _lib.workspace(device, nbytes)
    rc = getattr(lib, entry)(
        C.byref(p), _lib.ptr(ws), nbytes if ws is not None else 0, _lib.current_stream_handle(device)
    )
    _lib.check(rc, entry)


def unified_attention(
    q,
    k,
    v,
    out,
    cu_seqlens_q,
    max_seqlen_q,
    seqused_k,
    max_seqlen_k,
    avg_seqlen_q,
    avg_seqlen_k,
    softmax_scale,
    causal,
    window_size,
    block_table,
    softcap,
    q_descale,
    k_descale,
    v_descale,
    alibi_slopes=None,
    force_selection=None,  # None, 2, 3 to select kernel (9: generic correctness kernel)
    softmax_lse=None,      # extension: float32 [num_tokens, num_heads], receives log(sum(exp(scores))) per row
    decode_rows_hint=0,    # extension: query tokens of this step's decode rows when the caller knows (1 + speculative drafts); 0 = library picks
):
    """Causal paged attention over vLLM block tables; writes `out` in place and returns None, as
    the reference does. `avg_seqlen_q/k` only fed the reference's autotuner keys
    (triton_unified_attention.py:878-881) and are accepted and ignored."""
    assert causal, "Only causal attention is supported"
    assert q_descale is None, "Q scales not supported"
    block_size = v.shape[1]
    assert q.element_size() >= 2 or block_size >= 32, "Block size must be at least 32 for fp8"
    if not q.is_cuda:
        raise RuntimeError("mi355_attn.unified_attention needs tensors on an MI355X (cuda/hip) device; there is no CPU path")
    p, keep = fill_attn_params(
        q, k, v, out, cu_seqlens_q, max_seqlen_q, seqused_k, max_seqlen_k, softmax_scale, window_size,
        block_table, softcap, k_descale, v_descale, alibi_slopes, force_selection, lse=softmax_lse, decode_rows_hint=decode_rows_hint,
    )
    launch(p, q.device)
    del keep
    return None


def decode_attention_and_cache_write(q, key, value, k_cache, v_cache, out, seqused_k, max_seqlen_k, softmax_scale, block_table,
                                     k_descale=None, v_descale=None, cu_seqlens_q=None, slot_mapping=None):
    """One launch for a decode step (every sequence has ONE query token): the new token's key / value (row i of `key` /
    `value` belongs to position seqused_k[i] - 1 of sequence i) is stored into its cache page - quantised for an fp8
    cache exactly as reshape_and_cache_flash stores it - and attended over, by the wave that owns the sequence's last
    tile. Replaces the pair of calls at LIB/backend/triton_attn.py:393-405 + :437 for such steps.
    `slot_mapping` (the step's, as reshape_and_cache_flash would get it): a row whose slot is negative - a padding row of
    a captured graph, triton_attn.py:149-151 - is never stored, whatever its seqused_k / block_table row hold.
    Returns False (and does nothing) when this configuration is not served fused: the caller then issues the two calls."""
    if not q.is_cuda:
        raise RuntimeError("mi355_attn.decode_attention_and_cache_write needs tensors on an MI355X (cuda/hip) device; there is no CPU path")
    n = q.shape[0]
    if cu_seqlens_q is None:
        cu_seqlens_q = _arange_cu(n, q.device)
    p, keep = fill_attn_params(q, k_cache, v_cache, out, cu_seqlens_q, 1, seqused_k, max_seqlen_k, softmax_scale, (-1, -1), block_table, 0.0,
                               k_descale, v_descale, None, None, k_new=key, v_new=value, write_new_kv=True, slot_mapping=slot_mapping)
    if not _lib.load().mi355_decode_write_fusable(C.byref(p)):
        return False
    launch(p, q.device)
    del keep
    return True


def prefill_attention_and_cache_write(q, key, value, k_cache, v_cache, out, cu_seqlens_q, max_seqlen_q, seqused_k, max_seqlen_k, softmax_scale,
                                      block_table, slot_mapping=None):
    """One launch for a PREFILL step the short-prompt kernel serves (library 0.6.0; csrc/prefill_lat.hip): the step's new
    keys / values (`key`, `value` [num_tokens, Hk, D]: row t belongs to query token t) are attended over straight from
    these tensors and stored into their cache pages by the Q block that owns the token - `slot_mapping` (the step's, as
    reshape_and_cache_flash would get it; a negative slot is not stored) or, without one, the position through the block
    table. Replaces the pair of calls at LIB/backend/triton_attn.py:393-405 + :437 for such steps.
    Returns False (and does nothing) when the step is not served fused - more than one launch, a long prompt, an fp8
    cache, features - and the caller issues the two calls."""
    if not q.is_cuda:
        raise RuntimeError("mi355_attn.prefill_attention_and_cache_write needs tensors on an MI355X (cuda/hip) device; there is no CPU path")
    if k_cache.dtype != q.dtype or key.dtype != q.dtype:
        return False
    n = q.shape[0]
    p, keep = fill_attn_params(q, k_cache, v_cache, out, cu_seqlens_q, max_seqlen_q, seqused_k, max_seqlen_k, softmax_scale, (-1, -1), block_table, 0.0,
                               None, None, None, None, k_new=key[:n], v_new=value[:n], write_new_kv=True, slot_mapping=slot_mapping)
    if not _lib.load().mi355_decode_write_fusable(C.byref(p)):
        return False
    launch(p, q.device)
    del keep
    return True


_cu_cache: dict = {}


def _arange_cu(n: int, device: torch.device) -> torch.Tensor:
    key = (device.type, device.index, n)
    t = _cu_cache.get(key)
    if t is None:
        t = torch.arange(n + 1, dtype=torch.int32, device=device)
        _cu_cache[key] = t
    return t
