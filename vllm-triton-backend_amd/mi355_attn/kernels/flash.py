"""`prefill_flash_attention` — the reference package's non-paged variable-length prefill op
(`triton_wrapper_forward_prefill`, LIB/kernels/triton_flash_attention.py:1326-1484, exported at
LIB/kernels/__init__.py:65-67), served by the paged MFMA kernels.

Self-attention prefill (Q and K/V share `cu_seqlens`: the reference harness's use, scripts/callers/triton_3d.py:100-112)
is ONE C-ABI call: the library reads the linear `k`, `v` as its "new token" source (`k_new` / `v_new` of
`mi355_attn_params`, the mechanism behind `context_attention_fwd`) with an empty context, gathers them into the
flash-layout scratch inside the caller's workspace and runs the matrix-core kernels on that. One torch op on the host
side (the per-sequence lengths), no scratch tensor of this module's own, nothing that moves under a captured graph.
When the key ranges differ from the query ranges (`cu_seqlens_k is not cu_seqlens_q`: cross-length causal alignment,
:954-960) K/V are laid out as 16-token pages in a scratch cache kept by this module (one pass of
`reshape_and_cache_flash`) and `unified_attention` runs over them; device-side torch arithmetic only, no host sync.

Served: the "thd" variable-length layout (`q [total_q, Hq, D]`, `k, v [total_k, Hk, D]`), causal masking with the
reference's bottom-right alignment (query t of a sequence sees keys j <= t + seqlen_k - seqlen_q), non-causal attention
(every query row sees its sequence's whole key range), grouped-query heads. Not served (raise `NotImplementedError`,
nothing silently ignored): softmax encodings, dropout; `bias` raises `AssertionError` exactly where the reference's own wrapper
does (its argument check rejects a bias for the variable-length layout it always sets, :126-128, :1341-1342).
"""

from __future__ import annotations

import torch

from .cache import reshape_and_cache_flash
from .unified import fill_attn_params, launch, unified_attention

_PAGE = 16
_scratch: dict = {}
_retired: list = []     # replaced scratch buffers stay alive: a graph captured on one replays into its raw address (_lib.workspace)
_dummy_bt: dict = {}
_dummy_cache: dict = {}


def _scratch_cache(dev, num_pages, hk, d, dtype):
    """The paged scratch K/V of a call whose key ranges differ from its query ranges, kept per (device, STREAM, heads,
    head size, dtype) and grown geometrically. Stream-ordered reuse on one stream: the next call's cache write is
    enqueued behind this call's attention; calls on two streams never share a buffer (one stream's cache write could
    otherwise overwrite pages the other stream's attention is still reading). A capturing stream never allocates
    here: an eager call on the same stream sizes the buffer first."""
    key = (dev.type, dev.index, torch.cuda.current_stream(dev).cuda_stream, hk, d, dtype)
    buf = _scratch.get(key)
    if buf is None or buf.shape[1] < num_pages:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("prefill_flash_attention: run one eager call of the largest shape on this stream before graph capture")
        if buf is not None:
            _retired.append(buf)
        grow = max(num_pages, 2 * (buf.shape[1] if buf is not None else 0))
        buf = torch.empty((2, grow, _PAGE, hk, d), dtype=dtype, device=dev)
        _scratch[key] = buf
    return buf[0, :num_pages], buf[1, :num_pages]


def _same_ranges(cu_q, cu_k) -> bool:
    return cu_q is cu_k or (cu_q.data_ptr() == cu_k.data_ptr() and cu_q.shape == cu_k.shape and cu_q.dtype == cu_k.dtype)


def _self_attention(q, k, v, out, cu, max_seqlen, sm_scale, causal):
    """Q and K/V share their ranges: context length 0, every key from the linear tensors (`k_new` / `v_new`). The cache
    and block-table pointers of the parameter block are never dereferenced then (repack.hip reads the cache for
    positions below the context length only); they are set to live tensors all the same."""
    dev = q.device
    cu32 = cu if cu.dtype == torch.int32 else cu.to(torch.int32)
    lens = cu32[1:] - cu32[:-1]
    bt = _dummy_bt.get((dev.type, dev.index))
    if bt is None:
        bt = torch.zeros((1, 1), dtype=torch.int32, device=dev)
        _dummy_bt[(dev.type, dev.index)] = bt
    hk, d = k.shape[1], k.shape[2]
    key = (dev.type, dev.index, hk, d, k.dtype)
    stand_in = _dummy_cache.get(key)
    if stand_in is None:
        stand_in = torch.zeros((1, _PAGE, hk, d), dtype=k.dtype, device=dev)
        _dummy_cache[key] = stand_in
    p, keep = fill_attn_params(q, stand_in, stand_in, out, cu32, int(max_seqlen), lens, int(max_seqlen), float(sm_scale), (-1, -1),
                               bt, 0.0, None, None, None, None, k_new=k, v_new=v, non_causal=not causal, new_kv_all_rows=True)
    p.block_table_stride = 0
    launch(p, dev)
    del keep


def prefill_flash_attention(
    q,
    k,
    v,
    max_seqlen_q,
    max_seqlen_k,
    cu_seqlens_q,
    cu_seqlens_k,
    causal=False,
    sm_scale=1.0,
    bias=None,
    config=None,
    in_place_output=None,
    do_not_return_softmax_encodings=True,
):
    if not q.is_cuda:
        raise RuntimeError("mi355_attn.prefill_flash_attention needs tensors on an MI355X (cuda/hip) device; there is no CPU path")
    if bias is not None:
        # The reference's wrapper sets the variable-length parameters unconditionally (triton_flash_attention.py:1341-1342) and its
        # argument check then asserts `bias is None` for that layout (:126-128, "TODO: Remove once bias is supported with
        # varlen"): a bias reaches neither kernel there. Same behaviour here: an AssertionError before any launch.
        raise AssertionError("prefill_flash_attention: bias is not supported with the variable-length layout (as in the reference, "
                             "triton_flash_attention.py:126-128)")
    if not do_not_return_softmax_encodings:
        raise NotImplementedError("prefill_flash_attention: softmax encodings are not produced")
    if cu_seqlens_q is None or cu_seqlens_k is None:
        raise NotImplementedError("prefill_flash_attention: only the variable-length (thd) layout is served")
    if q.dim() != 3 or k.dim() != 3 or v.dim() != 3 or k.shape != v.shape:
        raise ValueError("q must be [total_q, Hq, D] and k, v [total_k, Hk, D]")
    total_k, hk, d = k.shape
    dev = q.device
    out = in_place_output if in_place_output is not None else torch.empty_like(q)
    if _same_ranges(cu_seqlens_q, cu_seqlens_k) and q.dtype in (torch.float16, torch.bfloat16) and k.dtype == q.dtype and d % 8 == 0:
        _self_attention(q, k, v, out, cu_seqlens_q, max(int(max_seqlen_q), int(max_seqlen_k)), sm_scale, causal)
        return out
    cu_k = cu_seqlens_k.to(torch.int64)
    num_seqs = cu_k.numel() - 1
    lens = cu_k[1:] - cu_k[:-1]
    pages_per_seq = (lens + (_PAGE - 1)) // _PAGE
    page_base = torch.cumsum(pages_per_seq, 0) - pages_per_seq
    tok_seq = torch.repeat_interleave(torch.arange(num_seqs, device=dev), lens, output_size=total_k)
    tok_pos = torch.arange(total_k, device=dev) - cu_k[tok_seq]
    slot_mapping = page_base[tok_seq] * _PAGE + tok_pos
    num_pages = total_k // _PAGE + num_seqs                      # host-known upper bound of sum(ceil(len / 16))
    k_cache, v_cache = _scratch_cache(dev, num_pages, hk, d, k.dtype)
    reshape_and_cache_flash(k, v, k_cache, v_cache, slot_mapping, "auto", None, None)
    max_pages = (int(max_seqlen_k) + _PAGE - 1) // _PAGE
    block_table = (page_base[:, None] + torch.arange(max_pages, device=dev)[None, :]).clamp_(max=num_pages - 1).to(torch.int32)
    if causal:
        unified_attention(
            q=q, k=k_cache, v=v_cache, out=out, cu_seqlens_q=cu_seqlens_q.to(torch.int32), max_seqlen_q=int(max_seqlen_q),
            seqused_k=lens.to(torch.int32), max_seqlen_k=int(max_seqlen_k), avg_seqlen_q=0.0, avg_seqlen_k=0.0,
            softmax_scale=float(sm_scale), causal=True, window_size=(-1, -1), block_table=block_table, softcap=0.0,
            q_descale=None, k_descale=None, v_descale=None,
        )
    else:
        # `unified_attention` keeps the reference op's "only causal" assert; the parameter block has the switch
        p, keep = fill_attn_params(q, k_cache, v_cache, out, cu_seqlens_q.to(torch.int32), int(max_seqlen_q), lens.to(torch.int32),
                                   int(max_seqlen_k), float(sm_scale), (-1, -1), block_table, 0.0, None, None, None, None, non_causal=True)
        launch(p, q.device)
        del keep
    return out
