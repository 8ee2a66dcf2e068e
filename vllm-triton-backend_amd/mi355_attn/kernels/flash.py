"""`prefill_flash_attention` — the reference package's non-paged variable-length prefill op
(`triton_wrapper_forward_prefill`, LIB/kernels/triton_flash_attention.py:1326-1484, exported at
LIB/kernels/__init__.py:65-67), served by the paged MFMA kernels: K/V are laid out as 16-token pages in a scratch cache
(one pass of `reshape_and_cache_flash`, pages of a sequence contiguous) and `unified_attention` runs over them.
Everything is device-side torch arithmetic on `cu_seqlens_*`: no host synchronisation; the scratch cache is kept between calls.

Served: the "thd" variable-length layout (`q [total_q, Hq, D]`, `k, v [total_k, Hk, D]`), causal masking with the
reference's bottom-right alignment (query t of a sequence sees keys j <= t + seqlen_k - seqlen_q, :954-960) on the MFMA
kernels, non-causal attention (every query row sees its sequence's whole key range) on the shape-agnostic kernel - a
correctness path, not a fast one -, grouped-query heads. Not served (raise `NotImplementedError`, nothing silently
ignored): `bias`, softmax encodings, dropout.
"""

from __future__ import annotations

import torch

from .cache import reshape_and_cache_flash
from .unified import fill_attn_params, launch, unified_attention

_PAGE = 16
_scratch: dict = {}


def _scratch_cache(dev, num_pages, hk, d, dtype):
    """The paged scratch K/V of a call, kept per (device, stream, heads, head size, dtype) and grown geometrically: a
    serving loop calls this op with the same shapes over and over, and a per-call torch.empty pair would go through the
    caching allocator every time (and move under a captured graph). Stream-ordered reuse: the next call's cache write is
    enqueued behind this call's attention on the same stream."""
    key = (dev.type, dev.index, torch.cuda.current_stream(dev).cuda_stream, hk, d, dtype)
    buf = _scratch.get(key)
    if buf is None or buf.shape[1] < num_pages:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("prefill_flash_attention: run one eager call of the largest shape before graph capture")
        grow = max(num_pages, 2 * (buf.shape[1] if buf is not None else 0))
        buf = torch.empty((2, grow, _PAGE, hk, d), dtype=dtype, device=dev)
        _scratch[key] = buf
    return buf[0, :num_pages], buf[1, :num_pages]


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
        raise NotImplementedError("prefill_flash_attention: bias is not supported")
    if not do_not_return_softmax_encodings:
        raise NotImplementedError("prefill_flash_attention: softmax encodings are not produced")
    if cu_seqlens_q is None or cu_seqlens_k is None:
        raise NotImplementedError("prefill_flash_attention: only the variable-length (thd) layout is served")
    if q.dim() != 3 or k.dim() != 3 or v.dim() != 3 or k.shape != v.shape:
        raise ValueError("q must be [total_q, Hq, D] and k, v [total_k, Hk, D]")
    total_k, hk, d = k.shape
    dev = q.device
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
    out = in_place_output if in_place_output is not None else torch.empty_like(q)
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
