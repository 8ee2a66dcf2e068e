"""Helpers shared by the -m gpu parity tests: run the HIP path (through the C ABI) on fixtures."""

import torch

from mi355_attn import _lib
from mi355_attn.kernels import unified_attention

DEV = torch.device("cuda:0")


def to_dev(d):
    return {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in d.items()}


def run_unified(t, scale, *, window=0, softcap=0.0, kv_scale=None, v_scale=None, force=None, out=None, lse=None):
    """t: dict with q, k_cache, v_cache, cu_seqlens_q, seqused_k, block_table[, alibi_slopes] on DEV."""
    q = t["q"]
    if out is None:
        out = torch.full_like(q, float("nan"))
    ql = t["cu_seqlens_q"][1:] - t["cu_seqlens_q"][:-1]
    ks = None if kv_scale is None else torch.tensor([kv_scale], dtype=torch.float32, device=q.device)
    vs = ks if v_scale is None else torch.tensor([v_scale], dtype=torch.float32, device=q.device)
    unified_attention(
        q=q, k=t["k_cache"], v=t["v_cache"], out=out, cu_seqlens_q=t["cu_seqlens_q"], max_seqlen_q=int(ql.max()),
        seqused_k=t["seqused_k"], max_seqlen_k=int(t["seqused_k"].max()), avg_seqlen_q=float(ql.float().mean()),
        avg_seqlen_k=float(t["seqused_k"].float().mean()), softmax_scale=scale, causal=True,
        window_size=(window - 1, 0) if window else (-1, -1), block_table=t["block_table"], softcap=softcap,
        q_descale=None, k_descale=ks, v_descale=vs, alibi_slopes=t.get("alibi_slopes"), force_selection=force, softmax_lse=lse,
    )
    torch.cuda.synchronize()
    return out, _lib.last_kernel()
