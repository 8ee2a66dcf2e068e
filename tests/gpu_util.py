"""Helpers shared by the -m gpu parity tests: run the HIP path (through the C ABI) on fixtures."""

import torch

from mi355_attn import _lib
from mi355_attn.kernels import unified_attention

DEV = torch.device("cuda:0")


def to_dev(d):
    return {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in d.items()}


def oracle_row(orc, q_row, k, v, bt_row, n_keys, scale, **kw):
    """The oracle for ONE query token over the first n_keys keys of one sequence of a BIG cache. The sequence's pages are
    gathered into a small contiguous cache first (identity block table): the oracle's per-head gathers out of a
    multi-gigabyte host tensor cost tens of seconds per row on the GPU box's host (0.1 s this way)."""
    pages = bt_row.reshape(-1)[: (n_keys + k.shape[1] - 1) // k.shape[1]].long()
    kc, vc = k[pages].contiguous(), v[pages].contiguous()
    if kc.dtype in (torch.float8_e4m3fn, torch.float8_e5m2):
        # the oracle's dequantisation, (fp8 -> f32) * scale -> query dtype, once per cache instead of once per query head
        kc = orc._dequant(kc, kw.pop("k_scale", 1.0), q_row.dtype).to(q_row.dtype)
        vc = orc._dequant(vc, kw.pop("v_scale", 1.0), q_row.dtype).to(q_row.dtype)
    ident = torch.arange(pages.numel(), dtype=torch.int32).view(1, -1)
    return orc.unified_attention_oracle(q_row, kc, vc, torch.tensor([0, 1], dtype=torch.int32), torch.tensor([n_keys], dtype=torch.int32), ident, scale, **kw)


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
