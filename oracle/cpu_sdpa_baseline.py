"""CPU baseline: the reference-style PyTorch SDPA path over the paged KV cache, timed on host cores.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (used by bench.py's `cpu_baseline` leg and by tests; never by
the product path). It is our restatement ("port") of what the reference harness does on CPU:
per sequence, gather the sequence's pages into contiguous [kv, Hk, D] tensors (ref_paged_attn,
scripts/vllm_utils.py:456-462) and call torch.nn.functional.scaled_dot_product_attention with GQA
broadcast and a bottom-right aligned causal mask (PytorchNativeAttentionPrefillCaller,
scripts/callers/pytorch_native.py:24-56,:110-143).
"""

from __future__ import annotations

import time

import torch
import torch.nn.functional as F


def paged_sdpa_cpu(q, k_cache, v_cache, cu_seqlens_q, seqused_k, block_table, scale):
    """Returns (out [T,Hq,D], gather_seconds, sdpa_seconds)."""
    T, Hq, D = q.shape
    page = k_cache.shape[1]
    out = torch.empty_like(q)
    t_gather = t_sdpa = 0.0
    cu = cu_seqlens_q.tolist()
    for i in range(len(seqused_k)):
        q0, q1 = cu[i], cu[i + 1]
        q_len, kv_len = q1 - q0, int(seqused_k[i])
        if q_len == 0:
            continue
        t0 = time.perf_counter()
        pages = block_table[i, : (kv_len + page - 1) // page].long()
        k = k_cache[pages].reshape(-1, k_cache.shape[2], D)[:kv_len].transpose(0, 1).unsqueeze(0)  # [1,Hk,kv,D]
        v = v_cache[pages].reshape(-1, v_cache.shape[2], D)[:kv_len].transpose(0, 1).unsqueeze(0)
        qi = q[q0:q1].transpose(0, 1).unsqueeze(0)                                                    # [1,Hq,q,D]
        t1 = time.perf_counter()
        if q_len == kv_len:
            o = F.scaled_dot_product_attention(qi, k, v, is_causal=True, scale=scale, enable_gqa=True)
        elif q_len == 1:
            o = F.scaled_dot_product_attention(qi, k, v, is_causal=False, scale=scale, enable_gqa=True)
        else:
            mask = torch.ones(q_len, kv_len, dtype=torch.bool).tril(diagonal=kv_len - q_len)
            o = F.scaled_dot_product_attention(qi, k, v, attn_mask=mask, scale=scale, enable_gqa=True)
        t2 = time.perf_counter()
        out[q0:q1] = o[0].transpose(0, 1)
        t_gather += t1 - t0
        t_sdpa += t2 - t1
    return out, t_gather, t_sdpa


def time_paged_sdpa_cpu(inputs, scale, warmup=1, reps=3, budget_s=25.0):
    """Median wall time of paged_sdpa_cpu over `reps` (stops early when the budget is spent)."""
    times, gathers = [], []
    out = None
    t_start = time.perf_counter()
    for it in range(warmup + reps):
        t0 = time.perf_counter()
        out, tg, ts = paged_sdpa_cpu(inputs["q"], inputs["k_cache"], inputs["v_cache"], inputs["cu_seqlens_q"], inputs["seqused_k"],
                                     inputs["block_table"], scale)
        dt = time.perf_counter() - t0
        if it >= warmup:
            times.append(dt)
            gathers.append(tg)
        if time.perf_counter() - t_start > budget_s and times:
            break
    times.sort()
    return out, times[len(times) // 2], sorted(gathers)[len(gathers) // 2], len(times)
