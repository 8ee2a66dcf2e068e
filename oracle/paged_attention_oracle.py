"""CPU oracle for the paged-attention hot path.

TEST INFRASTRUCTURE ONLY. Nothing under `vllm-triton-backend_amd/` may import this module; only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` use it, and there only
as the checker. It is a restatement (torch on CPU, fp32 arithmetic, plain loops over sequences and
KV heads) of the reference's Triton kernels, written from their semantics:

  unified_attention_oracle(mode="2d")  <- kernel_unified_attention_2d
                                          LIB/kernels/triton_unified_attention.py:275-523
  unified_attention_oracle(mode="3d")  <- kernel_unified_attention_3d :526-754 + reduce_segments :757-836
  reshape_and_cache_flash_oracle       <- scripts/vllm_utils.py:377-401 (+ slot < 0 skip, LIB/backend/triton_attn.py:149-151)
  context_attention_fwd_oracle         <- _fwd_kernel LIB/kernels/legacy/triton_prefix_prefill.py:26-301
  paged_attention_v0_oracle            <- kernel_paged_attention_2d LIB/kernels/legacy/triton_paged_decode_attention_2d.py:99-280
  dense_attention_fp64                 <- independent textbook softmax (no tiling), used to cross-check the above

Pinned by: tests/golden/*.npz, produced by tests/golden/make_golden.py from the reference's own
Triton kernels run under TRITON_INTERPRET=1 in the build container (tests/test_oracle_golden.py).
(LIB/ = ibm-triton-lib/ibm_triton_lib/ in the reference repository.)
"""

from __future__ import annotations

import math
from typing import Optional

import torch

NEG_INF = float("-inf")


def _dequant(x: torch.Tensor, scale: float, q_dtype: torch.dtype) -> torch.Tensor:
    """fp8 K/V: (fp8 -> f32) * scale -> Q dtype (triton_unified_attention.py:434-455); other
    dtypes pass through. Result is returned as fp32 values representable in q_dtype."""
    if x.dtype in (torch.float8_e4m3fn, torch.float8_e5m2):
        return (x.to(torch.float32) * scale).to(q_dtype).to(torch.float32)
    return x.to(torch.float32)


def _softcap(s: torch.Tensor, cap: float) -> torch.Tensor:
    # apply_softcap (:24-29): x * (e^{s/x} - e^{-s/x}) / (e^{s/x} + e^{-s/x}) == x * tanh(s/x)
    return cap * torch.tanh(s / cap)


def _gather_kv(cache: torch.Tensor, pages: torch.Tensor, n: int, head: int) -> torch.Tensor:
    """flash layout [num_pages, page, Hk, D] -> [n, D] rows of one KV head."""
    return cache[pages.long(), :, head, :].reshape(-1, cache.shape[-1])[:n]


def _tile_update(S, V_tile, M, L, acc, p_dtype):
    """One online-softmax step (:484-508). S already masked. Returns updated (M, L, acc)."""
    m_j = torch.maximum(M, S.max(dim=1).values)
    m_j = torch.where(m_j > NEG_INF, m_j, torch.zeros_like(m_j))          # :489
    P = torch.exp(S - m_j[:, None])
    l_j = P.sum(dim=1)
    alpha = torch.exp(M - m_j)
    acc = acc * alpha[:, None]
    L = L * alpha + l_j
    acc = acc + P.to(p_dtype).to(torch.float32) @ V_tile                 # P cast to V dtype (:508)
    return m_j, L, acc


def unified_attention_oracle(
    q: torch.Tensor,               # [T, Hq, D]
    k_cache: torch.Tensor,         # [num_pages, page, Hk, D]
    v_cache: torch.Tensor,
    cu_seqlens_q: torch.Tensor,    # [S+1]
    seqused_k: torch.Tensor,       # [S]
    block_table: torch.Tensor,     # [S, max_pages]
    scale: float,
    sliding_window: int = 0,       # kernel constant SLIDING_WINDOW = 1 + window_size[0] (:915); 0 = off
    softcap: float = 0.0,
    alibi_slopes: Optional[torch.Tensor] = None,
    k_scale: float = 1.0,
    v_scale: float = 1.0,
    mode: str = "2d",
    block_n: int = 16,             # KV tile of the 2D kernel (autotuned BLOCK_N in the reference)
    num_segments: int = 16,        # NUM_SEGMENTS of the 3D path (:948)
) -> torch.Tensor:
    T, Hq, D = q.shape
    page, Hk = k_cache.shape[1], k_cache.shape[2]
    G = Hq // Hk
    out = torch.zeros(T, Hq, D, dtype=torch.float32)
    qf = q.to(torch.float32)
    p_dtype = q.dtype if k_cache.dtype in (torch.float8_e4m3fn, torch.float8_e5m2) else v_cache.dtype
    cu = cu_seqlens_q.tolist()
    for i in range(len(seqused_k)):
        q0, q1 = cu[i], cu[i + 1]
        q_len, seq_len = q1 - q0, int(seqused_k[i])
        if q_len <= 0:
            continue
        ctx = seq_len - q_len                                             # :374
        n_pages = (seq_len + page - 1) // page
        pages = block_table[i, :n_pages]
        for h in range(Hk):
            K = _dequant(_gather_kv(k_cache, pages, seq_len, h), k_scale, q.dtype)   # [seq_len, D]
            V = _dequant(_gather_kv(v_cache, pages, seq_len, h), v_scale, q.dtype)
            # rows ordered (token, head-in-group) like offs_m // G, offs_m % G (:343-346)
            Q = qf[q0:q1, h * G:(h + 1) * G, :].reshape(q_len * G, D)
            q_pos = torch.arange(q_len).repeat_interleave(G)              # query_pos per row
            key_pos = torch.arange(seq_len)
            slopes = None
            if alibi_slopes is not None:
                slopes = alibi_slopes[h * G:(h + 1) * G].to(torch.float32).repeat(q_len)

            def masked_scores(k0, k1):
                S = scale * (Q @ K[k0:k1].T)                              # :465
                if softcap > 0:
                    S = _softcap(S, softcap)                              # :467-468
                kp = key_pos[k0:k1]
                S = torch.where(kp[None, :] < ctx + q_pos[:, None] + 1, S, NEG_INF)           # :460,:470-472
                if sliding_window > 0:
                    S = torch.where((ctx + q_pos[:, None] - kp[None, :]) < sliding_window, S, NEG_INF)  # :474-479
                if slopes is not None:
                    S = S + slopes[:, None] * (kp[None, :] - ctx).to(torch.float32)          # :481-482
                return S

            rows = q_len * G
            if mode == "2d":
                M = torch.full((rows,), NEG_INF)
                L = torch.ones(rows)                                      # :367
                acc = torch.zeros(rows, D)
                for k0 in range(0, seq_len, block_n):                     # the kernel stops at max_seq_prefix_len (:384-400);
                    k1 = min(k0 + block_n, seq_len)                       # tiles beyond it are fully masked, so this is equivalent
                    M, L, acc = _tile_update(masked_scores(k0, k1), V[k0:k1], M, L, acc, p_dtype)
                o = acc / L[:, None]                                      # :511
            elif mode == "3d":
                bps = (seq_len + num_segments * page - 1) // (num_segments * page)            # :592
                seg_acc, seg_M, seg_L = [], [], []
                for s in range(num_segments):
                    if s * bps * page >= seq_len:                         # :594-595
                        break
                    M = torch.full((rows,), NEG_INF)
                    L = torch.ones(rows)
                    acc = torch.zeros(rows, D)
                    for j in range(s * bps, min((s + 1) * bps, n_pages)):  # :640-643, tile = one page
                        k0, k1 = j * page, min((j + 1) * page, seq_len)
                        M, L, acc = _tile_update(masked_scores(k0, k1), V[k0:k1], M, L, acc, p_dtype)
                    seg_acc.append(acc); seg_M.append(M); seg_L.append(L)
                SM = torch.stack(seg_M)                                   # reduce_segments (:804-828)
                overall_max = SM.max(dim=0).values
                w = torch.exp(SM - overall_max[None, :])
                overall_sum = (torch.stack(seg_L) * w).sum(dim=0)
                acc_sum = (torch.stack(seg_acc) * w[:, :, None]).sum(dim=0)
                o = torch.where(overall_sum[:, None] == 0, torch.zeros_like(acc_sum), acc_sum / overall_sum[:, None])
            else:
                raise ValueError(mode)
            out[q0:q1, h * G:(h + 1) * G, :] = o.reshape(q_len, G, D)
    return out.to(q.dtype)


def dense_attention_fp64(
    q, k_cache, v_cache, cu_seqlens_q, seqused_k, block_table, scale, sliding_window=0, softcap=0.0,
    alibi_slopes=None, k_scale=1.0, v_scale=1.0, return_lse=False,
):
    """Independent check: gather, one dense softmax per (sequence, head) in float64. With `return_lse` also the
    log-sum-exp of every row's masked scores [T, Hq] (-inf for a row that sees no key): the library's optional second
    output, which the reference does not have."""
    T, Hq, D = q.shape
    page, Hk = k_cache.shape[1], k_cache.shape[2]
    G = Hq // Hk
    out = torch.zeros(T, Hq, D, dtype=torch.float64)
    lse = torch.full((T, Hq), NEG_INF, dtype=torch.float64)
    cu = cu_seqlens_q.tolist()
    for i in range(len(seqused_k)):
        q0, q1 = cu[i], cu[i + 1]
        q_len, seq_len = q1 - q0, int(seqused_k[i])
        if q_len <= 0:
            continue
        ctx = seq_len - q_len
        pages = block_table[i, : (seq_len + page - 1) // page]
        qp = torch.arange(q_len)[:, None] + ctx
        kp = torch.arange(seq_len)[None, :]
        mask = kp <= qp
        if sliding_window > 0:
            mask &= (qp - kp) < sliding_window
        for hq in range(Hq):
            h = hq // G
            K = _dequant(_gather_kv(k_cache, pages, seq_len, h), k_scale, q.dtype).double()
            V = _dequant(_gather_kv(v_cache, pages, seq_len, h), v_scale, q.dtype).double()
            S = scale * (q[q0:q1, hq].double() @ K.T)
            if softcap > 0:
                S = softcap * torch.tanh(S / softcap)
            if alibi_slopes is not None:
                S = S + float(alibi_slopes[hq]) * (kp - ctx).double()
            S = S.masked_fill(~mask, NEG_INF)
            P = torch.softmax(S, dim=-1)
            P = torch.nan_to_num(P, nan=0.0)
            out[q0:q1, hq] = P @ V
            if seq_len > 0:
                lse[q0:q1, hq] = torch.logsumexp(S, dim=-1)
    return (out, lse) if return_lse else out


def prefill_flash_attention_oracle(q, k, v, cu_seqlens_q, cu_seqlens_k, sm_scale, block_n: int = 16, causal: bool = True):
    """The reference's non-paged variable-length prefill op, causal or not (attn_fwd via
    triton_wrapper_forward_prefill, LIB/kernels/triton_flash_attention.py:1326-1484): per sequence and query head, keys
    in blocks of BLOCK_N with an online softmax in f32, the causal mask aligned bottom-right (query t of a sequence
    sees keys j <= t + seqlen_k - seqlen_q, :954-960), P rounded to V's type before P.V, grouped-query heads
    (k head = q head // (HQ // HK), :1007). q [total_q, Hq, D]; k, v [total_k, Hk, D]. Returns f32 [total_q, Hq, D]."""
    T, Hq, D = q.shape
    G = Hq // k.shape[1]
    out = torch.zeros(T, Hq, D, dtype=torch.float32)
    cq, ck = [int(x) for x in cu_seqlens_q], [int(x) for x in cu_seqlens_k]
    for i in range(len(cq) - 1):
        q0, q1, k0, k1 = cq[i], cq[i + 1], ck[i], ck[i + 1]
        lq, lk = q1 - q0, k1 - k0
        if lq <= 0:
            continue
        last = torch.arange(lq) + (lk - lq) if causal else torch.full((lq,), lk - 1)   # last visible key of each query row
        for hq in range(Hq):
            h = hq // G
            Q = q[q0:q1, hq].to(torch.float32)
            M = torch.full((lq,), float("-inf"))
            L = torch.zeros(lq)
            acc = torch.zeros(lq, D)
            for j0 in range(0, lk, block_n):
                j1 = min(j0 + block_n, lk)
                S = (Q @ k[k0 + j0:k0 + j1, h].to(torch.float32).T) * sm_scale
                S = S.masked_fill(torch.arange(j0, j1)[None, :] > last[:, None], float("-inf"))
                M, L, acc = _tile_update(S, v[k0 + j0:k0 + j1, h].to(torch.float32), M, L, acc, v.dtype)
            out[q0:q1, hq] = torch.where(L[:, None] > 0, acc / L[:, None], torch.zeros_like(acc))
    return out


def reshape_and_cache_flash_oracle(key, value, key_cache, value_cache, slot_mapping, k_scale=1.0, v_scale=1.0):
    """In-place scatter (scripts/vllm_utils.py:377-401); slot < 0 = padding (triton_attn.py:149-151);
    fp8 caches store saturating fp8(x / scale)."""
    page = key_cache.shape[1]
    for t, slot in enumerate(slot_mapping.tolist()):
        if slot < 0 or t >= key.shape[0]:
            continue
        b, o = slot // page, slot % page
        for src, dst, sc in ((key, key_cache, k_scale), (value, value_cache, v_scale)):
            if dst.dtype in (torch.float8_e4m3fn, torch.float8_e5m2):
                lim = torch.finfo(dst.dtype).max
                dst[b, o] = (src[t].to(torch.float32) / sc).clamp(-lim, lim).to(dst.dtype)
            else:
                dst[b, o] = src[t].to(dst.dtype)


def v0_to_flash(k_cache_v0: torch.Tensor, v_cache_v0: torch.Tensor):
    """legacy K [nb, Hk, D/x, page, x] (or [nb, Hk, D, page]), V [nb, Hk, D, page]
    (triton_paged_decode_attention_2d.py:103-104,:198-211) -> flash [nb, page, Hk, D]."""
    if k_cache_v0.dim() == 5:
        nb, Hk, Dx, page, x = k_cache_v0.shape
        k = k_cache_v0.permute(0, 3, 1, 2, 4).reshape(nb, page, Hk, Dx * x)
    else:
        k = k_cache_v0.permute(0, 3, 1, 2)
    v = v_cache_v0.permute(0, 3, 1, 2)
    return k.contiguous(), v.contiguous()


def paged_attention_v0_oracle(query, key_cache, value_cache, scale, block_tables, seq_lens, alibi_slopes=None,
                              k_scale=1.0, v_scale=1.0, num_segments=0):
    """Decode over the legacy cache layout: one query token per sequence, all seq_len keys visible
    (kernel_paged_attention_2d, triton_paged_decode_attention_2d.py:179-269; num_segments=4 gives the
    split-KV variant, triton_paged_decode_attention_3d.py:366)."""
    k, v = v0_to_flash(key_cache, value_cache)
    S = query.shape[0]
    cu = torch.arange(S + 1, dtype=torch.int32)
    return unified_attention_oracle(
        query, k, v, cu, seq_lens, block_tables, scale, alibi_slopes=alibi_slopes, k_scale=k_scale, v_scale=v_scale,
        mode="3d" if num_segments else "2d", block_n=k.shape[1], num_segments=num_segments or 16,
    )


def context_attention_fwd_oracle(q, k, v, k_cache, v_cache, b_loc, b_start_loc, b_seq_len, sm_scale=None,
                                 alibi_slopes=None, sliding_window=0, k_scale=1.0, v_scale=1.0, block=16):
    """Chunked prefill over the legacy layout (_fwd_kernel, triton_prefix_prefill.py:26-301): context
    keys from the paged cache (no causal mask, :122-217), new keys from linear k/v with a causal mask
    (:236-288); normalises at every step (:184-200); sliding window masks with -10000 (:177-182);
    rows of sequences with query_len == 1 are NOT written (:83-84) and come back as zeros here."""
    T, Hq, D = q.shape
    Hk = k.shape[1]
    G = Hq // Hk
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(D)                                     # :652-653
    kf, vf = v0_to_flash(k_cache, v_cache)
    page = kf.shape[1]
    out = torch.zeros(T, Hq, D, dtype=torch.float32)
    starts = b_start_loc.tolist()
    for i in range(len(b_seq_len)):
        q0, q1 = starts[i], starts[i + 1]
        q_len, seq_len = q1 - q0, int(b_seq_len[i])
        if q_len == 1 or q_len <= 0:
            continue
        ctx = seq_len - q_len
        pages = b_loc[i, : (ctx + page - 1) // page]
        for hq in range(Hq):
            h = hq // G
            Kc = _dequant(_gather_kv(kf, pages, ctx, h), k_scale, q.dtype) if ctx > 0 else torch.zeros(0, D)
            Vc = _dequant(_gather_kv(vf, pages, ctx, h), v_scale, q.dtype) if ctx > 0 else torch.zeros(0, D)
            Kall = torch.cat([Kc, k[q0:q1, h].to(torch.float32)])
            Vall = torch.cat([Vc, v[q0:q1, h].to(torch.float32)])
            Q = q[q0:q1, hq].to(torch.float32)
            qpos = torch.arange(q_len)[:, None] + ctx
            m_i = torch.full((q_len,), NEG_INF)
            l_i = torch.zeros(q_len)
            acc = torch.zeros(q_len, D)
            # the kernel runs the context loop and the new-token loop with separate tiles; a tile never
            # straddles the context boundary
            bounds = list(range(0, ctx, block)) + [ctx + t for t in range(0, q_len, block)]
            ends = [min(b + block, ctx) for b in range(0, ctx, block)] + [min(ctx + t + block, seq_len) for t in range(0, q_len, block)]
            for k0, k1 in zip(bounds, ends):
                kp = torch.arange(k0, k1)[None, :]
                qk = (Q @ Kall[k0:k1].T) * sm_scale
                qk = torch.where(kp <= qpos, qk, NEG_INF)
                if alibi_slopes is not None:
                    qk = qk + float(alibi_slopes[hq]) * (kp - qpos).to(torch.float32)
                if sliding_window > 0:
                    qk = torch.where((qpos - kp) < sliding_window, qk, torch.full_like(qk, -10000.0))
                m_ij = qk.max(dim=1).values
                valid = m_ij > NEG_INF
                m_safe = torch.where(valid, m_ij, torch.zeros_like(m_ij))
                p = torch.exp(qk - m_safe[:, None])
                l_ij = p.sum(dim=1)
                m_new = torch.maximum(m_i, m_ij)
                m_new_safe = torch.where(m_new > NEG_INF, m_new, torch.zeros_like(m_new))
                alpha = torch.exp(m_i - m_new_safe)
                beta = torch.exp(m_safe - m_new_safe) * valid
                l_new = alpha * l_i + beta * l_ij
                denom = torch.where(l_new > 0, l_new, torch.ones_like(l_new))
                p = p * (beta / denom)[:, None]
                acc = acc * (l_i / denom * alpha)[:, None]
                acc = acc + p.to(q.dtype).to(torch.float32) @ Vall[k0:k1]
                l_i, m_i = l_new, m_new
            out[q0:q1, hq] = acc
    return out.to(q.dtype)


def make_paged_inputs(seed, query_lens, kv_lens, num_q_heads, num_kv_heads, head_size, page_size, dtype,
                      kv_dtype=None, num_pages=None, max_value=1.0, kv_scale=1.0):
    """Seeded synthetic inputs in the reference harness's style (U(-max,max) values,
    scripts/benchmark.py:136,:1168-1174) but with a random PERMUTATION of pages (the harness's
    unseeded random.randint with replacement, :1199-1203, aliases pages and is not reproduced)."""
    g = torch.Generator().manual_seed(seed)
    S = len(query_lens)
    T = sum(query_lens)
    pages_per_seq = [(n + page_size - 1) // page_size for n in kv_lens]
    need = sum(pages_per_seq)
    if num_pages is None:
        num_pages = max(need + 3, int(need * 1.25))
    q = (torch.rand(T, num_q_heads, head_size, generator=g) * 2 - 1).mul(max_value).to(dtype)
    kf = (torch.rand(num_pages, page_size, num_kv_heads, head_size, generator=g) * 2 - 1).mul(max_value)
    vf = (torch.rand(num_pages, page_size, num_kv_heads, head_size, generator=g) * 2 - 1).mul(max_value)
    if kv_dtype is None or kv_dtype == dtype:
        k_cache, v_cache = kf.to(dtype), vf.to(dtype)
    else:
        k_cache, v_cache = (kf / kv_scale).to(kv_dtype), (vf / kv_scale).to(kv_dtype)
    perm = torch.randperm(num_pages, generator=g).to(torch.int32)
    max_pages = max(pages_per_seq) if pages_per_seq else 1
    block_table = torch.zeros(S, max_pages, dtype=torch.int32)
    o = 0
    for i, n in enumerate(pages_per_seq):
        block_table[i, :n] = perm[o:o + n]
        o += n
    cu = torch.zeros(S + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(torch.tensor(query_lens, dtype=torch.int32), 0)
    seqused = torch.tensor(kv_lens, dtype=torch.int32)
    return dict(q=q, k_cache=k_cache, v_cache=v_cache, cu_seqlens_q=cu, seqused_k=seqused, block_table=block_table,
                scale=1.0 / math.sqrt(head_size))
