"""Multi-GPU partitioning of the attention hot path (one process per GPU).

The path shards over independent units — (sequence, KV head) — with NO data-path collective
(the reference's launch grid has exactly these axes and no cross-program traffic,
LIB/kernels/triton_unified_attention.py:321-322,:567-569). Two partitionings:

  * batch sharding (default): rank r owns a cost-balanced subset of the sequences, only those
    sequences' KV pages (re-indexed into a compact local cache) and the matching slices of q/out;
  * KV-head sharding (= what vLLM tensor parallelism hands the backend): rank r owns KV heads
    [r*Hk/n, (r+1)*Hk/n) and their query heads; block table and lengths are replicated.

The only collective is optional: gathering the per-rank outputs (torch.distributed all_gather over
RCCL/xGMI with backend "nccl"; "gloo" in CPU tests) for verification or for a caller that wants the
full tensor everywhere.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def attention_cost(query_len: int, kv_len: int) -> float:
    """Relative cost of one sequence: q*ctx + q(q+1)/2 score entries."""
    ctx = kv_len - query_len
    return float(query_len * ctx + query_len * (query_len + 1) / 2)


def assign_sequences(query_lens: Sequence[int], kv_lens: Sequence[int], world_size: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of sequences to ranks. Deterministic (ties by index),
    identical on every rank, preserves the original order inside each rank."""
    order = sorted(range(len(query_lens)), key=lambda i: (-attention_cost(query_lens[i], kv_lens[i]), i))
    load = [0.0] * world_size
    owned: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda x: (load[x], x))
        owned[r].append(i)
        load[r] += attention_cost(query_lens[i], kv_lens[i])
    return [sorted(o) for o in owned]


@dataclass
class LocalBatch:
    """What one rank needs to run its share with `unified_attention`."""

    seq_ids: List[int]                # global sequence indices owned by this rank
    token_index: torch.Tensor         # [T_local] global token index of every local query token
    q: torch.Tensor                   # [T_local, Hq, D]
    cu_seqlens_q: torch.Tensor        # [S_local + 1] int32
    seqused_k: torch.Tensor           # [S_local] int32
    block_table: torch.Tensor         # [S_local, max_pages] int32, indices into the LOCAL caches
    k_cache: torch.Tensor             # [local_pages, page, Hk, D]
    v_cache: torch.Tensor
    max_seqlen_q: int
    max_seqlen_k: int
    # global token indices of EVERY rank's share, in rank order (host tensors): the assignment is deterministic and
    # identical on every rank, so assembling the output needs no exchange of maps or counts
    rank_token_index: Optional[List[torch.Tensor]] = None


def shard_batch(rank: int, world_size: int, q, k_cache, v_cache, cu_seqlens_q, seqused_k, block_table) -> LocalBatch:
    """Batch sharding: slice out rank `rank`'s sequences and compact their KV pages. The plan - which tokens, which pages,
    the local block table - is tensor arithmetic over the batch (no per-page or per-token Python loop: a 1000-sequence
    batch with 100k pages plans in milliseconds); only the LPT assignment walks the sequences, on host lists."""
    cu_h = cu_seqlens_q.to("cpu", torch.int64)
    kv_h = seqused_k.to("cpu", torch.int64)
    S = int(kv_h.numel())
    qlens_h = cu_h[1:] - cu_h[:-1]
    owned = assign_sequences(qlens_h.tolist(), kv_h.tolist(), world_size)
    page = k_cache.shape[1]

    def tokens_of(seqs: torch.Tensor) -> torch.Tensor:          # global token indices of the sequences, in their order
        if seqs.numel() == 0:
            return torch.zeros(0, dtype=torch.long)
        n = qlens_h[seqs]
        starts = torch.repeat_interleave(cu_h[seqs], n)
        first = torch.repeat_interleave(torch.cumsum(n, 0) - n, n)
        return starts + (torch.arange(int(n.sum())) - first)

    rank_tok = [tokens_of(torch.tensor(o, dtype=torch.long)) for o in owned]
    mine = owned[rank]
    mine_t = torch.tensor(mine, dtype=torch.long)
    n_pages = (kv_h[mine_t] + page - 1) // page if mine else torch.zeros(0, dtype=torch.long)       # pages per local sequence
    max_pages = max(int(n_pages.max()) if mine else 1, 1)
    base = torch.cumsum(n_pages, 0) - n_pages                                                           # first LOCAL page of each
    col = torch.arange(max_pages)[None, :]
    valid = col < n_pages[:, None]                                                                      # [S_local, max_pages]
    bt = torch.where(valid, base[:, None] + col, torch.zeros((), dtype=torch.long)).to(torch.int32)
    # the global pages behind the local ones, in local order: row-major over the valid entries of the owned rows
    bt_rows = block_table.to("cpu")[mine_t][:, :max_pages] if mine else torch.zeros((0, max_pages), dtype=torch.int32)
    if bt_rows.shape[1] < max_pages:                         # (a block table narrower than this rank's longest sequence cannot happen in a valid batch)
        raise ValueError("block_table has fewer columns than a sequence has pages")
    page_idx = bt_rows[valid].to(torch.long).to(k_cache.device)
    token_index = rank_tok[rank]
    cu_local = torch.zeros(len(mine) + 1, dtype=torch.int32)
    if mine:
        cu_local[1:] = torch.cumsum(qlens_h[mine_t], 0).to(torch.int32)
    dev = q.device
    return LocalBatch(
        seq_ids=mine,
        token_index=token_index,
        q=q[token_index.to(dev)],
        cu_seqlens_q=cu_local.to(dev),
        seqused_k=kv_h[mine_t].to(torch.int32).to(dev) if mine else torch.zeros(0, dtype=torch.int32, device=dev),
        block_table=bt.to(dev),
        k_cache=k_cache[page_idx] if page_idx.numel() else k_cache[:0],
        v_cache=v_cache[page_idx] if page_idx.numel() else v_cache[:0],
        max_seqlen_q=int(qlens_h[mine_t].max()) if mine else 0,
        max_seqlen_k=int(kv_h[mine_t].max()) if mine else 0,
        rank_token_index=rank_tok,
    )


def shard_kv_heads(rank: int, world_size: int, q, k_cache, v_cache):
    """KV-head sharding (tensor-parallel layout): returns this rank's views (q_r, k_cache_r, v_cache_r)
    and the query-head slice it owns."""
    Hq, Hk = q.shape[1], k_cache.shape[2]
    if Hk % world_size != 0:
        raise ValueError(f"num_kv_heads {Hk} is not divisible by world size {world_size}")
    hk = Hk // world_size
    g = Hq // Hk
    ks = slice(rank * hk, (rank + 1) * hk)
    qs = slice(rank * hk * g, (rank + 1) * hk * g)
    return q[:, qs], k_cache[:, :, ks], v_cache[:, :, ks], qs


def gather_outputs(local_out: torch.Tensor, local: LocalBatch, total_tokens: int, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Assemble the full [T, Hq, D] output on every rank from the batch-sharded pieces: ONE all_gather_into_tensor of
    the padded pieces (RCCL over xGMI under backend "nccl"). Counts and token maps are host arithmetic every rank
    already has (`LocalBatch.rank_token_index`): no exchange of metadata, no device-to-host read."""
    world = dist.get_world_size(group)
    maps = local.rank_token_index
    assert maps is not None and len(maps) == world, "LocalBatch was built for another world size"
    counts = [int(m.numel()) for m in maps]
    t_max = max(max(counts), 1)
    dev = local_out.device
    pad_out = torch.zeros((t_max,) + tuple(local_out.shape[1:]), dtype=local_out.dtype, device=dev)
    pad_out[: local_out.shape[0]] = local_out
    gathered = torch.empty((world * t_max,) + tuple(local_out.shape[1:]), dtype=local_out.dtype, device=dev)
    dist.all_gather_into_tensor(gathered, pad_out, group=group)
    src = torch.cat([torch.arange(r * t_max, r * t_max + n) for r, n in enumerate(counts)]).to(dev)
    dst = torch.cat(maps).to(dev)
    full = torch.zeros((total_tokens,) + tuple(local_out.shape[1:]), dtype=local_out.dtype, device=dev)
    full[dst] = gathered[src]
    return full


# ---------------------------------------------------------------------------------------------
# Context parallelism for one very long sequence (SURVEY.md §8e "cross-GPU split-KV", §8f-3): the
# sequence's KV pages are striped over the ranks, every rank attends its own key range and the
# partial results meet in ONE exchange step - the only place on this path where a collective
# carries data. The merge is the arithmetic of `reduce_segments`
# (LIB/kernels/triton_unified_attention.py:804-828) with the log-sum-exp in place of (max, sum).
# ---------------------------------------------------------------------------------------------
def merge_partial_attention(outs: torch.Tensor, lses: torch.Tensor) -> tuple:
    """outs [R, T, H, D] (any float dtype), lses [R, T, H] float32: partial attention results of R disjoint key
    ranges and the log-sum-exp of each range's scores (`softmax_lse` of `unified_attention`). Returns (out [T, H, D]
    float32, lse [T, H]). A range that saw no key has lse = -inf and contributes nothing; a row no range saw gets 0."""
    lses = lses.to(torch.float32)
    lse = torch.logsumexp(lses, dim=0)                                   # [T, H]; -inf where every range is empty
    w = torch.exp(lses - torch.where(torch.isinf(lse), torch.zeros_like(lse), lse)[None])
    w = torch.where(torch.isinf(lses) & (lses < 0), torch.zeros_like(w), w)
    out = (outs.to(torch.float32) * w[..., None]).sum(dim=0)
    return out, lse


def split_key_range(seq_len: int, page_size: int, world_size: int) -> List[tuple]:
    """Page-aligned contiguous key ranges [(first_key, end_key)] of one sequence, one per rank (possibly empty)."""
    pages = (seq_len + page_size - 1) // page_size
    per = (pages + world_size - 1) // world_size
    out = []
    for r in range(world_size):
        k0 = min(r * per * page_size, seq_len)
        k1 = min((r + 1) * per * page_size, seq_len)
        out.append((k0, k1))
    return out


def all_gather_and_merge(local_out: torch.Tensor, local_lse: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> tuple:
    """The exchange step: ONE all_gather_into_tensor of a byte buffer holding the rank's partial output IN ITS OWN
    16-bit type and its f32 lse - RCCL over xGMI with backend "nccl", "gloo" in the CPU tests - then the merge on every
    rank: on the GPU one launch of the library's merge kernel (`mi355_merge_attention_partials`, the key-split merge of
    the prefill path = reduce_segments on normalised partials), result in the query type; with CPU tensors (the gloo
    tests) the torch restatement `merge_partial_attention`. Payload per rank: T*H*(2*D + 4) bytes. No host sync.
    Returns (out [T, H, D], lse [T, H] f32); out is f32 on the CPU path, the query type on the GPU path."""
    world = dist.get_world_size(group)
    T, H, D = local_out.shape
    dev = local_out.device
    fast = local_out.is_cuda and local_out.dtype in (torch.bfloat16, torch.float16) and world <= 8 and D % 8 == 0
    o = local_out.contiguous() if fast else local_out.to(torch.float32).contiguous()
    l32 = local_lse.to(torch.float32).contiguous()
    ob, lb = o.numel() * o.element_size(), l32.numel() * 4
    ob_pad = (ob + 255) & ~255                                   # keeps every rank's lse block (and the next rank's out) 256-byte aligned
    per = ob_pad + ((lb + 255) & ~255)
    send = torch.empty(per, dtype=torch.uint8, device=dev)
    send[:ob].view(o.dtype).view(T, H, D).copy_(o)
    send[ob_pad:ob_pad + lb].view(torch.float32).view(T, H).copy_(l32)
    recv = torch.empty(world * per, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, per)
    outs = recv[:, :ob].view(o.dtype).view(world, T, H, D)
    lses = recv[:, ob_pad:ob_pad + lb].view(torch.float32).view(world, T, H)
    if not fast:
        return merge_partial_attention(outs, lses)
    return merge_partial_attention_device(outs, lses)


def merge_partial_attention_device(outs: torch.Tensor, lses: torch.Tensor) -> tuple:
    """`merge_partial_attention` as ONE launch of the library's merge kernel (`mi355_merge_attention_partials`): outs
    [R, T, H, D] bf16 / f16 on the GPU, lses [R, T, H] f32, R <= 8. Returns (out [T, H, D] in the partials' type, lse f32)."""
    from . import _lib

    R, T, H, D = outs.shape
    dev = outs.device
    part_out = outs.contiguous()           # (gathered blocks sit `per` bytes apart: packed once)
    part_lse = lses.to(torch.float32).contiguous()
    out = torch.empty((T, H, D), dtype=outs.dtype, device=dev)
    lse = torch.empty((T, H), dtype=torch.float32, device=dev)
    rc = _lib.load().mi355_merge_attention_partials(part_out.data_ptr(), part_lse.data_ptr(), R, out.data_ptr(), lse.data_ptr(),
                                                    _lib.dtype_code(outs.dtype), T, H, D, H * D, D, H, _lib.current_stream_handle(dev))
    _lib.check(rc, "mi355_merge_attention_partials")
    return out, lse
