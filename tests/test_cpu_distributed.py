"""World-size-2 `gloo` test of the multi-GPU path (batch sharding) on CPU: each rank
takes its share with parallel.shard_batch, computes it (with the oracle standing in for the kernel —
the partitioning and the gather are what is under test), and the gathered result must equal the
single-process result bit for bit. KV-head sharding is checked the same way."""

import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q_out):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "vllm-triton-backend_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mi355_attn import parallel
        from oracle import paged_attention_oracle as orc

        query_lens, kv_lens = [7, 1, 1, 40, 9, 1, 33], [70, 45, 33, 70, 33, 200, 33]
        inp = orc.make_paged_inputs(3, query_lens, kv_lens, 8, 2, 64, 16, torch.float32)
        full = orc.unified_attention_oracle(inp["q"], inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"],
                                            inp["scale"])
        # --- batch sharding + gather
        lb = parallel.shard_batch(rank, world, inp["q"], inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"])
        assert lb.k_cache.shape[0] == sum((kv_lens[i] + 15) // 16 for i in lb.seq_ids)      # only this rank's pages
        local = orc.unified_attention_oracle(lb.q, lb.k_cache, lb.v_cache, lb.cu_seqlens_q, lb.seqused_k, lb.block_table, inp["scale"])
        got = parallel.gather_outputs(local, lb, inp["q"].shape[0])
        ok_batch = torch.equal(got, full)
        # --- KV-head sharding (tensor-parallel layout): concatenating the head slices gives the full output
        q_r, k_r, v_r, qs = parallel.shard_kv_heads(rank, world, inp["q"], inp["k_cache"], inp["v_cache"])
        part = orc.unified_attention_oracle(q_r.contiguous(), k_r.contiguous(), v_r.contiguous(), inp["cu_seqlens_q"], inp["seqused_k"],
                                            inp["block_table"], inp["scale"])
        parts = [torch.empty_like(part) for _ in range(world)]
        dist.all_gather(parts, part)
        ok_heads = torch.equal(torch.cat(parts, dim=1), full)
        q_out.put((rank, ok_batch, ok_heads, lb.seq_ids))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_batch_and_head_sharding_world2_gloo():
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q_out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q_out.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    owned = []
    for rank, ok_batch, ok_heads, seq_ids in res:
        assert ok_batch, f"rank {rank}: gathered batch-sharded output differs from the single-process output"
        assert ok_heads, f"rank {rank}: concatenated head-sharded output differs"
        owned += seq_ids
    assert sorted(owned) == list(range(7))


def _cp_worker(rank, world, port, q_out):
    """Context parallelism: each rank attends its page-aligned key range of every (decode) sequence; the exchange step
    (all_gather of partial outputs + log-sum-exps over gloo, then the merge) must give the full result on every rank."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "vllm-triton-backend_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mi355_attn import parallel
        from oracle import paged_attention_oracle as orc

        page, kv_lens = 16, [1000, 40, 17, 1]
        inp = orc.make_paged_inputs(5, [1] * len(kv_lens), kv_lens, 8, 2, 64, page, torch.float32)
        full, full_lse = orc.dense_attention_fp64(inp["q"], inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"],
                                                  inp["block_table"], inp["scale"], return_lse=True)
        ranges = [parallel.split_key_range(n, page, world)[rank] for n in kv_lens]
        width = max(max((k1 - k0 + page - 1) // page for k0, k1 in ranges), 1)
        bt = torch.zeros((len(kv_lens), width), dtype=torch.int32)
        for i, (k0, k1) in enumerate(ranges):
            n_pages = (k1 - k0 + page - 1) // page
            bt[i, :n_pages] = inp["block_table"][i, k0 // page: k0 // page + n_pages]
        lens = torch.tensor([k1 - k0 for k0, k1 in ranges], dtype=torch.int32)
        # the oracle stands in for the kernel: the split, the exchange and the merge are what is under test
        part, part_lse = orc.dense_attention_fp64(inp["q"], inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], lens, bt, inp["scale"],
                                                  return_lse=True)
        merged, merged_lse = parallel.all_gather_and_merge(part, part_lse.to(torch.float32))
        q_out.put((rank, float((merged.double() - full).abs().max()), float((merged_lse.double() - full_lse).abs().max()), lens.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_context_parallel_exchange_world2_gloo():
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cp_worker, args=(r, 2, port, q_out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q_out.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err_out, err_lse, lens in sorted(res):
        assert err_out < 1e-5 and err_lse < 1e-5, (rank, err_out, err_lse)
    assert sorted(res)[0][3] == [512, 32, 16, 1] and sorted(res)[1][3] == [488, 8, 1, 0]   # rank 1 holds no key of the last sequence


def test_shard_batch_plan_equals_the_per_page_walk():
    """The vectorised shard plan (tokens, compacted pages, local block table) against the obvious per-sequence, per-page walk
    on a ragged batch: more ranks than heavy sequences, a sequence without query tokens, one without keys, 1 .. 5 ranks."""
    import torch
    from mi355_attn import parallel

    g = torch.Generator().manual_seed(3)
    q_lens = [7, 1, 0, 40, 1, 129, 3]
    kv_lens = [70, 45, 16, 70, 0, 300, 3]
    page, Hq, Hk, D = 16, 4, 2, 8
    S, T = len(q_lens), sum(q_lens)
    pps = [(n + page - 1) // page for n in kv_lens]
    nb = sum(pps) + 5
    k = torch.rand(nb, page, Hk, D, generator=g)
    v = torch.rand(nb, page, Hk, D, generator=g)
    q = torch.rand(T, Hq, D, generator=g)
    perm = torch.randperm(nb, generator=g).to(torch.int32)
    bt = torch.full((S, max(pps)), -7, dtype=torch.int32)          # (entries past a sequence's pages hold junk)
    o = 0
    for i, n in enumerate(pps):
        bt[i, :n] = perm[o:o + n]
        o += n
    cu = torch.zeros(S + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(torch.tensor(q_lens, dtype=torch.int32), 0)
    sk = torch.tensor(kv_lens, dtype=torch.int32)
    for world in (1, 2, 3, 5):
        owned = parallel.assign_sequences(q_lens, kv_lens, world)
        assert sorted(i for o_ in owned for i in o_) == list(range(S))
        seen_tokens = []
        for rank in range(world):
            loc = parallel.shard_batch(rank, world, q, k, v, cu, sk, bt)
            assert loc.seq_ids == owned[rank]
            tok, pages = [], []
            for i in owned[rank]:
                tok += list(range(int(cu[i]), int(cu[i + 1])))
                pages += bt[i, : pps[i]].tolist()
            assert loc.token_index.tolist() == tok
            seen_tokens += tok
            assert torch.equal(loc.q, q[torch.tensor(tok, dtype=torch.long)] if tok else q[:0])
            assert torch.equal(loc.k_cache, k[torch.tensor(pages, dtype=torch.long)] if pages else k[:0])
            assert torch.equal(loc.v_cache, v[torch.tensor(pages, dtype=torch.long)] if pages else v[:0])
            # every local sequence's pages, through the LOCAL block table, are its global pages in order
            for row, i in enumerate(owned[rank]):
                assert int(loc.seqused_k[row]) == kv_lens[i] and int(loc.cu_seqlens_q[row + 1] - loc.cu_seqlens_q[row]) == q_lens[i]
                lp = loc.block_table[row, : pps[i]].long()
                assert torch.equal(loc.k_cache[lp], k[bt[i, : pps[i]].long()])
            assert [m.tolist() for m in loc.rank_token_index] == [[t for i in o_ for t in range(int(cu[i]), int(cu[i + 1]))] for o_ in owned]
        assert sorted(seen_tokens) == list(range(T))
