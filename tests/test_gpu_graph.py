"""-m gpu: the ops are HIP-graph capturable (no allocation, no sync, static launch grids): capture
one decode step and one mixed step, then replay with new data in the same buffers."""

import pytest
import torch

from oracle import paged_attention_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("query_lens,kv_lens", [([1] * 6, [300, 17, 2048, 1, 129, 700]), ([1, 40, 1, 129], [90, 70, 513, 400])])
def test_unified_attention_and_cache_write_under_graph_capture(query_lens, kv_lens):
    import gpu_util
    from mi355_attn.kernels import reshape_and_cache_flash, unified_attention

    dev = gpu_util.DEV
    Hq, Hk, D, page = 16, 4, 128, 16
    inp = orc.make_paged_inputs(41, query_lens, kv_lens, Hq, Hk, D, page, torch.bfloat16)
    d = gpu_util.to_dev(inp)
    T = sum(query_lens)
    # the new tokens' K/V are written into the cache inside the graph, like a vLLM step does
    slots = []
    for i, (ql, kl) in enumerate(zip(query_lens, kv_lens)):
        for j in range(kl - ql, kl):
            slots.append(int(inp["block_table"][i, j // page]) * page + j % page)
    slot_mapping = torch.tensor(slots, dtype=torch.int64, device=dev)
    k_new = torch.zeros(T, Hk, D, dtype=torch.bfloat16, device=dev)
    v_new = torch.zeros_like(k_new)
    out = torch.zeros_like(d["q"])

    def step():
        reshape_and_cache_flash(k_new, v_new, d["k_cache"], d["v_cache"], slot_mapping, "auto", None, None)
        unified_attention(q=d["q"], k=d["k_cache"], v=d["v_cache"], out=out, cu_seqlens_q=d["cu_seqlens_q"], max_seqlen_q=max(query_lens),
                          seqused_k=d["seqused_k"], max_seqlen_k=max(kv_lens), avg_seqlen_q=1, avg_seqlen_k=1, softmax_scale=inp["scale"],
                          causal=True, window_size=(-1, -1), block_table=d["block_table"], softcap=0, q_descale=None, k_descale=None,
                          v_descale=None)

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step()                       # warm-up outside the capture: sizes the workspace, sets kernel attributes
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        step()
    # replay twice with fresh data written into the SAME buffers
    for seed in (1, 2):
        g = torch.Generator().manual_seed(seed)
        q2 = (torch.rand(T, Hq, D, generator=g) * 2 - 1).to(torch.bfloat16)
        kn = (torch.rand(T, Hk, D, generator=g) * 2 - 1).to(torch.bfloat16)
        vn = (torch.rand(T, Hk, D, generator=g) * 2 - 1).to(torch.bfloat16)
        d["q"].copy_(q2.to(dev)); k_new.copy_(kn.to(dev)); v_new.copy_(vn.to(dev))
        out.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        kc, vc = d["k_cache"].cpu(), d["v_cache"].cpu()
        ref = orc.unified_attention_oracle(q2, kc, vc, inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"], inp["scale"])
        assert not torch.isnan(out).any()
        torch.testing.assert_close(out.float().cpu(), ref.float(), atol=2e-2, rtol=2e-2)
        # the cache really holds the tokens written inside the graph
        flat_k = kc.view(-1, Hk, D)
        assert torch.equal(flat_k[torch.tensor(slots)].view(torch.int16), kn.view(torch.int16))


def test_replay_after_a_later_eager_call_has_grown_the_workspace():
    """A captured decode graph holds the raw address of its workspace (arrival counters, split partials). A later
    eager call that needs more bytes (a key-split prefill here) makes the binding move on to a bigger buffer; the one
    the graph points into must stay alive and untouched, and the replay must still be right."""
    import gpu_util
    from mi355_attn import _lib
    from mi355_attn.kernels import unified_attention

    dev = gpu_util.DEV
    Hq, Hk, D, page = 16, 4, 128, 16
    _lib._workspaces.clear()                     # start from no workspace: the capture below sizes it for the decode call only
    query_lens, kv_lens = [1] * 5, [3000, 17, 2048, 129, 700]
    inp = orc.make_paged_inputs(43, query_lens, kv_lens, Hq, Hk, D, page, torch.bfloat16)
    d = gpu_util.to_dev(inp)
    out = torch.zeros_like(d["q"])

    def step():
        unified_attention(q=d["q"], k=d["k_cache"], v=d["v_cache"], out=out, cu_seqlens_q=d["cu_seqlens_q"], max_seqlen_q=1,
                          seqused_k=d["seqused_k"], max_seqlen_k=max(kv_lens), avg_seqlen_q=1, avg_seqlen_k=1, softmax_scale=inp["scale"],
                          causal=True, window_size=(-1, -1), block_table=d["block_table"], softcap=0, q_descale=None, k_descale=None,
                          v_descale=None)

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        step()
    key = (dev.type, dev.index, s.cuda_stream)   # workspaces are per (device, stream)
    small = _lib._workspaces[key]
    small_ptr, small_bytes = small.data_ptr(), small.numel()
    # an eager call with a much larger workspace: one 512-token chunk over 8k keys is key-split (partials for every split)
    big = orc.make_paged_inputs(44, [512], [8192], Hq, Hk, D, page, torch.bfloat16)
    bd = gpu_util.to_dev(big)
    with torch.cuda.stream(s):
        bout, kernel = gpu_util.run_unified(bd, big["scale"])
    assert kernel.endswith("_ksplit"), kernel
    grown = _lib._workspaces[key]
    assert grown.numel() > small_bytes and grown.data_ptr() != small_ptr
    # another stream of the same device works in a buffer of its own (calls on two streams may overlap)
    bout2, _ = gpu_util.run_unified(bd, big["scale"])
    other = _lib._workspaces[(dev.type, dev.index, torch.cuda.current_stream(dev).cuda_stream)]
    assert other.data_ptr() != grown.data_ptr()
    assert torch.equal(bout.view(torch.int16), bout2.view(torch.int16))
    assert any(t.data_ptr() == small_ptr for t in _lib._retired)          # the captured buffer is still owned
    junk = [torch.full((small_bytes,), 0xA5, dtype=torch.uint8, device=dev) for _ in range(4)]   # would land in it had it been freed
    # replay with new data
    g = torch.Generator().manual_seed(5)
    q2 = (torch.rand(5, Hq, D, generator=g) * 2 - 1).to(torch.bfloat16)
    d["q"].copy_(q2.to(dev))
    out.fill_(float("nan"))
    graph.replay()
    torch.cuda.synchronize()
    ref = orc.unified_attention_oracle(q2, inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"], inp["scale"])
    torch.testing.assert_close(out.float().cpu(), ref.float(), atol=2e-2, rtol=2e-2)
    del junk


def test_decode_graph_captured_at_max_model_len_replays_any_length():
    """vLLM captures full graphs with max_seq_len = max_model_len (seq_lens = 1 during capture) and replays them on
    whatever the batch holds. The split COUNT is frozen by the capture; how many tiles a split walks is decided on the
    device from each row's own length (reference: kernel_unified_attention_3d :592), so every replay is right AND a
    512-key replay is not left to a single split sized for 131072 keys."""
    import gpu_util
    from mi355_attn.kernels import unified_attention

    dev = gpu_util.DEV
    Hq, Hk, D, page, B = 8, 2, 128, 16, 4
    cap_len = 131072
    pages_per_seq = cap_len // page
    g = torch.Generator().manual_seed(47)
    nb = B * 6300 + 8
    k = (torch.rand(nb, page, Hk, D, generator=g) * 2 - 1).to(torch.bfloat16)
    v = (torch.rand(nb, page, Hk, D, generator=g) * 2 - 1).to(torch.bfloat16)
    bt = torch.zeros(B, pages_per_seq, dtype=torch.int32)
    perm = torch.randperm(nb, generator=g).to(torch.int32)
    for i in range(B):
        bt[i, :6300] = perm[i * 6300:(i + 1) * 6300]
    q = (torch.rand(B, Hq, D, generator=g) * 2 - 1).to(torch.bfloat16)
    cu = torch.arange(B + 1, dtype=torch.int32)
    d = dict(q=q.to(dev), k=k.to(dev), v=v.to(dev), bt=bt.to(dev), cu=cu.to(dev), sl=torch.ones(B, dtype=torch.int32, device=dev))
    out = torch.zeros_like(d["q"])
    scale = 1.0 / (D ** 0.5)

    def step(max_k):
        unified_attention(q=d["q"], k=d["k"], v=d["v"], out=out, cu_seqlens_q=d["cu"], max_seqlen_q=1, seqused_k=d["sl"], max_seqlen_k=max_k,
                          avg_seqlen_q=1, avg_seqlen_k=1, softmax_scale=scale, causal=True, window_size=(-1, -1), block_table=d["bt"], softcap=0,
                          q_descale=None, k_descale=None, v_descale=None)

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step(cap_len)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        step(cap_len)

    def timed(fn, n=40):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    for lens in ([17, 1, 5, 30], [512, 300, 512, 17], [8192, 8000, 100, 8192], [100000, 512, 65536, 99999]):
        d["sl"].copy_(torch.tensor(lens, dtype=torch.int32))
        out.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        ref = torch.cat([gpu_util.oracle_row(orc, q[i:i + 1], k, v, bt[i], lens[i], scale) for i in range(len(lens))])
        assert not torch.isnan(out).any(), lens
        torch.testing.assert_close(out.float().cpu(), ref.float(), atol=2e-2, rtol=2e-2)
        if lens[0] == 512:
            t_replay = timed(graph.replay)
            t_eager = timed(lambda: step(512))                  # planned on the host for 512 keys
            assert t_replay <= 1.5 * t_eager + 0.01, (t_replay, t_eager)


def test_capture_with_the_plain_torch_idiom_gets_a_workspace_of_its_own():
    """`torch.cuda.graph(g)` without `stream=` captures on torch's private capture stream, on which no eager call ever
    ran (what `triton.testing.do_bench_cudagraph` - the reference harness's CUDA_GRAPHS mode, scripts/benchmark.py
    :1733-1738 - and vLLM's full-graph capture do). The binding gives every capture a workspace of its own, allocated
    inside the capture (no eager warm-up needed), so a replay on one stream and eager calls on another never count
    arrivals or park partials in the same bytes."""
    import gpu_util
    from mi355_attn import _lib
    from mi355_attn.kernels import unified_attention

    dev = gpu_util.DEV
    Hq, Hk, D, page = 16, 4, 128, 16
    query_lens, kv_lens = [1] * 5, [3000, 17, 2048, 129, 700]
    inp = orc.make_paged_inputs(45, query_lens, kv_lens, Hq, Hk, D, page, torch.bfloat16)
    d = gpu_util.to_dev(inp)
    out = torch.zeros_like(d["q"])
    q_eager, out_eager = d["q"].clone(), torch.zeros_like(d["q"])

    def step(q=None, o=None):
        unified_attention(q=d["q"] if q is None else q, k=d["k_cache"], v=d["v_cache"], out=out if o is None else o, cu_seqlens_q=d["cu_seqlens_q"],
                          max_seqlen_q=1, seqused_k=d["seqused_k"], max_seqlen_k=max(kv_lens), avg_seqlen_q=1, avg_seqlen_k=1,
                          softmax_scale=inp["scale"], causal=True, window_size=(-1, -1), block_table=d["block_table"], softcap=0,
                          q_descale=None, k_descale=None, v_descale=None)

    _lib._workspaces.clear()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):                 # torch's own capture stream, nothing warmed up
        step()
    cap_keys = [k for k in _lib._workspaces if k[2] == "capture"]
    assert len(cap_keys) == 1 and cap_keys[0][3] != 0, cap_keys       # keyed by the capture's id
    graph2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph2):                # a second capture on the same private stream: bytes of its own
        step()
    assert len([k for k in _lib._workspaces if k[2] == "capture"]) == 2
    ptrs = {ws.data_ptr() for k, ws in _lib._workspaces.items()}
    side = torch.cuda.Stream()
    for seed in (1, 2):
        g = torch.Generator().manual_seed(seed)
        q2 = (torch.rand(5, Hq, D, generator=g) * 2 - 1).to(torch.bfloat16)
        d["q"].copy_(q2.to(dev))
        q_eager.copy_(q2.to(dev))
        out.fill_(float("nan"))
        out_eager.fill_(float("nan"))
        torch.cuda.synchronize()
        for _ in range(20):                       # replays on the current stream, eager calls on another, free to overlap
            graph.replay()
            with torch.cuda.stream(side):
                step(q_eager, out_eager)
        torch.cuda.synchronize()
        ref = orc.unified_attention_oracle(q2, inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"], inp["scale"])
        torch.testing.assert_close(out.float().cpu(), ref.float(), atol=2e-2, rtol=2e-2)
        torch.testing.assert_close(out_eager.float().cpu(), ref.float(), atol=2e-2, rtol=2e-2)
    assert len({ws.data_ptr() for ws in _lib._workspaces.values()}) == len(ptrs) + 1     # (the side stream's eager workspace)


def test_multi_token_decode_graph_replays_other_lengths_and_draft_counts():
    """A speculative-decoding verification step captured for B sequences x (1 + k) tokens at max_model_len: replays see
    other context lengths AND sequences that carry fewer tokens than the capture's maximum (rejected drafts / padding:
    cu_seqlens_q changes inside the same buffers, trailing token rows unused). Everything the packed decode launch is
    sized from is host-known and fixed by the capture."""
    import gpu_util
    from mi355_attn import _lib
    from mi355_attn.kernels import unified_attention

    dev = gpu_util.DEV
    Hq, Hk, D, page, B, Tq = 32, 8, 128, 16, 5, 4
    cap_len = 32768
    first = dict(q_lens=[Tq] * B, kv_lens=[3000, 40, 700, 4, 1290])
    inp = orc.make_paged_inputs(51, first["q_lens"], first["kv_lens"], Hq, Hk, D, page, torch.bfloat16)
    d = gpu_util.to_dev(inp)
    out = torch.zeros_like(d["q"])

    def step():
        unified_attention(q=d["q"], k=d["k_cache"], v=d["v_cache"], out=out, cu_seqlens_q=d["cu_seqlens_q"], max_seqlen_q=Tq,
                          seqused_k=d["seqused_k"], max_seqlen_k=cap_len, avg_seqlen_q=1, avg_seqlen_k=1, softmax_scale=inp["scale"],
                          causal=True, window_size=(-1, -1), block_table=d["block_table"], softcap=0, q_descale=None, k_descale=None,
                          v_descale=None)

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step()
    torch.cuda.synchronize()
    assert "pack" in _lib.last_kernel()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        step()
    for q_lens, kv_lens in ([[4, 4, 4, 4, 4], [3000, 40, 700, 4, 1290]], [[4, 1, 3, 2, 4], [2999, 1, 650, 2, 100]], [[1, 1, 1, 1, 1], [17, 3000, 5, 64, 1]],
                            [[2, 4, 0, 4, 3], [33, 1200, 0, 500, 3]]):
        cu = torch.zeros(B + 1, dtype=torch.int32)
        cu[1:] = torch.cumsum(torch.tensor(q_lens, dtype=torch.int32), 0)
        d["cu_seqlens_q"].copy_(cu)
        d["seqused_k"].copy_(torch.tensor(kv_lens, dtype=torch.int32))
        out.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        T = int(cu[-1])
        ref = orc.unified_attention_oracle(inp["q"][:T], inp["k_cache"], inp["v_cache"], cu, torch.tensor(kv_lens, dtype=torch.int32), inp["block_table"],
                                           inp["scale"], mode="3d")
        assert not torch.isnan(out[:T]).any(), q_lens
        torch.testing.assert_close(out[:T].float().cpu(), ref.float(), atol=2e-2, rtol=2e-2)
        assert torch.isnan(out[T:]).all()          # token rows past the batch stay untouched


@pytest.mark.parametrize("query_lens,kv_lens,expect", [([500], [500], "prefill_mfma_lat"), ([300, 200, 260], [300, 712, 260], "prefill_mfma"),
                                                        ([1, 130, 1, 64], [300, 130, 77, 320], "+decode")])
def test_prefill_step_with_the_cache_write_inside_replays_with_new_data(query_lens, kv_lens, expect):
    """Round 4 (library 0.6.0): a prefill step as ONE call - `prefill_attention_and_cache_write`: the step's keys attended over
    from the linear tensors, stored by the launch - captured in a HIP graph and replayed with new queries / keys / values in
    the same buffers (what `tools/e2e_proxy.py` times): output against the oracle over the cache as the replay left it, the
    new rows' bytes in the cache. A one-prompt step (the latency kernel), a ragged step (the LDS-DMA kernel), a mixed step."""
    import gpu_util
    from mi355_attn import _lib
    from mi355_attn.kernels.unified import prefill_attention_and_cache_write

    dev = gpu_util.DEV
    Hq, Hk, D, page = 32, 8, 128, 16
    inp = orc.make_paged_inputs(43, query_lens, kv_lens, Hq, Hk, D, page, torch.bfloat16)
    d = gpu_util.to_dev(inp)
    T = sum(query_lens)
    slots = []
    for i, (ql, kl) in enumerate(zip(query_lens, kv_lens)):
        for j in range(kl - ql, kl):
            slots.append(int(inp["block_table"][i, j // page]) * page + j % page)
    slot_mapping = torch.tensor(slots, dtype=torch.int64, device=dev)
    k_new = torch.zeros(T, Hk, D, dtype=torch.bfloat16, device=dev)
    v_new = torch.zeros_like(k_new)
    out = torch.zeros_like(d["q"])

    def step():
        ok = prefill_attention_and_cache_write(d["q"], k_new, v_new, d["k_cache"], d["v_cache"], out, d["cu_seqlens_q"], max(query_lens), d["seqused_k"],
                                               max(kv_lens), inp["scale"], d["block_table"], slot_mapping)
        assert ok, "this step is served with the write inside the attention launches"

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step()
    torch.cuda.synchronize()
    assert expect in _lib.last_kernel(), _lib.last_kernel()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        step()
    for seed in (1, 2):
        g = torch.Generator().manual_seed(seed)
        q2 = (torch.rand(T, Hq, D, generator=g) * 2 - 1).to(torch.bfloat16)
        kn = (torch.rand(T, Hk, D, generator=g) * 2 - 1).to(torch.bfloat16)
        vn = (torch.rand(T, Hk, D, generator=g) * 2 - 1).to(torch.bfloat16)
        d["q"].copy_(q2.to(dev)); k_new.copy_(kn.to(dev)); v_new.copy_(vn.to(dev))
        out.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        kc, vc = d["k_cache"].cpu(), d["v_cache"].cpu()
        ref = orc.unified_attention_oracle(q2, kc, vc, inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"], inp["scale"])
        assert not torch.isnan(out).any()
        torch.testing.assert_close(out.float().cpu(), ref.float(), atol=2e-2, rtol=2e-2)
        idx = torch.tensor(slots)
        assert torch.equal(kc.view(-1, Hk, D)[idx].view(torch.int16), kn.view(torch.int16))
        assert torch.equal(vc.view(-1, Hk, D)[idx].view(torch.int16), vn.view(torch.int16))


@pytest.mark.parametrize("query_lens,kv_lens,expect", [([500], [500], "prefill_mfma_lat_fp8"), ([2100, 1], [2300, 900], "prefill_mfma_pw_fp8")])
def test_prefill_over_an_fp8_cache_replays_with_new_queries(query_lens, kv_lens, expect):
    """The fp8 forms of the prefill kernels (round 4) under capture: a short prompt and a long chunk + a decode row, replayed
    with new queries over the same fp8 cache."""
    import gpu_util
    from mi355_attn import _lib

    dev = gpu_util.DEV
    ks = 0.31
    inp = orc.make_paged_inputs(44, query_lens, kv_lens, 8, 2, 128, 16, torch.bfloat16, kv_dtype=torch.float8_e4m3fn, kv_scale=ks)
    d = gpu_util.to_dev(inp)
    out = torch.zeros_like(d["q"])
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        gpu_util.run_unified(d, inp["scale"], kv_scale=ks, out=out)
    assert _lib.last_kernel().startswith(expect), _lib.last_kernel()
    graph = torch.cuda.CUDAGraph()
    from mi355_attn.kernels import unified_attention
    kst = torch.tensor([ks], dtype=torch.float32, device=dev)
    with torch.cuda.graph(graph, stream=s):
        unified_attention(q=d["q"], k=d["k_cache"], v=d["v_cache"], out=out, cu_seqlens_q=d["cu_seqlens_q"], max_seqlen_q=max(query_lens),
                          seqused_k=d["seqused_k"], max_seqlen_k=max(kv_lens), avg_seqlen_q=1, avg_seqlen_k=1, softmax_scale=inp["scale"],
                          causal=True, window_size=(-1, -1), block_table=d["block_table"], softcap=0, q_descale=None, k_descale=kst, v_descale=kst)
    for seed in (1, 2):
        g = torch.Generator().manual_seed(seed)
        q2 = (torch.rand(*inp["q"].shape, generator=g) * 2 - 1).to(torch.bfloat16)
        d["q"].copy_(q2.to(dev))
        out.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        ref = orc.unified_attention_oracle(q2, inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"], inp["scale"],
                                           k_scale=ks, v_scale=ks)
        assert not torch.isnan(out).any()
        torch.testing.assert_close(out.float().cpu(), ref.float(), atol=3e-2, rtol=3e-2)
