"""-m gpu: the paged-cache write fused into the decode launch (replaces the reshape_and_cache_flash
launch in front of the attention, LIB/backend/triton_attn.py:393-405, for steps in which every sequence has one query
token). The cache must hold exactly what the separate write would have stored (bit for bit, fp8 quantisation
included), the output must match the oracle run on that updated cache, and nothing else in the cache may change."""

import math
import types

import pytest
import torch

import golden_io
from oracle import paged_attention_oracle as orc

pytestmark = pytest.mark.gpu


def _case(seed, kv_lens, hq, hk, d, page, dtype, kv_dtype=None, k_scale=1.0, v_scale=1.0):
    inp = orc.make_paged_inputs(seed, [1] * len(kv_lens), kv_lens, hq, hk, d, page, dtype, kv_dtype=kv_dtype, kv_scale=k_scale)
    g = torch.Generator().manual_seed(seed + 100)
    n = len(kv_lens)
    inp["k_new"] = ((torch.rand(n, hk, d, generator=g) * 2 - 1) * 1.2).to(dtype)
    inp["v_new"] = (torch.rand(n, hk, d, generator=g) * 2 - 1).to(dtype)
    inp["k_new"][0, 0, :4] = torch.tensor([0.0, -0.0, 1e-8, -1e-8], dtype=dtype)       # zeros and values that round to zero
    if kv_dtype is not None:                                                          # and a few that saturate the fp8 range
        big = min(k_scale * torch.finfo(kv_dtype).max * 1.5, 60000.0)
        inp["v_new"][-1, 0, :2] = torch.tensor([big, -big], dtype=dtype)
    inp["slots"] = torch.tensor([int(inp["block_table"][i, (kl - 1) // page]) * page + (kl - 1) % page for i, kl in enumerate(kv_lens)], dtype=torch.int64)
    return inp


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("kv", ["same", "e4m3", "e5m2"])
@pytest.mark.parametrize("hq,hk,d,page", [(32, 8, 128, 16), (8, 2, 64, 16), (4, 1, 256, 32), (6, 2, 96, 16), (40, 2, 128, 16)])
def test_fused_decode_write_matches_separate_write_and_oracle(dtype, kv, hq, hk, d, page):
    import gpu_util
    from mi355_attn import _lib
    from mi355_attn.kernels import reshape_and_cache_flash
    from mi355_attn.kernels.unified import decode_attention_and_cache_write

    dev = gpu_util.DEV
    kv_dtype = {"same": None, "e4m3": torch.float8_e4m3fn, "e5m2": torch.float8_e5m2}[kv]
    if kv_dtype is not None and d % 16:
        pytest.skip("fp8 caches need a head size that is a multiple of 16")
    ks, vs = (0.0237, 0.041) if kv_dtype is not None else (1.0, 1.0)
    kv_lens = [1, 16, 17, 32, 33, 300, 1000, 2049, 64, 5]
    inp = _case(51, kv_lens, hq, hk, d, page, dtype, kv_dtype, ks, vs)
    d_ = gpu_util.to_dev(inp)
    kst, vst = (torch.tensor([ks], device=dev), torch.tensor([vs], device=dev)) if kv_dtype is not None else (None, None)
    # reference: the separate write, then the oracle on the cache it produced
    kc_ref, vc_ref = d_["k_cache"].clone(), d_["v_cache"].clone()
    reshape_and_cache_flash(d_["k_new"], d_["v_new"], kc_ref, vc_ref, d_["slots"], "auto" if kv_dtype is None else "fp8", kst, vst)
    torch.cuda.synchronize()
    ref = orc.unified_attention_oracle(inp["q"], kc_ref.cpu(), vc_ref.cpu(), inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"], inp["scale"],
                                       k_scale=ks, v_scale=vs, mode="3d")
    # fused: poison the slots being written so that a kernel that attends over the OLD cache contents shows
    kc, vc = d_["k_cache"].clone(), d_["v_cache"].clone()
    flat_k, flat_v = kc.view(-1, hk, d), vc.view(-1, hk, d)
    if kv_dtype is None:
        flat_k[d_["slots"]] = float("nan")
        flat_v[d_["slots"]] = float("nan")
    else:
        flat_k.view(torch.uint8)[d_["slots"]] = 0x7F
        flat_v.view(torch.uint8)[d_["slots"]] = 0x7F
    out = torch.full_like(d_["q"], float("nan"))
    fused = decode_attention_and_cache_write(d_["q"], d_["k_new"], d_["v_new"], kc, vc, out, d_["seqused_k"], max(kv_lens), inp["scale"],
                                             d_["block_table"], kst, vst)
    torch.cuda.synchronize()
    assert fused, "the fused decode kernel should serve this configuration"
    assert _lib.last_kernel().startswith("decode"), _lib.last_kernel()
    assert torch.equal(kc.view(torch.uint8), kc_ref.view(torch.uint8))          # the whole cache, bit for bit
    assert torch.equal(vc.view(torch.uint8), vc_ref.view(torch.uint8))
    atol, rtol = golden_io.tolerance(dtype, kv_dtype)
    assert not torch.isnan(out).any()
    torch.testing.assert_close(out.float().cpu(), ref.float(), atol=atol, rtol=rtol)


def test_fused_decode_write_is_declined_where_it_does_not_apply():
    import ctypes as C

    import gpu_util
    from mi355_attn import _lib
    from mi355_attn.kernels.unified import fill_attn_params

    dev = gpu_util.DEV
    inp = _case(52, [40, 70], 4, 4, 128, 16, torch.float32)       # fp32: no matrix-core decode kernel
    d_ = gpu_util.to_dev(inp)
    out = torch.zeros_like(d_["q"])
    p, keep = fill_attn_params(d_["q"], d_["k_cache"], d_["v_cache"], out, d_["cu_seqlens_q"], 1, d_["seqused_k"], 70, inp["scale"], (-1, -1),
                               d_["block_table"], 0.0, None, None, None, None, k_new=d_["k_new"], v_new=d_["v_new"], write_new_kv=True)
    lib = _lib.load()
    assert lib.mi355_decode_write_fusable(C.byref(p)) == 0
    assert lib.mi355_unified_attention(C.byref(p), None, 0, None) == _lib.MI355_ERR_UNSUPPORTED
    p.max_seqlen_q = 2        # (not a decode step: since library 0.6.0 a request the short-prompt prefill kernel may serve - not in f32)
    assert lib.mi355_decode_write_fusable(C.byref(p)) == 0
    assert lib.mi355_unified_attention(C.byref(p), None, 0, None) == _lib.MI355_ERR_UNSUPPORTED


@pytest.mark.parametrize("kv_cache_dtype", ["auto", "fp8"])
def test_impl_forward_issues_one_launch_for_a_decode_step(kv_cache_dtype, monkeypatch):
    """MI355AttentionImpl.forward on a decode-only step: ONE C call (counted), cache and output as with the two calls."""
    import gpu_util
    from mi355_attn import _lib
    from mi355_attn.backend import attn

    dev = gpu_util.DEV
    Hq, Hk, D, page = 16, 4, 128, 16
    fp8 = kv_cache_dtype == "fp8"
    cache_dtype = torch.float8_e4m3fn if fp8 else torch.bfloat16
    ks, vs = (0.03, 0.05) if fp8 else (1.0, 1.0)
    kv_lens = [300, 17, 2048, 1, 129, 64]
    inp = _case(53, kv_lens, Hq, Hk, D, page, torch.bfloat16, cache_dtype if fp8 else None, ks, vs)
    T = len(kv_lens)
    kc, vc = inp["k_cache"].clone(), inp["v_cache"].clone()
    kv_cache = torch.stack([kc, vc]).to(dev)
    if fp8:
        kv_cache = kv_cache.view(torch.uint8)
    kc_ref, vc_ref = inp["k_cache"].clone(), inp["v_cache"].clone()
    orc.reshape_and_cache_flash_oracle(inp["k_new"], inp["v_new"], kc_ref, vc_ref, inp["slots"], k_scale=ks, v_scale=vs)
    ref = orc.unified_attention_oracle(inp["q"], kc_ref, vc_ref, inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"], inp["scale"],
                                       k_scale=ks, v_scale=vs, mode="3d")
    impl = attn.MI355AttentionImpl(Hq, D, inp["scale"], Hk, None, None, kv_cache_dtype)
    layer = types.SimpleNamespace(_k_scale=torch.tensor(ks, device=dev), _v_scale=torch.tensor(vs, device=dev), _q_scale=torch.tensor(1.0, device=dev))
    pad = 2
    slot_mapping = torch.cat([inp["slots"], torch.full((pad,), -1, dtype=torch.int64)])
    md = attn.MI355AttentionMetadata(
        num_actual_tokens=T, max_query_len=1, avg_query_len=1, avg_seq_len=sum(kv_lens) // T, query_start_loc=inp["cu_seqlens_q"].to(dev),
        max_seq_len=max(kv_lens), seq_lens=inp["seqused_k"].to(dev), block_table=inp["block_table"].to(dev), slot_mapping=slot_mapping.to(dev),
        use_cascade=False, common_prefix_len=0, cu_prefix_query_lens=None, prefix_kv_lens=None, suffix_kv_lens=None)
    q_pad = torch.zeros(T + pad, Hq, D, dtype=torch.bfloat16, device=dev); q_pad[:T] = inp["q"].to(dev)
    k_pad = torch.zeros(T + pad, Hk, D, dtype=torch.bfloat16, device=dev); k_pad[:T] = inp["k_new"].to(dev)
    v_pad = torch.zeros(T + pad, Hk, D, dtype=torch.bfloat16, device=dev); v_pad[:T] = inp["v_new"].to(dev)
    output = torch.full((T + pad, Hq * D), float("nan"), dtype=torch.bfloat16, device=dev)
    lib = _lib.load()
    calls = {"attn": 0, "cache": 0}
    real_attn, real_cache = lib.mi355_unified_attention, lib.mi355_reshape_and_cache_flash

    class Counting:
        def __init__(self, fn, key):
            self.fn, self.key = fn, key
            self.restype, self.argtypes = fn.restype, fn.argtypes

        def __call__(self, *a):
            calls[self.key] += 1
            return self.fn(*a)

    monkeypatch.setattr(lib, "mi355_unified_attention", Counting(real_attn, "attn"), raising=False)
    monkeypatch.setattr(lib, "mi355_reshape_and_cache_flash", Counting(real_cache, "cache"), raising=False)
    ret = impl.forward(layer, q_pad, k_pad, v_pad, kv_cache, md, output=output)
    torch.cuda.synchronize()
    assert ret is output
    assert calls == {"attn": 1, "cache": 0}, calls
    atol, rtol = golden_io.tolerance(torch.bfloat16, cache_dtype if fp8 else None)
    torch.testing.assert_close(output[:T].view(T, Hq, D).float().cpu(), ref.float(), atol=atol, rtol=rtol)
    assert torch.isnan(output[T:]).all()
    got_k = kv_cache[0].view(cache_dtype) if fp8 else kv_cache[0]
    got_v = kv_cache[1].view(cache_dtype) if fp8 else kv_cache[1]
    assert torch.equal(got_k.cpu().view(torch.uint8), kc_ref.view(torch.uint8))
    assert torch.equal(got_v.cpu().view(torch.uint8), vc_ref.view(torch.uint8))


@pytest.mark.parametrize("kv_cache_dtype", ["auto", "fp8"])
def test_padded_decode_graph_leaves_the_cache_of_padding_rows_untouched(kv_cache_dtype):
    """Full-graph mode (LIB/backend/triton_attn.py:107,:120-128,:149-151): a decode step is captured at batch 8 and
    replayed with 5 live sequences; the 3 padding rows carry slot -1, and their seq_lens / block-table rows hold whatever
    an earlier step left there. The reference's cache write skips slot -1; the fused write must as well - both for
    padding rows with seq_len 0 (vLLM zeroes them) and for rows whose stale seq_len and block-table row point INTO a
    live sequence's pages."""
    import gpu_util
    from mi355_attn.backend import attn

    dev = gpu_util.DEV
    Hq, Hk, D, page, B, live = 16, 4, 128, 16, 8, 5
    fp8 = kv_cache_dtype == "fp8"
    cache_dtype = torch.float8_e4m3fn if fp8 else torch.bfloat16
    ks, vs = (0.03, 0.05) if fp8 else (1.0, 1.0)
    kv_lens = [300, 17, 2048, 33, 129, 64, 500, 16]
    inp = _case(57, kv_lens, Hq, Hk, D, page, torch.bfloat16, cache_dtype if fp8 else None, ks, vs)
    kv_cache = torch.stack([inp["k_cache"], inp["v_cache"]]).to(dev)
    if fp8:
        kv_cache = kv_cache.view(torch.uint8)
    impl = attn.MI355AttentionImpl(Hq, D, inp["scale"], Hk, None, None, kv_cache_dtype)
    layer = types.SimpleNamespace(_k_scale=torch.tensor(ks, device=dev), _v_scale=torch.tensor(vs, device=dev), _q_scale=torch.tensor(1.0, device=dev))
    seq_lens = inp["seqused_k"].to(dev).clone()
    slot_mapping = inp["slots"].to(dev).clone()
    block_table = inp["block_table"].to(dev).clone()
    md = attn.MI355AttentionMetadata(
        num_actual_tokens=B, max_query_len=1, avg_query_len=1, avg_seq_len=100, query_start_loc=inp["cu_seqlens_q"].to(dev),
        max_seq_len=max(kv_lens), seq_lens=seq_lens, block_table=block_table, slot_mapping=slot_mapping,
        use_cascade=False, common_prefix_len=0, cu_prefix_query_lens=None, prefix_kv_lens=None, suffix_kv_lens=None)
    q, kn, vn = inp["q"].to(dev), inp["k_new"].to(dev), inp["v_new"].to(dev)
    output = torch.zeros(B, Hq * D, dtype=torch.bfloat16, device=dev)
    snapshot = kv_cache.clone()
    impl.forward(layer, q, kn, vn, kv_cache, md, output=output)       # warm-up (sizes the workspace), then restore the cache
    torch.cuda.synchronize()
    kv_cache.copy_(snapshot)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):                                     # the plain torch idiom: torch's own capture stream
        impl.forward(layer, q, kn, vn, kv_cache, md, output=output)
    kv_cache.copy_(snapshot)
    for stale in (False, True):
        # rows 5..7 are padding: slot -1; seq_len 0 (what vLLM writes) or - the harder case - stale values whose last
        # position lies inside live sequence 2's pages
        slot_mapping[live:] = -1
        if stale:
            seq_lens[live:] = torch.tensor([700, 1024, 2048], dtype=torch.int32, device=dev)
            block_table[live:] = block_table[2]
        else:
            seq_lens[live:] = 0
        kv_cache.copy_(snapshot)
        output.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        # expected cache: the separate write with the same slot mapping (slot -1 skipped)
        kc_ref, vc_ref = inp["k_cache"].clone(), inp["v_cache"].clone()
        orc.reshape_and_cache_flash_oracle(inp["k_new"], inp["v_new"], kc_ref, vc_ref, slot_mapping.cpu(), k_scale=ks, v_scale=vs)
        got_k = kv_cache[0].view(cache_dtype) if fp8 else kv_cache[0]
        got_v = kv_cache[1].view(cache_dtype) if fp8 else kv_cache[1]
        assert torch.equal(got_k.cpu().view(torch.uint8), kc_ref.view(torch.uint8)), f"stale={stale}"
        assert torch.equal(got_v.cpu().view(torch.uint8), vc_ref.view(torch.uint8)), f"stale={stale}"
        ref = orc.unified_attention_oracle(inp["q"][:live], kc_ref, vc_ref, inp["cu_seqlens_q"][:live + 1], inp["seqused_k"][:live],
                                           inp["block_table"][:live], inp["scale"], k_scale=ks, v_scale=vs, mode="3d")
        atol, rtol = golden_io.tolerance(torch.bfloat16, cache_dtype if fp8 else None)
        torch.testing.assert_close(output[:live].view(live, Hq, D).float().cpu(), ref.float(), atol=atol, rtol=rtol)
