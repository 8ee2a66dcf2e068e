"""-m gpu: KV caches beyond 4 GiB. Page byte offsets no longer fit 32 bits there; every kernel family must carry
them in 64 bits. The oracle runs on a small logical cache; the same pages are then scattered to the FAR END of a cache
of > 4 GiB per tensor on the GPU (block table re-indexed) and the HIP result must not change."""

import pytest
import torch

import golden_io
from oracle import paged_attention_oracle as orc

pytestmark = pytest.mark.gpu


def _far_end_copy(inp, dev, kv_dtype=None):
    """Returns device tensors whose caches have ~6 GiB each, with the used pages at the highest page indices."""
    k_small, v_small = inp["k_cache"], inp["v_cache"]
    nb_small, page, hk, d = k_small.shape
    page_bytes = page * hk * d * k_small.element_size()
    nb_big = int(6.2 * (1 << 30)) // page_bytes
    assert nb_big * page_bytes > (1 << 32) + (1 << 30)
    k_big = torch.zeros((nb_big, page, hk, d), dtype=k_small.dtype, device=dev)
    v_big = torch.zeros((nb_big, page, hk, d), dtype=v_small.dtype, device=dev)
    shift = nb_big - nb_small                       # page i of the small cache -> page i + shift
    k_big[shift:] = k_small.to(dev)
    v_big[shift:] = v_small.to(dev)
    t = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in inp.items() if k not in ("k_cache", "v_cache")}
    t["k_cache"], t["v_cache"] = k_big, v_big
    t["block_table"] = (inp["block_table"].to(torch.int64) + shift).to(torch.int32).to(dev)
    assert int(t["block_table"].max()) * page_bytes > (1 << 32)
    return t


@pytest.mark.parametrize("case", ["decode", "prefill_dma", "prefill_feat", "decode_fp8", "generic"])
def test_pages_beyond_4_gib(case):
    import gpu_util

    dtype = torch.bfloat16
    kv_dtype, kv_scale, window, force = None, None, 0, None
    if case.startswith("decode"):
        query_lens, kv_lens = [1] * 5, [700, 33, 1023, 257, 1]
        expect = "decode"
        if case == "decode_fp8":
            kv_dtype, kv_scale, expect = torch.float8_e4m3fn, 0.5, "decode_splitkv_fp8"
    else:
        query_lens, kv_lens = [129, 64, 200, 5], [129, 257, 777, 5]
        expect = "prefill_mfma"
        if case == "prefill_feat":
            window, expect = 100, "prefill_mfma_feat"
        if case == "generic":
            force, expect = 9, "generic"
    kw = dict(kv_dtype=kv_dtype, kv_scale=kv_scale) if kv_dtype is not None else {}
    inp = orc.make_paged_inputs(50, query_lens, kv_lens, 8, 2, 128, 16, dtype, **kw)
    ref = orc.unified_attention_oracle(inp["q"], inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"],
                                       inp["scale"], sliding_window=window, k_scale=kv_scale or 1.0, v_scale=kv_scale or 1.0,
                                       mode="3d" if case.startswith("decode") else "2d", block_n=64)
    t = _far_end_copy(inp, gpu_util.DEV)
    out, kernel = gpu_util.run_unified(t, inp["scale"], window=window, kv_scale=kv_scale, force=force)
    assert kernel.startswith(expect), kernel
    atol, rtol = golden_io.tolerance(dtype, kv_dtype)
    assert not torch.isnan(out).any()
    torch.testing.assert_close(out.float().cpu(), ref.float(), atol=atol, rtol=rtol)
    del t
    torch.cuda.empty_cache()


@pytest.mark.parametrize("case", ["decode", "chunked_prefill"])
def test_long_context_100k_keys(case):
    """One sequence with 100k-131k keys (6250-8192 pages): decode (64 splits), and a 128-token chunk over that context (the wide
    prefill kernel, whose block table is staged in LDS, runs the same case in tests/test_gpu_variants.py)."""
    import gpu_util

    ctx = 100_003 if case != "decode" else 131_071
    dtype = torch.bfloat16
    if case == "decode":
        query_lens, kv_lens, mode = [1], [ctx], "3d"
    else:
        query_lens, kv_lens, mode = [128, 200], [ctx + 128, 4096 + 200], "2d"
    inp = orc.make_paged_inputs(60, query_lens, kv_lens, 4, 1, 128, 16, dtype)
    ref = orc.unified_attention_oracle(inp["q"], inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"],
                                       inp["scale"], mode=mode, block_n=64)
    t = gpu_util.to_dev(inp)
    out, kernel = gpu_util.run_unified(t, inp["scale"])
    assert kernel.startswith("decode" if case == "decode" else "prefill_mfma"), kernel
    assert not torch.isnan(out).any()
    atol, rtol = golden_io.tolerance(dtype)
    torch.testing.assert_close(out.float().cpu(), ref.float(), atol=atol, rtol=rtol)
