"""-m gpu: the optional second output (log-sum-exp per row, `softmax_lse`) of every kernel family against a dense
float64 computation, and its use: partial results over disjoint key ranges of one long sequence - what ranks of a
context-parallel group would each compute - merged with `parallel.merge_partial_attention` equal the full result."""

import math

import pytest
import torch

import golden_io
from oracle import paged_attention_oracle as orc

pytestmark = pytest.mark.gpu


def _run(t, scale, *, window=0, softcap=0.0, kv_scale=None, force=None):
    from mi355_attn import _lib
    from mi355_attn.kernels import unified_attention

    q = t["q"]
    out = torch.full_like(q, float("nan"))
    lse = torch.full((q.shape[0], q.shape[1]), float("nan"), dtype=torch.float32, device=q.device)
    ql = t["cu_seqlens_q"][1:] - t["cu_seqlens_q"][:-1]
    ks = None if kv_scale is None else torch.tensor([kv_scale], dtype=torch.float32, device=q.device)
    unified_attention(q=q, k=t["k_cache"], v=t["v_cache"], out=out, cu_seqlens_q=t["cu_seqlens_q"], max_seqlen_q=int(ql.max()),
                      seqused_k=t["seqused_k"], max_seqlen_k=max(int(t["seqused_k"].max()), 1), avg_seqlen_q=0.0, avg_seqlen_k=0.0,
                      softmax_scale=scale, causal=True, window_size=(window - 1, 0) if window else (-1, -1), block_table=t["block_table"],
                      softcap=softcap, q_descale=None, k_descale=ks, v_descale=ks, alibi_slopes=t.get("alibi_slopes"),
                      force_selection=force, softmax_lse=lse)
    torch.cuda.synchronize()
    return out, lse, _lib.last_kernel()


CASES = {
    "decode_split": dict(q=[1] * 5, kv=[700, 33, 1023, 257, 1], expect="decode_splitkv"),
    "decode_fused_merge": dict(q=[1] * 64, kv=[600] * 64, expect="decode_splitkv", hq=8, hk=2),
    "decode_single": dict(q=[1] * 3, kv=[20, 5, 32], expect="decode_single"),
    "decode_fp8": dict(q=[1] * 4, kv=[700, 33, 300, 17], expect="decode_splitkv_fp8", kv_dtype=torch.float8_e4m3fn),
    "prefill": dict(q=[129, 64, 200, 5], kv=[129, 257, 777, 5], expect="prefill_mfma"),
    "prefill_feat": dict(q=[129, 64, 200, 5], kv=[129, 257, 777, 5], expect="prefill_mfma_feat", window=100, softcap=30.0),
    "prefill_fp8": dict(q=[129, 64, 200, 5], kv=[129, 257, 777, 5], expect="prefill_mfma_fp8", kv_dtype=torch.float8_e4m3fn),
    "mixed": dict(q=[1, 64, 1, 200, 1], kv=[900, 64, 17, 333, 1], expect="prefill_mfma+decode"),
    "generic": dict(q=[1, 40, 9], kv=[70, 45, 33], expect="generic", force=9),
}


@pytest.mark.parametrize("name", list(CASES))
def test_lse_of_every_kernel_family(name):
    import gpu_util

    c = CASES[name]
    dtype = torch.bfloat16
    kv_dtype = c.get("kv_dtype")
    kv_scale = 0.5 if kv_dtype is not None else None
    kw = dict(kv_dtype=kv_dtype, kv_scale=kv_scale) if kv_dtype is not None else {}
    inp = orc.make_paged_inputs(80, c["q"], c["kv"], c.get("hq", 8), c.get("hk", 2), 128, 16, dtype, **kw)
    ref, ref_lse = orc.dense_attention_fp64(inp["q"], inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"],
                                            inp["scale"], sliding_window=c.get("window", 0), softcap=c.get("softcap", 0.0),
                                            k_scale=kv_scale or 1.0, v_scale=kv_scale or 1.0, return_lse=True)
    t = gpu_util.to_dev(inp)
    out, lse, kernel = _run(t, inp["scale"], window=c.get("window", 0), softcap=c.get("softcap", 0.0), kv_scale=kv_scale, force=c.get("force"))
    # (prefill_mfma_pw[_sw][_sc]: the 64-rows-per-wave kernel and its window / soft-cap forms, when a variant test pins it)
    assert kernel.replace("_pw_sw_sc", "_feat").replace("_pw", "").replace("_lat", "").startswith(c["expect"]), kernel
    atol, rtol = golden_io.tolerance(dtype, kv_dtype)
    torch.testing.assert_close(out.double().cpu(), ref, atol=atol, rtol=rtol)
    assert not torch.isnan(lse).any()
    # scores are O(10) with 16-bit inputs: the lse is accurate to ~1e-2 absolute (fp8 K: the dequantised K is rounded to bf16)
    torch.testing.assert_close(lse.double().cpu(), ref_lse, atol=2e-2 if kv_dtype is None else 4e-2, rtol=0)


def test_key_ranges_of_one_long_sequence_merge_to_the_full_result():
    """Four 'ranks' each attend one page-aligned quarter of a 5000-key sequence (decode), one of a second, short sequence
    gets nothing: merged partials == the single-GPU result (tolerance of one bf16 rounding of the partial outputs)."""
    import gpu_util
    from mi355_attn import parallel

    dtype, page, world = torch.bfloat16, 16, 4
    kv_lens = [5000, 40]
    inp = orc.make_paged_inputs(81, [1, 1], kv_lens, 8, 2, 128, page, dtype)
    t = gpu_util.to_dev(inp)
    full_out, full_lse, kernel = _run(t, inp["scale"])
    assert kernel.startswith("decode")
    outs, lses = [], []
    for r in range(world):
        ranges = [parallel.split_key_range(n, page, world)[r] for n in kv_lens]
        loc = dict(t)
        width = max(max((k1 - k0 + page - 1) // page for k0, k1 in ranges), 1)
        bt = torch.zeros((len(kv_lens), width), dtype=torch.int32, device=gpu_util.DEV)
        for i, (k0, k1) in enumerate(ranges):
            n_pages = (k1 - k0 + page - 1) // page
            bt[i, :n_pages] = t["block_table"][i, k0 // page: k0 // page + n_pages]
        loc["block_table"] = bt
        loc["seqused_k"] = torch.tensor([k1 - k0 for k0, k1 in ranges], dtype=torch.int32, device=gpu_util.DEV)
        o, l, _ = _run(loc, inp["scale"])
        outs.append(o)
        lses.append(l)
    assert torch.isinf(lses[3][1]).all() and (outs[3][1] == 0).all()          # rank 3 holds no key of the short sequence
    merged, merged_lse = parallel.merge_partial_attention(torch.stack(outs), torch.stack(lses))
    torch.testing.assert_close(merged, full_out.float(), atol=1e-2, rtol=1e-2)
    torch.testing.assert_close(merged_lse, full_lse, atol=1e-3, rtol=0)
    ref = orc.dense_attention_fp64(inp["q"], inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"], inp["scale"])
    torch.testing.assert_close(merged.double().cpu(), ref, atol=2e-2, rtol=2e-2)
    # the same merge as ONE launch of the library's kernel on the partials in their own 16-bit type (what the exchange
    # step of parallel.all_gather_and_merge runs on a GPU; mi355_merge_attention_partials): output in the query type
    dev_out, dev_lse = parallel.merge_partial_attention_device(torch.stack(outs), torch.stack(lses))
    torch.cuda.synchronize()
    assert dev_out.dtype == dtype
    torch.testing.assert_close(dev_out.float(), merged, atol=8e-3, rtol=8e-3)
    torch.testing.assert_close(dev_lse, merged_lse, atol=1e-4, rtol=1e-5)
    assert torch.isinf(dev_lse).sum() == torch.isinf(merged_lse).sum()
