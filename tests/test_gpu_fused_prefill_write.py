"""-m gpu: the paged-cache write fused into a PREFILL launch (library 0.6.0; SURVEY.md 8f-2: the pair of calls at
LIB/backend/triton_attn.py:393-405 + :437 as one) - the short-prompt kernel (csrc/prefill_lat.hip) attends over the step's
new keys / values straight from the linear tensors and the Q block that owns a token stores its rows. The cache must hold
exactly what the separate write would have stored, nothing else in it may change, the output must match the oracle run on
the updated cache, and the slots being written are poisoned first so that a kernel reading them from the cache shows."""

import math
import types

import pytest
import torch

import golden_io
from oracle import paged_attention_oracle as orc

pytestmark = pytest.mark.gpu


def _case(seed, q_lens, kv_lens, hq, hk, d, page, dtype):
    inp = orc.make_paged_inputs(seed, q_lens, kv_lens, hq, hk, d, page, dtype)
    g = torch.Generator().manual_seed(seed + 100)
    T = sum(q_lens)
    inp["k_new"] = ((torch.rand(T, hk, d, generator=g) * 2 - 1) * 1.2).to(dtype)
    inp["v_new"] = (torch.rand(T, hk, d, generator=g) * 2 - 1).to(dtype)
    slots = []
    for i, (ql, kl) in enumerate(zip(q_lens, kv_lens)):
        for t in range(ql):
            pos = kl - ql + t
            slots.append(int(inp["block_table"][i, pos // page]) * page + pos % page)
    inp["slots"] = torch.tensor(slots, dtype=torch.int64)
    return inp


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("q_lens,kv_lens", [([500], [500]), ([512], [512]), ([129], [700]), ([37], [37]), ([200], [1000]), ([1000], [1500]),
                                            ([64, 64, 64], [64, 300, 77]), ([17], [1029])])
@pytest.mark.parametrize("hq,hk,page", [(32, 8, 16), (8, 2, 32), (6, 2, 16)])
def test_fused_prefill_write_matches_separate_write_and_oracle(dtype, q_lens, kv_lens, hq, hk, page):
    import gpu_util
    from mi355_attn import _lib
    from mi355_attn.kernels import reshape_and_cache_flash
    from mi355_attn.kernels.unified import prefill_attention_and_cache_write

    if hq == 32 and sum(q_lens) > 520 and max(kv_lens) < 640:
        pytest.skip("beyond one workgroup per CU and not one long sequence: not this kernel's step")
    dev = gpu_util.DEV
    d = 128
    inp = _case(61, q_lens, kv_lens, hq, hk, d, page, dtype)
    d_ = gpu_util.to_dev(inp)
    # reference: the separate write, then the oracle on the cache it produced
    kc_ref, vc_ref = d_["k_cache"].clone(), d_["v_cache"].clone()
    reshape_and_cache_flash(d_["k_new"], d_["v_new"], kc_ref, vc_ref, d_["slots"], "auto", None, None)
    torch.cuda.synchronize()
    ref = orc.unified_attention_oracle(inp["q"], kc_ref.cpu(), vc_ref.cpu(), inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"], inp["scale"],
                                       mode="2d", block_n=64)
    # fused: poison the slots being written so that a kernel that attends over the OLD cache contents shows
    kc, vc = d_["k_cache"].clone(), d_["v_cache"].clone()
    kc.view(-1, hk, d)[d_["slots"]] = float("nan")
    vc.view(-1, hk, d)[d_["slots"]] = float("nan")
    out = torch.full_like(d_["q"], float("nan"))
    for slot_mapping in (d_["slots"], None, d_["slots"].to(torch.int32)):
        kc2, vc2 = kc.clone(), vc.clone()
        ok = prefill_attention_and_cache_write(d_["q"], d_["k_new"], d_["v_new"], kc2, vc2, out, d_["cu_seqlens_q"], max(q_lens), d_["seqused_k"],
                                               max(kv_lens), inp["scale"], d_["block_table"], slot_mapping)
        torch.cuda.synchronize()
        assert ok, "this step should be served fused"
        assert _lib.last_kernel() == "prefill_mfma_lat", _lib.last_kernel()
        assert torch.equal(kc2.view(torch.int16), kc_ref.view(torch.int16)) and torch.equal(vc2.view(torch.int16), vc_ref.view(torch.int16))
        atol, rtol = golden_io.tolerance(dtype, None)
        assert not torch.isnan(out).any()
        torch.testing.assert_close(out.float().cpu(), ref.float(), atol=atol, rtol=rtol)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("q_lens,kv_lens,expect", [
    ([512] * 4, [512] * 4, "prefill_mfma"),                                        # several prompts: the 4-wave LDS-DMA kernel
    ([300, 40, 129], [300, 100, 129], "prefill_mfma"),                             # unequal prompts: two launches, no one-token row
    ([40, 1, 9, 1, 300], [70, 45, 33, 900, 333], "prefill_mfma"),                  # a mixed step: one-token rows on the decode kernel's fused write
    ([1, 1, 700, 1], [1500, 17, 1200, 1], "prefill_mfma"),                         # chunk over a context beside decode rows (one with a single key)
    ([1024] * 8, [1024] * 8, "prefill_mfma"),                                      # 8 x 1024: the LDS-DMA kernel at full width
])
def test_fused_write_in_steps_of_several_sequences(dtype, q_lens, kv_lens, expect):
    """Steps with several sequences: the prefill rows on an LDS-DMA kernel (or the short-prompt kernel) with the write inside,
    the one-token rows of a mixed step on the split-KV decode kernel's own fused write - ONE op, no separate cache write, the
    cache and the output as the pair of calls gives them."""
    import gpu_util
    from mi355_attn import _lib
    from mi355_attn.kernels import reshape_and_cache_flash
    from mi355_attn.kernels.unified import prefill_attention_and_cache_write

    hq, hk, d, page = 32, 8, 128, 16
    inp = _case(65, q_lens, kv_lens, hq, hk, d, page, dtype)
    d_ = gpu_util.to_dev(inp)
    kc_ref, vc_ref = d_["k_cache"].clone(), d_["v_cache"].clone()
    reshape_and_cache_flash(d_["k_new"], d_["v_new"], kc_ref, vc_ref, d_["slots"], "auto", None, None)
    torch.cuda.synchronize()
    kc, vc = d_["k_cache"].clone(), d_["v_cache"].clone()
    kc.view(-1, hk, d)[d_["slots"]] = float("nan")
    vc.view(-1, hk, d)[d_["slots"]] = float("nan")
    out = torch.full_like(d_["q"], float("nan"))
    ok = prefill_attention_and_cache_write(d_["q"], d_["k_new"], d_["v_new"], kc, vc, out, d_["cu_seqlens_q"], max(q_lens), d_["seqused_k"], max(kv_lens),
                                           inp["scale"], d_["block_table"], d_["slots"])
    torch.cuda.synchronize()
    assert ok and _lib.last_kernel().startswith(expect), _lib.last_kernel()
    assert torch.equal(kc.view(torch.int16), kc_ref.view(torch.int16)) and torch.equal(vc.view(torch.int16), vc_ref.view(torch.int16))
    assert not torch.isnan(out).any()
    atol, rtol = golden_io.tolerance(dtype, None)
    if sum(q_lens) <= 2200:
        ref = orc.unified_attention_oracle(inp["q"], kc_ref.cpu(), vc_ref.cpu(), inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"], inp["scale"],
                                           mode="2d", block_n=64)
        torch.testing.assert_close(out.float().cpu(), ref.float(), atol=atol, rtol=rtol)
    else:                                         # 8 x 1024: sampled rows
        cu = inp["cu_seqlens_q"].tolist()
        for s_ in (0, 3, 7):
            for t in (0, 511, 1023):
                row = cu[s_] + t
                ref = gpu_util.oracle_row(orc, inp["q"][row:row + 1], kc_ref.cpu(), vc_ref.cpu(), inp["block_table"][s_], kv_lens[s_] - q_lens[s_] + t + 1, inp["scale"])
                torch.testing.assert_close(out[row:row + 1].float().cpu(), ref.float(), atol=atol, rtol=rtol)


def test_negative_slots_are_not_stored_but_attended_over():
    """A token whose slot is negative (a padding token, triton_attn.py:149-151) is attended over like any other - its key comes
    from the linear tensor - and never stored."""
    import gpu_util
    from mi355_attn.kernels.unified import prefill_attention_and_cache_write

    dtype, hq, hk, d, page = torch.bfloat16, 8, 2, 128, 16
    inp = _case(62, [100], [260], hq, hk, d, page, dtype)
    d_ = gpu_util.to_dev(inp)
    slots = d_["slots"].clone()
    skip = torch.tensor([3, 50, 99], device=gpu_util.DEV)
    slots[skip] = -1
    kc, vc = d_["k_cache"].clone(), d_["v_cache"].clone()
    before_k = kc.view(-1, hk, d)[d_["slots"][skip]].clone()
    out = torch.full_like(d_["q"], float("nan"))
    assert prefill_attention_and_cache_write(d_["q"], d_["k_new"], d_["v_new"], kc, vc, out, d_["cu_seqlens_q"], 100, d_["seqused_k"], 260, inp["scale"],
                                             d_["block_table"], slots)
    torch.cuda.synchronize()
    assert torch.equal(kc.view(-1, hk, d)[d_["slots"][skip]].view(torch.int16), before_k.view(torch.int16))     # untouched
    kept = torch.ones(100, dtype=torch.bool, device=gpu_util.DEV)
    kept[skip] = False
    assert torch.equal(kc.view(-1, hk, d)[d_["slots"][kept]].view(torch.int16), d_["k_new"][kept].view(torch.int16))
    # the output is that of a cache holding ALL the new rows
    kc_full, vc_full = inp["k_cache"].clone(), inp["v_cache"].clone()
    kc_full.view(-1, hk, d)[inp["slots"]] = inp["k_new"]
    vc_full.view(-1, hk, d)[inp["slots"]] = inp["v_new"]
    ref = orc.unified_attention_oracle(inp["q"], kc_full, vc_full, inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"], inp["scale"], mode="2d", block_n=64)
    torch.testing.assert_close(out.float().cpu(), ref.float(), atol=2e-2, rtol=2e-2)


def test_steps_that_are_not_served_fused_say_so_and_the_op_falls_back():
    """A long prompt, several chunks over long contexts (the 64-rows-per-wave kernel's): `prefill_attention_and_cache_write`
    returns False and touches nothing; the registered op then issues the pair of calls - same cache, same output."""
    import gpu_util
    from mi355_attn import _lib, ops  # noqa: F401
    from mi355_attn.kernels.unified import prefill_attention_and_cache_write

    dtype, hq, hk, d, page = torch.bfloat16, 8, 2, 128, 16
    for q_lens, kv_lens in (([2100], [2100]), ([600, 700], [2000, 1900])):          # the long-prefill kernel's steps
        inp = _case(63, q_lens, kv_lens, hq, hk, d, page, dtype)
        d_ = gpu_util.to_dev(inp)
        kc, vc = d_["k_cache"].clone(), d_["v_cache"].clone()
        out = torch.full_like(d_["q"], float("nan"))
        assert not prefill_attention_and_cache_write(d_["q"], d_["k_new"], d_["v_new"], kc, vc, out, d_["cu_seqlens_q"], max(q_lens), d_["seqused_k"],
                                                     max(kv_lens), inp["scale"], d_["block_table"], d_["slots"])
        assert torch.equal(kc.view(torch.int16), d_["k_cache"].view(torch.int16)) and torch.isnan(out).all()
        torch.ops.mi355_attn.prefill_attention_and_cache_write(d_["q"], d_["k_new"], d_["v_new"], kc, vc, out, d_["cu_seqlens_q"], max(q_lens), d_["seqused_k"],
                                                               max(kv_lens), inp["scale"], d_["block_table"], d_["slots"], None, None, "auto", 0)
        torch.cuda.synchronize()
        kc_ref = inp["k_cache"].clone(); vc_ref = inp["v_cache"].clone()
        kc_ref.view(-1, hk, d)[inp["slots"]] = inp["k_new"]
        vc_ref.view(-1, hk, d)[inp["slots"]] = inp["v_new"]
        assert torch.equal(kc.cpu().view(torch.int16), kc_ref.view(torch.int16))
        ref = orc.unified_attention_oracle(inp["q"], kc_ref, vc_ref, inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"], inp["scale"], mode="2d", block_n=64)
        torch.testing.assert_close(out.float().cpu(), ref.float(), atol=2e-2, rtol=2e-2)


def test_impl_forward_takes_the_fused_prefill_write_for_a_short_prompt():
    """MI355AttentionImpl.forward on a 300-token prompt: one library call (the op's fused branch), cache and output as the pair
    of calls gives them."""
    import gpu_util
    from mi355_attn import _lib
    from mi355_attn.backend import attn

    dev = gpu_util.DEV
    dtype, hq, hk, d, page = torch.bfloat16, 32, 8, 128, 16
    inp = _case(64, [300], [300], hq, hk, d, page, dtype)
    d_ = gpu_util.to_dev(inp)
    nb = inp["k_cache"].shape[0]
    kv_cache = torch.stack([d_["k_cache"], d_["v_cache"]]).contiguous()
    impl = attn.MI355AttentionImpl(hq, d, inp["scale"], hk, None, None, "auto")
    layer = types.SimpleNamespace(_k_scale=torch.ones((), device=dev), _v_scale=torch.ones((), device=dev), _q_scale=torch.ones((), device=dev), _q_scale_float=1.0)
    md = attn.MI355AttentionMetadata(num_actual_tokens=300, max_query_len=300, avg_query_len=300, avg_seq_len=300, query_start_loc=d_["cu_seqlens_q"],
                                     max_seq_len=300, seq_lens=d_["seqused_k"], block_table=d_["block_table"], slot_mapping=d_["slots"], use_cascade=False,
                                     common_prefix_len=0, cu_prefix_query_lens=None, prefix_kv_lens=None, suffix_kv_lens=None)
    out = torch.full((300, hq * d), float("nan"), dtype=dtype, device=dev)
    impl.forward(layer, d_["q"], d_["k_new"], d_["v_new"], kv_cache, md, output=out)
    torch.cuda.synchronize()
    assert _lib.last_kernel() == "prefill_mfma_lat"
    kc_ref = inp["k_cache"].clone(); vc_ref = inp["v_cache"].clone()
    kc_ref.view(-1, hk, d)[inp["slots"]] = inp["k_new"]
    vc_ref.view(-1, hk, d)[inp["slots"]] = inp["v_new"]
    assert torch.equal(kv_cache[0].cpu().view(torch.int16), kc_ref.view(torch.int16)) and torch.equal(kv_cache[1].cpu().view(torch.int16), vc_ref.view(torch.int16))
    ref = orc.unified_attention_oracle(inp["q"], kc_ref, vc_ref, inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"], inp["scale"], mode="2d", block_n=64)
    torch.testing.assert_close(out.view(300, hq, d).float().cpu(), ref.float(), atol=2e-2, rtol=2e-2)
