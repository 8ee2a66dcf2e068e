"""-m gpu: the short-prompt (latency) prefill kernel, `prefill_lat_kernel` (csrc/prefill_lat.hip), through the C ABI against
the CPU oracle (reference: kernel_unified_attention_2d, LIB/kernels/triton_unified_attention.py:275-523) at the shapes the
reference's latency protocol lives on (scripts/bench_vllm_latency_range.py:48-50: batch 1, ~500 input tokens) and around
them: ragged batches, chunked prefill over a context, GQA group sizes that do not divide the 64-row Q block, pages, both
16-bit types, non-causal, lse, mixed steps whose decode rows ride the split-KV kernel."""

import math

import pytest
import torch

import golden_io
from oracle import paged_attention_oracle as orc

pytestmark = pytest.mark.gpu


def _check(inp, dtype, *, expect="prefill_mfma_lat", lse=False):
    import gpu_util

    ref = orc.unified_attention_oracle(inp["q"], inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"],
                                       inp["scale"], mode="2d", block_n=64)
    d = gpu_util.to_dev(inp)
    lse_t = torch.full((inp["q"].shape[0], inp["q"].shape[1]), float("nan"), dtype=torch.float32, device=gpu_util.DEV) if lse else None
    out, kernel = gpu_util.run_unified(d, inp["scale"], lse=lse_t)
    assert kernel.startswith(expect), kernel
    atol, rtol = golden_io.tolerance(dtype, None)
    assert not torch.isnan(out).any()
    torch.testing.assert_close(out.float().cpu(), ref.float(), atol=atol, rtol=rtol)
    return out, lse_t


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("tokens", [512, 500, 1024, 64, 17])
def test_one_short_prompt_at_the_llama_shape(dtype, tokens):
    """Batch 1, Hq 32 / Hk 8 / D 128, 16-token pages: 1 x 512 and 1 x 1024 are the shapes of VERDICT r03 item 1; 500 is the
    reference protocol's input length; 17 leaves most of the only Q block's rows empty."""
    inp = orc.make_paged_inputs(300 + tokens, [tokens], [tokens], 32, 8, 128, 16, dtype)
    _check(inp, dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hq,hk", [(8, 2), (4, 4), (8, 1), (6, 2), (10, 2), (32, 1), (64, 1)])
def test_ragged_batches_and_group_sizes(dtype, hq, hk):
    """Prefill-only ragged batches (no one-token rows: those go to the decode kernel, below) with contexts (chunked
    prefill), G = 1 .. 64: G = 3 and 5 leave padding rows in every Q block, G = 64 makes a Q block one token."""
    query_lens = [5, 129, 64, 33, 200, 2]
    kv_lens = [5, 129, 257, 100, 777, 1500]
    inp = orc.make_paged_inputs(321, query_lens, kv_lens, hq, hk, 128, 16, dtype)
    _check(inp, dtype, expect="prefill_mfma" if hq // hk == 64 else "prefill_mfma_lat")     # (G = 64: 439 Q blocks of one token, beyond what the dispatch gives this kernel)
    if hq // hk == 64:                                                                          # ... so one sequence of that shape, which it takes
        one = orc.make_paged_inputs(322, [200], [777], hq, hk, 128, 16, dtype)
        _check(one, dtype)


@pytest.mark.parametrize("page", [16, 32, 128])
def test_page_sizes(page):
    inp = orc.make_paged_inputs(322, [70, 3, 300], [70, 513, 411], 8, 2, 128, page, torch.bfloat16)
    _check(inp, torch.bfloat16)


@pytest.mark.parametrize("batch,expect", [(2, "prefill_mfma"), (8, "prefill_mfma")])
def test_several_prompts_of_512_tokens(batch, expect):
    """2 x 512 and 8 x 512 (the third shape of VERDICT r03 item 1): several short prompts are 128-row Q blocks' work - the
    dispatch keeps them on the 4-wave kernel, which round 4 gave this kernel's LDS-DMA issue (launch_prefill) - sampled
    rows of every sequence against the oracle."""
    import gpu_util

    lens = [512] * batch
    inp = orc.make_paged_inputs(323, lens, lens, 32, 8, 128, 16, torch.bfloat16)
    d = gpu_util.to_dev(inp)
    out, kernel = gpu_util.run_unified(d, inp["scale"])
    assert kernel == expect, kernel
    assert not torch.isnan(out).any()
    for s in range(batch):
        for t in (0, 1, 63, 64, 300, 511):
            row = s * 512 + t
            ref = gpu_util.oracle_row(orc, inp["q"][row:row + 1], inp["k_cache"], inp["v_cache"], inp["block_table"][s], t + 1, inp["scale"])
            torch.testing.assert_close(out[row:row + 1].float().cpu(), ref.float(), atol=2e-2, rtol=2e-2)


def test_mixed_step_prefill_rows_here_decode_rows_on_the_split_kv_kernel():
    query_lens = [1, 7, 1, 40, 1, 300]
    kv_lens = [900, 70, 17, 70, 1, 333]
    inp = orc.make_paged_inputs(324, query_lens, kv_lens, 8, 2, 128, 16, torch.bfloat16)
    _check(inp, torch.bfloat16, expect="prefill_mfma_lat+decode")


def test_lse_of_the_latency_kernel_against_float64():
    import gpu_util

    query_lens, kv_lens = [40, 9, 130], [70, 33, 400]
    inp = orc.make_paged_inputs(325, query_lens, kv_lens, 8, 2, 128, 16, torch.float16)
    out, lse = _check(inp, torch.float16, lse=True)
    G = 4
    cu = inp["cu_seqlens_q"].tolist()
    for s, (ql, kl) in enumerate(zip(query_lens, kv_lens)):
        pages = inp["block_table"][s, : (kl + 15) // 16].long()
        k = inp["k_cache"][pages].reshape(-1, 2, 128)[:kl].double()
        for t in (0, ql - 1):
            n = kl - ql + t + 1
            for h in (0, 5):
                sc = inp["scale"] * (k[:n, h // G] @ inp["q"][cu[s] + t, h].double())
                want = torch.logsumexp(sc, 0).item()
                assert abs(lse[cu[s] + t, h].item() - want) < 2e-3, (s, t, h)


def test_non_causal_short_sequences_on_the_matrix_core_kernels():
    """prefill_flash_attention(causal=False) over the two-range form (a scratch cache + the unified launch): every row sees
    its sequence's whole key range. The dispatch sends non-causal calls to prefill_pw_kernel where that applies; this
    kernel serves `non_causal` too (a pinned build, MI355_PREFILL=lat, runs this test on it)."""
    from mi355_attn import _lib
    from mi355_attn.kernels import prefill_flash_attention

    hq, hk, d = 8, 2, 128
    g = torch.Generator().manual_seed(326)
    lens = [129, 64, 200, 17]
    cu = [0] + torch.tensor(lens).cumsum(0).tolist()
    q = (torch.rand(cu[-1], hq, d, generator=g) * 2 - 1).to(torch.bfloat16)
    k = (torch.rand(cu[-1], hk, d, generator=g) * 2 - 1).to(torch.bfloat16)
    v = (torch.rand(cu[-1], hk, d, generator=g) * 2 - 1).to(torch.bfloat16)
    scale = 1.0 / math.sqrt(d)
    ref = torch.zeros(q.shape, dtype=torch.float64)
    for i in range(len(lens)):
        a, b = cu[i], cu[i + 1]
        for h in range(hq):
            s = scale * (q[a:b, h].double() @ k[a:b, h // (hq // hk)].double().T)
            ref[a:b, h] = torch.softmax(s, dim=-1) @ v[a:b, h // (hq // hk)].double()
    dev = torch.device("cuda:0")
    cud = torch.tensor(cu, dtype=torch.int32, device=dev)
    out = prefill_flash_attention(q.to(dev), k.to(dev), v.to(dev), max(lens), max(lens), cud, cud.clone(), causal=False, sm_scale=scale)
    torch.cuda.synchronize()
    assert _lib.last_kernel().startswith("prefill_mfma"), _lib.last_kernel()
    torch.testing.assert_close(out.double().cpu(), ref, atol=2e-2, rtol=2e-2)


def test_rows_whose_scores_rise_late_and_rows_far_from_zero():
    """A true running maximum per row: a needle key deep in the context scoring +30 nats over the rest, and rows whose
    scores all sit far from zero (a query aligned with an attention sink), cost this kernel nothing special."""
    import gpu_util

    tokens, hq, hk, d = 512, 8, 2, 128
    for dtype in (torch.bfloat16, torch.float16):
        inp = orc.make_paged_inputs(327, [tokens], [tokens], hq, hk, d, 16, dtype)
        kc = inp["k_cache"].float()
        page, slot = int(inp["block_table"][0, 300 // 16]), 300 % 16
        direction = torch.nn.functional.normalize(torch.randn(d, generator=torch.Generator().manual_seed(5)), dim=0)
        kc[page, slot, :, :] = direction * 8.0                      # the needle key at position 300
        inp["k_cache"] = kc.to(dtype)
        q = inp["q"].float()
        q[:, 0, :] = direction * (30.0 / inp["scale"] / 8.0)        # head 0: every row scores +30 nats on the needle
        q[:, 3, :] += direction * 40.0                              # head 3: a strong component along it as well
        inp["q"] = q.to(dtype)
        _check(inp, dtype)
