"""-m gpu: row H of the scope table - the microbenchmark harness (tools/microbench.py, our counterpart of the
reference's scripts/benchmark.py prefix test) runs end to end: its generator builds the batch, the call is checked
against the CPU oracle, and each of the three timing modes of the reference's measure_benchmarks
(scripts/benchmark.py:1708-1750: events with a cache flush between repetitions, graph replay, end-to-end wall clock)
returns an ordered (median, p20, p80) in milliseconds."""

import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_harness_timing_modes_smoke():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gpu_util
    import microbench
    from mi355_attn import _lib
    from oracle import paged_attention_oracle as orc

    dev = gpu_util.DEV
    q_lens, ctx_lens = microbench.make_prefix_batch(8, 256, [1.0], 0.5, 0.5, "ALTERNATING", 16)
    assert (q_lens.count(1), sum(1 for c in ctx_lens if c == 0)) == (4, 2)
    inp = microbench.build_inputs(q_lens, ctx_lens, 8, 2, 128, 16, torch.bfloat16, dev, seed=0)
    out = torch.zeros_like(inp["q"])
    call = microbench.make_call(inp, out, None)
    call()
    torch.cuda.synchronize()
    assert _lib.last_kernel().startswith("prefill_mfma"), _lib.last_kernel()
    ref = orc.unified_attention_oracle(inp["q"].cpu(), inp["k_cache"].cpu(), inp["v_cache"].cpu(), inp["cu_seqlens_q"].cpu(), inp["seqused_k"].cpu(),
                                       inp["block_table"].cpu(), inp["scale"], block_n=64)
    torch.testing.assert_close(out.float().cpu(), ref.float(), atol=2e-2, rtol=2e-2)
    times = {}
    for mode in ("events", "graphs", "end2end"):
        med, p20, p80 = microbench.measure(mode, call, dev, warmup_ms=2, rep_ms=10)
        assert 0.0 < p20 <= med <= p80 < 50.0, (mode, med, p20, p80)
        times[mode] = med
    assert times["graphs"] <= times["end2end"]            # a replay excludes the launch path the wall clock includes
    # the result is still right after the graph capture and the flushes
    out.zero_()
    call()
    torch.cuda.synchronize()
    torch.testing.assert_close(out.float().cpu(), ref.float(), atol=2e-2, rtol=2e-2)
