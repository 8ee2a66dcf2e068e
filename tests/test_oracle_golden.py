"""Pin the CPU oracle against the golden vectors produced by the reference's own Triton kernels
(tests/golden/make_golden.py) and against an independent dense fp64 softmax. CPU only."""

import pytest
import torch

import golden_io
from oracle import paged_attention_oracle as orc

UNIFIED = [n for n in golden_io.names() if golden_io.load(n)[0]["kind"] == "unified"]


def _run_oracle(meta, t, fn=orc.unified_attention_oracle, **kw):
    return fn(
        t["q"], t["k_cache"], t["v_cache"], t["cu_seqlens_q"], t["seqused_k"], t["block_table"], meta["scale"],
        sliding_window=meta["window"], softcap=meta["softcap"], alibi_slopes=t.get("alibi_slopes"),
        k_scale=meta["kv_scale"], v_scale=meta.get("v_scale", meta["kv_scale"]), **kw,
    )


@pytest.mark.parametrize("name", UNIFIED)
def test_oracle_matches_reference_kernels(name):
    meta, t = golden_io.load(name)
    t = golden_io.widen(meta, t)          # (16-bit-representable fp32 fixtures: the reference computed in fp32)
    out = _run_oracle(meta, t, mode=meta["path"], block_n=meta["tile"][1])
    atol, rtol = golden_io.tolerance(t["q"].dtype, t["k_cache"].dtype)
    # same algorithm, same tiling -> much tighter than the stated cross-implementation tolerance
    tight = {torch.float32: 2e-6, torch.float16: 1e-3}[t["q"].dtype]
    torch.testing.assert_close(out.float(), t["out"].float(), atol=min(atol, tight), rtol=rtol)


@pytest.mark.parametrize("name", UNIFIED)
def test_oracle_matches_dense_fp64(name):
    meta, t = golden_io.load(name)
    t = golden_io.widen(meta, t)
    dense = _run_oracle(meta, t, fn=orc.dense_attention_fp64)
    for mode in ("2d", "3d"):
        out = _run_oracle(meta, t, mode=mode, block_n=t["k_cache"].shape[1])
        atol, rtol = golden_io.tolerance(t["q"].dtype, t["k_cache"].dtype)
        torch.testing.assert_close(out.double(), dense, atol=atol, rtol=rtol)


@pytest.mark.parametrize("name", golden_io.names("legacy_paged"))
def test_legacy_decode_oracle(name):
    meta, t = golden_io.load(name)
    out = orc.paged_attention_v0_oracle(t["q"], t["k_cache_v0"], t["v_cache_v0"], meta["scale"], t["block_table"],
                                        t["seqused_k"], alibi_slopes=t.get("alibi_slopes"), num_segments=meta["segments"])
    atol, rtol = golden_io.tolerance(t["q"].dtype)
    torch.testing.assert_close(out.float(), t["out"].float(), atol=atol, rtol=rtol)


@pytest.mark.parametrize("name", golden_io.names("legacy_ctxfwd"))
def test_legacy_context_fwd_oracle(name):
    meta, t = golden_io.load(name)
    out = orc.context_attention_fwd_oracle(t["q"], t["k_new"], t["v_new"], t["k_cache_v0"], t["v_cache_v0"], t["block_table"],
                                           t["cu_seqlens_q"], t["seqused_k"], sm_scale=meta["scale"],
                                           sliding_window=meta["window"], block=64 if t["q"].dtype == torch.float32 else 128)
    atol, rtol = golden_io.tolerance(t["q"].dtype)
    torch.testing.assert_close(out.float(), t["out"].float(), atol=atol, rtol=rtol)
    # decode rows (query_len == 1) are left untouched by the reference (zeros in the fixture)
    cu = t["cu_seqlens_q"].tolist()
    for i, ql in enumerate(meta["query_lens"]):
        if ql == 1:
            assert torch.count_nonzero(t["out"][cu[i]]) == 0


@pytest.mark.parametrize("name", golden_io.names("reshape_and_cache"))
def test_cache_write_oracle(name):
    meta, t = golden_io.load(name)
    kc = torch.zeros_like(t["k_cache_out"])
    vc = torch.zeros_like(t["v_cache_out"])
    orc.reshape_and_cache_flash_oracle(t["key"], t["value"], kc, vc, t["slot_mapping"])
    assert torch.equal(kc.view(torch.uint8), t["k_cache_out"].view(torch.uint8))
    assert torch.equal(vc.view(torch.uint8), t["v_cache_out"].view(torch.uint8))


def test_cache_write_oracle_skips_negative_slots_and_quantises():
    g = torch.Generator().manual_seed(0)
    key = torch.randn(5, 2, 16, generator=g) * 300
    value = torch.randn(5, 2, 16, generator=g)
    slots = torch.tensor([3, -1, 0, 17, -1])
    kc = torch.zeros(2, 16, 2, 16, dtype=torch.float8_e4m3fn)
    vc = torch.zeros_like(kc)
    orc.reshape_and_cache_flash_oracle(key, value, kc, vc, slots, k_scale=0.5, v_scale=2.0)
    assert torch.count_nonzero(kc.float()[0, 1]) == 0  # untouched
    assert kc.float().abs().max() <= 448.0             # saturated, never inf/nan
    torch.testing.assert_close(vc.float()[1, 1], (value[3] / 2.0).to(torch.float8_e4m3fn).float())


@pytest.mark.parametrize("name", golden_io.names("flash_varlen"))
def test_prefill_flash_attention_oracle(name):
    """The oracle's restatement of the reference's non-paged varlen prefill op against the outputs of the reference's own
    attn_fwd kernel (run under the Triton interpreter by tests/golden/make_golden.py::flash_cases)."""
    meta, t = golden_io.load(name)
    out = orc.prefill_flash_attention_oracle(t["q"], t["k"], t["v"], t["cu_seqlens_q"], t["cu_seqlens_k"], meta["scale"], causal=meta["causal"])
    atol, rtol = golden_io.tolerance(t["q"].dtype)
    torch.testing.assert_close(out, t["out"].float(), atol=atol, rtol=rtol)
