"""-m gpu: the HIP path (through the C ABI) against the golden vectors of the reference's Triton
kernels, for every kernel family that accepts the case."""

import pytest
import torch

import golden_io

pytestmark = pytest.mark.gpu

UNIFIED = [n for n in golden_io.names() if golden_io.load(n)[0]["kind"] == "unified"]


@pytest.mark.parametrize("force", [None, 2, 3, 9], ids=["auto", "2d", "3d", "generic"])
@pytest.mark.parametrize("name", UNIFIED)
def test_unified_attention_vs_reference_golden(name, force):
    import gpu_util

    meta, t = golden_io.load(name)
    d = gpu_util.to_dev(t)
    fp8 = t["k_cache"].dtype in (torch.float8_e4m3fn, torch.float8_e5m2)
    out, kernel = gpu_util.run_unified(d, meta["scale"], window=meta["window"], softcap=meta["softcap"],
                                       kv_scale=meta["kv_scale"] if fp8 else None, v_scale=meta.get("v_scale") if fp8 else None, force=force)
    atol, rtol = golden_io.tolerance(t["q"].dtype, t["k_cache"].dtype)
    assert not torch.isnan(out).any(), f"{kernel}: NaN in output (unwritten rows?)"
    torch.testing.assert_close(out.float().cpu(), t["out"].float(), atol=atol, rtol=rtol, msg=lambda m: f"[{kernel}] {m}")


@pytest.mark.parametrize("name", golden_io.names("reshape_and_cache"))
def test_reshape_and_cache_flash_golden(name):
    import gpu_util
    from mi355_attn.kernels import reshape_and_cache_flash

    meta, t = golden_io.load(name)
    d = gpu_util.to_dev(t)
    kc, vc = torch.zeros_like(d["k_cache_out"]), torch.zeros_like(d["v_cache_out"])
    reshape_and_cache_flash(d["key"], d["value"], kc, vc, d["slot_mapping"], "auto", None, None)
    torch.cuda.synchronize()
    assert torch.equal(kc.cpu().view(torch.uint8), t["k_cache_out"].view(torch.uint8))
    assert torch.equal(vc.cpu().view(torch.uint8), t["v_cache_out"].view(torch.uint8))


@pytest.mark.parametrize("name", golden_io.names("long_chunk"))
def test_64_rows_per_wave_prefill_kernel_vs_the_reference_2d_kernel(name):
    """`prefill_pw_kernel` (16-bit only, chosen from 2048 keys on) against the reference's own 2D kernel: the fixture is
    the reference's fp32 run (BLOCK_M = BLOCK_N = 64) on inputs a 16-bit type holds exactly - bf16 does not run under the
    Triton interpreter - a 256-token chunk over a 2048-token context. `num_segments = 1` asks for the
    single pass over the key range (the reference's 2D kernel); the auto plan would deal this small grid's keys to
    several workgroups (covered by the parametrised test above)."""
    import gpu_util
    from mi355_attn import _lib
    from mi355_attn.kernels.unified import fill_attn_params, launch

    meta, t = golden_io.load(name)
    assert t["q"].dtype in (torch.bfloat16, torch.float16) and t["out"].dtype == torch.float32
    d = gpu_util.to_dev(t)
    out = torch.full_like(d["q"], float("nan"))
    ql = meta["query_lens"]
    window = int(meta.get("window", 0))                      # (`long_chunk_sw*`: the sliding-window instantiation, reference :474-479)
    p, keep = fill_attn_params(d["q"], d["k_cache"], d["v_cache"], out, d["cu_seqlens_q"], max(ql), d["seqused_k"], max(meta["kv_lens"]), meta["scale"],
                               (window - 1, 0) if window else (-1, -1), d["block_table"], 0.0, None, None, None, None, num_segments=1)
    launch(p, gpu_util.DEV)
    torch.cuda.synchronize()
    import os
    pinned = os.environ.get("MI355_PREFILL", "pw")          # (tests/test_gpu_variants.py pins other prefill kernels on this file)
    expect = ("prefill_mfma_pw_sw" if window else "prefill_mfma_pw") if pinned == "pw" else ("prefill_mfma_feat" if window else "prefill_mfma")
    assert _lib.last_kernel() == expect, _lib.last_kernel()
    atol, rtol = golden_io.tolerance(t["q"].dtype)
    torch.testing.assert_close(out.float().cpu(), t["out"], atol=atol, rtol=rtol)
