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
