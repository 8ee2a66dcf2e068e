"""-m gpu: legacy op signatures (v0 cache layout) against the reference-kernel goldens, and the
vLLM-facing Impl.forward (cache write + attention) against the oracle."""

import types

import pytest
import torch

import golden_io
from oracle import paged_attention_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", golden_io.names("legacy_paged"))
def test_legacy_paged_attention_golden(name):
    import gpu_util
    from mi355_attn import _lib
    from mi355_attn.kernels.legacy import paged_attention_2d, paged_attention_3d

    meta, t = golden_io.load(name)
    d = gpu_util.to_dev(t)
    S, Hq, D = t["q"].shape
    Hk = t["v_cache_v0"].shape[1]
    out = torch.full_like(d["q"], float("nan"))
    one = torch.ones(1, dtype=torch.float32, device=gpu_util.DEV)
    fn = paged_attention_3d if meta["segments"] else paged_attention_2d
    fn(out, d["q"], d["k_cache_v0"], d["v_cache_v0"], meta["scale"], one, one, "auto", d["block_table"], d["seqused_k"],
       d.get("alibi_slopes"), t["v_cache_v0"].shape[3], S, Hq, Hq // Hk, D)
    torch.cuda.synchronize()
    atol, rtol = golden_io.tolerance(t["q"].dtype)
    torch.testing.assert_close(out.float().cpu(), t["out"].float(), atol=atol, rtol=rtol)
    # 16-bit 5-D caches run on the MFMA decode kernel reading that layout, other 16-bit caches (4-D keys) on the same
    # kernel behind the repack pass, fp32 goldens on the generic one
    sixteen_bit = t["q"].dtype in (torch.bfloat16, torch.float16)
    fast = sixteen_bit and t["k_cache_v0"].dim() == 5 and t["k_cache_v0"].shape[4] == 8 and D in (64, 128, 256)
    name = _lib.last_kernel()
    assert name.endswith("_v0") if fast else name.startswith("repack+decode") if sixteen_bit else name == "generic", name


def _flash_to_v0(k, v):
    """flash [nb, page, Hk, D] -> legacy K [nb, Hk, D/8, page, 8], V [nb, Hk, D, page]"""
    nb, page, hk, d = k.shape
    k0 = k.view(nb, page, hk, d // 8, 8).permute(0, 2, 3, 1, 4).contiguous()
    v0 = v.permute(0, 2, 3, 1).contiguous()
    return k0, v0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hq,hk,d,page", [(32, 8, 128, 16), (8, 8, 64, 16), (16, 1, 128, 32), (40, 8, 128, 16), (8, 2, 256, 16), (6, 2, 64, 32)])
def test_legacy_paged_decode_on_the_mfma_kernel(dtype, hq, hk, d, page):
    """paged_attention_2d/3d over a 16-bit v0-layout cache: the split-KV MFMA decode kernel reads that layout directly."""
    import gpu_util
    from mi355_attn import _lib
    from mi355_attn.kernels.legacy import paged_attention_2d, paged_attention_3d

    kv_lens = [1, 15, 16, 17, 33, 257, 1023, 700]
    inp = orc.make_paged_inputs(90 + hq + d, [1] * len(kv_lens), kv_lens, hq, hk, d, page, dtype)
    alibi = torch.tensor([2.0 ** (-(i % 8 + 1)) for i in range(hq)], dtype=torch.float32)
    dev = gpu_util.DEV
    k0, v0 = _flash_to_v0(inp["k_cache"], inp["v_cache"])
    one = torch.ones(1, dtype=torch.float32, device=dev)
    for fn, slopes in ((paged_attention_2d, None), (paged_attention_3d, alibi)):
        ref = orc.unified_attention_oracle(inp["q"], inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"],
                                           inp["scale"], alibi_slopes=slopes, mode="3d")
        out = torch.full_like(inp["q"].to(dev), float("nan"))
        fn(out, inp["q"].to(dev), k0.to(dev), v0.to(dev), inp["scale"], one, one, "auto", inp["block_table"].to(dev), inp["seqused_k"].to(dev),
           None if slopes is None else slopes.to(dev), page, len(kv_lens), hq, hq // hk, d)
        torch.cuda.synchronize()
        assert _lib.last_kernel() in ("decode_splitkv_v0", "decode_single_v0"), _lib.last_kernel()
        assert not torch.isnan(out).any()
        atol, rtol = golden_io.tolerance(dtype)
        torch.testing.assert_close(out.float().cpu(), ref.float(), atol=atol, rtol=rtol)


def test_legacy_paged_decode_ignores_stale_nan_slots():
    import gpu_util
    from mi355_attn.kernels.legacy import paged_attention_2d

    kv_lens, page = [33, 100, 5, 1000], 16
    inp = orc.make_paged_inputs(95, [1] * len(kv_lens), kv_lens, 8, 2, 128, page, torch.bfloat16)
    ref = orc.unified_attention_oracle(inp["q"], inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"],
                                       inp["scale"], mode="3d")
    used = torch.zeros(inp["k_cache"].shape[:2], dtype=torch.bool)
    for i, n in enumerate(kv_lens):
        for j in range(n):
            used[inp["block_table"][i, j // page], j % page] = True
    inp["k_cache"][~used] = float("nan")
    inp["v_cache"][~used] = float("nan")
    dev = gpu_util.DEV
    k0, v0 = _flash_to_v0(inp["k_cache"], inp["v_cache"])
    one = torch.ones(1, dtype=torch.float32, device=dev)
    out = torch.full_like(inp["q"].to(dev), float("nan"))
    paged_attention_2d(out, inp["q"].to(dev), k0.to(dev), v0.to(dev), inp["scale"], one, one, "auto", inp["block_table"].to(dev),
                       inp["seqused_k"].to(dev), None, page, len(kv_lens), 8, 4, 128)
    torch.cuda.synchronize()
    assert not torch.isnan(out).any()
    torch.testing.assert_close(out.float().cpu(), ref.float(), atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("name", golden_io.names("legacy_ctxfwd"))
def test_legacy_context_attention_fwd_golden(name):
    import gpu_util
    from mi355_attn import _lib
    from mi355_attn.kernels.legacy import chunked_prefill_paged_decode, context_attention_fwd

    meta, t = golden_io.load(name)
    d = gpu_util.to_dev(t)
    one = torch.ones(1, dtype=torch.float32, device=gpu_util.DEV)
    out = torch.zeros_like(d["q"])
    context_attention_fwd(d["q"], d["k_new"], d["v_new"], out, "auto", d["k_cache_v0"], d["v_cache_v0"], d["block_table"], d["cu_seqlens_q"],
                          d["seqused_k"], max(meta["query_lens"]), one, one, sliding_window=meta["window"] or None)
    torch.cuda.synchronize()
    atol, rtol = golden_io.tolerance(t["q"].dtype)
    torch.testing.assert_close(out.float().cpu(), t["out"].float(), atol=atol, rtol=rtol)   # decode rows stay zero, as in the reference
    sixteen_bit = t["q"].dtype in (torch.bfloat16, torch.float16)
    assert _lib.last_kernel().startswith("repack+prefill") == sixteen_bit, _lib.last_kernel()   # fp32: generic kernel
    # chunked_prefill_paged_decode = the same prefill rows + the decode rows from the cache. The cache
    # must then hold the new tokens too: write them with our cache op into a flash-layout copy.
    kf, vf = orc.v0_to_flash(t["k_cache_v0"], t["v_cache_v0"])
    cu = t["cu_seqlens_q"].tolist()
    slots = []
    for i, (ql, cl) in enumerate(zip(meta["query_lens"], meta["ctx_lens"])):
        for j in range(cl, cl + ql):
            slots.append(int(t["block_table"][i, j // 16]) * 16 + j % 16)
    orc.reshape_and_cache_flash_oracle(t["k_new"], t["v_new"], kf, vf, torch.tensor(slots))
    ref = orc.unified_attention_oracle(t["q"], kf, vf, t["cu_seqlens_q"], t["seqused_k"], t["block_table"], meta["scale"],
                                       sliding_window=meta["window"])
    x = t["k_cache_v0"].shape[4]
    nb, page, hk, dd = kf.shape
    k0 = kf.view(nb, page, hk, dd // x, x).permute(0, 2, 3, 1, 4).contiguous().to(gpu_util.DEV)
    v0 = vf.permute(0, 2, 3, 1).contiguous().to(gpu_util.DEV)
    out2 = torch.full_like(d["q"], float("nan"))
    chunked_prefill_paged_decode(d["q"], d["k_new"], d["v_new"], out2, "auto", k0, v0, d["block_table"], d["cu_seqlens_q"], d["seqused_k"],
                                 max(meta["query_lens"]), one, one, None, meta["window"] or None, meta["scale"])
    torch.cuda.synchronize()
    torch.testing.assert_close(out2.float().cpu(), ref.float(), atol=atol, rtol=rtol)
    if sixteen_bit:   # prefill rows repacked, decode rows straight from the v0 cache
        assert _lib.last_kernel().startswith("repack+prefill") and _lib.last_kernel().endswith("_v0"), _lib.last_kernel()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("D,x", [(128, 8), (64, 8), (80, 8), (128, 1), (256, 8)])
@pytest.mark.parametrize("feat", ["plain", "alibi", "window"])
def test_legacy_prefill_ops_repacked_match_generic_kernel(dtype, D, x, feat):
    """context_attention_fwd / chunked_prefill_paged_decode over v0 caches (5-D x=8 and 4-D) run on the repack +
    matrix-core path; same call forced onto the shape-agnostic kernel is the reference (itself pinned by the goldens
    above). Unused cache slots are NaN, rows of query_len == 1 must stay untouched in context_attention_fwd."""
    import gpu_util
    from mi355_attn import _lib
    from mi355_attn.kernels.legacy import chunked_prefill_paged_decode, context_attention_fwd
    from mi355_attn.kernels.unified import fill_attn_params, launch

    if feat != "plain" and (D, x) not in ((128, 8), (80, 8)):
        pytest.skip("feature variants on two shapes only")
    dev = gpu_util.DEV
    Hq, Hk, page = 8, 2, 16
    query_lens, ctx_lens = [70, 1, 300, 128, 1, 33], [0, 100, 37, 1000, 17, 16]
    kv_lens = [a + b for a, b in zip(query_lens, ctx_lens)]
    inp = orc.make_paged_inputs(77, query_lens, kv_lens, Hq, Hk, D, page, dtype)
    T = sum(query_lens)
    g = torch.Generator().manual_seed(78)
    k_new = (torch.rand(T, Hk, D, generator=g) * 2 - 1).to(dtype).to(dev)
    v_new = (torch.rand(T, Hk, D, generator=g) * 2 - 1).to(dtype).to(dev)
    used = torch.zeros(inp["k_cache"].shape[:2], dtype=torch.bool)
    for i, n in enumerate(kv_lens):
        for j in range(n):
            used[inp["block_table"][i, j // page], j % page] = True
    inp["k_cache"][~used] = float("nan")
    inp["v_cache"][~used] = float("nan")
    nb = inp["k_cache"].shape[0]
    k0 = inp["k_cache"].view(nb, page, Hk, D // x, x).permute(0, 2, 3, 1, 4).contiguous()
    if x == 1:
        k0 = k0.view(nb, Hk, D, page)
    v0 = inp["v_cache"].permute(0, 2, 3, 1).contiguous()
    k0, v0 = k0.to(dev), v0.to(dev)
    q, bt, cu, sl = inp["q"].to(dev), inp["block_table"].to(dev), inp["cu_seqlens_q"].to(dev), inp["seqused_k"].to(dev)
    one = torch.ones(1, dtype=torch.float32, device=dev)
    slopes = torch.tensor([0.5 ** (i + 1) for i in range(Hq)], dtype=torch.float32, device=dev) if feat == "alibi" else None
    window = 48 if feat == "window" else None
    atol, rtol = golden_io.tolerance(dtype)
    bound = bt.shape[1] * page + max(query_lens)

    def generic(skip_decodes):
        ref = torch.full_like(q, 7.0)
        p, keep = fill_attn_params(q, k0, v0, ref, cu, max(query_lens), sl, bound, inp["scale"], (window - 1, 0) if window else (-1, -1), bt,
                                   0.0, one, one, slopes, 9, k_new=k_new, v_new=v_new, skip_decodes=skip_decodes, legacy_v0_layout=True)
        launch(p, dev)
        torch.cuda.synchronize()
        assert _lib.last_kernel() == "generic"
        return ref

    out = torch.full_like(q, 7.0)
    context_attention_fwd(q, k_new, v_new, out, "auto", k0, v0, bt, cu, sl, max(query_lens), one, one, alibi_slopes=slopes,
                          sliding_window=window, sm_scale=inp["scale"])
    torch.cuda.synchronize()
    assert _lib.last_kernel().startswith("repack+prefill"), _lib.last_kernel()
    ref = generic(True)
    assert not torch.isnan(out).any()
    for i, ql in enumerate(query_lens):
        if ql == 1:
            assert (out[int(inp["cu_seqlens_q"][i])] == 7.0).all()
    torch.testing.assert_close(out.float(), ref.float(), atol=atol, rtol=rtol)

    out2 = torch.full_like(q, float("nan"))
    chunked_prefill_paged_decode(q, k_new, v_new, out2, "auto", k0, v0, bt, cu, sl, max(query_lens), one, one, slopes, window, inp["scale"])
    torch.cuda.synchronize()
    name = _lib.last_kernel()
    assert name.startswith("repack+prefill"), name
    if x == 8 and D in (64, 128, 256):
        assert name.endswith("_v0"), name        # decode rows straight from the caller's cache
    ref2 = generic(False)
    torch.testing.assert_close(out2.float(), ref2.float(), atol=atol, rtol=rtol)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("kv_cache_dtype", ["fp8", "fp8_e5m2"])
def test_legacy_prefill_ops_fp8_cache_repacked_match_generic_kernel(dtype, kv_cache_dtype):
    """fp8 v0 caches (uint8 tensors + kv_cache_dtype, x = 16): the repack pass dequantises the context the way the
    reference's kernels do on load, new keys stay 16-bit. Reference: the same call on the shape-agnostic kernel."""
    import gpu_util
    from mi355_attn import _lib
    from mi355_attn.kernels.legacy import chunked_prefill_paged_decode, context_attention_fwd
    from mi355_attn.kernels.unified import fill_attn_params, launch

    dev = gpu_util.DEV
    Hq, Hk, D, page, x = 8, 2, 128, 16, 16
    f8 = torch.float8_e4m3fn if kv_cache_dtype == "fp8" else torch.float8_e5m2
    query_lens, ctx_lens = [70, 1, 300, 1, 33], [0, 100, 37, 17, 16]
    kv_lens = [a + b for a, b in zip(query_lens, ctx_lens)]
    inp = orc.make_paged_inputs(81, query_lens, kv_lens, Hq, Hk, D, page, dtype)
    T = sum(query_lens)
    g = torch.Generator().manual_seed(82)
    k_new = (torch.rand(T, Hk, D, generator=g) * 2 - 1).to(dtype).to(dev)
    v_new = (torch.rand(T, Hk, D, generator=g) * 2 - 1).to(dtype).to(dev)
    nb = inp["k_cache"].shape[0]
    ks, vs = 0.5, 0.25
    k8 = (inp["k_cache"].float() / ks).to(f8)
    v8 = (inp["v_cache"].float() / vs).to(f8)
    k0 = k8.view(torch.uint8).view(nb, page, Hk, D // x, x).permute(0, 2, 3, 1, 4).contiguous().to(dev)
    v0 = v8.view(torch.uint8).permute(0, 2, 3, 1).contiguous().to(dev)
    q, bt, cu, sl = inp["q"].to(dev), inp["block_table"].to(dev), inp["cu_seqlens_q"].to(dev), inp["seqused_k"].to(dev)
    k_scale = torch.tensor([ks], dtype=torch.float32, device=dev)
    v_scale = torch.tensor([vs], dtype=torch.float32, device=dev)
    atol, rtol = golden_io.tolerance(dtype, f8)
    bound = bt.shape[1] * page + max(query_lens)

    def generic(skip_decodes):
        ref = torch.full_like(q, 7.0)
        p, keep = fill_attn_params(q, k0.view(f8), v0.view(f8), ref, cu, max(query_lens), sl, bound, inp["scale"], (-1, -1), bt, 0.0, k_scale, v_scale,
                                   None, 9, k_new=k_new, v_new=v_new, skip_decodes=skip_decodes, legacy_v0_layout=True)
        launch(p, dev)
        torch.cuda.synchronize()
        assert _lib.last_kernel() == "generic"
        return ref

    out = torch.full_like(q, 7.0)
    context_attention_fwd(q, k_new, v_new, out, kv_cache_dtype, k0, v0, bt, cu, sl, max(query_lens), k_scale, v_scale, sm_scale=inp["scale"])
    torch.cuda.synchronize()
    assert _lib.last_kernel().startswith("repack+prefill"), _lib.last_kernel()
    torch.testing.assert_close(out.float(), generic(True).float(), atol=atol, rtol=rtol)
    out2 = torch.full_like(q, float("nan"))
    chunked_prefill_paged_decode(q, k_new, v_new, out2, kv_cache_dtype, k0, v0, bt, cu, sl, max(query_lens), k_scale, v_scale, None, None, inp["scale"])
    torch.cuda.synchronize()
    assert _lib.last_kernel().startswith("repack+prefill"), _lib.last_kernel()
    torch.testing.assert_close(out2.float(), generic(False).float(), atol=atol, rtol=rtol)


def test_legacy_context_attention_fwd_c2_size_equals_unified_attention():
    """Size-independent property at the C2 shape (Hq 32 / Hk 8 / D 128, 4096 keys, bf16): context_attention_fwd over
    a v0 cache + linear new K/V gives what unified_attention gives over a flash-layout cache holding the same keys."""
    import gpu_util
    from mi355_attn import _lib
    from mi355_attn.kernels import reshape_and_cache_flash, unified_attention
    from mi355_attn.kernels.legacy import context_attention_fwd

    dev = gpu_util.DEV
    B, ctx, QL, Hq, Hk, D, page = 2, 1024, 3072, 32, 8, 128, 16
    L = ctx + QL
    pps = L // page
    nb = B * pps + 5
    g = torch.Generator(device="cpu").manual_seed(5)
    kf = (torch.rand(nb, page, Hk, D, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
    vf = (torch.rand(nb, page, Hk, D, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
    q = (torch.rand(B * QL, Hq, D, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
    k_new = (torch.rand(B * QL, Hk, D, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
    v_new = (torch.rand(B * QL, Hk, D, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
    bt = torch.randperm(nb, generator=g)[: B * pps].to(torch.int32).view(B, pps).to(dev)
    cu = (torch.arange(B + 1, dtype=torch.int32) * QL).to(dev)
    sl = torch.full((B,), L, dtype=torch.int32, device=dev)
    # v0 view of the cache BEFORE the new tokens are written: the legacy op must take them from the linear tensors
    k0 = kf.view(nb, page, Hk, D // 8, 8).permute(0, 2, 3, 1, 4).contiguous()
    v0 = vf.permute(0, 2, 3, 1).contiguous()
    pos = torch.arange(ctx, L, device=dev)
    slots = torch.cat([bt[i].long()[pos // page] * page + pos % page for i in range(B)])
    reshape_and_cache_flash(k_new, v_new, kf, vf, slots, "auto", None, None)
    scale = 1.0 / D ** 0.5
    out_u = torch.empty_like(q)
    unified_attention(q, kf, vf, out_u, cu, QL, sl, L, QL, L, scale, True, (-1, -1), bt, 0.0, None, None, None)
    one = torch.ones(1, dtype=torch.float32, device=dev)
    out_l = torch.empty_like(q)
    context_attention_fwd(q, k_new, v_new, out_l, "auto", k0, v0, bt, cu, sl, QL, one, one, sm_scale=scale)
    torch.cuda.synchronize()
    assert _lib.last_kernel().startswith("repack+prefill"), _lib.last_kernel()
    torch.testing.assert_close(out_l.float(), out_u.float(), atol=1e-2, rtol=0)   # bf16 tolerance of the suite: 2e-2


@pytest.mark.parametrize("kv_cache_dtype,dtype", [("auto", torch.bfloat16), ("auto", torch.float16), ("fp8", torch.bfloat16), ("fp8_e5m2", torch.float16)])
def test_impl_forward_writes_cache_then_attends(kv_cache_dtype, dtype):
    """What vLLM calls per layer: forward(layer, q, k, v, kv_cache, metadata, output)."""
    import gpu_util
    from mi355_attn.backend import attn

    dev = gpu_util.DEV
    Hq, Hk, D, page = 8, 2, 128, 16
    query_lens, ctx_lens = [5, 1, 40, 1], [0, 44, 30, 299]
    kv_lens = [a + b for a, b in zip(query_lens, ctx_lens)]
    inp = orc.make_paged_inputs(31, query_lens, kv_lens, Hq, Hk, D, page, dtype)
    T = sum(query_lens)
    g = torch.Generator().manual_seed(32)
    k_new = (torch.rand(T, Hk, D, generator=g) * 2 - 1).to(dtype)
    v_new = (torch.rand(T, Hk, D, generator=g) * 2 - 1).to(dtype)
    slots = []
    for i, (ql, cl) in enumerate(zip(query_lens, ctx_lens)):
        for j in range(cl, cl + ql):
            slots.append(int(inp["block_table"][i, j // page]) * page + j % page)
    pad = 3  # padded tokens at the end of the step (full-graph mode): slot -1
    slot_mapping = torch.tensor(slots + [-1] * pad, dtype=torch.int64)
    fp8 = kv_cache_dtype.startswith("fp8")
    cache_dtype = {"auto": dtype, "fp8": torch.float8_e4m3fn, "fp8_e5m2": torch.float8_e5m2}[kv_cache_dtype]
    k_scale, v_scale = (0.5, 0.25) if fp8 else (1.0, 1.0)
    # reference state: context already in the cache (quantised for fp8), then the oracle's cache write
    kc = torch.zeros(inp["k_cache"].shape, dtype=cache_dtype)
    vc = torch.zeros_like(kc)
    if fp8:
        kc.copy_((inp["k_cache"].float() / k_scale).to(cache_dtype))
        vc.copy_((inp["v_cache"].float() / v_scale).to(cache_dtype))
    else:
        kc.copy_(inp["k_cache"]); vc.copy_(inp["v_cache"])
    kv_cache = torch.stack([kc, vc]).to(dev)
    if fp8:
        kv_cache = kv_cache.view(torch.uint8)           # vLLM hands fp8 caches over as uint8
    orc.reshape_and_cache_flash_oracle(k_new, v_new, kc, vc, slot_mapping[:T], k_scale=k_scale, v_scale=v_scale)
    ref = orc.unified_attention_oracle(inp["q"], kc, vc, inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"], inp["scale"],
                                       k_scale=k_scale, v_scale=v_scale)

    impl = attn.MI355AttentionImpl(Hq, D, inp["scale"], Hk, None, None, kv_cache_dtype)
    layer = types.SimpleNamespace(_k_scale=torch.tensor(k_scale, device=dev), _v_scale=torch.tensor(v_scale, device=dev),
                                  _q_scale=torch.tensor(1.0, device=dev))
    md = attn.MI355AttentionMetadata(
        num_actual_tokens=T, max_query_len=max(query_lens), avg_query_len=T // 4, avg_seq_len=sum(kv_lens) // 4,
        query_start_loc=inp["cu_seqlens_q"].to(dev), max_seq_len=max(kv_lens), seq_lens=inp["seqused_k"].to(dev),
        block_table=inp["block_table"].to(dev), slot_mapping=slot_mapping.to(dev), use_cascade=False, common_prefix_len=0,
        cu_prefix_query_lens=None, prefix_kv_lens=None, suffix_kv_lens=None)
    q_pad = torch.zeros(T + pad, Hq, D, dtype=dtype, device=dev); q_pad[:T] = inp["q"].to(dev)
    k_pad = torch.zeros(T + pad, Hk, D, dtype=dtype, device=dev); k_pad[:T] = k_new.to(dev)
    v_pad = torch.zeros(T + pad, Hk, D, dtype=dtype, device=dev); v_pad[:T] = v_new.to(dev)
    output = torch.full((T + pad, Hq * D), float("nan"), dtype=dtype, device=dev)
    ret = impl.forward(layer, q_pad, k_pad, v_pad, kv_cache, md, output=output)
    torch.cuda.synchronize()
    assert ret is output
    atol, rtol = golden_io.tolerance(dtype, cache_dtype if fp8 else None)
    torch.testing.assert_close(output[:T].view(T, Hq, D).float().cpu(), ref.float(), atol=atol, rtol=rtol)
    assert torch.isnan(output[T:]).all()                                # padded rows untouched
    got_k = kv_cache[0].view(cache_dtype) if fp8 else kv_cache[0]
    assert torch.equal(got_k.cpu().view(torch.uint8), kc.view(torch.uint8))   # cache write is bit-exact (incl. fp8 quantisation)
    got_v = kv_cache[1].view(cache_dtype) if fp8 else kv_cache[1]
    assert torch.equal(got_v.cpu().view(torch.uint8), vc.view(torch.uint8))


@pytest.mark.parametrize("cache_dtype,kv_cache_dtype", [(torch.float8_e4m3fn, "fp8_e4m3"), (torch.float8_e5m2, "fp8_e5m2")])
@pytest.mark.parametrize("scale", [1.0, 0.0237])
def test_fp8_cache_write_of_edge_values_is_bit_exact(cache_dtype, kv_cache_dtype, scale):
    """The quantising store sat_fp8(x / scale) on the values a random fill never produces: +-0, values that round to
    zero, fp8 subnormals, ties, the largest finite value and everything beyond it. Bit for bit torch's conversion."""
    import gpu_util
    from mi355_attn.kernels import reshape_and_cache_flash

    dev = gpu_util.DEV
    fi = torch.finfo(cache_dtype)
    tiny = fi.smallest_normal
    sub = tiny / (8 if cache_dtype == torch.float8_e4m3fn else 4)      # smallest subnormal
    vals = [0.0, -0.0, sub * 0.49, -sub * 0.49, sub * 0.5, sub * 0.51, sub, -sub, sub * 1.5, sub * 2.5, tiny * 0.999, tiny, -tiny, tiny * 1.0625,
            1.0, -1.0, 1.0625, 1.1875, 3.0, fi.max * 0.999, fi.max, -fi.max, fi.max * 1.03, fi.max * 1.2, -fi.max * 7.0, 1e30, -1e30]
    g = torch.Generator().manual_seed(3)
    x = torch.tensor(vals, dtype=torch.float32) * scale
    x = torch.cat([x, (torch.rand(2 * 128 - len(vals), generator=g) * 2 - 1) * fi.max * 1.1 * scale]).view(1, 2, 128)
    key = x.to(torch.bfloat16)
    value = key.flip(2).contiguous()
    kc = torch.zeros(2, 16, 2, 128, dtype=torch.uint8, device=dev)
    vc = torch.zeros_like(kc)
    slot = torch.tensor([19], dtype=torch.int64, device=dev)
    reshape_and_cache_flash(key.to(dev), value.to(dev), kc, vc, slot, kv_cache_dtype, torch.tensor([scale], device=dev), torch.tensor([scale], device=dev))
    torch.cuda.synchronize()

    def want(t):       # the reference's store: (x / scale) saturated to the finite range, then the conversion (scripts/vllm_utils.py:377-401)
        return (t.float() / scale).clamp(-fi.max, fi.max).to(cache_dtype).view(torch.uint8)

    assert torch.equal(kc[1, 3].cpu(), want(key)[0])
    assert torch.equal(vc[1, 3].cpu(), want(value)[0])
    assert int(kc[0].sum()) == 0 and int(kc[1, :3].sum()) == 0            # nothing else written


def test_impl_forward_local_attention_branch_with_hand_built_virtual_batches():
    """The iRoPE / chunked local attention branch of forward (LIB/backend/triton_attn.py:157-190 builds the virtual
    batches, :423-444 swaps them in): with `use_irope` and `local_attn_metadata` the attention call runs over VIRTUAL
    sequences - one per (sequence, attention chunk) - while the cache write still uses the step's real slot mapping.
    The virtual metadata is built by hand here (vLLM's make_local_attention_virtual_batches is not needed for that) and
    the result is checked against the oracle on the virtual batch AND against a dense float64 softmax with the chunk
    mask written out, on the real sequences."""
    import gpu_util
    from mi355_attn.backend import attn

    dev = gpu_util.DEV
    Hq, Hk, D, page, chunk = 8, 2, 128, 16, 32
    dtype = torch.bfloat16
    query_lens, ctx_lens = [50, 1, 7, 33], [40, 70, 0, 31]
    kv_lens = [a + b for a, b in zip(query_lens, ctx_lens)]
    inp = orc.make_paged_inputs(61, query_lens, kv_lens, Hq, Hk, D, page, dtype)
    T = sum(query_lens)
    g = torch.Generator().manual_seed(62)
    k_new = (torch.rand(T, Hk, D, generator=g) * 2 - 1).to(dtype)
    v_new = (torch.rand(T, Hk, D, generator=g) * 2 - 1).to(dtype)
    slots = []
    for i, (ql, cl) in enumerate(zip(query_lens, ctx_lens)):
        for j in range(cl, cl + ql):
            slots.append(int(inp["block_table"][i, j // page]) * page + j % page)
    slot_mapping = torch.tensor(slots, dtype=torch.int64)
    kc, vc = inp["k_cache"].clone(), inp["v_cache"].clone()
    orc.reshape_and_cache_flash_oracle(k_new, v_new, kc, vc, slot_mapping)
    # virtual batches: for every sequence, every attention chunk its query tokens fall into
    v_qlens, v_klens, v_bt = [], [], []
    ppc = chunk // page
    for i, (ql, cl) in enumerate(zip(query_lens, ctx_lens)):
        pos = cl
        while pos < cl + ql:
            c0 = (pos // chunk) * chunk
            hi = min(c0 + chunk, cl + ql)
            v_qlens.append(hi - pos)
            v_klens.append(hi - c0)
            row = inp["block_table"][i, c0 // page: c0 // page + ppc]
            v_bt.append(torch.cat([row, torch.zeros(ppc - row.numel(), dtype=torch.int32)]))
            pos = hi
    v_cu = torch.tensor([0] + torch.tensor(v_qlens).cumsum(0).tolist(), dtype=torch.int32)
    v_sk = torch.tensor(v_klens, dtype=torch.int32)
    v_bt = torch.stack(v_bt).contiguous()
    assert int(v_cu[-1]) == T and len(v_qlens) > len(query_lens)
    ref = orc.unified_attention_oracle(inp["q"], kc, vc, v_cu, v_sk, v_bt, inp["scale"])
    # independent: dense float64 softmax over the real sequences with the chunk mask
    dense = torch.zeros(T, Hq, D, dtype=torch.float64)
    t0 = 0
    for i, (ql, cl) in enumerate(zip(query_lens, ctx_lens)):
        L = cl + ql
        pages = inp["block_table"][i, : (L + page - 1) // page].long()
        ks = kc[pages].reshape(-1, Hk, D)[:L].double()
        vs = vc[pages].reshape(-1, Hk, D)[:L].double()
        qpos = torch.arange(cl, L)[:, None]
        kpos = torch.arange(L)[None, :]
        mask = (kpos <= qpos) & (kpos >= (qpos // chunk) * chunk)
        for h in range(Hq):
            s = inp["scale"] * (inp["q"][t0:t0 + ql, h].double() @ ks[:, h // (Hq // Hk)].T)
            dense[t0:t0 + ql, h] = torch.softmax(s.masked_fill(~mask, float("-inf")), dim=-1) @ vs[:, h // (Hq // Hk)]
        t0 += ql
    torch.testing.assert_close(ref.double(), dense, atol=2e-2, rtol=2e-2)

    impl = attn.MI355AttentionImpl(Hq, D, inp["scale"], Hk, None, None, "auto", use_irope=True)
    layer = types.SimpleNamespace(_k_scale=torch.tensor(1.0, device=dev), _v_scale=torch.tensor(1.0, device=dev), _q_scale=torch.tensor(1.0, device=dev))
    local = attn.MI355AttentionMetadata.LocalAttentionMetadata(
        local_query_start_loc=v_cu.to(dev), local_seqused_k=v_sk.to(dev), local_block_table=v_bt.to(dev),
        local_max_query_len=max(v_qlens), local_max_seq_len=max(v_klens), local_avg_query_len=T // len(v_qlens),
        local_avg_seq_len=sum(v_klens) // len(v_klens), local_scheduler_metadata=None)
    md = attn.MI355AttentionMetadata(
        num_actual_tokens=T, max_query_len=max(query_lens), avg_query_len=T // 4, avg_seq_len=sum(kv_lens) // 4,
        query_start_loc=inp["cu_seqlens_q"].to(dev), max_seq_len=max(kv_lens), seq_lens=inp["seqused_k"].to(dev),
        block_table=inp["block_table"].to(dev), slot_mapping=slot_mapping.to(dev), use_cascade=False, common_prefix_len=0,
        cu_prefix_query_lens=None, prefix_kv_lens=None, suffix_kv_lens=None, local_attn_metadata=local)
    kv_cache = torch.stack([inp["k_cache"], inp["v_cache"]]).to(dev)
    output = torch.full((T, Hq * D), float("nan"), dtype=dtype, device=dev)
    impl.forward(layer, inp["q"].to(dev), k_new.to(dev), v_new.to(dev), kv_cache, md, output=output)
    torch.cuda.synchronize()
    assert torch.equal(kv_cache[0].cpu().view(torch.int16), kc.view(torch.int16))        # the write used the REAL slot mapping
    torch.testing.assert_close(output.view(T, Hq, D).float().cpu(), ref.float(), atol=2e-2, rtol=2e-2)
    # the same layer without local metadata attends over the whole context: a different result (the branch was really taken)
    md.local_attn_metadata = None
    out2 = torch.full((T, Hq * D), float("nan"), dtype=dtype, device=dev)
    impl.forward(layer, inp["q"].to(dev), k_new.to(dev), v_new.to(dev), kv_cache, md, output=out2)
    torch.cuda.synchronize()
    assert (out2.float() - output.float()).abs().max().item() > 0.05

    # a decode step with local metadata must not take the fused write (its kernel knows nothing of virtual batches)
    dec = attn.MI355AttentionMetadata(
        num_actual_tokens=1, max_query_len=1, avg_query_len=1, avg_seq_len=71, query_start_loc=torch.tensor([0, 1], dtype=torch.int32, device=dev),
        max_seq_len=71, seq_lens=torch.tensor([71], dtype=torch.int32, device=dev), block_table=inp["block_table"][1:2].to(dev),
        slot_mapping=slot_mapping[50:51].to(dev), use_cascade=False, common_prefix_len=0, cu_prefix_query_lens=None, prefix_kv_lens=None,
        suffix_kv_lens=None,
        local_attn_metadata=attn.MI355AttentionMetadata.LocalAttentionMetadata(
            local_query_start_loc=torch.tensor([0, 1], dtype=torch.int32, device=dev), local_seqused_k=torch.tensor([7], dtype=torch.int32, device=dev),
            local_block_table=inp["block_table"][1:2, 4:6].contiguous().to(dev), local_max_query_len=1, local_max_seq_len=7,
            local_avg_query_len=1, local_avg_seq_len=7, local_scheduler_metadata=None))
    o1 = torch.full((1, Hq * D), float("nan"), dtype=dtype, device=dev)
    impl.forward(layer, inp["q"][50:51].to(dev), k_new[50:51].to(dev), v_new[50:51].to(dev), kv_cache, dec, output=o1)
    torch.cuda.synchronize()
    torch.testing.assert_close(o1.view(1, Hq, D).float().cpu(), ref[50:51].float(), atol=2e-2, rtol=2e-2)
