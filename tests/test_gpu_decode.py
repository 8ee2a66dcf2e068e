"""-m gpu: split-KV decode kernel vs the CPU oracle on seeded inputs (sizes the oracle finishes in
seconds), plus size-independent properties at BASELINE's C3 size."""

import math

import pytest
import torch

import golden_io
from oracle import paged_attention_oracle as orc

pytestmark = pytest.mark.gpu


def _check(inp, dtype, *, force, window=0, softcap=0.0, alibi=None, expect=None, kv_dtype=None, kv_scale=None):
    import gpu_util

    ref = orc.unified_attention_oracle(inp["q"], inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"],
                                       inp["scale"], sliding_window=window, softcap=softcap, alibi_slopes=alibi,
                                       k_scale=kv_scale or 1.0, v_scale=kv_scale or 1.0, mode="3d")
    d = gpu_util.to_dev(inp)
    if alibi is not None:
        d["alibi_slopes"] = alibi.to(gpu_util.DEV)
    out, kernel = gpu_util.run_unified(d, inp["scale"], window=window, softcap=softcap, force=force, kv_scale=kv_scale)
    if expect is not None:
        assert kernel.startswith(expect), kernel
    atol, rtol = golden_io.tolerance(dtype, kv_dtype)
    assert not torch.isnan(out).any()
    torch.testing.assert_close(out.float().cpu(), ref.float(), atol=atol, rtol=rtol)
    return kernel


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hq,hk", [(32, 8), (8, 8), (16, 1), (40, 8), (6, 2)])
@pytest.mark.parametrize("d", [64, 128, 256])
def test_decode_heads_and_head_sizes(dtype, hq, hk, d):
    if dtype == torch.float16 and d != 128:
        pytest.skip("fp16 is covered at head size 128; the other head sizes run in bf16 (suite run time)")
    kv_lens = [1, 15, 16, 17, 33, 257, 1023]
    inp = orc.make_paged_inputs(5, [1] * len(kv_lens), kv_lens, hq, hk, d, 16, dtype)
    _check(inp, dtype, force=None, expect="decode")


@pytest.mark.parametrize("hq,hk,d,kv", [(32, 1, 128, None), (40, 2, 64, None), (48, 1, 128, "e4m3"), (72, 2, 128, None), (34, 2, 256, None)])
def test_decode_more_than_16_query_heads_per_kv_head(hq, hk, d, kv):
    """A wave's MFMA columns hold 16 query heads; a KV head with more of them takes cdiv(G, 16) waves (the last one
    partly filled: G = 20, 36, 17), through the in-kernel merge and through the merge launch (segments)."""
    import gpu_util
    kv_dtype = torch.float8_e4m3fn if kv else None
    scale = 0.5 if kv else None
    kv_lens = [1, 16, 17, 33, 257, 1023, 2500]
    inp = orc.make_paged_inputs(21, [1] * len(kv_lens), kv_lens, hq, hk, d, 16, torch.bfloat16, **({"kv_dtype": kv_dtype, "kv_scale": scale} if kv else {}))
    _check(inp, torch.bfloat16, force=None, expect="decode_splitkv", kv_dtype=kv_dtype, kv_scale=scale)
    from mi355_attn.kernels import unified as ua_mod
    dev = gpu_util.to_dev(inp)
    ks = None if scale is None else torch.tensor([scale], dtype=torch.float32, device=gpu_util.DEV)
    outs = []
    for n in (1, 5, 64):
        out = torch.full_like(dev["q"], float("nan"))
        p, keep = ua_mod.fill_attn_params(dev["q"], dev["k_cache"], dev["v_cache"], out, dev["cu_seqlens_q"], 1, dev["seqused_k"], max(kv_lens),
                                          inp["scale"], (-1, -1), dev["block_table"], 0.0, ks, ks, None, 3, num_segments=n)
        ua_mod.launch(p, gpu_util.DEV)
        torch.cuda.synchronize()
        outs.append(out.float())
    torch.testing.assert_close(outs[1], outs[0], atol=2e-3, rtol=1.6e-2)
    torch.testing.assert_close(outs[2], outs[0], atol=2e-3, rtol=1.6e-2)


@pytest.mark.parametrize("d", [32, 80, 96, 160, 192, 224])
def test_decode_head_sizes_that_run_padded(d):
    """Head sizes between the built ones run on the next built size; the padding columns are never read or written
    (the reference pads to the next power of two, triton_unified_attention.py:353,:912)."""
    kv_lens = [1, 17, 33, 700, 1023, 257]
    inp = orc.make_paged_inputs(30 + d, [1] * len(kv_lens), kv_lens, 8, 2, d, 16, torch.bfloat16)
    _check(inp, torch.bfloat16, force=None, expect="decode")
    if d % 16 == 0:
        inp8 = orc.make_paged_inputs(31 + d, [1] * len(kv_lens), kv_lens, 8, 2, d, 16, torch.float16, kv_dtype=torch.float8_e4m3fn, kv_scale=0.5)
        _check(inp8, torch.float16, force=None, expect="decode_splitkv_fp8", kv_dtype=torch.float8_e4m3fn, kv_scale=0.5)


def test_decode_padded_head_size_leaves_neighbouring_output_columns_alone():
    """out has a wider row than the head: bytes between heads must keep their contents."""
    import gpu_util

    d, dw = 96, 128
    kv_lens = [300, 17, 1023]
    inp = orc.make_paged_inputs(33, [1] * 3, kv_lens, 8, 2, d, 16, torch.bfloat16)
    ref = orc.unified_attention_oracle(inp["q"], inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"],
                                       inp["scale"], mode="3d")
    t = gpu_util.to_dev(inp)
    wide = torch.full((3, 8, dw), 7.0, dtype=torch.bfloat16, device=gpu_util.DEV)
    out, kernel = gpu_util.run_unified(t, inp["scale"], out=wide[:, :, :d])
    assert kernel.startswith("decode")
    torch.testing.assert_close(wide[:, :, :d].float().cpu(), ref.float(), atol=2e-2, rtol=2e-2)
    assert torch.all(wide[:, :, d:] == 7.0)


@pytest.mark.parametrize("page", [16, 32, 64, 128])
def test_decode_page_sizes(page):
    kv_lens = [5, 129, 640, 77, 300]
    inp = orc.make_paged_inputs(6, [1] * len(kv_lens), kv_lens, 16, 4, 128, page, torch.bfloat16)
    _check(inp, torch.bfloat16, force=None, expect="decode")


@pytest.mark.parametrize("segments", [1, 3, 64])
def test_decode_split_counts_agree(segments):
    """Any split count must give the same answer (merge is exact up to fp32 rounding)."""
    import gpu_util
    from mi355_attn.kernels import unified as ua_mod

    kv_lens = [2000, 33, 1, 4096, 515]
    inp = orc.make_paged_inputs(7, [1] * len(kv_lens), kv_lens, 32, 8, 128, 16, torch.bfloat16)
    d = gpu_util.to_dev(inp)
    ref = orc.unified_attention_oracle(inp["q"], inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"],
                                       inp["scale"], mode="3d")
    out = torch.full_like(d["q"], float("nan"))
    p, keep = ua_mod.fill_attn_params(d["q"], d["k_cache"], d["v_cache"], out, d["cu_seqlens_q"], 1, d["seqused_k"], max(kv_lens),
                                      inp["scale"], (-1, -1), d["block_table"], 0.0, None, None, None, 3, num_segments=segments)
    ua_mod.launch(p, gpu_util.DEV)
    torch.cuda.synchronize()
    torch.testing.assert_close(out.float().cpu(), ref.float(), atol=2e-2, rtol=2e-2)


def test_decode_features_window_softcap_alibi():
    kv_lens = [300, 17, 256, 1, 129, 1000]
    inp = orc.make_paged_inputs(8, [1] * len(kv_lens), kv_lens, 8, 2, 128, 16, torch.float16)
    alibi = torch.tensor([2.0 ** (-(i + 1)) for i in range(8)], dtype=torch.float32)
    _check(inp, torch.float16, force=None, window=8, expect="decode")
    _check(inp, torch.float16, force=None, window=100, expect="decode")
    _check(inp, torch.float16, force=None, softcap=30.0, expect="decode")
    _check(inp, torch.float16, force=None, alibi=alibi, expect="decode")
    _check(inp, torch.float16, force=None, window=64, softcap=20.0, alibi=alibi, expect="decode")


def test_decode_kernel_on_multi_token_queries():
    """force 3D on a batch with query_len > 1: every token is handled as its own causal decode."""
    inp = orc.make_paged_inputs(9, [7, 1, 40, 3], [70, 45, 70, 300], 8, 2, 128, 16, torch.bfloat16)
    _check(inp, torch.bfloat16, force=3, expect="decode")


def test_decode_stale_nan_beyond_sequence_is_ignored():
    """Slots past seq_len inside the last page (and unused pages) may hold anything, incl. NaN."""
    import gpu_util

    kv_lens = [33, 100, 5]
    inp = orc.make_paged_inputs(10, [1] * 3, kv_lens, 8, 2, 128, 16, torch.bfloat16)
    ref = orc.unified_attention_oracle(inp["q"], inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"],
                                       inp["scale"], mode="3d")
    used = torch.zeros(inp["k_cache"].shape[:2], dtype=torch.bool)
    for i, n in enumerate(kv_lens):
        for j in range(n):
            used[inp["block_table"][i, j // 16], j % 16] = True
    inp["k_cache"][~used] = float("nan")
    inp["v_cache"][~used] = float("nan")
    d = gpu_util.to_dev(inp)
    out, kernel = gpu_util.run_unified(d, inp["scale"])
    assert kernel.startswith("decode")
    assert not torch.isnan(out).any()
    torch.testing.assert_close(out.float().cpu(), ref.float(), atol=2e-2, rtol=2e-2)


def test_decode_c3_full_size_properties():
    """BASELINE C3 (B=64, Hq=32, Hk=8, D=128, kv=8192, bf16, page 16) — too big for the oracle, so check
    size-independent properties: (1) V == const  =>  out == const exactly-ish (softmax weights sum to 1);
    (2) permuting the physical pages (with the block table) leaves the output bit-identical;
    (3) a sample of (seq, head) rows equals the oracle."""
    import gpu_util

    dev = gpu_util.DEV
    B, Hq, Hk, D, kv, page = 64, 32, 8, 128, 8192, 16
    g = torch.Generator(device="cpu").manual_seed(0)
    pps = kv // page
    nb = B * pps + 7
    k = (torch.rand(nb, page, Hk, D, generator=g) * 2 - 1).to(torch.bfloat16)
    v = (torch.rand(nb, page, Hk, D, generator=g) * 2 - 1).to(torch.bfloat16)
    q = (torch.rand(B, Hq, D, generator=g) * 2 - 1).to(torch.bfloat16)
    bt = torch.randperm(nb, generator=g)[: B * pps].to(torch.int32).view(B, pps)
    t = dict(q=q, k_cache=k, v_cache=v, block_table=bt, cu_seqlens_q=torch.arange(B + 1, dtype=torch.int32),
             seqused_k=torch.full((B,), kv, dtype=torch.int32))
    scale = 1.0 / math.sqrt(D)
    d = gpu_util.to_dev(t)
    out, kernel = gpu_util.run_unified(d, scale)
    assert kernel == "decode_splitkv"
    # (3) sample rows vs oracle
    for i in (0, 37, 63):
        ref = gpu_util.oracle_row(orc, q[i:i + 1], k, v, bt[i], kv, scale, mode="3d")
        torch.testing.assert_close(out[i:i + 1].float().cpu(), ref.float(), atol=2e-2, rtol=2e-2)
    # (2) page permutation invariance, bit-exact
    perm = torch.randperm(nb, generator=g)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(nb)
    d2 = dict(d)
    d2["k_cache"] = d["k_cache"][perm.to(dev)]
    d2["v_cache"] = d["v_cache"][perm.to(dev)]
    d2["block_table"] = inv.to(dev)[d["block_table"].long()].to(torch.int32)
    out2, _ = gpu_util.run_unified(d2, scale)
    assert torch.equal(out.view(torch.int16), out2.view(torch.int16))
    # (1) constant V
    d3 = dict(d)
    d3["v_cache"] = torch.full_like(d["v_cache"], 0.5)
    out3, _ = gpu_util.run_unified(d3, scale)
    torch.testing.assert_close(out3.float(), torch.full_like(out3, 0.5).float(), atol=4e-3, rtol=0)


def test_decode_c5_full_size_properties():
    """BASELINE C5 (Llama-3-70B shape: Hq=64, Hk=8, D=128, B=16, kv=32768 = 2048 pages per sequence, fp8-e4m3 KV with
    scalar scales that are no powers of two, bf16 Q): sampled (sequence, head) rows against the oracle, page-permutation
    invariance bit for bit, constant V => constant output, and ragged lengths under the same (captured) split plan."""
    import gpu_util

    dev = gpu_util.DEV
    B, Hq, Hk, D, kv, page = 16, 64, 8, 128, 32768, 16
    ks, vs = 0.0237, 0.041
    g = torch.Generator(device="cpu").manual_seed(5)
    pps = kv // page
    nb = B * pps + 5
    # (the 2 x 0.5 GB of cache values are drawn on the device: a minute of host time otherwise)
    gd = torch.Generator(device=dev).manual_seed(5)
    k = ((torch.rand(nb, page, Hk, D, generator=gd, device=dev) * 2 - 1) / ks).to(torch.float8_e4m3fn).cpu()
    v = ((torch.rand(nb, page, Hk, D, generator=gd, device=dev) * 2 - 1) / vs).to(torch.float8_e4m3fn).cpu()
    q = (torch.rand(B, Hq, D, generator=g) * 2 - 1).to(torch.bfloat16)
    bt = torch.randperm(nb, generator=g)[: B * pps].to(torch.int32).view(B, pps)
    t = dict(q=q, k_cache=k, v_cache=v, block_table=bt, cu_seqlens_q=torch.arange(B + 1, dtype=torch.int32),
             seqused_k=torch.full((B,), kv, dtype=torch.int32))
    scale = 1.0 / math.sqrt(D)
    d = gpu_util.to_dev(t)
    out, kernel = gpu_util.run_unified(d, scale, kv_scale=ks, v_scale=vs)
    assert kernel == "decode_splitkv_fp8", kernel
    assert not torch.isnan(out).any()
    atol, rtol = golden_io.tolerance(torch.bfloat16, torch.float8_e4m3fn)
    for i in (0, 9, 15):
        ref = gpu_util.oracle_row(orc, q[i:i + 1], k, v, bt[i], kv, scale, k_scale=ks, v_scale=vs, mode="3d")
        torch.testing.assert_close(out[i:i + 1].float().cpu(), ref.float(), atol=atol, rtol=rtol)
    perm = torch.randperm(nb, generator=g)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(nb)
    d2 = dict(d)
    d2["k_cache"] = d["k_cache"].view(torch.uint8)[perm.to(dev)].view(torch.float8_e4m3fn)
    d2["v_cache"] = d["v_cache"].view(torch.uint8)[perm.to(dev)].view(torch.float8_e4m3fn)
    d2["block_table"] = inv.to(dev)[d["block_table"].long()].to(torch.int32)
    out2, _ = gpu_util.run_unified(d2, scale, kv_scale=ks, v_scale=vs)
    assert torch.equal(out.view(torch.int16), out2.view(torch.int16))
    d3 = dict(d)
    d3["v_cache"] = torch.full((nb, page, Hk, D), 0.5 / vs, device=dev).to(torch.float8_e4m3fn)
    out3, _ = gpu_util.run_unified(d3, scale, kv_scale=ks, v_scale=vs)
    const = float(torch.tensor(0.5 / vs).to(torch.float8_e4m3fn).float()) * vs
    torch.testing.assert_close(out3.float(), torch.full_like(out3, const).float(), atol=6e-3, rtol=0)
    # ragged lengths, same max_seqlen_k (what a graph captured at this size replays): rows vs the oracle
    lens = [32768, 1, 17, 5000, 32767, 16384, 100, 31999, 2, 8192, 4096, 33, 32768, 777, 12345, 20000]
    d4 = dict(d)
    d4["seqused_k"] = torch.tensor(lens, dtype=torch.int32, device=dev)
    out4, _ = _run_with_max_k(d4, scale, kv, ks, vs)
    for i in (1, 2, 3, 4, 13):
        ref = gpu_util.oracle_row(orc, q[i:i + 1], k, v, bt[i], lens[i], scale, k_scale=ks, v_scale=vs, mode="3d")
        torch.testing.assert_close(out4[i:i + 1].float().cpu(), ref.float(), atol=atol, rtol=rtol)


def _run_with_max_k(t, scale, max_k, ks, vs):
    from mi355_attn import _lib
    from mi355_attn.kernels import unified_attention

    q = t["q"]
    out = torch.full_like(q, float("nan"))
    kt = torch.tensor([ks], dtype=torch.float32, device=q.device)
    vt = torch.tensor([vs], dtype=torch.float32, device=q.device)
    unified_attention(q=q, k=t["k_cache"], v=t["v_cache"], out=out, cu_seqlens_q=t["cu_seqlens_q"], max_seqlen_q=1, seqused_k=t["seqused_k"],
                      max_seqlen_k=max_k, avg_seqlen_q=1, avg_seqlen_k=1, softmax_scale=scale, causal=True, window_size=(-1, -1),
                      block_table=t["block_table"], softcap=0, q_descale=None, k_descale=kt, v_descale=vt)
    torch.cuda.synchronize()
    return out, _lib.last_kernel()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("kv_dtype", [torch.float8_e4m3fn, torch.float8_e5m2])
@pytest.mark.parametrize("hq,hk,d", [(32, 8, 128), (64, 8, 128), (8, 2, 64), (4, 1, 256)])
def test_decode_fp8_kv_cache_on_the_mfma_path(dtype, kv_dtype, hq, hk, d):
    """C5-style: fp8 KV cache, 16-bit queries, scalar k/v scales (LIB/kernels/triton_unified_attention.py:434-455)."""
    kv_lens = [1, 15, 16, 17, 33, 700, 1023, 257]
    inp = orc.make_paged_inputs(11, [1] * len(kv_lens), kv_lens, hq, hk, d, 16, dtype, kv_dtype=kv_dtype, kv_scale=0.5)
    _check(inp, dtype, force=None, expect="decode_splitkv_fp8", kv_dtype=kv_dtype, kv_scale=0.5)


def test_decode_fp8_page32_and_features():
    kv_lens = [300, 17, 256, 1, 129]
    inp = orc.make_paged_inputs(12, [1] * len(kv_lens), kv_lens, 8, 2, 128, 32, torch.float16, kv_dtype=torch.float8_e4m3fn, kv_scale=0.25)
    alibi = torch.tensor([2.0 ** (-(i + 1)) for i in range(8)], dtype=torch.float32)
    _check(inp, torch.float16, force=None, expect="decode", kv_dtype=torch.float8_e4m3fn, kv_scale=0.25)
    _check(inp, torch.float16, force=None, window=50, softcap=25.0, alibi=alibi, expect="decode", kv_dtype=torch.float8_e4m3fn, kv_scale=0.25)


@pytest.mark.parametrize("spike_key", [0, 5, 12, 21, 30, 37, 63, 100, 255])
def test_decode_running_max_exchange_with_a_spiked_key(spike_key):
    """One key scores ~+60 above the rest, placed so that it lands in each of the four lane groups
    / both 16-key groups of a tile in turn: the running max must be exchanged across lane groups
    (a wrong exchange overflows exp2 or loses the spike's weight)."""
    import gpu_util

    kv_len, Hq, Hk, D = 256, 8, 2, 128
    inp = orc.make_paged_inputs(13, [1], [kv_len], Hq, Hk, D, 16, torch.bfloat16)
    page, slot = int(inp["block_table"][0, spike_key // 16]), spike_key % 16
    qn = inp["q"][0].float()                                       # [Hq, D]
    for h in range(Hk):                                            # key = 6 * normalised mean query of its group
        qm = qn[h * 4:(h + 1) * 4].mean(0)
        inp["k_cache"][page, slot, h] = (qm / qm.norm() * 60.0).to(torch.bfloat16)
    ref = orc.unified_attention_oracle(inp["q"], inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"],
                                       1.0, mode="3d")
    d = gpu_util.to_dev(inp)
    for force in (None, 2):
        out, kernel = gpu_util.run_unified(d, 1.0, force=force)
        assert not torch.isnan(out).any() and not torch.isinf(out).any(), kernel
        torch.testing.assert_close(out.float().cpu(), ref.float(), atol=2e-2, rtol=2e-2, msg=lambda m: f"[{kernel}] {m}")


def test_mixed_batch_splits_into_prefill_and_decode_launches():
    import gpu_util

    query_lens = [1, 64, 1, 1, 200, 1, 33, 1]
    kv_lens = [900, 64, 17, 2048, 333, 1, 100, 513]
    inp = orc.make_paged_inputs(14, query_lens, kv_lens, 32, 8, 128, 16, torch.bfloat16)
    ref = orc.unified_attention_oracle(inp["q"], inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"],
                                       inp["scale"], mode="2d", block_n=64)
    d = gpu_util.to_dev(inp)
    out, kernel = gpu_util.run_unified(d, inp["scale"])
    # (the prefill half may be the key-split launch: few Q blocks, and the longest sequence of the batch has 2048 keys)
    # (the decode half: the packed instantiation - G = 4 here, rows of up to 4 query tokens are decode rows)
    assert kernel.replace("_pw", "").replace("_pack", "") in ("prefill_mfma+decode_splitkv", "prefill_mfma+decode_single", "prefill_mfma_ksplit+decode_splitkv"), kernel
    assert not torch.isnan(out).any()
    torch.testing.assert_close(out.float().cpu(), ref.float(), atol=2e-2, rtol=2e-2)
    # and the single-kernel 2D path gives the same answer
    out2, kernel2 = gpu_util.run_unified(d, inp["scale"], force=2)
    assert kernel2.startswith("prefill_mfma") and "+" not in kernel2, kernel2
    torch.testing.assert_close(out2.float().cpu(), ref.float(), atol=2e-2, rtol=2e-2)


# ---- multi-token decode steps on the PACK kernels (16 / G query tokens of a sequence share a wave's matrix columns) ----------
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hq,hk", [(32, 8), (8, 8), (16, 2), (6, 2), (64, 8), (10, 2), (32, 2), (24, 2)])
@pytest.mark.parametrize("d", [64, 128, 256])
def test_multi_token_decode_steps_share_one_key_stream(dtype, hq, hk, d):
    """Speculative-decoding / MTP verification batches (a few query tokens per sequence, causal among themselves, over a
    long context) run on the decode kernel with several tokens per work unit; every row must match the oracle - the
    per-column key limits, ragged chunks (query lengths that are not multiples of the chunk), sequences of one token
    and contexts of zero keys included."""
    q_lens = [4, 1, 3, 2, 4, 4, 2, 1]
    kv_lens = [700, 45, 3, 2, 4, 1030, 65, 1]            # contexts of 696, 44, 0, 0, 0, 1026, 63, 0 keys
    inp = orc.make_paged_inputs(21, q_lens, kv_lens, hq, hk, d, 16, dtype)
    # (selected explicitly: the dispatch itself packs only what ONE work unit per sequence holds, see the test below)
    kernel = _check(inp, dtype, force=3, expect="decode_s")
    g = hq // hk
    want = "pack2" if (g > 4 and d <= 128) else "pack" if g <= 8 else "decode_splitkv"      # 4 tokens: one column group up to G = 4, two up to head size 128
    assert want in kernel and (want != "pack" or "pack2" not in kernel), kernel


@pytest.mark.parametrize("q_len", [2, 5, 8, 16])
@pytest.mark.parametrize("segments", [1, 2, 7, 40])
def test_multi_token_decode_split_counts_and_chunks_agree(q_len, segments):
    """More tokens than one chunk holds (two work units per sequence and more - forced onto the decode path) and every
    merge route: no split, the in-kernel merge, the merge launch."""
    import gpu_util
    from mi355_attn.kernels import unified as ua_mod

    q_lens = [q_len, max(1, q_len - 1), q_len, 1]
    kv_lens = [1500 + q_len, 33 + q_len, 257, 300]
    inp = orc.make_paged_inputs(22 + q_len, q_lens, kv_lens, 32, 8, 128, 16, torch.bfloat16)
    d = gpu_util.to_dev(inp)
    ref = orc.unified_attention_oracle(inp["q"], inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"],
                                       inp["scale"], mode="3d")
    out = torch.full_like(d["q"], float("nan"))
    lse = torch.full((d["q"].shape[0], 32), float("nan"), dtype=torch.float32, device=gpu_util.DEV)
    p, keep = ua_mod.fill_attn_params(d["q"], d["k_cache"], d["v_cache"], out, d["cu_seqlens_q"], max(q_lens), d["seqused_k"], max(kv_lens),
                                      inp["scale"], (-1, -1), d["block_table"], 0.0, None, None, None, 3, num_segments=segments, lse=lse)
    ua_mod.launch(p, gpu_util.DEV)
    torch.cuda.synchronize()
    from mi355_attn import _lib
    assert "pack" in _lib.last_kernel()
    torch.testing.assert_close(out.float().cpu(), ref.float(), atol=2e-2, rtol=2e-2)
    assert torch.isfinite(lse).all()


@pytest.mark.parametrize("kv_dtype", [torch.float8_e4m3fn, torch.float8_e5m2])
def test_multi_token_decode_fp8_cache(kv_dtype):
    q_lens, kv_lens = [4, 2, 3, 1], [900, 130, 3, 64]
    inp = orc.make_paged_inputs(31, q_lens, kv_lens, 32, 8, 128, 16, torch.bfloat16, kv_dtype=kv_dtype, kv_scale=0.5)
    kernel = _check(inp, torch.bfloat16, force=None, expect="decode_s", kv_dtype=kv_dtype, kv_scale=0.5)
    assert "pack" in kernel and "fp8" in kernel


def test_multi_token_decode_dispatch():
    """The dispatch packs a batch whose longest query fits one work unit (16 / G tokens, rounded down to a power of two);
    windows / soft-cap / ALiBi, more than 8 query heads per KV head and longer queries stay on their former paths."""
    for hq, hk, q_lens, want in [(32, 8, [4, 1, 3], "pack"), (16, 2, [2, 2, 1], "pack"), (8, 8, [16, 9], "pack"), (6, 2, [4, 4], "pack"),
                                 (16, 2, [3, 2], "pack2"), (32, 8, [8, 5, 1], "pack2"), (32, 2, [2, 2], "pack2"), (8, 8, [32, 17], "pack2")]:
        inp = orc.make_paged_inputs(35, q_lens, [300 + q for q in q_lens], hq, hk, 128, 16, torch.bfloat16)
        kernel = _check(inp, torch.bfloat16, force=None)
        assert want in kernel and (want != "pack" or "pack2" not in kernel), (hq, hk, q_lens, kernel)
    inp = orc.make_paged_inputs(36, [5, 2], [300, 200], 16, 2, 128, 16, torch.bfloat16)      # G = 8: four tokens per unit at most
    assert _check(inp, torch.bfloat16, force=None).startswith("prefill")                     # (a mixed batch: the 2-token row is the decode launch's)
    inp = orc.make_paged_inputs(36, [5, 5], [300, 200], 16, 2, 128, 16, torch.bfloat16)
    assert "pack" not in _check(inp, torch.bfloat16, force=None)
    inp = orc.make_paged_inputs(37, [4, 4], [300, 200], 16, 2, 256, 16, torch.bfloat16)      # head size 256: one column group only
    assert "pack" not in _check(inp, torch.bfloat16, force=None)
    inp = orc.make_paged_inputs(32, [4, 4], [300, 200], 8, 2, 128, 16, torch.bfloat16)
    assert "pack" in _check(inp, torch.bfloat16, force=None, window=64)                      # features: one column group
    inp = orc.make_paged_inputs(32, [8, 8], [300, 200], 8, 2, 128, 16, torch.bfloat16)
    assert "pack" not in _check(inp, torch.bfloat16, force=None, window=64)                  # ... and no second one
    inp = orc.make_paged_inputs(33, [2, 2], [300, 200], 34, 2, 128, 16, torch.bfloat16)      # G = 17
    assert "pack" not in _check(inp, torch.bfloat16, force=None)
    inp = orc.make_paged_inputs(34, [40, 33], [300, 200], 32, 8, 128, 16, torch.bfloat16)
    assert _check(inp, torch.bfloat16, force=None).startswith("prefill")                      # (with a decode launch that finds no row of up to 4 tokens)
    inp = orc.make_paged_inputs(34, [40, 40], [300, 200], 32, 8, 128, 16, torch.bfloat16)
    assert "pack" not in _check(inp, torch.bfloat16, force=None)


def test_mixed_batch_sends_multi_token_decode_rows_to_the_decode_launch():
    """A step that mixes prefill chunks with speculative-decoding rows (1 + k tokens each): rows of up to 16 / G tokens
    are the decode launch's (packed into a wave's matrix columns), longer ones the prefill launch's; every row is
    computed exactly once (the output starts as NaN) and matches the oracle."""
    import gpu_util
    query_lens = [300, 4, 1, 3, 5, 2, 4, 64, 1]
    kv_lens = [900, 640, 17, 2048, 333, 2, 4, 64, 513]
    for hq, hk, thr in [(32, 8, 4), (16, 2, 2), (8, 8, 16)]:
        inp = orc.make_paged_inputs(61, query_lens, kv_lens, hq, hk, 128, 16, torch.bfloat16)
        ref = orc.unified_attention_oracle(inp["q"], inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"],
                                           inp["scale"], mode="2d", block_n=64)
        d = gpu_util.to_dev(inp)
        lse = torch.full((sum(query_lens), hq), float("nan"), dtype=torch.float32, device=gpu_util.DEV)
        out, kernel = gpu_util.run_unified(d, inp["scale"], lse=lse)
        assert "+decode" in kernel and "pack" in kernel, kernel
        assert not torch.isnan(out).any() and torch.isfinite(lse).all(), (hq, hk)
        torch.testing.assert_close(out.float().cpu(), ref.float(), atol=2e-2, rtol=2e-2)
    # with a sliding window: the packed kernel's feature instantiation
    inp = orc.make_paged_inputs(62, query_lens, kv_lens, 32, 8, 128, 16, torch.bfloat16)
    ref = orc.unified_attention_oracle(inp["q"], inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"],
                                       inp["scale"], sliding_window=100, mode="2d", block_n=64)
    out, kernel = gpu_util.run_unified(gpu_util.to_dev(inp), inp["scale"], window=100)
    assert "+decode" in kernel and "pack" in kernel, kernel
    torch.testing.assert_close(out.float().cpu(), ref.float(), atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("hq,hk,q_lens", [(32, 8, [4, 1, 3, 2, 4, 4]), (16, 2, [2, 1, 2, 2, 1, 2]), (8, 8, [16, 5, 9, 1, 12, 16])])
def test_multi_token_decode_features_window_softcap_alibi(hq, hk, q_lens):
    """The packed kernel's feature instantiation: per-column window starts (token i of a chunk sees keys from
    ctx_len + i - window + 1 on), soft-cap, ALiBi slopes per column's head; windows shorter than a tile, shorter than
    the chunk, and longer than the context."""
    kv_lens = [700 + q_lens[0], 45, 3 + q_lens[2], q_lens[3], 1030, 65 + q_lens[5]]
    inp = orc.make_paged_inputs(71, q_lens, kv_lens, hq, hk, 128, 16, torch.float16)
    alibi = torch.tensor([2.0 ** (-(i % 8 + 1)) for i in range(hq)], dtype=torch.float32)
    for kw in (dict(window=3), dict(window=8), dict(window=33), dict(window=100), dict(window=5000), dict(softcap=30.0), dict(alibi=alibi),
               dict(window=64, softcap=20.0, alibi=alibi)):
        kernel = _check(inp, torch.float16, force=None, expect="decode_s", **kw)
        assert "pack" in kernel, (kernel, kw)


def test_mixed_batch_with_a_decode_row_hint_uses_two_column_groups():
    """A caller that knows its decode rows carry 1 + k tokens (speculative decoding) says so: rows of up to that many
    tokens - beyond what one column group holds - ride the decode launch on two column groups; a hint the packed kernels
    cannot hold (features, too long) falls back to the library's own threshold. Every row exactly once, against the oracle."""
    import gpu_util
    from mi355_attn import _lib
    from mi355_attn.kernels import unified_attention

    query_lens = [300, 4, 1, 3, 5, 2, 4, 64, 1, 8]
    kv_lens = [900, 640, 17, 2048, 333, 2, 4, 64, 513, 1000]
    for hq, hk, hint, window, want in [(64, 8, 4, 0, "pack2"), (32, 8, 8, 0, "pack2"), (32, 8, 4, 0, "pack"), (32, 8, 8, 100, "pack"), (32, 8, 64, 0, "pack")]:
        inp = orc.make_paged_inputs(63, query_lens, kv_lens, hq, hk, 128, 16, torch.bfloat16)
        ref = orc.unified_attention_oracle(inp["q"], inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"],
                                           inp["scale"], sliding_window=window, mode="2d", block_n=64)
        d = gpu_util.to_dev(inp)
        out = torch.full_like(d["q"], float("nan"))
        unified_attention(q=d["q"], k=d["k_cache"], v=d["v_cache"], out=out, cu_seqlens_q=d["cu_seqlens_q"], max_seqlen_q=max(query_lens),
                          seqused_k=d["seqused_k"], max_seqlen_k=max(kv_lens), avg_seqlen_q=1, avg_seqlen_k=1, softmax_scale=inp["scale"], causal=True,
                          window_size=(window - 1, 0) if window else (-1, -1), block_table=d["block_table"], softcap=0, q_descale=None,
                          k_descale=None, v_descale=None, decode_rows_hint=hint)
        torch.cuda.synchronize()
        kernel = _lib.last_kernel()
        assert "+decode" in kernel and want in kernel and (want != "pack" or "pack2" not in kernel), (kernel, hq, hint, window)
        assert not torch.isnan(out).any(), (hq, hint)
        torch.testing.assert_close(out.float().cpu(), ref.float(), atol=2e-2, rtol=2e-2)
