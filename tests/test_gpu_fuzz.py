"""-m gpu: seeded random differential test. The matrix-core kernels (whatever the dispatcher picks) against the
shape-agnostic VALU kernel (`force_selection=9`, independent code, fp32 math) over random batches, head counts, head
sizes, page sizes, dtypes, fp8 caches and features. The generic kernel itself is pinned by the golden fixtures and the
oracle (tests/test_gpu_golden.py); here it stands in for the oracle so that many cases run in seconds."""

import math
import os
import random

import pytest
import torch

import golden_io
from oracle import paged_attention_oracle as orc

pytestmark = pytest.mark.gpu

# (soak runs: MI355_FUZZ_CASES / MI355_FUZZ_SEED widen and move the sample)
CASES, SEED = int(os.environ.get("MI355_FUZZ_CASES", "200")), int(os.environ.get("MI355_FUZZ_SEED", "1000"))


def _random_case(rng):
    n_seq = rng.randint(1, 6)
    kind = rng.choice(["decode", "prefill", "mixed", "mixed"])
    q_lens, kv_lens = [], []
    for _ in range(n_seq):
        if kind == "decode" or (kind == "mixed" and rng.random() < 0.5):
            ql = 1
        else:
            ql = rng.choice([2, 5, 16, 31, 64, 65, 129, 200, 300])
        ctx = rng.choice([0, 0, 1, 15, 16, 17, 63, 64, 255, 700, 1500, 4097, 6000])
        q_lens.append(ql)
        kv_lens.append(ql + ctx)
    hk = rng.choice([1, 2, 4, 8])
    g = rng.choice([1, 2, 3, 4, 5, 8, 16])
    d = rng.choice([64, 64, 96, 128, 128, 128, 256, 32, 160, 224, 80])
    page = rng.choice([16, 16, 32, 64, 128])
    dtype = rng.choice([torch.bfloat16, torch.float16])
    kv_dtype = rng.choice([None, None, torch.float8_e4m3fn, torch.float8_e5m2]) if d % 16 == 0 else None
    window = rng.choice([0, 0, 0, 7, 64, 300])
    softcap = rng.choice([0.0, 0.0, 0.0, 25.0])
    use_alibi = rng.random() < 0.2
    return dict(q_lens=q_lens, kv_lens=kv_lens, hq=hk * g, hk=hk, d=d, page=page, dtype=dtype, kv_dtype=kv_dtype, window=window,
                softcap=softcap, use_alibi=use_alibi)


@pytest.mark.parametrize("case_id", range(CASES))
def test_fast_kernels_agree_with_the_generic_kernel(case_id):
    import gpu_util

    rng = random.Random(SEED + case_id)
    c = _random_case(rng)
    kw = dict(kv_dtype=c["kv_dtype"], kv_scale=0.5) if c["kv_dtype"] is not None else {}
    inp = orc.make_paged_inputs(SEED + 1000 + case_id, c["q_lens"], c["kv_lens"], c["hq"], c["hk"], c["d"], c["page"], c["dtype"], **kw)
    t = gpu_util.to_dev(inp)
    if c["use_alibi"]:
        t["alibi_slopes"] = torch.tensor([2.0 ** (-(i % 8 + 1)) for i in range(c["hq"])], dtype=torch.float32, device=gpu_util.DEV)
    scale = 1.0 / math.sqrt(c["d"])
    kv_scale = 0.5 if c["kv_dtype"] is not None else None
    n_tok = t["q"].shape[0]
    ref_lse = torch.full((n_tok, c["hq"]), float("nan"), dtype=torch.float32, device=gpu_util.DEV)
    lse = torch.full_like(ref_lse, float("nan"))
    ref, ref_kernel = gpu_util.run_unified(t, scale, window=c["window"], softcap=c["softcap"], kv_scale=kv_scale, force=9, lse=ref_lse)
    assert ref_kernel == "generic"
    out, kernel = gpu_util.run_unified(t, scale, window=c["window"], softcap=c["softcap"], kv_scale=kv_scale, lse=lse)
    assert not torch.isnan(out).any(), (kernel, c)
    atol, rtol = golden_io.tolerance(c["dtype"], c["kv_dtype"])
    torch.testing.assert_close(out.float(), ref.float(), atol=atol, rtol=rtol, msg=lambda m: f"[{kernel}] {c}\n{m}")
    # the second output: log-sum-exp per row (finite everywhere here: every query sees at least its own key)
    assert not torch.isnan(lse).any() and not torch.isnan(ref_lse).any(), (kernel, c)
    torch.testing.assert_close(lse, ref_lse, atol=5e-2, rtol=0, msg=lambda m: f"[{kernel}] lse {c}\n{m}")


# (soak runs: MI355_FUZZ_PACK_CASES / MI355_FUZZ_PACK_SEED widen and move the sample)
PACK_CASES, PACK_SEED = int(os.environ.get("MI355_FUZZ_PACK_CASES", "120")), int(os.environ.get("MI355_FUZZ_PACK_SEED", "7000"))


@pytest.mark.parametrize("case_id", range(PACK_CASES))
def test_packed_multi_token_decode_agrees_with_the_generic_kernel(case_id):
    """Multi-token decode steps (speculative decoding / MTP verification) on the PACK decode kernels: random query heads
    per KV head (1 .. 16), head sizes, page sizes, dtypes, fp8 caches, query lengths up to what two column groups hold
    (and beyond: then selected explicitly, several chunks per sequence), split counts through every merge route."""
    import gpu_util
    from mi355_attn import _lib
    from mi355_attn.kernels import unified as ua_mod

    rng = random.Random(PACK_SEED + case_id)
    g = rng.choice([1, 2, 3, 4, 4, 5, 6, 8, 8, 12, 16])
    hk = rng.choice([1, 2, 8])
    d = rng.choice([64, 128, 128, 256])
    max_q = rng.choice([2, 3, 4, 8, 11, 32])
    n_seq = rng.randint(1, 7)
    q_lens = [rng.randint(1, max_q) for _ in range(n_seq)]
    q_lens[rng.randrange(n_seq)] = max_q
    kv_lens = [ql + rng.choice([0, 0, 1, 14, 15, 16, 17, 31, 32, 33, 63, 255, 700, 1500, 4097]) for ql in q_lens]
    dtype = rng.choice([torch.bfloat16, torch.float16])
    kv_dtype = rng.choice([None, None, torch.float8_e4m3fn, torch.float8_e5m2])
    page = rng.choice([16, 16, 32, 128])
    segments = rng.choice([0, 0, 1, 2, 3, 8, 33])
    window = rng.choice([0, 0, 0, 0, 3, 40, 300])
    softcap = rng.choice([0.0, 0.0, 0.0, 25.0])
    use_alibi = rng.random() < 0.15
    feat = window > 0 or softcap > 0 or use_alibi
    kw = dict(kv_dtype=kv_dtype, kv_scale=0.5) if kv_dtype is not None else {}
    inp = orc.make_paged_inputs(PACK_SEED + case_id, q_lens, kv_lens, hk * g, hk, d, page, dtype, **kw)
    t = gpu_util.to_dev(inp)
    scale = 1.0 / math.sqrt(d)
    n_tok = t["q"].shape[0]
    ref_lse = torch.full((n_tok, hk * g), float("nan"), dtype=torch.float32, device=gpu_util.DEV)
    if use_alibi:
        t["alibi_slopes"] = torch.tensor([2.0 ** (-(i % 8 + 1)) for i in range(hk * g)], dtype=torch.float32, device=gpu_util.DEV)
    ref, ref_kernel = gpu_util.run_unified(t, scale, window=window, softcap=softcap, kv_scale=0.5 if kv_dtype is not None else None, force=9, lse=ref_lse)
    assert ref_kernel == "generic"
    out = torch.full_like(t["q"], float("nan"))
    lse = torch.full_like(ref_lse, float("nan"))
    descale = torch.tensor([0.5], dtype=torch.float32, device=gpu_util.DEV) if kv_dtype is not None else None
    p, keep = ua_mod.fill_attn_params(t["q"], t["k_cache"], t["v_cache"], out, t["cu_seqlens_q"], max(q_lens), t["seqused_k"], max(kv_lens),
                                      scale, (window - 1, 0) if window else (-1, -1), t["block_table"], softcap, descale, descale, t.get("alibi_slopes"), 3,
                                      num_segments=segments, lse=lse)
    ua_mod.launch(p, gpu_util.DEV)
    torch.cuda.synchronize()
    kernel = _lib.last_kernel()
    packable = max(q_lens) > 1 and sum(q_lens) > n_seq and (g <= 8 or (d <= 128 and not feat))      # features: one column group
    assert ("pack" in kernel) == packable, (kernel, g, d, q_lens)
    assert not torch.isnan(out).any(), (kernel, q_lens, kv_lens)
    atol, rtol = golden_io.tolerance(dtype, kv_dtype)
    case = dict(g=g, hk=hk, d=d, q_lens=q_lens, kv_lens=kv_lens, page=page, segments=segments, kv_dtype=kv_dtype, window=window, softcap=softcap, alibi=use_alibi)
    torch.testing.assert_close(out.float(), ref.float(), atol=atol, rtol=rtol, msg=lambda m: f"[{kernel}] {case}\n{m}")
    torch.testing.assert_close(lse, ref_lse, atol=5e-2, rtol=0, msg=lambda m: f"[{kernel}] lse {case}\n{m}")


@pytest.mark.parametrize("q_lens,kv_lens", [([5, 0, 1, 0, 64], [70, 33, 45, 0, 64]), ([0, 1, 1], [16, 300, 1]), ([0, 200], [0, 777])])
def test_sequences_without_query_tokens_are_skipped(q_lens, kv_lens):
    """A sequence may contribute no query token to a step (equal neighbours in cu_seqlens_q): nothing is computed for it and
    the other sequences' rows are unaffected - fast kernels vs the generic kernel."""
    import gpu_util

    dtype = torch.bfloat16
    keep = [i for i, n in enumerate(q_lens) if n > 0]
    inp = orc.make_paged_inputs(3000, [q_lens[i] for i in keep], [kv_lens[i] for i in keep], 8, 2, 128, 16, dtype)
    # re-insert the empty sequences: same q / cache, longer metadata
    cu, sl, bt_rows, j = [0], [], [], 0
    for i, n in enumerate(q_lens):
        cu.append(cu[-1] + n)
        sl.append(kv_lens[i])
        if n > 0:
            bt_rows.append(inp["block_table"][j])
            j += 1
        else:
            bt_rows.append(torch.zeros_like(inp["block_table"][0]))
    t = gpu_util.to_dev(inp)
    t["cu_seqlens_q"] = torch.tensor(cu, dtype=torch.int32, device=gpu_util.DEV)
    t["seqused_k"] = torch.tensor(sl, dtype=torch.int32, device=gpu_util.DEV)
    t["block_table"] = torch.stack(bt_rows).to(gpu_util.DEV)
    scale = 1.0 / math.sqrt(128)
    ref, _ = gpu_util.run_unified(t, scale, force=9)
    out, kernel = gpu_util.run_unified(t, scale)
    assert kernel != "generic"
    assert not torch.isnan(out).any() and not torch.isnan(ref).any()
    torch.testing.assert_close(out.float(), ref.float(), atol=2e-2, rtol=2e-2, msg=lambda m: f"[{kernel}] {m}")


# ---- the same differential fuzz against the ORACLE (the legs above are kernel-vs-kernel). Small shapes so that
# the CPU restatement of the reference's kernels (oracle/paged_attention_oracle.py, pinned by tests/golden) takes about a second
# per case; 48 cases, every feature and cache type of the generator above, whatever kernel the dispatcher picks.
ORACLE_CASES, ORACLE_SEED = int(os.environ.get("MI355_FUZZ_ORACLE_CASES", "48")), int(os.environ.get("MI355_FUZZ_ORACLE_SEED", "9000"))


def _small_case(rng):
    n_seq = rng.randint(1, 4)
    kind = rng.choice(["decode", "prefill", "mixed", "mixed"])
    q_lens, kv_lens = [], []
    for _ in range(n_seq):
        ql = 1 if (kind == "decode" or (kind == "mixed" and rng.random() < 0.5)) else rng.choice([2, 5, 16, 31, 64, 65, 129])
        q_lens.append(ql)
        kv_lens.append(ql + rng.choice([0, 0, 1, 15, 16, 17, 63, 64, 255, 600]))
    hk = rng.choice([1, 2, 4])
    g = rng.choice([1, 2, 3, 4, 8])
    d = rng.choice([64, 96, 128, 128, 128, 256, 80])
    page = rng.choice([16, 16, 32, 128])
    dtype = rng.choice([torch.bfloat16, torch.float16])
    kv_dtype = rng.choice([None, None, torch.float8_e4m3fn, torch.float8_e5m2]) if d % 16 == 0 else None
    return dict(q_lens=q_lens, kv_lens=kv_lens, hq=hk * g, hk=hk, d=d, page=page, dtype=dtype, kv_dtype=kv_dtype,
                window=rng.choice([0, 0, 0, 7, 64, 300]), softcap=rng.choice([0.0, 0.0, 0.0, 25.0]), use_alibi=rng.random() < 0.2)


@pytest.mark.parametrize("case_id", range(ORACLE_CASES))
def test_dispatched_kernels_agree_with_the_oracle(case_id):
    import gpu_util

    rng = random.Random(ORACLE_SEED + case_id)
    c = _small_case(rng)
    kv_scale = 0.5 if c["kv_dtype"] is not None else None
    kw = dict(kv_dtype=c["kv_dtype"], kv_scale=0.5) if c["kv_dtype"] is not None else {}
    inp = orc.make_paged_inputs(ORACLE_SEED + 500 + case_id, c["q_lens"], c["kv_lens"], c["hq"], c["hk"], c["d"], c["page"], c["dtype"], **kw)
    alibi = torch.tensor([2.0 ** (-(i % 8 + 1)) for i in range(c["hq"])], dtype=torch.float32) if c["use_alibi"] else None
    # (mode: the reference's 2D kernel for batches with a prefill, its 3D kernel + reduce_segments for decode batches, :884)
    ref = orc.unified_attention_oracle(inp["q"], inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"], inp["scale"],
                                       sliding_window=c["window"], softcap=c["softcap"], alibi_slopes=alibi, k_scale=kv_scale or 1.0, v_scale=kv_scale or 1.0,
                                       mode="3d" if max(c["q_lens"]) == 1 else "2d")
    t = gpu_util.to_dev(inp)
    if alibi is not None:
        t["alibi_slopes"] = alibi.to(gpu_util.DEV)
    out, kernel = gpu_util.run_unified(t, inp["scale"], window=c["window"], softcap=c["softcap"], kv_scale=kv_scale)
    assert not torch.isnan(out).any(), (kernel, c)
    atol, rtol = golden_io.tolerance(c["dtype"], c["kv_dtype"])
    torch.testing.assert_close(out.float().cpu(), ref.float(), atol=atol, rtol=rtol, msg=lambda m: f"[{kernel}] {c}\n{m}")
