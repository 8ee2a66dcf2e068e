"""-m gpu: seeded random differential test. The matrix-core kernels (whatever the dispatcher picks) against the
shape-agnostic VALU kernel (`force_selection=9`, independent code, fp32 math) over random batches, head counts, head
sizes, page sizes, dtypes, fp8 caches and features. The generic kernel itself is pinned by the golden fixtures and the
oracle (tests/test_gpu_golden.py); here it stands in for the oracle so that many cases run in seconds."""

import math
import random

import pytest
import torch

import golden_io
from oracle import paged_attention_oracle as orc

pytestmark = pytest.mark.gpu

CASES = 200


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

    rng = random.Random(1000 + case_id)
    c = _random_case(rng)
    kw = dict(kv_dtype=c["kv_dtype"], kv_scale=0.5) if c["kv_dtype"] is not None else {}
    inp = orc.make_paged_inputs(2000 + case_id, c["q_lens"], c["kv_lens"], c["hq"], c["hk"], c["d"], c["page"], c["dtype"], **kw)
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
