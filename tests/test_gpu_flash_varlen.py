"""-m gpu: `prefill_flash_attention` (the reference's non-paged varlen prefill op, LIB/kernels/triton_flash_attention.py:
1326-1484) against an independent dense float64 softmax, incl. seqlen_q < seqlen_k (bottom-right aligned causal mask),
grouped-query heads, lengths that are not multiples of the 16-token scratch page, and in-place output."""

import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _dense_reference(q, k, v, cu_q, cu_k, scale):
    Hq, Hk = q.shape[1], k.shape[1]
    G = Hq // Hk
    out = torch.zeros(q.shape, dtype=torch.float64)
    for i in range(len(cu_q) - 1):
        q0, q1, k0, k1 = cu_q[i], cu_q[i + 1], cu_k[i], cu_k[i + 1]
        lq, lk = q1 - q0, k1 - k0
        qp = torch.arange(lq)[:, None] + (lk - lq)
        mask = torch.arange(lk)[None, :] <= qp
        for h in range(Hq):
            s = scale * (q[q0:q1, h].double() @ k[k0:k1, h // G].double().T)
            s = s.masked_fill(~mask, float("-inf"))
            p = torch.softmax(s, dim=-1)
            p = torch.nan_to_num(p, nan=0.0)            # rows that see no key (lq > lk) return 0
            out[q0:q1, h] = p @ v[k0:k1, h // G].double()
    return out


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hq,hk,d", [(8, 2, 128), (4, 4, 64), (8, 1, 96)])
def test_varlen_causal_prefill(dtype, hq, hk, d):
    from mi355_attn import _lib
    from mi355_attn.kernels import prefill_flash_attention

    g = torch.Generator().manual_seed(70 + hq + d)
    q_lens = [129, 1, 64, 200, 17]
    k_lens = [129, 45, 257, 200, 33]
    cu_q = [0] + torch.tensor(q_lens).cumsum(0).tolist()
    cu_k = [0] + torch.tensor(k_lens).cumsum(0).tolist()
    q = (torch.rand(cu_q[-1], hq, d, generator=g) * 2 - 1).to(dtype)
    k = (torch.rand(cu_k[-1], hk, d, generator=g) * 2 - 1).to(dtype)
    v = (torch.rand(cu_k[-1], hk, d, generator=g) * 2 - 1).to(dtype)
    scale = 1.0 / math.sqrt(d)
    ref = _dense_reference(q, k, v, cu_q, cu_k, scale)
    dev = torch.device("cuda:0")
    cq = torch.tensor(cu_q, dtype=torch.int32, device=dev)
    ck = torch.tensor(cu_k, dtype=torch.int32, device=dev)
    out = prefill_flash_attention(q.to(dev), k.to(dev), v.to(dev), max(q_lens), max(k_lens), cq, ck, causal=True, sm_scale=scale)
    torch.cuda.synchronize()
    assert _lib.last_kernel().startswith("prefill_mfma"), _lib.last_kernel()
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-3
    torch.testing.assert_close(out.double().cpu(), ref, atol=tol, rtol=tol)
    # in-place output buffer
    buf = torch.full_like(q.to(dev), float("nan"))
    ret = prefill_flash_attention(q.to(dev), k.to(dev), v.to(dev), max(q_lens), max(k_lens), cq, ck, causal=True, sm_scale=scale,
                                  in_place_output=buf)
    assert ret is buf
    torch.testing.assert_close(buf.double().cpu(), ref, atol=tol, rtol=tol)


def test_varlen_unsupported_modes_raise():
    from mi355_attn.kernels import prefill_flash_attention

    dev = torch.device("cuda:0")
    q = torch.zeros(16, 4, 64, dtype=torch.bfloat16, device=dev)
    cu = torch.tensor([0, 16], dtype=torch.int32, device=dev)
    # bias: the reference's own wrapper asserts it away for the variable-length layout it always sets
    # (triton_flash_attention.py:126-128, :1341-1342): AssertionError there, AssertionError here
    with pytest.raises(AssertionError):
        prefill_flash_attention(q, q, q, 16, 16, cu, cu, causal=False, bias=torch.zeros(1, device=dev))
    with pytest.raises(AssertionError):
        prefill_flash_attention(q, q, q, 16, 16, cu, cu, causal=True, bias=torch.zeros(1, device=dev))
    with pytest.raises(NotImplementedError):
        prefill_flash_attention(q, q, q, 16, 16, cu, cu, causal=True, do_not_return_softmax_encodings=False)


import golden_io  # noqa: E402


@pytest.mark.parametrize("name", golden_io.names("flash_varlen"))
def test_varlen_prefill_against_the_reference_kernels_outputs(name):
    """Fixtures written by the reference's own attn_fwd (triton_wrapper_forward_prefill under the Triton interpreter,
    tests/golden/make_golden.py::flash_cases): fp32 on the shape-agnostic kernel, fp16 on the MFMA prefill kernels."""
    from mi355_attn import _lib
    from mi355_attn.kernels import prefill_flash_attention

    meta, t = golden_io.load(name)
    dev = torch.device("cuda:0")
    out = prefill_flash_attention(t["q"].to(dev), t["k"].to(dev), t["v"].to(dev), meta["max_seqlen_q"], meta["max_seqlen_k"],
                                  t["cu_seqlens_q"].to(dev), t["cu_seqlens_k"].to(dev), causal=meta["causal"], sm_scale=meta["scale"])
    torch.cuda.synchronize()
    if t["q"].dtype == torch.float16 and meta["causal"]:
        assert _lib.last_kernel().startswith("prefill_mfma"), _lib.last_kernel()
    if not meta["causal"] and t["q"].dtype == torch.float32:
        assert _lib.last_kernel() == "generic", _lib.last_kernel()      # f32: the shape-agnostic kernel
    atol, rtol = golden_io.tolerance(t["q"].dtype)
    torch.testing.assert_close(out.float().cpu(), t["out"].float(), atol=atol, rtol=rtol)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("causal", [True, False])
def test_self_attention_prefill_reads_the_linear_tensors_in_one_library_call(dtype, causal, monkeypatch):
    """Q and K/V share `cu_seqlens` (the reference harness's use, scripts/callers/triton_3d.py:100-112): ONE C-ABI call
    with the linear k / v as the library's new-token source, no scratch tensor or cache write on the Python side - a
    sequence of length 1 included (its only key is in the linear tensors too)."""
    from mi355_attn import _lib
    from mi355_attn.kernels import flash, prefill_flash_attention

    hq, hk, d = 8, 2, 128
    g = torch.Generator().manual_seed(91)
    lens = [129, 1, 64, 200, 17, 1]
    cu = [0] + torch.tensor(lens).cumsum(0).tolist()
    q = (torch.rand(cu[-1], hq, d, generator=g) * 2 - 1).to(dtype)
    k = (torch.rand(cu[-1], hk, d, generator=g) * 2 - 1).to(dtype)
    v = (torch.rand(cu[-1], hk, d, generator=g) * 2 - 1).to(dtype)
    scale = 1.0 / math.sqrt(d)
    if causal:
        ref = _dense_reference(q, k, v, cu, cu, scale)
    else:
        ref = torch.zeros(q.shape, dtype=torch.float64)
        for i in range(len(lens)):
            a, b = cu[i], cu[i + 1]
            for h in range(hq):
                s = scale * (q[a:b, h].double() @ k[a:b, h // (hq // hk)].double().T)
                ref[a:b, h] = torch.softmax(s, dim=-1) @ v[a:b, h // (hq // hk)].double()
    dev = torch.device("cuda:0")
    cud = torch.tensor(cu, dtype=torch.int32, device=dev)
    lib = _lib.load()
    calls = {"attn": 0, "cache": 0}

    class Counting:
        def __init__(self, fn, key):
            self.fn, self.key = fn, key
            self.restype, self.argtypes = fn.restype, fn.argtypes

        def __call__(self, *a):
            calls[self.key] += 1
            return self.fn(*a)

    monkeypatch.setattr(lib, "mi355_unified_attention", Counting(lib.mi355_unified_attention, "attn"), raising=False)
    monkeypatch.setattr(lib, "mi355_reshape_and_cache_flash", Counting(lib.mi355_reshape_and_cache_flash, "cache"), raising=False)
    n_scratch = len(flash._scratch)
    out = prefill_flash_attention(q.to(dev), k.to(dev), v.to(dev), max(lens), max(lens), cud, cud, causal=causal, sm_scale=scale)
    torch.cuda.synchronize()
    assert calls == {"attn": 1, "cache": 0}, calls
    assert len(flash._scratch) == n_scratch
    assert _lib.last_kernel().startswith("repack+prefill_mfma_pw" if not causal else "repack+prefill_"), _lib.last_kernel()
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-3
    torch.testing.assert_close(out.double().cpu(), ref, atol=tol, rtol=tol)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("lens", [[2] * 6, [3] * 5, [4] * 7, [2, 3, 4, 1, 4, 2], [2, 300, 4, 3, 129, 1, 2, 700], [4, 4, 4, 2048]])
def test_non_causal_sequences_of_a_few_tokens_see_all_their_keys(dtype, lens):
    """Non-causal varlen attention whose sequences carry 2..4 tokens - uniform, ragged, and mixed with long ones - is
    what a packed multi-token DECODE step looks like to the dispatch (a few query tokens per sequence), and the decode
    kernels mask causally: such a call must never reach them (only one-token rows are the same under both
    masks). Against a dense float64 softmax over every key of the sequence."""
    from mi355_attn import _lib
    from mi355_attn.kernels import prefill_flash_attention

    hq, hk, d = 32, 8, 128             # G = 4: one packed column group holds 4 tokens, two hold 8
    g = torch.Generator().manual_seed(97 + len(lens))
    cu = [0] + torch.tensor(lens).cumsum(0).tolist()
    q = (torch.rand(cu[-1], hq, d, generator=g) * 2 - 1).to(dtype)
    k = (torch.rand(cu[-1], hk, d, generator=g) * 2 - 1).to(dtype)
    v = (torch.rand(cu[-1], hk, d, generator=g) * 2 - 1).to(dtype)
    scale = 1.0 / math.sqrt(d)
    ref = torch.zeros(q.shape, dtype=torch.float64)
    for i in range(len(lens)):
        a, b = cu[i], cu[i + 1]
        for h in range(hq):
            s = scale * (q[a:b, h].double() @ k[a:b, h // (hq // hk)].double().T)
            ref[a:b, h] = torch.softmax(s, dim=-1) @ v[a:b, h // (hq // hk)].double()
    dev = torch.device("cuda:0")
    cud = torch.tensor(cu, dtype=torch.int32, device=dev)
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-3
    # self-attention (one library call over the linear tensors) and the two-range form (scratch pages + unified_attention's launch)
    out = prefill_flash_attention(q.to(dev), k.to(dev), v.to(dev), max(lens), max(lens), cud, cud, causal=False, sm_scale=scale)
    torch.cuda.synchronize()
    # (one-token rows may ride a decode launch: they are the same under both masks; no PACKED multi-token launch may appear)
    assert "_pack" not in _lib.last_kernel(), _lib.last_kernel()
    torch.testing.assert_close(out.double().cpu(), ref, atol=tol, rtol=tol)
    out2 = prefill_flash_attention(q.to(dev), k.to(dev), v.to(dev), max(lens), max(lens), cud, cud.clone(), causal=False, sm_scale=scale)
    torch.cuda.synchronize()
    assert "_pack" not in _lib.last_kernel(), _lib.last_kernel()
    torch.testing.assert_close(out2.double().cpu(), ref, atol=tol, rtol=tol)
