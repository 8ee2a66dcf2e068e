"""-m gpu: key-split prefill (few Q blocks over a long context: the key tiles of a Q block are dealt to several
workgroups, partial outputs merged by lse). Reference: the CPU oracle (which is pinned to the reference's kernels by the
golden fixtures), at the stated tolerance of the query type; the shape-agnostic kernel is checked beside it."""

import pytest
import torch

import golden_io
from oracle import paged_attention_oracle as orc

pytestmark = pytest.mark.gpu


def _run(inp, dev, force, lse=False, **kw):
    from mi355_attn import _lib
    from mi355_attn.kernels.unified import fill_attn_params, launch

    q = inp["q"].to(dev)
    out = torch.full_like(q, 5.0)
    lse_t = torch.full((q.shape[0], q.shape[1]), 9.0, dtype=torch.float32, device=dev) if lse else None
    p, keep = fill_attn_params(q, kw.pop("k"), kw.pop("v"), out, inp["cu_seqlens_q"].to(dev), kw.pop("max_q"), inp["seqused_k"].to(dev), kw.pop("max_k"),
                               inp["scale"], kw.pop("window", (-1, -1)), inp["block_table"].to(dev), kw.pop("softcap", 0.0), kw.pop("k_scale", None),
                               kw.pop("v_scale", None), kw.pop("alibi", None), force, lse=lse_t, **kw)
    launch(p, dev)
    torch.cuda.synchronize()
    return out, lse_t, _lib.last_kernel()


@pytest.mark.parametrize("case", ["plain", "alibi", "window", "softcap", "fp8", "d64", "d80", "d256", "fp16", "mixed", "two_seqs"])
def test_key_split_prefill_matches_the_oracle(case):
    import gpu_util

    dev = gpu_util.DEV
    dtype = torch.float16 if case == "fp16" else torch.bfloat16
    D = {"d64": 64, "d80": 80, "d256": 256}.get(case, 128)
    Hq, Hk, page = 8, 2, 16
    if case == "mixed":          # prefill chunk + decode rows: prefill rows key-split, decode rows on the split-KV kernel
        query_lens, ctx_lens = [1, 200, 1], [777, 3900, 40]
    elif case == "two_seqs":
        query_lens, ctx_lens = [130, 64], [2500, 4000]
    else:
        query_lens, ctx_lens = [300], [3800]
    kv_lens = [a + b for a, b in zip(query_lens, ctx_lens)]
    inp = orc.make_paged_inputs(101, query_lens, kv_lens, Hq, Hk, D, page, dtype)
    used = torch.zeros(inp["k_cache"].shape[:2], dtype=torch.bool)
    for i, n in enumerate(kv_lens):
        for j in range(n):
            used[inp["block_table"][i, j // page], j % page] = True
    inp["k_cache"][~used] = float("nan")
    inp["v_cache"][~used] = float("nan")
    k, v = inp["k_cache"].to(dev), inp["v_cache"].to(dev)
    kw = {}
    f8 = None
    if case == "fp8":
        f8 = torch.float8_e4m3fn
        k, v = (inp["k_cache"].nan_to_num(0.0).float() / 0.5).to(f8).to(dev), (inp["v_cache"].nan_to_num(0.0).float() / 0.25).to(f8).to(dev)
        kw.update(k_scale=torch.tensor([0.5], device=dev), v_scale=torch.tensor([0.25], device=dev))
    if case == "alibi":
        kw["alibi"] = torch.tensor([0.5 ** (i + 1) for i in range(Hq)], dtype=torch.float32, device=dev)
    if case == "window":
        kw["window"] = (999, 0)
    if case == "softcap":
        kw["softcap"] = 30.0
    common = dict(k=k, v=v, max_q=max(query_lens), max_k=max(kv_lens))
    out, lse, name = _run(inp, dev, None, lse=True, **common, **kw)
    ref, ref_lse, ref_name = _run(inp, dev, 9, lse=True, **common, **kw)
    assert ref_name == "generic"
    assert "_ksplit" in name, name
    if case == "mixed":
        assert "+decode" in name, name
    atol, rtol = golden_io.tolerance(dtype, f8)
    assert not torch.isnan(out).any()
    # the oracle on the same inputs (the fp8 case on the quantised cache the kernel read)
    okw = dict(sliding_window=1000 if case == "window" else 0, softcap=30.0 if case == "softcap" else 0.0,
               alibi_slopes=kw["alibi"].cpu() if case == "alibi" else None)
    if case == "fp8":
        okw.update(k_scale=0.5, v_scale=0.25)
    orc_out = orc.unified_attention_oracle(inp["q"], k.cpu().nan_to_num(0.0) if f8 is None else k.cpu(), v.cpu().nan_to_num(0.0) if f8 is None else v.cpu(),
                                           inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"], inp["scale"], mode="2d", block_n=64, **okw)
    if case == "mixed":      # query_len == 1 rows run on the split-KV decode kernel: the 3D restatement is their oracle
        orc3 = orc.unified_attention_oracle(inp["q"], k.cpu().nan_to_num(0.0), v.cpu().nan_to_num(0.0), inp["cu_seqlens_q"], inp["seqused_k"],
                                            inp["block_table"], inp["scale"], mode="3d")
        for row in (0, 201):
            orc_out[row] = orc3[row]
    torch.testing.assert_close(out.float().cpu(), orc_out.float(), atol=atol, rtol=rtol)
    torch.testing.assert_close(ref.float().cpu(), orc_out.float(), atol=atol, rtol=rtol)
    torch.testing.assert_close(lse, ref_lse, atol=2e-2 if f8 is None else 3e-2, rtol=1e-3)


def test_key_split_prefill_agrees_with_the_unsplit_kernel_at_chunk_size(monkeypatch):
    """512-token chunk of one sequence against 8192 keys (Hq 32 / Hk 8 / D 128, bf16): forced split counts 1 .. 8 agree."""
    import subprocess
    import sys
    import os

    code = r'''
import os, sys, math, torch
sys.path[:0] = [os.environ["MI355_ROOT"], os.path.join(os.environ["MI355_ROOT"], "vllm-triton-backend_amd")]
from mi355_attn import _lib
from mi355_attn.kernels import unified as ua
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(3)
L, QL, Hq, Hk, D, page = 8192, 512, 32, 8, 128, 16
nb = L // page + 3
k = (torch.rand(nb, page, Hk, D, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
v = (torch.rand(nb, page, Hk, D, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
q = (torch.rand(QL, Hq, D, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
bt = torch.randperm(nb, generator=g)[: L // page].to(torch.int32).view(1, -1).to(dev)
cu = torch.tensor([0, QL], dtype=torch.int32, device=dev)
sl = torch.tensor([L], dtype=torch.int32, device=dev)
out = torch.empty_like(q)
p, keep = ua.fill_attn_params(q, k, v, out, cu, QL, sl, L, 1 / math.sqrt(D), (-1, -1), bt, 0.0, None, None, None, None)
ua.launch(p, dev); torch.cuda.synchronize()
print(_lib.last_kernel()); torch.save(out.cpu(), os.environ["MI355_OUT"])
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for splits in ("1", "2", "5", "8", ""):
        env = dict(os.environ, MI355_ROOT=root, MI355_OUT=f"/tmp/ksplit_{splits or 'auto'}.pt", PYTHONDONTWRITEBYTECODE="1")
        if splits:
            env["MI355_PREFILL_KEY_SPLITS"] = splits
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        name = r.stdout.strip().splitlines()[-1]
        assert ("_ksplit" in name) == (splits != "1"), (splits, name)
        outs[splits] = torch.load(env["MI355_OUT"])
    for s, o in outs.items():
        torch.testing.assert_close(o.float(), outs["1"].float(), atol=1e-2, rtol=0, msg=lambda m: f"splits={s}: {m}")
