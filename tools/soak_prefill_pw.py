#!/usr/bin/env python3
"""Soak: random featured batches through prefill_pw_kernel (pinned where it applies) against the shape-agnostic kernel.
Not a test of the suite (minutes on the GPU): python tools/soak_prefill_pw.py [first_seed] [count]"""
import math
import os

os.environ.setdefault("MI355_LAB", "1")      # tools may pin kernels through the library's measurement switches
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vllm-triton-backend_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("MI355_PREFILL", "pw")
import torch  # noqa: E402

import golden_io  # noqa: E402
import gpu_util  # noqa: E402
from oracle import paged_attention_oracle as orc  # noqa: E402


def case(rng):
    n_seq = rng.randint(1, 5)
    q_lens, kv_lens = [], []
    for _ in range(n_seq):
        ql = 1 if rng.random() < 0.2 else rng.choice([2, 17, 64, 65, 129, 257, 500, 700, 1100])
        ctx = rng.choice([0, 0, 1, 63, 64, 300, 1500, 2100, 3000])
        q_lens.append(ql)
        kv_lens.append(ql + ctx)
    hk = rng.choice([1, 2, 4])
    g = rng.choice([1, 2, 3, 4, 5, 8, 12, 16])
    feat = rng.choice(["plain", "window", "softcap", "softcap+window", "alibi", "plain"])
    d = rng.choice([128, 128, 128, 64, 96])
    if d != 128:
        feat = "plain"                     # (head sizes 64 / 96 are on the 64-rows-per-wave kernel without features)
    return dict(d=d, q_lens=q_lens, kv_lens=kv_lens, hq=hk * g, hk=hk, page=rng.choice([16, 16, 32, 64]), dtype=rng.choice([torch.bfloat16, torch.float16]),
                window=rng.choice([9, 64, 300, 1000]) if "window" in feat else 0, softcap=rng.choice([20.0, 50.0]) if "softcap" in feat else 0.0,
                alibi=feat == "alibi", kv_dtype=rng.choice([None, None, torch.float8_e4m3fn, torch.float8_e5m2]))


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    kernels = {}
    for cid in range(first, first + count):
        rng = random.Random(77000 + cid)
        c = case(rng)
        kw = dict(kv_dtype=c["kv_dtype"], kv_scale=0.5) if c["kv_dtype"] is not None else {}
        inp = orc.make_paged_inputs(5000 + cid, c["q_lens"], c["kv_lens"], c["hq"], c["hk"], c["d"], c["page"], c["dtype"], **kw)
        t = gpu_util.to_dev(inp)
        if c["alibi"]:
            t["alibi_slopes"] = torch.tensor([2.0 ** (-(i % 8 + 1)) for i in range(c["hq"])], dtype=torch.float32, device=gpu_util.DEV)
        scale = 1.0 / math.sqrt(c["d"])
        kvs = 0.5 if c["kv_dtype"] is not None else None
        n_tok = t["q"].shape[0]
        ref_lse = torch.full((n_tok, c["hq"]), float("nan"), dtype=torch.float32, device=gpu_util.DEV)
        lse = torch.full_like(ref_lse, float("nan"))
        ref, _ = gpu_util.run_unified(t, scale, window=c["window"], softcap=c["softcap"], kv_scale=kvs, force=9, lse=ref_lse)
        out, kernel = gpu_util.run_unified(t, scale, window=c["window"], softcap=c["softcap"], kv_scale=kvs, lse=lse)
        kernels[kernel] = kernels.get(kernel, 0) + 1
        atol, rtol = golden_io.tolerance(c["dtype"], c["kv_dtype"])
        assert not torch.isnan(out).any(), (cid, kernel, c)
        torch.testing.assert_close(out.float(), ref.float(), atol=atol, rtol=rtol, msg=lambda m: f"case {cid} [{kernel}] {c}\n{m}")
        torch.testing.assert_close(lse, ref_lse, atol=5e-2, rtol=0, msg=lambda m: f"case {cid} [{kernel}] lse {c}\n{m}")
    print(f"{count} cases from seed {first}: all agree; kernels:", dict(sorted(kernels.items(), key=lambda kv: -kv[1])))


if __name__ == "__main__":
    main()
