#!/usr/bin/env python3
"""Microbenchmark + correctness harness: our counterpart of the reference's `scripts/benchmark.py`
prefix test (`test_prefix_vllm_v1_attention`, scripts/benchmark.py:967-1493) for the MI355X backend.

Reproduces, from their semantics (nothing copied):
  * the mixed-batch generator (decode / partial prefill / full prefill shares, prompt pattern,
    DEC_PRE | PRE_DEC | ALTERNATING composition; scripts/benchmark.py:1053-1112);
  * the three timing modes of `measure_benchmarks` (scripts/benchmark.py:1708-1750):
      events  = HIP events around each call with an L2/MALL flush between repetitions
                (triton.testing.do_bench style: ~25 ms warm-up, ~100 ms of repetitions; median, p20, p80)
      graphs  = HIP-graph replay of the call
      end2end = wall clock around call + synchronize, 256 MB flush buffer (scripts/torch_utils.py:35-73);
  * a correctness check against the CPU oracle (tight tolerances, unlike the reference's 2*max_value)
    whenever the batch is small enough for the oracle to finish in seconds;
  * one tab-separated result row per configuration (like the reference's CSV).
Unlike the reference, Python's `random` is seeded and block tables are permutations (no aliasing pages).

    python tools/microbench.py --batch 1 4 16 --seqlen 512 4096 --decode-share 0 0.5 1 --mode events graphs
"""
import argparse
import itertools
import math
import os

os.environ.setdefault("MI355_LAB", "1")      # tools may pin kernels through the library's measurement switches
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vllm-triton-backend_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mi355_attn import _lib  # noqa: E402
from mi355_attn.kernels import unified_attention  # noqa: E402


def make_prefix_batch(batch_size, seqlen, prompt_pattern, decode_share, partial_prefill_share, composition, block_size):
    """query/context lengths per sequence (reference: scripts/benchmark.py:1053-1112)."""
    frac = itertools.cycle(prompt_pattern)
    init = [int(math.ceil(seqlen * next(frac))) for _ in range(batch_size)]
    n_dec = int(math.ceil(batch_size * decode_share))
    n_pre = batch_size - n_dec
    n_part = int(math.ceil(n_pre * partial_prefill_share))
    half = itertools.cycle([p * 0.5 for p in prompt_pattern])
    part_ctx = []
    for l in init:
        c = int(math.ceil(l // block_size * next(half))) * block_size
        part_ctx.append(c - block_size if c == l else c)           # never leave an empty query
    q = [1] * n_dec + [init[i] - part_ctx[i] for i in range(n_dec, n_dec + n_part)] + init[n_dec + n_part:]
    ctx = [l - 1 for l in init[:n_dec]] + part_ctx[n_dec:n_dec + n_part] + [0] * (n_pre - n_part)
    if composition == "PRE_DEC":
        q.reverse(); ctx.reverse()
    elif composition == "ALTERNATING":
        order, rest = [], list(range(len(q)))
        for i in range(len(q) // 2):
            order += [i, len(q) - 1 - i]
            rest.remove(i); rest.remove(len(q) - 1 - i)
        order += rest
        q = [q[i] for i in order]; ctx = [ctx[i] for i in order]
    return q, ctx


def build_inputs(q_lens, ctx_lens, hq, hk, d, page, dtype, dev, seed):
    torch.manual_seed(seed); random.seed(seed); np.random.seed(seed)
    kv = [a + b for a, b in zip(q_lens, ctx_lens)]
    pps = [(n + page - 1) // page for n in kv]
    nb = int(sum(pps) * 1.25) + 4
    k = (torch.rand(nb, page, hk, d, device=dev) * 2 - 1).to(dtype)
    v = (torch.rand(nb, page, hk, d, device=dev) * 2 - 1).to(dtype)
    q = (torch.rand(sum(q_lens), hq, d, device=dev) * 2 - 1).to(dtype)
    perm = torch.randperm(nb, device=dev).to(torch.int32)
    bt = torch.zeros(len(kv), max(pps), dtype=torch.int32, device=dev)
    o = 0
    for i, n in enumerate(pps):
        bt[i, :n] = perm[o:o + n]; o += n
    cu = torch.zeros(len(kv) + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(torch.tensor(q_lens, dtype=torch.int32), 0)
    return dict(q=q, k_cache=k, v_cache=v, block_table=bt, cu_seqlens_q=cu.to(dev), seqused_k=torch.tensor(kv, dtype=torch.int32, device=dev),
                scale=1.0 / math.sqrt(d), q_lens=q_lens, kv_lens=kv)


def make_call(inp, out, force):
    def call():
        unified_attention(q=inp["q"], k=inp["k_cache"], v=inp["v_cache"], out=out, cu_seqlens_q=inp["cu_seqlens_q"], max_seqlen_q=max(inp["q_lens"]),
                          seqused_k=inp["seqused_k"], max_seqlen_k=max(inp["kv_lens"]), avg_seqlen_q=np.mean(inp["q_lens"]),
                          avg_seqlen_k=np.mean(inp["kv_lens"]), softmax_scale=inp["scale"], causal=True, window_size=(-1, -1),
                          block_table=inp["block_table"], softcap=0, q_descale=None, k_descale=None, v_descale=None, force_selection=force)
    return call


def quantiles(ts):
    ts = sorted(ts)
    pick = lambda f: ts[min(len(ts) - 1, int(f * len(ts)))]
    return pick(0.5), pick(0.2), pick(0.8)


def measure(mode, call, dev, warmup_ms=25, rep_ms=100):
    flush = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); call(); e1.record(); torch.cuda.synchronize()
    est = max(e0.elapsed_time(e1), 1e-3)
    n_warm, n_rep = max(1, int(warmup_ms / est)), max(3, min(1000, int(rep_ms / est)))
    if mode == "events":
        for _ in range(n_warm):
            call()
        ts = []
        for _ in range(n_rep):
            flush.zero_()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); call(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return quantiles(ts)
    if mode == "graphs":
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            call()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            call()
        for _ in range(n_warm):
            g.replay()
        ts = []
        for _ in range(n_rep):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); g.replay(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return quantiles(ts)
    if mode == "end2end":
        ts = []
        for _ in range(n_warm + n_rep):
            flush.zero_(); torch.cuda.synchronize()
            t0 = time.perf_counter(); call(); torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return quantiles(ts[n_warm:])
    raise ValueError(mode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[1, 16, 64])
    ap.add_argument("--seqlen", type=int, nargs="+", default=[512, 4096])
    ap.add_argument("--heads", type=str, nargs="+", default=["32,8"])
    ap.add_argument("--head-size", type=int, default=128)
    ap.add_argument("--block-size", type=int, default=16)
    ap.add_argument("--prompt-pattern", type=float, nargs="+", default=[1.0])
    ap.add_argument("--decode-share", type=float, nargs="+", default=[0.0, 0.5, 1.0])
    ap.add_argument("--partial-prefill-share", type=float, nargs="+", default=[0.5])
    ap.add_argument("--composition", nargs="+", default=["ALTERNATING"], choices=["DEC_PRE", "PRE_DEC", "ALTERNATING"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--mode", nargs="+", default=["events"], choices=["events", "graphs", "end2end"])
    ap.add_argument("--impl", nargs="+", default=["auto"], choices=["auto", "2d", "3d", "generic"])
    ap.add_argument("--check-tokens", type=int, default=3000, help="verify against the CPU oracle when the batch has at most this many query tokens * mean kv")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[args.dtype]
    force = {"auto": None, "2d": 2, "3d": 3, "generic": 9}
    cols = ["impl", "mode", "kernel", "batch", "seqlen", "hq", "hk", "d", "block", "decode_share", "partial_share", "composition", "tokens",
            "ms_median", "ms_p20", "ms_p80", "tflops", "kv_gbs", "checked", "max_abs_err"]
    rows = ["\t".join(cols)]
    print(rows[0])
    for heads, b, sl, ds, ps, comp in itertools.product(args.heads, args.batch, args.seqlen, args.decode_share, args.partial_prefill_share, args.composition):
        hq, hk = (int(x) for x in heads.split(","))
        q_lens, ctx_lens = make_prefix_batch(b, sl, args.prompt_pattern, ds, ps, comp, args.block_size)
        inp = build_inputs(q_lens, ctx_lens, hq, hk, args.head_size, args.block_size, dt, dev, seed=0)
        out = torch.zeros_like(inp["q"])
        flops = sum(4 * args.head_size * hq * (ql * cl + ql * (ql + 1) / 2) for ql, cl in zip(q_lens, ctx_lens))
        kv_bytes = sum(inp["kv_lens"]) * hk * args.head_size * 2 * inp["k_cache"].element_size()
        for impl in args.impl:
            call = make_call(inp, out, force[impl])
            call(); torch.cuda.synchronize()
            kernel = _lib.last_kernel()
            checked, err = False, float("nan")
            if sum(q_lens) * np.mean(inp["kv_lens"]) <= args.check_tokens * 1000:
                from oracle import paged_attention_oracle as orc
                ref = orc.unified_attention_oracle(inp["q"].cpu(), inp["k_cache"].cpu(), inp["v_cache"].cpu(), inp["cu_seqlens_q"].cpu(),
                                                   inp["seqused_k"].cpu(), inp["block_table"].cpu(), inp["scale"], block_n=64)
                err = (out.float().cpu() - ref.float()).abs().max().item()
                checked = True
                assert err <= (2e-2 if dt == torch.bfloat16 else 2e-3), f"{impl}/{kernel}: max abs err {err}"
            for mode in args.mode:
                med, p20, p80 = measure(mode, call, dev)
                row = [impl, mode, kernel, b, sl, hq, hk, args.head_size, args.block_size, ds, ps, comp, sum(q_lens), f"{med:.4f}", f"{p20:.4f}",
                       f"{p80:.4f}", f"{flops / (med * 1e-3) / 1e12:.1f}", f"{kv_bytes / (med * 1e-3) / 1e9:.1f}", checked, f"{err:.2e}"]
                rows.append("\t".join(str(x) for x in row))
                print(rows[-1], flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, "w").write("\n".join(rows) + "\n")


if __name__ == "__main__":
    main()
