#!/usr/bin/env python3
"""C4: Granite-3.1-8B-shape (Hq 32 / Hk 8 / D 128) mixed chunked-prefill + decode batch, generator of the
reference harness (scripts/benchmark.py:1053-1112: batch 64, seqlen 4096, decode_share 0.5,
partial_prefill_share 0.5, ALTERNATING) -> 32 decodes (ctx 4095), 16 partial prefills (ctx 2048 + q 2048),
16 full prefills (q 4096). Times the whole batch on one GPU and the share one of 8 GPUs would get
under batch sharding (mi355_attn.parallel.assign_sequences)."""
import math
import os

os.environ.setdefault("MI355_LAB", "1")      # tools may pin kernels through the library's measurement switches
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vllm-triton-backend_amd")]
import torch  # noqa: E402

from mi355_attn import _lib, parallel  # noqa: E402
from mi355_attn.kernels import unified as ua  # noqa: E402


def c4_lens(batch=64, seqlen=4096, page=16):
    dec = batch // 2
    pre = batch - dec
    part = pre // 2
    q = [1] * dec + [seqlen // 2] * part + [seqlen] * (pre - part)
    ctx = [seqlen - 1] * dec + [seqlen // 2] * part + [0] * (pre - part)
    order = []
    for i in range(batch // 2):
        order += [i, batch - 1 - i]
    q = [q[i] for i in order]
    ctx = [ctx[i] for i in order]
    return q, [a + b for a, b in zip(q, ctx)]


HQ, HK = 32, 8
WINDOW = 0          # sliding window (keys), 0 = none


def run(qlens, kvlens, dev, iters=10):
    Hq, Hk, D, page = HQ, HK, 128, 16
    S, T = len(qlens), sum(qlens)
    pps = [(n + page - 1) // page for n in kvlens]
    nb = sum(pps) + 8
    k = (torch.rand(nb, page, Hk, D, device=dev) * 2 - 1).bfloat16()
    v = (torch.rand(nb, page, Hk, D, device=dev) * 2 - 1).bfloat16()
    q = (torch.rand(T, Hq, D, device=dev) * 2 - 1).bfloat16()
    perm = torch.randperm(nb, device=dev).to(torch.int32)
    bt = torch.zeros(S, max(pps), dtype=torch.int32, device=dev)
    o = 0
    for i, n in enumerate(pps):
        bt[i, :n] = perm[o:o + n]
        o += n
    cu = torch.zeros(S + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(torch.tensor(qlens, dtype=torch.int32), 0)
    cu = cu.to(dev)
    sl = torch.tensor(kvlens, dtype=torch.int32, device=dev)
    out = torch.empty_like(q)
    p, keep = ua.fill_attn_params(q, k, v, out, cu, max(qlens), sl, max(kvlens), 1 / math.sqrt(D), (WINDOW - 1, 0) if WINDOW else (-1, -1), bt, 0.0, None, None, None, None)
    for _ in range(3):
        ua.launch(p, dev)
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ua.launch(p, dev); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    ts.sort()
    t = ts[len(ts) // 2]
    flops = sum(4 * D * Hq * (ql * (kl - ql) + ql * (ql + 1) / 2) for ql, kl in zip(qlens, kvlens))
    byts = sum(kl * Hk * D * 4 for kl in kvlens) + 2 * T * Hq * D * 2
    return t, flops, byts, _lib.last_kernel()


def main():
    dev = torch.device("cuda:0")
    qlens, kvlens = c4_lens()
    print(f"C4: {len(qlens)} seqs, {sum(qlens)} query tokens, {sum((n + 15) // 16 for n in kvlens)} pages")
    t, fl, by, kern = run(qlens, kvlens, dev)
    print(f"whole batch on 1 GPU : {t*1e3:8.3f} ms  {fl/t/1e12:7.1f} TFLOP/s  {by/t/1e9:7.1f} GB/s  kernel={kern}")
    owned = parallel.assign_sequences(qlens, kvlens, 8)
    worst = 0.0
    for r, ids in enumerate(owned):
        t, fl, by, kern = run([qlens[i] for i in ids], [kvlens[i] for i in ids], dev)
        worst = max(worst, t)
        print(f"rank {r} share ({len(ids)} seqs, {sum(qlens[i] for i in ids)} tokens): {t*1e3:7.3f} ms  {fl/t/1e12:7.1f} TFLOP/s")
    tot_fl = sum(4 * 128 * 32 * (ql * (kl - ql) + ql * (ql + 1) / 2) for ql, kl in zip(qlens, kvlens))
    print(f"8-way batch-sharded (max over ranks, each share measured on this one GPU): {worst*1e3:7.3f} ms -> {tot_fl/worst/1e12:7.1f} TFLOP/s aggregate")


if __name__ == "__main__":
    main()
