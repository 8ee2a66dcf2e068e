#!/usr/bin/env python3
"""A vLLM-shaped step over an fp8 cache (VERDICT r03 'missing' 2): 48 decode rows + one 2048-token chunk over 2048 keys of
context, Llama shape (Hq 32 / Hk 8 / D 128), e4m3 KV. HIP-graph replay of the one `unified_attention` call; the prefill
rows' FLOPs (visible keys) over the whole step's time.   python tools/bench_fp8_step.py [--kvdtype fp8|same] [--chunk 2048] [--ctx 2048] [--decodes 48]"""
import argparse
import math
import os

os.environ.setdefault("MI355_LAB", "1")
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vllm-triton-backend_amd")]
import torch  # noqa: E402

from mi355_attn import _lib  # noqa: E402
from mi355_attn.kernels import unified as ua_mod  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kvdtype", default="fp8", choices=["fp8", "same"])
    ap.add_argument("--chunk", type=int, default=2048)
    ap.add_argument("--ctx", type=int, default=2048)
    ap.add_argument("--decodes", type=int, default=48)
    ap.add_argument("--decode-kv", type=int, default=2048)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    hq, hk, d, page = 32, 8, 128, 16
    q_lens = [1] * args.decodes + [args.chunk]
    kv_lens = [args.decode_kv - 13 * i for i in range(args.decodes)] + [args.ctx + args.chunk]
    pages = [(n + page - 1) // page for n in kv_lens]
    nb = int(sum(pages) * 1.2)
    kvdt = torch.float8_e4m3fn if args.kvdtype == "fp8" else torch.bfloat16
    k = (torch.rand(nb, page, hk, d, device=dev) * 2 - 1).to(kvdt)
    v = (torch.rand(nb, page, hk, d, device=dev) * 2 - 1).to(kvdt)
    perm = torch.randperm(nb, device=dev).to(torch.int32)
    bt = torch.zeros(len(q_lens), max(pages), dtype=torch.int32, device=dev)
    o = 0
    for i, n in enumerate(pages):
        bt[i, :n] = perm[o:o + n]
        o += n
    T = sum(q_lens)
    q = (torch.rand(T, hq, d, device=dev) * 2 - 1).bfloat16()
    out = torch.empty_like(q)
    cu = torch.tensor([0] + list(torch.tensor(q_lens).cumsum(0)), dtype=torch.int32, device=dev)
    sl = torch.tensor(kv_lens, dtype=torch.int32, device=dev)
    sc = torch.ones(1, device=dev) if args.kvdtype == "fp8" else None
    p, keep = ua_mod.fill_attn_params(q, k, v, out, cu, max(q_lens), sl, max(kv_lens), 1.0 / math.sqrt(d), (-1, -1), bt, 0.0, sc, sc, None, None)
    import ctypes
    ws = _lib.load().mi355_attn_workspace_bytes(ctypes.byref(p))
    for _ in range(3):
        ua_mod.launch(p, dev)
    torch.cuda.synchronize()
    kernel = _lib.last_kernel()
    gs = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(gs):
        ua_mod.launch(p, dev)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=gs):
            for _ in range(20):
                ua_mod.launch(p, dev)
        for _ in range(5):
            g.replay()
        e0.record(gs)
        for _ in range(10):
            g.replay()
        e1.record(gs)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    flops = 4 * d * hq * (args.chunk * args.ctx + args.chunk * (args.chunk + 1) / 2)
    dec_bytes = sum(kv_lens[: args.decodes]) * hk * d * 2 * (1 if args.kvdtype == "fp8" else 2)
    print(f"{args.decodes} decode rows (~{args.decode_kv} keys) + a {args.chunk}-token chunk over {args.ctx} keys, kv {args.kvdtype}: kernel={kernel} "
          f"workspace {ws} B; {us:7.1f} us per step (graph replay) = {flops / us / 1e6:7.1f} TFLOP/s of the chunk's FLOPs over the whole step "
          f"({flops / us / 1e6 / 2500:5.3f} of 2.5 PF; the decode rows read {dec_bytes / 1e6:.1f} MB)", flush=True)


if __name__ == "__main__":
    main()
