#!/usr/bin/env python3
"""Quick prefill microbench (config C2 by default): HIP-event timing of mi355_attn.unified_attention."""
import argparse
import math
import os

os.environ.setdefault("MI355_LAB", "1")      # tools may pin kernels through the library's measurement switches
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vllm-triton-backend_amd")]
import torch  # noqa: E402

from mi355_attn import _lib  # noqa: E402
from mi355_attn.kernels import unified as ua_mod  # noqa: E402


def main():
    if os.environ.get("MI355_LIB"):        # A/B: benchmark another build of the library on the same box
        _lib.LIB_PATH = os.path.abspath(os.environ["MI355_LIB"])
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--seq", type=int, default=4096)
    ap.add_argument("--hq", type=int, default=32)
    ap.add_argument("--hk", type=int, default=8)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--page", type=int, default=16)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--kvdtype", default="same", choices=["same", "fp8", "fp8_e5m2"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--window", type=int, default=0, help="sliding window (keys per query row), 0 = off; FLOPs count the visible keys only")
    ap.add_argument("--softcap", type=float, default=0.0)
    ap.add_argument("--alibi", action="store_true", help="ALiBi slopes 2^(-8 (i + 1) / Hq)")
    ap.add_argument("--segments", type=int, default=0, help="num_segments of the call (1 = no key split)")
    ap.add_argument("--legacy", action="store_true", help="context_attention_fwd: v0 cache layout for the first --ctx keys, the rest from linear k/v")
    ap.add_argument("--ctx", type=int, default=0, help="context keys per sequence already in the cache (query length = seq - ctx)")
    ap.add_argument("--generic", action="store_true", help="with --legacy: force the shape-agnostic kernel (the path before the repack)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    torch.manual_seed(0)
    B, L, page = args.batch, args.seq, args.page
    pps = (L + page - 1) // page
    nb = int(B * pps * 1.25)
    kvdt = {"same": dt, "fp8": torch.float8_e4m3fn, "fp8_e5m2": torch.float8_e5m2}[args.kvdtype]
    k = (torch.rand(nb, page, args.hk, args.d, device=dev) * 2 - 1).to(kvdt)
    v = (torch.rand(nb, page, args.hk, args.d, device=dev) * 2 - 1).to(kvdt)
    ksc = torch.ones(1, device=dev) if kvdt != dt else None
    q = (torch.rand(B * L, args.hq, args.d, device=dev) * 2 - 1).to(dt)
    bt = torch.randperm(nb, device=dev)[: B * pps].to(torch.int32).view(B, pps)
    cu = (torch.arange(B + 1, dtype=torch.int32, device=dev) * L).to(torch.int32)
    sl = torch.full((B,), L, dtype=torch.int32, device=dev)
    out = torch.empty_like(q)
    flops = 4 * L * L * args.d * args.hq / 2 * B
    win = (args.window - 1, 0) if args.window else (-1, -1)
    if args.window:          # row at position i sees min(i + 1, window) keys
        W = args.window
        vis = sum(min(i + 1, W) for i in range(L))
        flops = 4 * args.d * args.hq * vis * B
    slopes = torch.tensor([2.0 ** (-(i + 1) * 8.0 / args.hq) for i in range(args.hq)], dtype=torch.float32, device=q.device) if args.alibi else None
    p, keep = ua_mod.fill_attn_params(q, k, v, out, cu, L, sl, L, 1.0 / math.sqrt(args.d), win, bt, args.softcap, ksc, ksc, slopes, None,
                                      num_segments=args.segments)
    if args.ctx and not args.legacy:     # chunked prefill through unified_attention: the last seq - ctx tokens are the queries
        QL = L - args.ctx
        q = q[: B * QL].contiguous()
        out = torch.empty_like(q)
        cu = (torch.arange(B + 1, dtype=torch.int32, device=dev) * QL).to(torch.int32)
        flops = 4 * args.d * args.hq * (QL * args.ctx + QL * (QL + 1) / 2) * B
        p, keep = ua_mod.fill_attn_params(q, k, v, out, cu, QL, sl, L, 1.0 / math.sqrt(args.d), (-1, -1), bt, 0.0, ksc, ksc, None, None)
    if args.legacy:
        QL = L - args.ctx
        q = q[: B * QL].contiguous()
        out = torch.empty_like(q)
        cu = (torch.arange(B + 1, dtype=torch.int32, device=dev) * QL).to(torch.int32)
        k0 = k.view(nb, page, args.hk, args.d // 8, 8).permute(0, 2, 3, 1, 4).contiguous()
        v0 = v.permute(0, 2, 3, 1).contiguous()
        k_new = (torch.rand(B * QL, args.hk, args.d, device=dev) * 2 - 1).to(dt)
        v_new = (torch.rand(B * QL, args.hk, args.d, device=dev) * 2 - 1).to(dt)
        flops = 4 * args.d * args.hq * (QL * args.ctx + QL * (QL + 1) / 2) * B
        p, keep = ua_mod.fill_attn_params(q, k0, v0, out, cu, QL, sl, pps * page + QL, 1.0 / math.sqrt(args.d), (-1, -1), bt, 0.0, None, None, None,
                                          9 if args.generic else None, k_new=k_new, v_new=v_new, skip_decodes=True, legacy_v0_layout=True)
    for _ in range(3):
        ua_mod.launch(p, dev)
    torch.cuda.synchronize()
    # ramping figure: every launch timed on its own, device idle in between (what a latency-sensitive caller sees first)
    ts = []
    for _ in range(args.iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ua_mod.launch(p, dev)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    ts.sort()
    med = ts[len(ts) // 2]
    # sustained figure: 0.25 s of back-to-back launches untimed (clock/power ramp), then N launches under one event pair
    import time
    t_end = time.perf_counter() + 0.25
    while time.perf_counter() < t_end:
        for _ in range(20):
            ua_mod.launch(p, dev)
        torch.cuda.synchronize()
    n = max(20, int(0.05 / med))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ua_mod.launch(p, dev)
    e1.record()
    torch.cuda.synchronize()
    sus = e0.elapsed_time(e1) * 1e-3 / n
    # GPU-side figure: the same launch 50 times in a HIP graph, replayed (no host issue between launches: a short kernel's
    # back-to-back figure above is bounded by the ~7 us a Python -> ctypes -> hipLaunchKernel call takes)
    gs = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(gs):
        ua_mod.launch(p, dev)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=gs):
            for _ in range(50):
                ua_mod.launch(p, dev)
        for _ in range(5):
            g.replay()
        e0.record(gs)
        for _ in range(10):
            g.replay()
        e1.record(gs)
    torch.cuda.synchronize()
    gr = e0.elapsed_time(e1) * 1e-3 / 500
    print(f"B={B} L={L} kernel={_lib.last_kernel()} median {med*1e6:8.1f} us  min {ts[0]*1e6:8.1f} us  {flops/med/1e12:7.1f} TFLOP/s "
          f"(min-time {flops/ts[0]/1e12:7.1f})  frac_of_2.5PF={flops/med/2.5e15:5.3f}  | sustained {sus*1e6:8.1f} us {flops/sus/1e12:7.1f} TFLOP/s"
          f"  | graph {gr*1e6:8.1f} us {flops/gr/1e12:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
