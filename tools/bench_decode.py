#!/usr/bin/env python3
"""Quick decode microbench (config C3 by default): HIP-event timing of mi355_attn.unified_attention.
Usage: python tools/bench_decode.py [--batch 64] [--kv 8192] [--hq 32] [--hk 8] [--segments N]"""
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
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--kv", type=int, default=8192)
    ap.add_argument("--hq", type=int, default=32)
    ap.add_argument("--hk", type=int, default=8)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--page", type=int, default=16)
    ap.add_argument("--segments", type=int, nargs="*", default=[0])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--kvdtype", default="same", choices=["same", "fp8", "fp8_e5m2"])
    ap.add_argument("--flush", default="write", choices=["write", "read", "none"])
    ap.add_argument("--legacy", action="store_true", help="legacy v0 cache layout through paged_attention_2d")
    ap.add_argument("--generic", action="store_true", help="force the shape-agnostic kernel")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[args.dtype]
    torch.manual_seed(0)
    B, kv, page = args.batch, args.kv, args.page
    ppseq = (kv + page - 1) // page
    nb = int(B * ppseq * 1.25)
    kvdt = {"same": dt, "fp8": torch.float8_e4m3fn, "fp8_e5m2": torch.float8_e5m2}[args.kvdtype]
    k = (torch.rand(nb, page, args.hk, args.d, device=dev) * 2 - 1).to(kvdt)
    v = (torch.rand(nb, page, args.hk, args.d, device=dev) * 2 - 1).to(kvdt)
    ksc = torch.ones(1, device=dev) if kvdt != dt else None
    q = (torch.rand(B, args.hq, args.d, device=dev) * 2 - 1).to(dt)
    bt = torch.randperm(nb, device=dev)[: B * ppseq].to(torch.int32).view(B, ppseq)
    cu = torch.arange(B + 1, dtype=torch.int32, device=dev)
    sl = torch.full((B,), kv, dtype=torch.int32, device=dev)
    out = torch.empty_like(q)
    algo_bytes = B * kv * args.hk * args.d * 2 * k.element_size() + 2 * q.numel() * q.element_size() + bt.numel() * 4 + (2 * B + 1) * 4
    if args.legacy:      # K [nb, Hk, D/8, page, 8], V [nb, Hk, D, page]
        x = 16 // k.element_size()
        k = k.view(nb, page, args.hk, args.d // x, x).permute(0, 2, 3, 1, 4).contiguous()
        v = v.permute(0, 2, 3, 1).contiguous()
    for seg in args.segments:
        p, keep = ua_mod.fill_attn_params(q, k, v, out, cu, 1, sl, kv, 1.0 / math.sqrt(args.d), (-1, -1), bt, 0.0, ksc, ksc, None, 9 if args.generic else None,
                                          num_segments=seg, legacy_v0_layout=args.legacy)
        for _ in range(3):
            ua_mod.launch(p, dev)
        torch.cuda.synchronize()
        # Cold caches between launches. "write" is what triton.testing.do_bench does (zero a buffer)
        # and leaves L2 + the 256 MB MALL full of DIRTY lines, whose write-back then competes with the
        # timed kernel's reads for HBM; "read" sweeps a 1 GiB buffer instead (clean lines, same eviction).
        flush = torch.zeros((1 << 30) if args.flush == "read" else (512 << 20), dtype=torch.uint8, device=dev)
        ts = []
        for _ in range(args.iters):
            if args.flush == "write":
                flush.zero_()
            elif args.flush == "read":
                flush.view(torch.int64).sum()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ua_mod.launch(p, dev)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e-3)
        ts.sort()
        med = ts[len(ts) // 2]
        print(f"segments={seg:3d} kernel={_lib.last_kernel()} median {med*1e6:8.1f} us  min {ts[0]*1e6:8.1f} us  "
              f"{algo_bytes/med/1e12:6.3f} TB/s (min-time {algo_bytes/ts[0]/1e12:6.3f})  frac_of_8TB/s={algo_bytes/med/8e12:5.3f}", flush=True)


if __name__ == "__main__":
    main()
