#!/usr/bin/env python3
"""Small-batch decode latency: per-call time of mi355_attn.unified_attention at batch 1/4 over context lengths,
issued back-to-back on a stream and replayed from a HIP graph of 50 calls (GPU-side time without host issue)."""
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
    if os.environ.get("MI355_LIB"):
        _lib.LIB_PATH = os.path.abspath(os.environ["MI355_LIB"])
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[1, 4])
    ap.add_argument("--kv", type=int, nargs="+", default=[512, 2048, 8192, 32768])
    ap.add_argument("--hq", type=int, default=32)
    ap.add_argument("--hk", type=int, default=8)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--segments", type=int, nargs="+", default=[0], help="forced split counts (0 = the library's plan)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dt, page = torch.bfloat16, 16
    torch.manual_seed(0)
    # floor: a one-element torch kernel, same two measurements
    x = torch.zeros(1, device=dev)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        x.add_(1.0)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(50):
                x.add_(1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"floor: one-element kernel in a graph {e0.elapsed_time(e1) * 1e3 / 500:5.1f} us/node", flush=True)
    for B in args.batch:
        for L in args.kv:
            pps = (L + page - 1) // page
            nb = B * pps + 8
            k = (torch.rand(nb, page, args.hk, args.d, device=dev) * 2 - 1).to(dt)
            v = (torch.rand(nb, page, args.hk, args.d, device=dev) * 2 - 1).to(dt)
            q = (torch.rand(B, args.hq, args.d, device=dev) * 2 - 1).to(dt)
            bt = torch.randperm(nb, device=dev)[: B * pps].to(torch.int32).view(B, pps)
            cu = torch.arange(B + 1, dtype=torch.int32, device=dev)
            sl = torch.full((B,), L, dtype=torch.int32, device=dev)
            out = torch.empty_like(q)
            for seg in args.segments:
              p, keep = ua_mod.fill_attn_params(q, k, v, out, cu, 1, sl, L, 1.0 / math.sqrt(args.d), (-1, -1), bt, 0.0, None, None, None, None,
                                                num_segments=seg)
              for _ in range(20):
                  ua_mod.launch(p, dev)
              torch.cuda.synchronize()
              n = 200
              e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
              e0.record()
              for _ in range(n):
                  ua_mod.launch(p, dev)
              e1.record()
              torch.cuda.synchronize()
              stream_us = e0.elapsed_time(e1) * 1e3 / n
              g = torch.cuda.CUDAGraph()
              s = torch.cuda.Stream()
              with torch.cuda.stream(s):
                  ua_mod.launch(p, dev)
                  torch.cuda.synchronize()
                  with torch.cuda.graph(g, stream=s):
                      for _ in range(50):
                          ua_mod.launch(p, dev)
              for _ in range(3):
                  g.replay()
              torch.cuda.synchronize()
              e0.record()
              for _ in range(10):
                  g.replay()
              e1.record()
              torch.cuda.synchronize()
              graph_us = e0.elapsed_time(e1) * 1e3 / 500
              nbytes = 2 * B * L * args.hk * args.d * 2
              print(f"B={B} kv={L:6d} seg={seg:2d} kernel={_lib.last_kernel():16s} stream {stream_us:7.1f} us/call   graph {graph_us:7.1f} us/call   "
                    f"({nbytes / graph_us / 1e3:7.1f} GB/s)", flush=True)
            # a layer's decode step in a graph: cache write + attention (two nodes) against the fused launch (one node)
            from mi355_attn.kernels import reshape_and_cache_flash
            from mi355_attn.kernels.unified import decode_attention_and_cache_write
            k_new = (torch.rand(B, args.hk, args.d, device=dev) * 2 - 1).to(dt)
            v_new = (torch.rand(B, args.hk, args.d, device=dev) * 2 - 1).to(dt)
            slots = (bt[:, (L - 1) // page].long() * page + (L - 1) % page)
            p, keep = ua_mod.fill_attn_params(q, k, v, out, cu, 1, sl, L, 1.0 / math.sqrt(args.d), (-1, -1), bt, 0.0, None, None, None, None)

            def two():
                reshape_and_cache_flash(k_new, v_new, k, v, slots, "auto", None, None)
                ua_mod.launch(p, dev)

            def one():
                assert decode_attention_and_cache_write(q, k_new, v_new, k, v, out, sl, L, 1.0 / math.sqrt(args.d), bt, None, None, cu)

            res = []
            for fn in (two, one):
                g = torch.cuda.CUDAGraph()
                s = torch.cuda.Stream()
                with torch.cuda.stream(s):
                    fn()
                    torch.cuda.synchronize()
                    with torch.cuda.graph(g, stream=s):
                        for _ in range(50):
                            fn()
                for _ in range(3):
                    g.replay()
                torch.cuda.synchronize()
                e0.record()
                for _ in range(10):
                    g.replay()
                e1.record()
                torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1) * 1e3 / 500)
            print(f"B={B} kv={L:6d} decode step in a graph: cache write + attention {res[0]:6.1f} us, fused into one launch {res[1]:6.1f} us "
                  f"(saves {res[0] - res[1]:4.1f} us per layer)", flush=True)


if __name__ == "__main__":
    main()
