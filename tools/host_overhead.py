#!/usr/bin/env python3
"""Host-side cost of one `unified_attention` call (Python argument checks + ctypes struct fill + C entry + launch), measured as
wall time per call over a back-to-back loop of tiny launches, next to the GPU time of the same launches."""
import math
import os

os.environ.setdefault("MI355_LAB", "1")      # tools may pin kernels through the library's measurement switches
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vllm-triton-backend_amd")]
import torch  # noqa: E402

from mi355_attn.kernels import unified as ua  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B, kv, Hq, Hk, D, page = 1, 256, 32, 8, 128, 16
    pps = kv // page
    k = torch.randn(B * pps + 4, page, Hk, D, device=dev, dtype=torch.bfloat16)
    v = torch.randn_like(k)
    q = torch.randn(B, Hq, D, device=dev, dtype=torch.bfloat16)
    bt = torch.arange(B * pps, device=dev, dtype=torch.int32).view(B, pps)
    cu = torch.arange(B + 1, device=dev, dtype=torch.int32)
    sl = torch.full((B,), kv, device=dev, dtype=torch.int32)
    out = torch.empty_like(q)
    scale = 1 / math.sqrt(D)

    def full_call():
        ua.unified_attention(q, k, v, out, cu, 1, sl, kv, 1.0, float(kv), scale, True, (-1, -1), bt, 0.0, None, None, None)

    p, keep = ua.fill_attn_params(q, k, v, out, cu, 1, sl, kv, scale, (-1, -1), bt, 0.0, None, None, None, None)

    def launch_only():
        ua.launch(p, dev)

    for name, fn in (("unified_attention (checks + struct fill + launch)", full_call), ("launch of a prepared struct", launch_only)):
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        n = 2000
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        print(f"{name}: host issue {t_issue / n * 1e6:.1f} us per call; GPU span {e0.elapsed_time(e1) / n * 1e3:.1f} us per call")


if __name__ == "__main__":
    main()
