#!/usr/bin/env python3
"""Reference point for the MFMA roofline: what a plain dense bf16 GEMM (hipBLASLt through torch.matmul) sustains on
this chip, with the clocks and package power rocm-smi reports meanwhile. Not part of the product."""
import subprocess
import sys
import threading
import time

import torch


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    dev = torch.device("cuda:0")
    a = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
    b = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        a @ b
    torch.cuda.synchronize()
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
            lines = [l for l in o.strip().splitlines() if l and not l.startswith("WARNING")]
            if len(lines) >= 2:
                rec = dict(zip(lines[0].split(","), lines[1].split(",")))
                samples.append((time.time(), rec.get("sclk clock speed:", "?"), next((v for k, v in rec.items() if "Power" in k), "?")))
            time.sleep(0.3)

    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.time()
    iters = 0
    while time.time() - t0 < 4.0:
        for _ in range(20):
            a @ b
        torch.cuda.synchronize()
        iters += 20
    t1 = time.time()
    stop.set()
    th.join()
    tf = 2.0 * n ** 3 * iters / (t1 - t0) / 1e12
    print(f"bf16 GEMM {n}^3: {tf:.0f} TFLOP/s sustained over {t1 - t0:.1f} s = {tf / 2500:.3f} of the 2.5 PF nominal peak")
    for ts, sclk, pw in samples:
        if t0 + 1.0 <= ts <= t1:
            print(f"  [{ts - t0:5.2f}s] sclk {sclk} power {pw} W")


if __name__ == "__main__":
    main()
