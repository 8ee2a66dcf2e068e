#!/usr/bin/env python3
"""Diagnostic: sample rocm-smi clocks/power while a kernel loop runs (is the chip power-limited under this load?).
    python tools/clock_watch.py prefill|decode [seconds]"""
import math
import os

os.environ.setdefault("MI355_LAB", "1")      # tools may pin kernels through the library's measurement switches
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vllm-triton-backend_amd")]
import torch  # noqa: E402

from mi355_attn.kernels import unified as ua  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "prefill"
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
    dev = torch.device("cuda:0")
    Hq, Hk, D, page = 32, 8, 128, 16
    if kind == "prefill":
        B, ql, kl = 16, 4096, 4096
    else:
        B, ql, kl = 64, 1, 8192
    pps = kl // page
    nb = int(B * pps * 1.25)
    k = (torch.rand(nb, page, Hk, D, device=dev) * 2 - 1).bfloat16()
    v = (torch.rand(nb, page, Hk, D, device=dev) * 2 - 1).bfloat16()
    q = (torch.rand(B * ql, Hq, D, device=dev) * 2 - 1).bfloat16()
    bt = torch.randperm(nb, device=dev)[: B * pps].to(torch.int32).view(B, pps)
    cu = (torch.arange(B + 1, device=dev) * ql).to(torch.int32)
    sl = torch.full((B,), kl, dtype=torch.int32, device=dev)
    out = torch.empty_like(q)
    p, keep = ua.fill_attn_params(q, k, v, out, cu, ql, sl, kl, 1 / math.sqrt(D), (-1, -1), bt, 0.0, None, None, None, None)
    ua.launch(p, dev)
    torch.cuda.synchronize()
    samples = []
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            try:
                o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--csv"], capture_output=True, text=True, timeout=5).stdout
                samples.append((time.time(), o))
            except Exception as e:  # noqa: BLE001
                samples.append((time.time(), f"ERR {e}"))
            time.sleep(0.3)

    th = threading.Thread(target=sampler)
    th.start()
    time.sleep(1.0)           # idle samples first
    t0 = time.time()
    n = 0
    while time.time() - t0 < secs:
        for _ in range(50):
            ua.launch(p, dev)
        torch.cuda.synchronize()
        n += 50
    t1 = time.time()
    time.sleep(0.5)
    stop.set()
    th.join()
    print(f"{kind}: {n} launches in {t1 - t0:.2f} s = {(t1 - t0) / n * 1e6:.1f} us per launch (sustained)")
    for ts, o in samples:
        tag = "busy" if t0 <= ts <= t1 else "idle"
        lines = [l for l in o.strip().splitlines() if l and not l.startswith("WARNING")]
        if len(lines) >= 2 and lines[0].startswith("device"):
            rec = dict(zip(lines[0].split(","), lines[1].split(",")))
            sclk = rec.get("sclk clock speed:", "?")
            pw = next((v for k, v in rec.items() if "Power" in k), "?")
            tj = rec.get("Temperature (Sensor junction) (C)", "?")
            tm = rec.get("Temperature (Sensor memory) (C)", "?")
            print(f"[{ts - t0:6.2f}s {tag}] sclk {sclk} power {pw} W  Tj {tj} C  Tmem {tm} C")
        else:
            print(f"[{ts - t0:6.2f}s {tag}] " + " | ".join(lines[:3]))


if __name__ == "__main__":
    main()
