#!/usr/bin/env python3
"""Multi-token decode steps (speculative decoding / MTP verification: a few query tokens per sequence over a long
context): the packed decode kernels (several tokens of a sequence in one wave's matrix columns) against the former
route (the prefill kernels), same tensors. MI355_DECODE_PACK is the library's A/B switch (0: off, 1: one column group only), read once per process:
the tool runs each setting in a child process of its own (--pack).
  python tools/bench_multitoken_decode.py [--hq 32 --hk 8]"""
import argparse
import os

os.environ.setdefault("MI355_LAB", "1")      # tools may pin kernels through the library's measurement switches
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-triton-backend_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_mixed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", nargs="*", default=["64x4x8192", "64x2x8192", "64x8x8192", "16x4x32768", "128x4x2048", "8x4x512", "64x3x8192", "64x16x8192"])
    ap.add_argument("--mixed", nargs="*", type=lambda x: tuple(int(v) for v in x.split("x")), default=[],
                    help="rows x tokens x keys x chunks x chunk-tokens, e.g. 32x4x8192x2x2048")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--hq", type=int, default=32)
    ap.add_argument("--hk", type=int, default=8)
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--pack", default=None, help="(internal) run one setting of MI355_DECODE_PACK in this process and print raw rows")
    a = ap.parse_args()
    if a.pack is None:       # parent: one child per setting (the library reads the switch once), rows joined side by side
        import subprocess
        outs = {}
        for pack in ("0", "2"):
            env = dict(os.environ, MI355_DECODE_PACK=pack)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), *sys.argv[1:], "--pack", pack], env=env, capture_output=True, text=True)
            if r.returncode != 0:
                sys.stderr.write(r.stderr)
                sys.exit(r.returncode)
            outs[pack] = [ln for ln in r.stdout.splitlines()]
        for l0, l2 in zip(outs["0"], outs["2"]):
            if l0.startswith("#"):
                print(l0)
            else:
                key, v0 = l0.split(": ", 1)
                print(f"{key}: {v0}   |   {l2.split(': ', 1)[1]}", flush=True)
        return
    dev = torch.device("cuda:0")
    bench_mixed.HQ, bench_mixed.HK, bench_mixed.WINDOW = a.hq, a.hk, a.window
    if a.window:
        print(f"# sliding window {a.window} (the stream rate column still counts ALL keys of the context)")
    print(f"# sequences x query tokens x keys (Hq {a.hq} / Hk {a.hk} / D 128, bf16): median us, K/V stream rate, kernel")
    for sh in a.shapes:
        b, ql, kv = (int(x) for x in sh.split("x"))
        row = []
        for pack in (a.pack,):
            torch.manual_seed(0)
            t, fl, by, kern = bench_mixed.run([ql] * b, [kv] * b, dev, iters=a.iters)
            row.append(f"{'packed' if pack != '0' else 'former'} {t * 1e6:8.1f} us {by / t / 1e12:5.2f} TB/s {kern}")
        print(f"{sh:>14}: " + "   |   ".join(row), flush=True)
    # a step that mixes prefill chunks with multi-token decode rows: rows x tokens x keys + chunks x tokens (no context)
    for rows, ql, kv, chunks, cl in a.mixed:
        row = []
        for pack in (a.pack,):
            torch.manual_seed(0)
            t, fl, by, kern = bench_mixed.run([ql] * rows + [cl] * chunks, [kv] * rows + [cl] * chunks, dev, iters=a.iters)
            row.append(f"{'packed' if pack != '0' else 'former'} {t * 1e6:8.1f} us {kern}")
        print(f"{rows}x{ql}x{kv} + {chunks}x{cl}: " + "   |   ".join(row), flush=True)


if __name__ == "__main__":
    main()
