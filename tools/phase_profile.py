#!/usr/bin/env python3
"""Diagnostic: where does a tile iteration of prefill_dma_kernel spend its cycles?
Builds a SEPARATE library with -DMI355_PROFILE_PHASES (s_memtime stamps around the phases of the
tile loop; never part of the product build), runs C2-shaped work and prints the phase shares.
Read the SHARES, not the run time (the stamps' fences forbid overlaps the real kernel has)."""
import ctypes as C
import math
import os

os.environ.setdefault("MI355_LAB", "1")      # tools may pin kernels through the library's measurement switches
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vllm-triton-backend_amd")]
CSRC = os.path.join(ROOT, "vllm-triton-backend_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_out", "libmi355_attn_prof.so")

import torch  # noqa: E402

from mi355_attn import _lib  # noqa: E402


def main():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    srcs = [os.path.join(CSRC, f) for f in ("api.hip", "generic_attn.hip", "cache_write.hip", "decode_splitkv.hip", "decode_splitkv_pack.hip", "prefill_mfma.hip", "prefill_lat.hip", "prefill_pw.hip", "prefill_pw_feat.hip", "prefill_pw_heads.hip", "prefill_pw_fp8.hip", "repack.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-DMI355_PROFILE_PHASES", "-DMI355_LAB",
                           "-o", OUT, *srcs])
    _lib.LIB_PATH = OUT
    from mi355_attn.kernels import unified as ua

    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    dev = torch.device("cuda:0")
    L, Hq, Hk, D, page = 4096, 32, 8, 128, 16
    pps = L // page
    nb = int(batch * pps * 1.25)
    k = (torch.rand(nb, page, Hk, D, device=dev) * 2 - 1).bfloat16()
    v = (torch.rand(nb, page, Hk, D, device=dev) * 2 - 1).bfloat16()
    q = (torch.rand(batch * L, Hq, D, device=dev) * 2 - 1).bfloat16()
    bt = torch.randperm(nb, device=dev)[: batch * pps].to(torch.int32).view(batch, pps)
    cu = (torch.arange(batch + 1, device=dev) * L).to(torch.int32)
    sl = torch.full((batch,), L, dtype=torch.int32, device=dev)
    out = torch.empty_like(q)
    dbg = torch.zeros(8, dtype=torch.int64, device=dev)
    p, keep = ua.fill_attn_params(q, k, v, out, cu, L, sl, L, 1 / math.sqrt(D), (-1, -1), bt, 0.0, None, None, None, 2)
    addr = dbg.data_ptr()
    p.reserved0 = C.c_int32(addr & 0xFFFFFFFF).value
    p.reserved1 = C.c_int32((addr >> 32) & 0xFFFFFFFF).value
    for _ in range(3):
        ua.launch(p, dev)
    torch.cuda.synchronize()
    dbg.zero_()
    ua.launch(p, dev)
    torch.cuda.synchronize()
    s = dbg.cpu().tolist()
    names = ["dma issue + page lookup", "QK (LDS reads + 16 MFMA, to completion)", "softmax", "PV (LDS reads + 16 MFMA, to completion)",
             "wait own DMA (vmcnt)", "barrier"]
    tot = sum(s[:6])
    print(f"waves {s[6]}, total stamped cycles/wave {tot / max(s[6], 1):.0f}")
    for n, x in zip(names, s[:6]):
        print(f"  {n:45s} {100.0 * x / tot:5.1f} %   {x / max(s[6], 1):10.0f} cycles/wave")


if __name__ == "__main__":
    main()
