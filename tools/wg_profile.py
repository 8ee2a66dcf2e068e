#!/usr/bin/env python3
"""Diagnostic: life of a prefill_dma_kernel workgroup (prologue / tile loop / epilogue) in s_memtime
ticks, summed over workgroups, plus the span first-entry -> last-exit. Builds a SEPARATE library with
-DMI355_PROFILE_WG (never part of the product build).   python tools/wg_profile.py [batch] [seq]"""
import ctypes as C
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vllm-triton-backend_amd")]
CSRC = os.path.join(ROOT, "vllm-triton-backend_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_out", "libmi355_attn_wgprof.so")

import torch  # noqa: E402

from mi355_attn import _lib  # noqa: E402


def main():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    srcs = [os.path.join(CSRC, f) for f in ("api.hip", "generic_attn.hip", "cache_write.hip", "decode_splitkv.hip", "prefill_mfma.hip", "prefill_w64.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-DMI355_PROFILE_WG",
                           "-I", os.path.join(ROOT, "include"), "-o", OUT, *srcs])
    _lib.LIB_PATH = OUT
    from mi355_attn.kernels import unified as ua

    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    dev = torch.device("cuda:0")
    Hq, Hk, D, page = 32, 8, 128, 16
    pps = L // page
    nb = int(batch * pps * 1.25)
    k = (torch.rand(nb, page, Hk, D, device=dev) * 2 - 1).bfloat16()
    v = (torch.rand(nb, page, Hk, D, device=dev) * 2 - 1).bfloat16()
    q = (torch.rand(batch * L, Hq, D, device=dev) * 2 - 1).bfloat16()
    bt = torch.randperm(nb, device=dev)[: batch * pps].to(torch.int32).view(batch, pps)
    cu = (torch.arange(batch + 1, device=dev) * L).to(torch.int32)
    sl = torch.full((batch,), L, dtype=torch.int32, device=dev)
    out = torch.empty_like(q)
    dbg = torch.zeros(16, dtype=torch.int64, device=dev)
    p, keep = ua.fill_attn_params(q, k, v, out, cu, L, sl, L, 1 / math.sqrt(D), (-1, -1), bt, 0.0, None, None, None, 2)
    addr = dbg.data_ptr()
    p.reserved0 = C.c_int32(addr & 0xFFFFFFFF).value
    p.reserved1 = C.c_int32((addr >> 32) & 0xFFFFFFFF).value
    for _ in range(3):
        ua.launch(p, dev)
    torch.cuda.synchronize()
    dbg.zero_()
    dbg[13] = (1 << 62)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ua.launch(p, dev)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    s = dbg.cpu().tolist()
    n = max(s[12], 1)
    span = s[14] - s[13]
    tick_us = us / span if span else 0.0
    print(f"B={batch} L={L}: event time {us:.1f} us, span {span} ticks (=> {tick_us*1e3:.2f} ns/tick if the span is the whole kernel)")
    print(f"  workgroups {s[12]}, tiles {s[11]} ({s[11]/n:.1f} per WG)")
    for name, x in (("prologue (entry -> first tile staged)", s[8]), ("tile loop", s[9]), ("epilogue (normalise, store, drain)", s[10])):
        print(f"  {name:40s} {x/n:9.1f} ticks/WG   {100.0*x/(s[8]+s[9]+s[10]):5.1f} %")
    print(f"  prologue split: entry->metadata {s[15]/n:.0f}, metadata->block table staged {s[7]/n:.0f}, first tiles' DMA {(s[8]-s[15]-s[7])/n:.0f} ticks/WG")
    print(f"  tile loop: {s[9]/max(s[11],1):.2f} ticks per tile; prologue+epilogue = {(s[8]+s[10])/n/(s[9]/max(s[11],1)):.2f} tiles' worth per WG")
    slots = 512
    print(f"  sum of WG lives / {slots} slots = {(s[8]+s[9]+s[10])/slots:.0f} ticks vs span {span}")


if __name__ == "__main__":
    main()
