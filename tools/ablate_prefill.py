#!/usr/bin/env python3
"""Diagnostic: which part of prefill_dma_kernel's tile loop costs what? Builds SEPARATE libraries with
one phase compiled out each (-DMI355_ABLATE_*; outputs are wrong, only the time matters; never part
of the product build) and times them against the full kernel in one process, interleaved rounds."""
import math
import os

os.environ.setdefault("MI355_LAB", "1")      # tools may pin kernels through the library's measurement switches
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vllm-triton-backend_amd")]
CSRC = os.path.join(ROOT, "vllm-triton-backend_amd", "csrc")
import ctypes as C  # noqa: E402

import torch  # noqa: E402

from mi355_attn import _lib  # noqa: E402
from mi355_attn.kernels import unified as ua  # noqa: E402

VARIANTS = {"full": [], "no_dma": ["-DMI355_ABLATE_DMA"], "no_barrier": ["-DMI355_ABLATE_BARRIER"], "no_qk": ["-DMI355_ABLATE_QK"],
            "no_softmax": ["-DMI355_ABLATE_SOFTMAX"], "no_pv": ["-DMI355_ABLATE_PV"],
            "no_qk_no_pv": ["-DMI355_ABLATE_QK", "-DMI355_ABLATE_PV"], "only_mfma": ["-DMI355_ABLATE_SOFTMAX", "-DMI355_ABLATE_DMA", "-DMI355_ABLATE_BARRIER"]}


def main():
    if os.environ.get("MI355_PREFILL"):
        print("variant:", os.environ["MI355_PREFILL"])
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    names = sys.argv[2:] or list(VARIANTS)
    out_dir = os.path.join(ROOT, "gpurun_out", "ablate")
    os.makedirs(out_dir, exist_ok=True)
    srcs = [os.path.join(CSRC, f) for f in ("api.hip", "generic_attn.hip", "cache_write.hip", "decode_splitkv.hip", "decode_splitkv_pack.hip", "prefill_mfma.hip", "prefill_lat.hip", "prefill_pw.hip", "prefill_pw_feat.hip", "prefill_pw_heads.hip", "prefill_pw_fp8.hip", "repack.hip")]
    procs = []
    for n in names:
        so = os.path.join(out_dir, f"lib_{n}.so")
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-DMI355_LAB", "-shared", *VARIANTS[n], "-o", so, *srcs]))
    for pr in procs:
        assert pr.wait() == 0
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
    p, keep = ua.fill_attn_params(q, k, v, out, cu, L, sl, L, 1 / math.sqrt(D), (-1, -1), bt, 0.0, None, None, None, 2)
    libs = {}
    for n in names:
        lib = C.CDLL(os.path.join(out_dir, f"lib_{n}.so"))
        lib.mi355_unified_attention.restype = C.c_int
        lib.mi355_unified_attention.argtypes = [C.POINTER(_lib.AttnParams), C.c_void_p, C.c_size_t, C.c_void_p]
        libs[n] = lib
    stream = torch.cuda.current_stream(dev).cuda_stream
    times = {n: [] for n in names}
    for rnd in range(6):
        for n in names:
            for _ in range(2):
                libs[n].mi355_unified_attention(C.byref(p), None, 0, stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                libs[n].mi355_unified_attention(C.byref(p), None, 0, stream)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                times[n].append(e0.elapsed_time(e1) / 3 * 1e3)
    base = sorted(times[names[0]])[len(times[names[0]]) // 2]
    for n in names:
        t = sorted(times[n])
        print(f"{n:14s} median {t[len(t)//2]:9.1f} us   min {t[0]:9.1f} us   delta vs {names[0]} {t[len(t)//2]-base:+9.1f} us ({100*(t[len(t)//2]-base)/base:+5.1f} %)")


if __name__ == "__main__":
    main()
