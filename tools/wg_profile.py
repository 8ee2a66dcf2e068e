#!/usr/bin/env python3
"""Diagnostic: life of a prefill_dma_kernel workgroup (prologue / tile loop / epilogue) in s_memtime
ticks, summed over workgroups, plus the span first-entry -> last-exit. Builds a SEPARATE library with
-DMI355_PROFILE_WG (never part of the product build).   python tools/wg_profile.py [batch] [seq]"""
import ctypes as C
import math
import os

os.environ.setdefault("MI355_LAB", "1")      # tools may pin kernels through the library's measurement switches
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
    srcs = [os.path.join(CSRC, f) for f in ("api.hip", "generic_attn.hip", "cache_write.hip", "decode_splitkv.hip", "decode_splitkv_pack.hip", "prefill_mfma.hip", "prefill_lat.hip", "prefill_pw.hip", "prefill_pw_feat.hip", "prefill_pw_heads.hip", "prefill_pw_fp8.hip", "repack.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-DMI355_PROFILE_WG", "-DMI355_LAB",
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
    max_wgs = (batch * L // 32 + batch) * Hk + 64
    dbg = torch.zeros(16 + 4 * max_wgs, dtype=torch.int64, device=dev)
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
    # ---- schedule reconstruction -------------------------------------------------------------------
    recs = [s[16 + 4 * i: 20 + 4 * i] for i in range(max_wgs)]
    recs = [(i, r) for i, r in enumerate(recs) if r[1] != 0]
    if not recs:
        return
    # records carry s_memrealtime stamps (100 MHz, one counter for the chip)
    def xcc_of(r):
        return (r[2] >> 32) & 0xF
    t_first = min(r[0] for _, r in recs)
    tmin = {x: t_first for x in range(16)}
    per_cu = {}
    for i, r in recs:
        hw = r[2] & 0xFFFFFFFF
        xcc = xcc_of(r)
        cu = (xcc, (hw >> 13) & 0x7, (hw >> 12) & 0x1, (hw >> 8) & 0xF)   # (xcc, se, sh, cu)
        per_cu.setdefault(cu, []).append((r[0] - tmin[xcc], r[1] - tmin[xcc], i, r[3]))
    span2 = max(e for v in per_cu.values() for _, e, _, _ in v)
    busy = sorted(sum(e - b for b, e, _, _ in v) / span2 for v in per_cu.values())
    ends = sorted(max(e for _, e, _, _ in v) / span2 for v in per_cu.values())
    starts = sorted(min(b for b, _, _, _ in v) / span2 for v in per_cu.values())
    nw = sorted(len(v) for v in per_cu.values())
    print(f"  schedule: {len(recs)} workgroups on {len(per_cu)} CUs, first entry -> last exit {span2 / 100:.1f} us (event time {us:.1f} us)")
    print(f"    workgroups per CU: min {nw[0]} median {nw[len(nw)//2]} max {nw[-1]}")
    print(f"    CU busy fraction of the span: min {busy[0]:.3f} p10 {busy[len(busy)//10]:.3f} median {busy[len(busy)//2]:.3f} max {busy[-1]:.3f} mean {sum(busy)/len(busy):.3f}")
    print(f"    first start: median {starts[len(starts)//2]:.3f} max {starts[-1]:.3f}; last end: min {ends[0]:.3f} p10 {ends[len(ends)//10]:.3f} median {ends[len(ends)//2]:.3f}")
    gaps = sorted((sorted(v)[1][0] - sorted(v)[0][1]) / span2 for v in per_cu.values() if len(v) >= 2)
    if gaps:
        print(f"    gap between a CU's first and second workgroup: median {gaps[len(gaps)//2]:.4f} max {gaps[-1]:.4f} of the span")
    xspan = {}
    for cu, v in per_cu.items():
        xspan[cu[0]] = max(xspan.get(cu[0], 0), max(e for _, e, _, _ in v))
    print("    last exit per XCC (us after the first entry):", {k: round(v / 100, 1) for k, v in sorted(xspan.items())})
    xs = {}
    for cu, v in per_cu.items():
        xs.setdefault(cu[0], []).append(sum(e - b for b, e, _, _ in v) / span2)
    print("    mean busy fraction per XCC:", {k: round(sum(v) / len(v), 3) for k, v in sorted(xs.items())}, "CUs per XCC:", {k: len(v) for k, v in sorted(xs.items())})
    worst = max(per_cu.items(), key=lambda kv: max(e for _, e, _, _ in kv[1]))
    print("    CU that finishes last:", worst[0], [(round(b / 100, 1), round(e / 100, 1), int(t)) for b, e, _, t in sorted(worst[1])][:8], "(start us, end us, tiles)")
    r1 = sorted(e for v in per_cu.values() for b, e, _, _ in v if b < span2 * 0.1)
    r2 = sorted(b for v in per_cu.values() for b, e, _, _ in v if b >= span2 * 0.1)
    if r1 and r2:
        print(f"    first round ends {r1[0]/100:.1f}..{r1[-1]/100:.1f} us, later rounds start {r2[0]/100:.1f}..{r2[-1]/100:.1f} us")
    heads = {}
    for cu, v in per_cu.items():
        for b, e, i, t in v:
            heads.setdefault(i % Hk, set()).add(cu[0])
    print("    XCCs each KV head ran on:", {h: sorted(x) for h, x in sorted(heads.items())})


if __name__ == "__main__":
    main()
