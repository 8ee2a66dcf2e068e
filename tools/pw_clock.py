#!/usr/bin/env python3
"""Diagnostic: in-kernel clock and cycles per KV tile of prefill_pw_kernel's tile loop (MI355X_MICROARCH.md, DVFS
give-back item 6). Needs a library built with -DMI355_PW_STAMP (tools/build_variant.sh pwstamp prefill_pw.hip
-DMI355_PW_STAMP), never the product build:
    MI355_LIB=tools/ab/pwstamp.so MI355_PREFILL=pw python tools/pw_clock.py [batch] [seq]"""
import ctypes as C
import math
import os

os.environ.setdefault("MI355_LAB", "1")      # tools may pin kernels through the library's measurement switches
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vllm-triton-backend_amd")]
import torch  # noqa: E402

from mi355_attn import _lib  # noqa: E402


def main():
    _lib.LIB_PATH = os.path.abspath(os.environ["MI355_LIB"])
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
    max_wgs = (batch * L * (Hq // Hk) // 256 + batch) * Hk + 64
    dbg = torch.zeros(12 * max_wgs, dtype=torch.int64, device=dev)
    win = int(os.environ.get("MI355_WIN", "0"))             # sliding window (the SW instantiation)
    p, keep = ua.fill_attn_params(q, k, v, out, cu, L, sl, L, 1 / math.sqrt(D), (win - 1, 0) if win else (-1, -1), bt, 0.0, None, None, None, 2)
    addr = dbg.data_ptr()
    p.reserved0 = C.c_int32(addr & 0xFFFFFFFF).value
    p.reserved1 = C.c_int32((addr >> 32) & 0xFFFFFFFF).value
    t_end = time.perf_counter() + 2.0            # 2 s of back-to-back launches: the clock the chip HOLDS under this kernel
    n = 0
    while time.perf_counter() < t_end:
        for _ in range(20):
            ua.launch(p, dev)
        torch.cuda.synchronize()
        n += 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ua.launch(p, dev)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    rec = dbg.cpu().view(-1, 12)
    rec = rec[rec[:, 2] > 0]
    cyc, rt, tiles = rec[:, 0].double(), rec[:, 1].double(), rec[:, 2].double()
    clk = (cyc / rt * 100.0)                      # MHz
    cpt = cyc / tiles
    big = tiles >= tiles.max() * 0.5
    flops = 4 * L * L * D * Hq / 2 * batch
    print(f"kernel={_lib.last_kernel()} B={batch} L={L}: {us:.1f} us per launch = {flops/us/1e6:.0f} TFLOP/s; {len(rec)} workgroups")
    print(f"  in-kernel clock (tile loop): median {clk.median():.0f} MHz  (p10 {clk.kthvalue(max(1,len(clk)//10)).values:.0f}, p90 {clk.kthvalue(max(1,len(clk)*9//10)).values:.0f})")
    print(f"  cycles per KV tile (64 MFMA per SIMD): median {cpt[big].median():.0f}  = {cpt[big].median()/64:.1f} per MFMA   (all workgroups: {cpt.median():.0f})")
    seg = rec[:, 3:8].double().sum(0) / tiles.sum()
    print("  cycles per tile by segment (stamps included, ~40 each): seg1 S_A|B exps|DMA %.0f, seg2 O_B|A max+exps|V reads %.0f, seg3 S_B|A exps|K reads %.0f, seg4 O_A|B max+exps %.0f, end wait+barrier %.0f" % tuple(seg.tolist()))
    pro, loop, epi = (rec[:, 9] - rec[:, 8]).double() * 0.01, (rec[:, 10] - rec[:, 9]).double() * 0.01, (rec[:, 11] - rec[:, 10]).double() * 0.01
    t0 = rec[:, 8].min()
    print(f"  workgroup life (us): entry->loop median {pro.median():.2f} (p90 {pro.kthvalue(max(1,len(pro)*9//10)).values:.2f}), loop {loop.median():.2f}, loop end->exit median {epi.median():.2f} (p90 {epi.kthvalue(max(1,len(epi)*9//10)).values:.2f})")
    print(f"  first entry -> last exit {(rec[:, 11].max() - t0) * 0.01:.1f} us; last entry at {(rec[:, 8].max() - t0) * 0.01:.1f} us; sum of lives / 256 = {((rec[:, 11] - rec[:, 8]).double().sum() * 0.01 / 256):.1f} us")
    if os.environ.get("MI355_PW_SEAM"):     # library built with -DMI355_PW_SEAM too: realtime stamps inside the seam instead of the segment sums
        r2 = rec[rec[:, 6] > 0]
        d = lambda a_, b_: ((r2[:, b_] - r2[:, a_]).double() * 0.01).median().item()
        print(f"  seam (us, medians; items followed by another): loop end -> drained {d(10, 3):.2f}, drained -> barrier {d(3, 4):.2f}, next item acquired {d(4, 5):.2f}, "
              f"its loads requested / begun {d(5, 6):.2f}, output stored (+ loads dealt over it) {d(6, 11):.2f}")
        nxt = {int(r[8]): r for r in rec}                # an item's entry stamp (8) is the previous item's exit stamp (11)
        conv = [float(nxt[int(r[11])][9] - r[11]) * 0.01 for r in r2 if int(r[11]) in nxt]
        if conv:
            conv.sort()
            print(f"  output stored -> next item's tile loop (Q conversion, waits, first K read): median {conv[len(conv) // 2]:.2f} us over {len(conv)} seams")
    print(f"  time per tile: {(rt[big]/tiles[big]).median()*10:.0f} ns; tile loop = {(rt.sum()*0.01)/ (us*256)*100:.1f} % of CU time (256 CUs x launch time)")


if __name__ == "__main__":
    main()
