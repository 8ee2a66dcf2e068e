#!/usr/bin/env python3
"""Diagnostic: the spiked-score prefill case (tests/test_gpu_prefill.py::_spiked_inputs, gains +-2000) through the
single-launch and the key-split form of the library, row by row against a float64 softmax.
    python tools/debug_spike.py run <out.pt>     (honours MI355_PREFILL / MI355_PREFILL_KEY_SPLITS)
    python tools/debug_spike.py cmp <a.pt> <b.pt>"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vllm-triton-backend_amd"), os.path.join(ROOT, "tests")]
import torch


def inputs():
    import test_gpu_prefill as tp
    return tp._spiked_inputs(31, [700, 270, 1], [2300, 2100, 2500], 8, 2, (0.0, 0.0, 0.0, 0.0, 2000.0, -2000.0))


def dense_ref(inp):
    q, cu, sl, bt = inp["q"].double(), inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"]
    Hq, Hk = q.shape[1], inp["k_cache"].shape[2]
    out = torch.zeros_like(q)
    gap = torch.zeros(q.shape[0], Hq, dtype=torch.float64)
    for s in range(len(sl)):
        n, q0, q1 = int(sl[s]), int(cu[s]), int(cu[s + 1])
        pages = bt[s, : (n + 15) // 16].long()
        k = inp["k_cache"][pages].reshape(-1, Hk, 128)[:n].double().repeat_interleave(Hq // Hk, 1)
        v = inp["v_cache"][pages].reshape(-1, Hk, 128)[:n].double().repeat_interleave(Hq // Hk, 1)
        for t in range(q0, q1):
            vis = n - (q1 - q0) + (t - q0) + 1
            sc = torch.einsum("hd,khd->hk", q[t], k[:vis]) * inp["scale"]
            top = sc.topk(min(2, vis), dim=1).values
            gap[t] = (top[:, 0] - top[:, -1])
            out[t] = torch.einsum("hk,khd->hd", torch.softmax(sc, dim=1), v[:vis])
    return out, gap


if len(sys.argv) < 2:
    pass
elif sys.argv[1] == "run":
    import gpu_util
    from mi355_attn import _lib
    inp = inputs()
    d = gpu_util.to_dev(inp)
    out, kernel = gpu_util.run_unified(d, inp["scale"])
    print("kernel", kernel)
    torch.save(out.cpu(), sys.argv[2])
else:
    a, b = torch.load(sys.argv[2]).double(), torch.load(sys.argv[3]).double()
    inp = inputs()
    ref, gap = dense_ref(inp)
    for name, o in (("a", a), ("b", b)):
        err = (o - ref).abs().amax(-1)
        bad = (err > 2e-2).nonzero()
        print(name, "rows off by > 2e-2 vs float64:", len(bad), "max", float(err.max()))
        for t, h in bad[:12].tolist():
            print(f"   token {t} head {h}: err {float(err[t, h]):.4f}  gap between the two largest scores {float(gap[t, h]):.4f} nats")
