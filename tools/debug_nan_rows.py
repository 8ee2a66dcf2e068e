#!/usr/bin/env python3
"""Diagnostic: which (token, head) rows of a plain causal prefill come out NaN / differ from a float reference."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vllm-triton-backend_amd"), os.path.join(ROOT, "tests")]
import torch
from mi355_attn import _lib
if os.environ.get('MI355_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['MI355_LIB'])
import gpu_util
L = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Hq, Hk, D, page = 32, 8, 128, 16
g = torch.Generator().manual_seed(3)
nb = L // page + 7
k = (torch.rand(nb, page, Hk, D, generator=g) * 2 - 1).to(torch.bfloat16)
v = (torch.rand(nb, page, Hk, D, generator=g) * 2 - 1).to(torch.bfloat16)
q = (torch.rand(L, Hq, D, generator=g) * 2 - 1).to(torch.bfloat16)
bt = torch.randperm(nb, generator=g)[: L // page].to(torch.int32).view(1, -1)
mode = os.environ.get("DBG_MODE", "")
if "q0" in mode: q = torch.zeros_like(q)                      # uniform softmax: out = mean of the visible V rows
if "vkey" in mode:                                            # V[key][d] = key index / 64 (exact in bf16 up to 256 keys... coarse): shows which keys a row averaged
    vv_ = torch.zeros_like(v)
    for pg_i in range(L // page):
        vv_[bt[0, pg_i]] = (torch.arange(pg_i * page, (pg_i + 1) * page).float() / 64.0)[:, None, None].to(torch.bfloat16)
    v = vv_
if "vhot" in mode:                                            # V[key][d] = (d == key % 128): out[row][d] = weight of key d
    vv_ = torch.zeros_like(v)
    for pg_i in range(L // page):
        for sl in range(page):
            vv_[bt[0, pg_i], sl, :, (pg_i * page + sl) % D] = 1.0
    v = vv_
if "kone" in mode: k = torch.ones_like(k) * 0.125
t = dict(q=q, k_cache=k, v_cache=v, block_table=bt, cu_seqlens_q=torch.tensor([0, L], dtype=torch.int32), seqused_k=torch.tensor([L], dtype=torch.int32))
d = gpu_util.to_dev(t)
out, kernel = gpu_util.run_unified(d, 1.0 / math.sqrt(D))
nanrow = torch.isnan(out.float()).any(-1).cpu()
print(kernel, "NaN rows:", int(nanrow.sum()), "of", nanrow.numel())
toks = nanrow.any(-1).nonzero().flatten()
print("tokens with NaN:", toks[:10].tolist(), "...", toks[-10:].tolist(), "count", len(toks))
if len(toks):
    print("heads NaN at first such token:", nanrow[toks[0]].nonzero().flatten().tolist())
    blocks = sorted(set((toks // 64).tolist()))
    print("64-token blocks touched:", blocks)
kk = k[bt[0].long()].reshape(L, Hk, D).float().repeat_interleave(Hq // Hk, 1).to(gpu_util.DEV)
vv = v[bt[0].long()].reshape(L, Hk, D).float().repeat_interleave(Hq // Hk, 1).to(gpu_util.DEV)
ref = torch.nn.functional.scaled_dot_product_attention(d["q"].float().transpose(0, 1), kk.transpose(0, 1), vv.transpose(0, 1), is_causal=True).transpose(0, 1)
err = (out.float() - ref).abs().amax(-1).cpu()
err[nanrow] = 0
badrow = err > 2e-2
print("finite rows off by > 2e-2:", int(badrow.sum()), "max err", float(err.max()))
if badrow.any():
    bt_ = badrow.any(-1).nonzero().flatten()
    print("tokens:", bt_[:10].tolist(), "...", bt_[-5:].tolist(), "blocks", sorted(set((bt_ // 64).tolist()))[:40])
# determinism and page-permutation invariance, bit-exact
out_b, _ = gpu_util.run_unified(d, 1.0 / math.sqrt(D))
diff = (out.view(torch.int16) != out_b.view(torch.int16)).any(-1).cpu()
if "hotd" in mode:
    for tk in (32, 33, 35, 36, 39, 48, 51, 63):
        for hh in (0, 3):
            print("tok", tk, "head", hh, "w*(tok+1):", "".join("%d" % min(9, round(float(x) * (tk + 1))) for x in out[tk, hh, :64].float().cpu()), " run2:", "".join("%d" % min(9, round(float(x) * (tk + 1))) for x in out_b[tk, hh, :64].float().cpu()))
if "dump" in mode:
    for hh in (0, 1, 4):
        print("head", hh, "out*64 col0:", [round(float(x) * 64, 1) for x in out[:64, hh, 0].float().cpu()])
        print("head", hh, "run2    col0:", [round(float(x) * 64, 1) for x in out_b[:64, hh, 0].float().cpu()])
        print("head", hh, "ref     col0:", [round(float(x) * 64, 1) for x in ref[:64, hh, 0].float().cpu()])
        print("head", hh, "out*64 col77:", [round(float(x) * 64, 1) for x in out[:64, hh, 77].float().cpu()])
print("same input twice: rows that differ:", int(diff.sum()), "tokens", diff.any(-1).nonzero().flatten()[:8].tolist())
perm = torch.randperm(nb, generator=g)
inv = torch.empty_like(perm)
inv[perm] = torch.arange(nb)
d2 = dict(d)
d2["k_cache"] = d["k_cache"][perm.to(gpu_util.DEV)]
d2["v_cache"] = d["v_cache"][perm.to(gpu_util.DEV)]
d2["block_table"] = inv.to(gpu_util.DEV)[d["block_table"].long()].to(torch.int32)
out2, _ = gpu_util.run_unified(d2, 1.0 / math.sqrt(D))
diff = (out.view(torch.int16) != out2.view(torch.int16)).any(-1).cpu()
tk = diff.any(-1).nonzero().flatten()
print("pages permuted: rows that differ:", int(diff.sum()), "tokens", tk[:8].tolist(), "...", tk[-8:].tolist(), "blocks", sorted(set((tk // 64).tolist())))
if len(tk):
    print("heads at first:", diff[tk[0]].nonzero().flatten().tolist(), " err vs ref of the permuted run:", float((out2.float() - ref).abs().max()))
# detail of the rows that are off (layout debugging): which heads, which columns
if badrow.any():
    cnt_by_tok = badrow.sum(-1)
    t0 = int(badrow.any(-1).nonzero().flatten()[0])
    print("first bad token", t0, "heads", badrow[t0].nonzero().flatten().tolist())
    h0 = int(badrow[t0].nonzero().flatten()[0])
    colerr = (out.float() - ref)[t0, h0].abs().cpu()
    print("columns off by > 1e-2:", (colerr > 1e-2).nonzero().flatten().tolist()[:64])
    wave_rows = collections.Counter() if False else None
    import collections
    c = collections.Counter()
    for t_, h_ in badrow.nonzero().tolist():
        m = (t_ % 64) * 4 + h_ % 4          # row of the 256-row Q block (G = 4)
        c[(m // 64, (m % 64) // 16)] += 1
    print("bad rows by (wave, 16-row tile of the wave):", sorted(c.items()))
