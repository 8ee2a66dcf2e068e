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
