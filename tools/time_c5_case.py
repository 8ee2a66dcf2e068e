#!/usr/bin/env python3
"""Diagnostic: where tests/test_gpu_decode.py::test_decode_c5_full_size_properties spends its time."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vllm-triton-backend_amd"), os.path.join(ROOT, "tests")]
import torch
import gpu_util
from oracle import paged_attention_oracle as orc
T0 = time.time()
def lap(s):
    global T0
    torch.cuda.synchronize(); print(f"{s}: {time.time() - T0:.1f} s", flush=True); T0 = time.time()
dev = gpu_util.DEV
B, Hq, Hk, D, kv, page = 16, 64, 8, 128, 32768, 16
ks, vs = 0.0237, 0.041
g = torch.Generator(device="cpu").manual_seed(5)
pps = kv // page
nb = B * pps + 5
gd = torch.Generator(device=dev).manual_seed(5)
k = ((torch.rand(nb, page, Hk, D, generator=gd, device=dev) * 2 - 1) / ks).to(torch.float8_e4m3fn).cpu()
v = ((torch.rand(nb, page, Hk, D, generator=gd, device=dev) * 2 - 1) / vs).to(torch.float8_e4m3fn).cpu()
lap("generate + to cpu")
q = (torch.rand(B, Hq, D, generator=g) * 2 - 1).to(torch.bfloat16)
bt = torch.randperm(nb, generator=g)[: B * pps].to(torch.int32).view(B, pps)
t = dict(q=q, k_cache=k, v_cache=v, block_table=bt, cu_seqlens_q=torch.arange(B + 1, dtype=torch.int32), seqused_k=torch.full((B,), kv, dtype=torch.int32))
d = gpu_util.to_dev(t)
lap("to_dev")
out, kernel = gpu_util.run_unified(d, 1 / math.sqrt(D), kv_scale=ks, v_scale=vs)
lap("kernel")
ref = orc.unified_attention_oracle(q[0:1], k, v, torch.tensor([0, 1], dtype=torch.int32), torch.tensor([kv], dtype=torch.int32), bt[0:1], 1 / math.sqrt(D), k_scale=ks, v_scale=vs, mode="3d")
lap("one oracle row")
perm = torch.randperm(nb, generator=g)
d2k = d["k_cache"].view(torch.uint8)[perm.to(dev)].view(torch.float8_e4m3fn)
lap("permute k on device")
d3 = torch.full((nb, page, Hk, D), 0.5 / vs, device=dev).to(torch.float8_e4m3fn)
lap("constant v")
