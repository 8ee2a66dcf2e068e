"""Scratch diagnostic for the in-kernel split merge: per-token error, NaN rows and arrival counters."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("tests", "", "vllm-triton-backend_amd"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import torch
import gpu_util
from oracle import paged_attention_oracle as orc
from mi355_attn import _lib

CASES = {
    "heads": dict(seed=5, q=[1] * 10, kv=[1, 15, 16, 17, 31, 32, 33, 700, 1023, 257], hq=32, hk=8, d=64, force=None),
    "multi": dict(seed=9, q=[7, 1, 40, 3], kv=[70, 45, 70, 300], hq=8, hk=2, d=128, force=3),
}
for name in sys.argv[1:] or list(CASES):
    c = CASES[name]
    inp = orc.make_paged_inputs(c["seed"], c["q"], c["kv"], c["hq"], c["hk"], c["d"], 16, torch.bfloat16)
    ref = orc.unified_attention_oracle(inp["q"], inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"], inp["scale"], mode="3d")
    dv = gpu_util.to_dev(inp)
    T = sum(c["q"])
    for it in range(2):
        out, kernel = gpu_util.run_unified(dv, inp["scale"], force=c["force"])
        torch.cuda.synchronize()
        o = out.float().cpu()
        nan_rows = torch.isnan(o).any(-1)
        err = (o - ref.float()).abs().amax(-1)
        print(name, "iter", it, kernel, "nan heads per token:", nan_rows.sum(-1).tolist())
        print("   max err per token:", [round(float(x), 3) for x in torch.nan_to_num(err, nan=-1).amax(-1)])
        ws = list(_lib._workspaces.values())[0]
        cnt = ws[:T * c["hk"] * 4].view(torch.int32).cpu().view(T, c["hk"])
        print("   nonzero counters:", cnt.nonzero().tolist(), cnt[cnt != 0].tolist())
