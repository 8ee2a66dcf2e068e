import sys, os, math, torch
ROOT="/root/repo"
sys.path[:0]=[ROOT, os.path.join(ROOT,"vllm-triton-backend_amd"), os.path.join(ROOT,"tests")]
from mi355_attn import _lib
if os.environ.get("MI355_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["MI355_LIB"])
import test_gpu_prefill as tp, gpu_util
from oracle import paged_attention_oracle as orc
gains=(0.0, 0.0, 0.0, 0.0, 2000.0, -2000.0)
ql, kl = [700,270,1],[2300,2100,2500]
inp = tp._spiked_inputs(31, ql, kl, 8, 2, gains)
ref = orc.unified_attention_oracle(inp["q"], inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"], inp["scale"], mode="2d", block_n=64)
d = gpu_util.to_dev(inp)
out, kernel = gpu_util.run_unified(d, inp["scale"])
out=out.float().cpu()
cu, sk, bt = inp["cu_seqlens_q"].tolist(), inp["seqused_k"].tolist(), inp["block_table"]
def f64(s_idx, tok, h):
    t = cu[s_idx]+tok
    n = sk[s_idx]-(cu[s_idx+1]-cu[s_idx])+tok+1
    pages = bt[s_idx,:(n+15)//16].long()
    K = inp["k_cache"][pages].reshape(-1,2,128)[:n].double(); V = inp["v_cache"][pages].reshape(-1,2,128)[:n].double()
    sc = (K[:,h//4] @ inp["q"][t,h].double())*inp["scale"]
    p = torch.softmax(sc,0)
    return (p[:,None]*V[:,h//4]).sum(0), sc
for (s_idx,tok) in ((0,4),(0,5),(0,10),(0,11),(0,2),(1,52)):
    t=cu[s_idx]+tok
    for h in (0,5):
        o64, sc = f64(s_idx,tok,h)
        print(kernel, f"seq{s_idx} tok{tok} h{h} gain={gains[t%6]} score range [{sc.min():.1f},{sc.max():.1f}] |kernel-f64|={ (out[t,h].double()-o64).abs().max():.4f} |oracle-f64|={(ref[t,h].double()-o64).abs().max():.4f}")
lse = torch.full((inp["q"].shape[0], 8), float("nan"), dtype=torch.float32, device=gpu_util.DEV)
out2, kernel = gpu_util.run_unified(d, inp["scale"], lse=lse)
for t in (4, 5, 322):
    print(t, gains[t % 6], "out", out2[t, 0, :4].float().cpu().tolist(), "ref", ref[t, 0, :4].float().tolist(), "lse", lse[t, 0].item(), lse[t, 5].item())
