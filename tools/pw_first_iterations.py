#!/usr/bin/env python3
"""Diagnostic: shader cycles of a work item's first two tile-loop iterations against the rest, prefill_pw_kernel (are the
first iterations waiting for the loads the seam issued? what does a masked iteration cost?). Needs a library built with
-DMI355_PW_STAMP -DMI355_PW_SEAM (tools/build_variant.sh pwseam prefill_pw.hip -DMI355_PW_STAMP -DMI355_PW_SEAM):
    MI355_LIB=tools/ab/pwseam.so MI355_PREFILL=pw [MI355_WIN=1024] python tools/pw_first_iterations.py <batch> <seq>"""
import ctypes as C, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"vllm-triton-backend_amd")]
import torch
from mi355_attn import _lib
_lib.LIB_PATH=os.path.abspath(os.environ["MI355_LIB"])
from mi355_attn.kernels import unified as ua
batch=int(sys.argv[1]); L=int(sys.argv[2]); win=int(os.environ.get("MI355_WIN","0"))
dev=torch.device("cuda:0"); Hq,Hk,D,page=32,8,128,16
pps=L//page; nb=int(batch*pps*1.25)
k=(torch.rand(nb,page,Hk,D,device=dev)*2-1).bfloat16(); v=(torch.rand(nb,page,Hk,D,device=dev)*2-1).bfloat16()
q=(torch.rand(batch*L,Hq,D,device=dev)*2-1).bfloat16()
bt=torch.randperm(nb,device=dev)[:batch*pps].to(torch.int32).view(batch,pps)
cu=(torch.arange(batch+1,device=dev)*L).to(torch.int32); sl=torch.full((batch,),L,dtype=torch.int32,device=dev)
out=torch.empty_like(q)
max_wgs=(batch*L*(Hq//Hk)//256+batch)*Hk+64
dbg=torch.zeros(12*max_wgs,dtype=torch.int64,device=dev)
p,keep=ua.fill_attn_params(q,k,v,out,cu,L,sl,L,1/math.sqrt(D),(win-1,0) if win else (-1,-1),bt,0.0,None,None,None,2)
addr=dbg.data_ptr(); p.reserved0=C.c_int32(addr&0xFFFFFFFF).value; p.reserved1=C.c_int32((addr>>32)&0xFFFFFFFF).value
t_end=time.perf_counter()+1.5
while time.perf_counter()<t_end:
    for _ in range(20): ua.launch(p,dev)
    torch.cuda.synchronize()
rec=dbg.cpu().view(-1,12); rec=rec[rec[:,2]>2]
cyc,tiles,first2=rec[:,0].double(),rec[:,2].double(),rec[:,7].double()
rest=(cyc-first2)/(tiles-2)
print(f"kernel={_lib.last_kernel()} B={batch} L={L} win={win}: items {len(rec)}, tiles/item median {tiles.median():.0f}; loop cycles/tile median {(cyc/tiles).median():.0f}; first two iterations median {first2.median():.0f} cycles (= {first2.median()/2:.0f} each); the other iterations median {rest.median():.0f} per tile")
