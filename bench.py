#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X paged-attention hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): "attn fwd TFLOPS (prefill) + KV GB/s (decode), Llama-3-8B GQA seq4k".
The timed region is K steps of the PREFILL workload C2 (BASELINE configs[1]: Hq 32 / Hk 8 / D 128,
one 4096-token sequence per GPU, bf16, paged KV with 16-token pages): `value` = attention-forward
TFLOP/s, counted FA-style (4*q*kv*D*Hq/2 = 137.44 GFLOP per sequence), whole job over all GPUs.
The DECODE workload C3 (configs[2]: batch 64, kv_len 8192 per GPU) is timed the same way right
after it and reported in the same JSON line under "decode" (KV GB/s) with its own roofline.
Multi-GPU: the path shards over sequences with no data-path collective: every rank
owns its sequences, their KV pages and a local block table -> weak scaling; the only collectives
are the timing barrier and the MAX over ranks of the measured time.
Inputs are synthetic (U(-1,1), seed 0 + rank), resident in HBM before the timed region starts.
"""

from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vllm-triton-backend_amd")]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak
PREWARM_S = float(os.environ.get("MI355_BENCH_PREWARM_S", "0.25"))   # untimed launches before the W warm-up steps


def make_workload(kind, device, seed, Hq=32, Hk=8, D=128, page=16):
    g = torch.Generator(device="cpu").manual_seed(seed)
    if kind == "prefill":      # C2
        B, q_len, kv_len = 1, 4096, 4096
    else:                      # C3
        B, q_len, kv_len = 64, 1, 8192
    pps = (kv_len + page - 1) // page
    nb = int(B * pps * 1.25)
    dt = torch.bfloat16
    # generate on the device to keep start-up short; values U(-1,1) (scripts/benchmark.py:136,:1168-1174)
    gen = torch.Generator(device=device).manual_seed(seed)
    k = (torch.rand(nb, page, Hk, D, device=device, generator=gen) * 2 - 1).to(dt)
    v = (torch.rand(nb, page, Hk, D, device=device, generator=gen) * 2 - 1).to(dt)
    q = (torch.rand(B * q_len, Hq, D, device=device, generator=gen) * 2 - 1).to(dt)
    bt = torch.randperm(nb, generator=g)[: B * pps].to(torch.int32).view(B, pps).to(device)
    cu = (torch.arange(B + 1, dtype=torch.int32) * q_len).to(device)
    sl = torch.full((B,), kv_len, dtype=torch.int32, device=device)
    out = torch.empty_like(q)
    w = dict(kind=kind, q=q, k_cache=k, v_cache=v, block_table=bt, cu_seqlens_q=cu, seqused_k=sl, out=out, B=B, q_len=q_len,
             kv_len=kv_len, Hq=Hq, Hk=Hk, D=D, page=page, scale=1.0 / math.sqrt(D))
    if kind == "prefill":
        w["flops"] = 4.0 * q_len * kv_len * D * Hq / 2 * B
        w["bytes"] = (2 * q.numel() + 2 * B * kv_len * Hk * D) * 2.0
    else:
        w["flops"] = 4.0 * B * Hq * kv_len * D
        w["bytes"] = B * kv_len * Hk * D * 2 * 2.0 + 2 * q.numel() * 2.0 + bt.numel() * 4.0 + (2 * B + 1) * 4.0
    return w


def build_call(w, device):
    from mi355_attn.kernels import unified as ua

    p, keep = ua.fill_attn_params(w["q"], w["k_cache"], w["v_cache"], w["out"], w["cu_seqlens_q"], w["q_len"], w["seqused_k"], w["kv_len"],
                                  w["scale"], (-1, -1), w["block_table"], 0.0, None, None, None, None)
    w["_keep"] = (p, keep)
    return lambda: ua.launch(p, device)


def timed_steps(call, steps, warmup, device, distributed):
    """W untimed steps, then EXACTLY K steps bracketed by barrier + synchronize. One HIP event pair brackets the K
    launches on the stream they are launched on: the kernel's average launch duration is that span / K (it includes
    the gaps between back-to-back launches, as a serving loop would see them; rocprofv3's per-kernel average under
    profiles/ is the same number minus those gaps). No event is recorded between launches: an event record is a
    barrier packet, and two of them per launch cost ~3 % of a 165 us kernel."""
    # Bring the device out of its idle power state first: a 165 us kernel timed 50 times right after the inputs were
    # generated measures the clock/power ramp, not the kernel (C2: 840 TFLOP/s over 50 steps, 990 over 500, 1015 over
    # 2000 on one box). PREWARM_S seconds of the same launches, untimed, precede the W warm-up steps.
    t_end = time.perf_counter() + PREWARM_S
    while time.perf_counter() < t_end:
        for _ in range(20):
            call()
        torch.cuda.synchronize(device)
    for _ in range(warmup):
        call()
    torch.cuda.synchronize(device)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        call()
    e1.record()
    torch.cuda.synchronize(device)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0
    return wall, e0.elapsed_time(e1) * 1e-3 / steps


def cpu_baseline(w, out_gpu):
    """Reference-style CPU SDPA path on this box's host cores, on the same inputs (rank 0, N=1)."""
    from oracle.cpu_sdpa_baseline import time_paged_sdpa_cpu

    if w["kind"] == "prefill":
        sample, nseq = "full C2 workload (1 seq x 4096 tokens, 32 q heads), median of <=3 reps", w["B"]
    else:
        nseq = 4
        sample = f"{nseq} of {w['B']} C3 sequences (kv_len 8192, 32 q heads), median of <=3 reps"
    cu = w["cu_seqlens_q"][: nseq + 1].cpu()
    inp = dict(q=w["q"][: int(cu[-1])].cpu(), k_cache=w["k_cache"].cpu(), v_cache=w["v_cache"].cpu(), cu_seqlens_q=cu,
               seqused_k=w["seqused_k"][:nseq].cpu(), block_table=w["block_table"][:nseq].cpu())
    out_cpu, t_med, t_gather, reps = time_paged_sdpa_cpu(inp, w["scale"], warmup=1, reps=3, budget_s=12.0)
    frac = nseq / w["B"]
    err = (out_cpu.float() - out_gpu[: out_cpu.shape[0]].float().cpu()).abs().max().item()
    if w["kind"] == "prefill":
        value, unit = w["flops"] * frac / t_med / 1e12, "TFLOP/s"
    else:
        value, unit = w["bytes"] * frac / t_med / 1e9, "GB/s"
    return {"value": round(value, 4), "unit": unit, "cores": torch.get_num_threads(), "host_cpus": os.cpu_count(), "kind": "port",
            "sample": sample, "seconds": round(t_med, 4), "gather_seconds": round(t_gather, 4), "reps": reps,
            "max_abs_diff_vs_gpu": err}


def measured_traffic(kernel_prefix):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (tools/collect_traffic.py ->
    profiles/r01/traffic.json: (2*FETCH_SIZE + WRITE_SIZE)*1024, the gfx950 correction of the guide);
    None when no profile of that kernel is committed."""
    path = os.path.join(ROOT, "profiles", "r01", "traffic.json")
    if not os.path.exists(path):
        return None
    try:
        for name, rec in json.load(open(path)).items():
            if name.startswith(kernel_prefix):
                return rec["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)   # launched by torch.distributed.run
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path for the product")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    n_gpus = world if distributed else 1
    if args.gpus != n_gpus and rank == 0:
        print(f"# note: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run for N>1", file=sys.stderr)

    import __graft_entry__ as ge
    from mi355_attn import _lib

    if not os.path.exists(_lib.LIB_PATH):
        if rank == 0:
            ge.build()
        if distributed:
            dist.barrier()

    results = {}
    for kind in ("prefill", "decode"):
        w = make_workload(kind, device, seed=rank)
        call = build_call(w, device)
        call()
        torch.cuda.synchronize(device)
        kernel = _lib.last_kernel()
        wall, per_launch = timed_steps(call, args.steps, args.warmup, device, distributed)
        t = torch.tensor([wall, per_launch], dtype=torch.float64, device=device)
        if distributed:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall_max, launch_max = t.tolist()
        results[kind] = dict(w=w, kernel=kernel, wall=wall_max, per_launch=launch_max)

    if rank == 0:
        pf, dc = results["prefill"], results["decode"]
        K = args.steps
        pf_val = pf["w"]["flops"] * n_gpus * K / pf["wall"] / 1e12
        dc_val = dc["w"]["bytes"] * n_gpus * K / dc["wall"] / 1e9
        pf_ach = pf["w"]["flops"] / pf["per_launch"] / 1e12
        dc_ach = dc["w"]["bytes"] / dc["per_launch"] / 1e9
        line = {
            "metric": "attn fwd TFLOPS (prefill) + KV GB/s (decode), Llama-3-8B GQA seq4k",
            "value": round(pf_val, 2), "unit": "TFLOP/s", "n_gpus": n_gpus, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(pf["wall"] / K * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "C2 prefill: Llama-3-8B shape Hq32/Hk8/D128, 1 seq x 4096 tokens per GPU, causal, paged KV (16-token pages)",
                       "global_batch": n_gpus, "seq_len": 4096, "parallelism": f"batch-sharded x{n_gpus}, no collective", "kernel": pf["kernel"]},
            "roofline": {"bound": "mfma", "achieved": round(pf_ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(pf_ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": measured_traffic("prefill_dma_kernel"), "kernel": pf["kernel"],
                         "kernel_us": round(pf["per_launch"] * 1e6, 2), "algorithmic_flops_per_launch": pf["w"]["flops"],
                         "note": "peak = nominal dense bf16 MFMA rate; the package is power-limited (1.4 kW) under MFMA load: a dense bf16 "
                                 "hipBLASLt GEMM sustains 1403 TFLOP/s on the same chip (profiles/r01/gemm_reference_point.log)"},
            "decode": {"metric": "KV GB/s (paged decode)", "value": round(dc_val, 1), "unit": "GB/s", "ms_per_step": round(dc["wall"] / K * 1e3, 4),
                       "config": {"workload": "C3 decode: Hq32/Hk8/D128, batch 64 x kv_len 8192 per GPU, bf16, 16-token pages",
                                  "global_batch": 64 * n_gpus, "kernel": dc["kernel"]},
                       "roofline": {"bound": "hbm", "achieved": round(dc_ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(dc_ach / HBM_PEAK_GBS, 4), "traffic": measured_traffic("decode_splitkv_kernel"), "kernel": dc["kernel"],
                                    "kernel_us": round(dc["per_launch"] * 1e6, 2), "algorithmic_bytes_per_launch": dc["w"]["bytes"],
                                    "note": "peak = nominal HBM3E rate; a plain streaming read of 2 GiB reaches 7.15 TB/s on this chip with nt loads, "
                                            "6.2 TB/s with ordinary ones (profiles/r01/hbm_read_reference_point.log)"}},
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(pf["w"], pf["w"]["out"])
            line["decode"]["cpu_baseline"] = cpu_baseline(dc["w"], dc["w"]["out"])
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
