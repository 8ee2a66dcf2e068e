#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X paged-attention hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): "attn fwd TFLOPS (prefill) + KV GB/s (decode), Llama-3-8B GQA seq4k".
The timed region is K steps of the PREFILL workload C2 (BASELINE configs[1]: Hq 32 / Hk 8 / D 128,
one 4096-token sequence per GPU, bf16, paged KV with 16-token pages): `value` = attention-forward
TFLOP/s, counted FA-style (4*q*kv*D*Hq/2 = 137.44 GFLOP per sequence), whole job over all GPUs.
Three more workloads are timed the same way right after it and reported in the same JSON line:
  "decode"      C3 (configs[2]): batch 64 x kv_len 8192 per GPU, bf16 KV            -> KV GB/s
  "decode_fp8"  C5 (configs[4]): Hq 64 / Hk 8, batch 16 x kv_len 32768 per GPU, fp8-e4m3 KV -> KV GB/s
  "mixed"       C4 (configs[3]): the reference harness's mixed batch (32 decodes, 16 partial and 16 full prefills,
                98 336 query tokens), ONE global batch dealt to the N ranks by mi355_attn.parallel.shard_batch
                (cost-balanced, each rank holds only its sequences' pages)          -> TFLOP/s, strong scaling
  "prefill_b8"  the C2 family's batch of 8: 8 sequences x 4096 tokens, ONE global batch dealt to the ranks
                the same way                                                         -> TFLOP/s, strong scaling
  "decode_b64"  C3's batch as ONE global batch of 64 sequences x 8192 keys dealt to the ranks the same way
                                                                                     -> KV GB/s, strong scaling
  "prefill_512" the reference's own latency regime (scripts/bench_vllm_latency_range.py:48-50): one 512-token prompt per GPU
                                                                                     -> us per launch (and TFLOP/s)
  "prefill_fp8" C2's shape over an fp8-e4m3 cache (round 4: the prefill kernel widens the fp8 tiles itself) -> TFLOP/s
`--verify-gather` (prefill_b8): the ranks' outputs assembled by ONE all_gather (parallel.gather_outputs) and compared with
the unsharded batch computed on rank 0 (bitwise: same kernel, same per-unit arithmetic order).
Multi-GPU: the path shards over sequences with no data-path collective: C2/C3/C5 are weak scaling
(every rank owns its own sequences, KV pages and block table), C4 and prefill_b8 are batch-sharded splits of one batch;
the only collectives are the timing barrier and the MAX over ranks of the measured time.
`python bench.py --gpus N` with N > 1 and no torch.distributed environment starts the N ranks itself
(torch.distributed.run as a child process, before this process touches a GPU) and fails loudly when the box has
fewer than N GPUs.
Inputs are synthetic (U(-1,1), seed 0 + rank), resident in HBM before the timed region starts.
"""

from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vllm-triton-backend_amd")]

HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak
# Untimed launches of the same call before the W warm-up steps (reported as "prewarm_s" in the line): a 120 us kernel timed
# 20 times right after its inputs were generated measures the device's clock/power ramp out of idle, not the kernel.
PREWARM_S = float(os.environ.get("MI355_BENCH_PREWARM_S", "0.25"))
PROFILE_DIR = os.path.join(ROOT, "profiles", "r04")
ALL_LEGS = ("prefill", "decode", "decode_fp8", "mixed", "prefill_b8", "decode_b64", "prefill_512", "prefill_fp8")
GLOBAL_LEGS = ("mixed", "prefill_b8", "decode_b64")       # ONE batch (same seed on every rank) dealt to the ranks: strong scaling


def spawn_ranks(args) -> int:
    """--gpus N > 1 without a torch.distributed environment: run the N ranks as a child torch.distributed.run.
    Nothing in this process has touched a GPU yet (device_count() does not initialise the runtime)."""
    import torch

    have = torch.cuda.device_count()
    if have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} asked for, but this box has {have} GPU(s): refusing to report a smaller job under that label",
              file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup)]
    cmd += ["--legs", args.legs]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    if args.verify_gather:
        cmd.append("--verify-gather")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def c4_lens(batch=64, seqlen=4096):
    """The reference harness's mixed batch (scripts/benchmark.py:1053-1112 with decode_share 0.5, partial_prefill_share
    0.5, pattern [1.0], ALTERNATING; restated in tools/microbench.py::make_prefix_batch and pinned by
    tests/test_cpu_host.py): 32 decodes (ctx 4095), 16 partial prefills (ctx 2048 + 2048 new), 16 full prefills."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from microbench import make_prefix_batch

    q, ctx = make_prefix_batch(batch, seqlen, [1.0], 0.5, 0.5, "ALTERNATING", 16)
    return q, [a + b for a, b in zip(q, ctx)]


def make_workload(kind, device, seed, rank, world):
    import torch

    page = 16
    dt = torch.bfloat16
    gen = torch.Generator(device=device).manual_seed(seed)
    g = torch.Generator(device="cpu").manual_seed(seed)
    if kind in GLOBAL_LEGS:                               # one GLOBAL batch (same seed on every rank), sharded below
        Hq, Hk, D = 32, 8, 128
        if kind == "mixed":                               # C4
            qlens, kvlens = c4_lens()
        elif kind == "decode_b64":                        # C3's batch, dealt to the ranks (the weak-scaling "decode" leg replicates it)
            qlens, kvlens = [1] * 64, [8192] * 64
        else:                                             # the C2 family's batch of 8 (B in {2,4,8} for the multi-GPU curve)
            qlens, kvlens = [4096] * 8, [4096] * 8
        S, T = len(qlens), sum(qlens)
        pps = [(n + page - 1) // page for n in kvlens]
        nb = sum(pps) + 8
        k = (torch.rand(nb, page, Hk, D, device=device, generator=gen) * 2 - 1).to(dt)
        v = (torch.rand(nb, page, Hk, D, device=device, generator=gen) * 2 - 1).to(dt)
        q = (torch.rand(T, Hq, D, device=device, generator=gen) * 2 - 1).to(dt)
        perm = torch.randperm(nb, generator=g).to(torch.int32)
        bt = torch.zeros(S, max(pps), dtype=torch.int32)
        o = 0
        for i, n in enumerate(pps):
            bt[i, :n] = perm[o:o + n]
            o += n
        cu = torch.zeros(S + 1, dtype=torch.int32)
        cu[1:] = torch.cumsum(torch.tensor(qlens, dtype=torch.int32), 0)
        from mi355_attn import parallel

        loc = parallel.shard_batch(rank, world, q, k, v, cu, torch.tensor(kvlens, dtype=torch.int32), bt)
        whole = dict(q=q, k_cache=k, v_cache=v, cu_seqlens_q=cu.to(device), seqused_k=torch.tensor(kvlens, dtype=torch.int32, device=device),
                     block_table=bt.to(device), q_len=max(qlens), kv_len=max(kvlens)) if (VERIFY_GATHER and kind == "prefill_b8") else None
        del q, k, v
        lq = (loc.cu_seqlens_q[1:] - loc.cu_seqlens_q[:-1]).tolist()
        lk = loc.seqused_k.tolist()
        flops_all = sum(4.0 * D * Hq * (a * (b - a) + a * (a + 1) / 2) for a, b in zip(qlens, kvlens))
        bytes_all = sum(b * Hk * D * 4.0 for b in kvlens) + 2.0 * T * Hq * D * 2
        return dict(kind=kind, q=loc.q.contiguous(), k_cache=loc.k_cache.contiguous(), v_cache=loc.v_cache.contiguous(), block_table=loc.block_table,
                    cu_seqlens_q=loc.cu_seqlens_q, seqused_k=loc.seqused_k, out=torch.empty_like(loc.q), q_len=max(lq), kv_len=max(lk),
                    Hq=Hq, Hk=Hk, D=D, scale=1.0 / math.sqrt(D), k_scale=None, flops=flops_all, bytes=bytes_all, global_work=True,
                    local=dict(seqs=len(lq), tokens=sum(lq)), shard=loc, whole=whole, total_tokens=T)
    if kind == "prefill":      # C2
        Hq, Hk, D, B, q_len, kv_len, kvdt = 32, 8, 128, 1, 4096, 4096, dt
    elif kind == "prefill_512":   # the latency regime: one 512-token prompt
        Hq, Hk, D, B, q_len, kv_len, kvdt = 32, 8, 128, 1, 512, 512, dt
    elif kind == "prefill_fp8":   # C2 over an fp8-e4m3 cache: the prefill kernel reads (and widens) the fp8 tiles itself, no 16-bit scratch
        Hq, Hk, D, B, q_len, kv_len, kvdt = 32, 8, 128, 1, 4096, 4096, torch.float8_e4m3fn
    elif kind == "decode":     # C3
        Hq, Hk, D, B, q_len, kv_len, kvdt = 32, 8, 128, 64, 1, 8192, dt
    else:                      # C5 "decode_fp8": Llama-3-70B shape, fp8-e4m3 KV (batch not given in BASELINE: 16)
        Hq, Hk, D, B, q_len, kv_len, kvdt = 64, 8, 128, 16, 1, 32768, torch.float8_e4m3fn
    pps = (kv_len + page - 1) // page
    nb = int(B * pps * 1.25)
    # generated on the device to keep start-up short; values U(-1,1) (scripts/benchmark.py:136,:1168-1174)
    k = (torch.rand(nb, page, Hk, D, device=device, generator=gen) * 2 - 1).to(kvdt)
    v = (torch.rand(nb, page, Hk, D, device=device, generator=gen) * 2 - 1).to(kvdt)
    q = (torch.rand(B * q_len, Hq, D, device=device, generator=gen) * 2 - 1).to(dt)
    bt = torch.randperm(nb, generator=g)[: B * pps].to(torch.int32).view(B, pps).to(device)
    cu = (torch.arange(B + 1, dtype=torch.int32) * q_len).to(device)
    sl = torch.full((B,), kv_len, dtype=torch.int32, device=device)
    w = dict(kind=kind, q=q, k_cache=k, v_cache=v, block_table=bt, cu_seqlens_q=cu, seqused_k=sl, out=torch.empty_like(q), B=B, q_len=q_len,
             kv_len=kv_len, Hq=Hq, Hk=Hk, D=D, scale=1.0 / math.sqrt(D), global_work=False,
             k_scale=torch.ones(1, device=device) if kvdt != dt else None)
    if kind in ("prefill", "prefill_512", "prefill_fp8"):
        w["flops"] = 4.0 * q_len * kv_len * D * Hq / 2 * B
        w["bytes"] = (2 * q.numel() + 2 * B * kv_len * Hk * D) * 2.0
    else:
        w["flops"] = 4.0 * B * Hq * kv_len * D
        w["bytes"] = B * kv_len * Hk * D * 2 * float(k.element_size()) + 2 * q.numel() * 2.0 + bt.numel() * 4.0 + (2 * B + 1) * 4.0
    return w


VERIFY_GATHER = False


def verify_gather(w, device, distributed):
    """prefill_b8 --verify-gather: every rank's share through the kernels, ONE all_gather_into_tensor of the padded shares
    (parallel.gather_outputs), compared on rank 0 with the whole batch computed there unsharded. Bitwise: the same
    kernel computes the same rows in the same order whatever rank holds them."""
    import torch
    from mi355_attn import parallel
    from mi355_attn.kernels import unified as ua

    loc, whole = w["shard"], w["whole"]
    torch.cuda.synchronize(device)
    full = parallel.gather_outputs(w["out"], loc, w["total_tokens"]) if distributed else None
    ref = torch.empty_like(whole["q"])
    p, keep = ua.fill_attn_params(whole["q"], whole["k_cache"], whole["v_cache"], ref, whole["cu_seqlens_q"], whole["q_len"], whole["seqused_k"],
                                  whole["kv_len"], w["scale"], (-1, -1), whole["block_table"], 0.0, None, None, None, None)
    ua.launch(p, device)
    torch.cuda.synchronize(device)
    if full is None:                                       # one GPU: the "gather" is the identity on rank 0's share = the batch
        full = torch.empty_like(ref)
        full[loc.token_index.to(device)] = w["out"]
    same = bool(torch.equal(full.view(torch.int16), ref.view(torch.int16)))
    return {"bitwise_equal_to_unsharded": same, "max_abs_diff": float((full.float() - ref.float()).abs().max())}


def build_call(w, device):
    from mi355_attn.kernels import unified as ua

    p, keep = ua.fill_attn_params(w["q"], w["k_cache"], w["v_cache"], w["out"], w["cu_seqlens_q"], w["q_len"], w["seqused_k"], w["kv_len"],
                                  w["scale"], (-1, -1), w["block_table"], 0.0, w["k_scale"], w["k_scale"], None, None)
    w["_keep"] = (p, keep)
    return lambda: ua.launch(p, device)


def timed_steps(call, steps, warmup, device, distributed):
    """PREWARM_S seconds of untimed launches (disclosed in the line), W untimed steps, then EXACTLY K steps bracketed by
    barrier + synchronize. One HIP event pair brackets the K launches on the stream they are launched on: the kernel's
    average launch duration is that span / K (it includes the gaps between back-to-back launches, as a serving loop
    sees them; rocprofv3's per-kernel average under profiles/ is the same number minus those gaps). No event is
    recorded between launches: an event record is a barrier packet, and two per launch cost ~3 % of a 120 us kernel."""
    import torch
    import torch.distributed as dist

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


def host_cpu_share():
    """CPUs this process may actually use: the cgroup CPU quota (a one-GPU job on the GPU box sees 256 CPUs and is given
    16), else the affinity mask, else the CPU count."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return max(1, int(quota) // int(period))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        if q > 0:
            return max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))
    except (OSError, ValueError):
        pass
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


def cpu_baseline(w, out_gpu):
    """Reference-style CPU SDPA path on this box's host cores, on the same inputs (rank 0, N=1). One torch thread per CPU
    the job is GIVEN (more only fight over the quota: 128 threads on a 16-CPU share ran the same SDPA 2.7x slower)."""
    import torch

    torch.set_num_threads(min(torch.get_num_threads(), host_cpu_share()))

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


# device symbol behind each name mi355_last_kernel() reports for the benchmarked calls (profiles are keyed by symbol)
# (symbol prefix, substring the instantiation must carry, substring it must not carry)
KERNEL_SYMBOL = {"prefill_mfma_pw": ("prefill_pw_kernel", "", "\0"), "decode_splitkv": ("decode_splitkv_kernel", "", "e4m3"),
                 "decode_splitkv_fp8": ("decode_splitkv_kernel", "e4m3", "\0")}


def kernel_source_digest():
    """sha256 over the kernel sources (csrc/*.hip, *.h, include/*.h), in name order: ties a counter pass to the code."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "vllm-triton-backend_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def measured_traffic(leg, kernel_name):
    """HBM bytes per launch of the leg's dominant kernel: (2*FETCH_SIZE + WRITE_SIZE)*1024 from rocprofv3 --pmc passes of
    THIS command with --legs <leg> (tools/collect_traffic.py -> profiles/r02/traffic.json, the guide's gfx950
    correction). Hardware counters cannot be read from inside the timed run, so the number comes from that separate,
    committed pass and is reported only when it was taken on the kernel symbol this run launched AND on the kernel sources
    this run was built from (the pass records a digest of csrc/, see kernel_source_digest); otherwise null."""
    path = os.path.join(PROFILE_DIR, "traffic.json")
    symbol = KERNEL_SYMBOL.get(kernel_name.split("+")[0])
    if symbol is None or not os.path.exists(path):
        return None
    try:
        doc = json.load(open(path))
        if doc.get("csrc_sha256") != kernel_source_digest():
            return None                                          # counters of an older kernel: stale, not reported
        prefix, need, forbid = symbol
        for name, rec in doc["legs"].get(leg, {}).items():
            if name.startswith(prefix) and need in name and forbid not in name:
                return rec["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def roofline(bound, work, peak, unit, scale, w, res, steps, extra):
    """`achieved` / `frac`: algorithmic work of one launch over the kernel's average launch duration taken from the SAME
    clock as the leg's `value` (wall clock of the K timed steps between the barriers / K, max over ranks) - so the line's
    `frac` is `value` per GPU over the peak, and never better than what the driver's own clock sees. `*_events`: the same
    with the HIP-event bracket around the K launches on their stream (it leaves out the launch latency in front of the
    first kernel and the wake-up behind the last synchronize: ~60 us per timed region, 3 % of 20 steps of 112 us).
    The rocprofv3 per-kernel average under profiles/ leaves out the gaps between launches as well."""
    t_wall, t_ev = res["wall"] / steps, res["per_launch"]
    achieved, achieved_ev = work / t_wall / scale, work / t_ev / scale
    r = {"bound": bound, "achieved": round(achieved, 2), "peak": peak, "unit": unit, "frac": round(achieved / peak, 4),
         "traffic": measured_traffic(w["kind"], res["kernel"]), "kernel": res["kernel"], "clock": "wall clock of the timed steps / K (the clock of `value`)",
         "kernel_us": round(t_wall * 1e6, 2), "achieved_events": round(achieved_ev, 2), "frac_events": round(achieved_ev / peak, 4),
         "kernel_us_events": round(t_ev * 1e6, 2)}
    r.update(extra)
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify-gather", action="store_true",
                    help="prefill_b8: assemble the ranks' outputs with one all_gather and compare with the unsharded batch (rank 0)")
    ap.add_argument("--legs", default=",".join(ALL_LEGS),
                    help="comma list of the workloads to run (profiles of ONE kernel: --legs prefill | decode | decode_fp8 | mixed | prefill_b8); "
                         "the driver's default runs all of them, the headline value is the prefill leg's")
    args = ap.parse_args()

    global VERIFY_GATHER
    VERIFY_GATHER = args.verify_gather
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ       # inside torch.distributed.run
    if args.gpus > 1 and not launched:
        raise SystemExit(spawn_ranks(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = launched
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path for the product")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        assert dist.get_world_size() == args.gpus
    n_gpus = world

    import __graft_entry__ as ge
    from mi355_attn import _lib

    if not os.path.exists(_lib.LIB_PATH):
        if rank == 0:
            ge.build()
        if distributed:
            dist.barrier()

    legs = [x for x in args.legs.split(",") if x]
    unknown = set(legs) - set(ALL_LEGS)
    if unknown or not legs:
        raise SystemExit(f"bench.py: --legs takes {', '.join(ALL_LEGS)}; got {args.legs!r}")
    results = {}
    for kind in ALL_LEGS:
        if kind not in legs:
            continue
        w = make_workload(kind, device, seed=0 if kind in GLOBAL_LEGS else rank, rank=rank, world=world)
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
        if kind == "prefill_fp8":
            import ctypes
            results[kind]["workspace_bytes"] = int(_lib.load().mi355_attn_workspace_bytes(ctypes.byref(w["_keep"][0])))
        if kind == "prefill_b8" and args.verify_gather:
            results[kind]["gather"] = verify_gather(w, device, distributed)
        for key in ("shard", "whole"):
            w.pop(key, None)
        if kind not in ("prefill", "decode"):
            for key in ("q", "k_cache", "v_cache", "out", "_keep"):     # free the big legs before the next one
                w.pop(key, None)
            torch.cuda.empty_cache()

    if rank == 0:
        K = args.steps
        pf, dc, d8, mx = results.get("prefill"), results.get("decode"), results.get("decode_fp8"), results.get("mixed")
        b8, d64, p512 = results.get("prefill_b8"), results.get("decode_b64"), results.get("prefill_512")
        pf8 = results.get("prefill_fp8")
        line = {"legs": legs}
        if pf:
            pf_val = pf["w"]["flops"] * n_gpus * K / pf["wall"] / 1e12
            line = {
                "metric": "attn fwd TFLOPS (prefill) + KV GB/s (decode), Llama-3-8B GQA seq4k",
                "value": round(pf_val, 2), "unit": "TFLOP/s", "n_gpus": n_gpus, "steps": K, "warmup": args.warmup, "prewarm_s": PREWARM_S,
                "ms_per_step": round(pf["wall"] / K * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "bf16", "data": "synthetic",
                "cache_state": "timed steps reuse the same inputs back to back, no L2 / Infinity-Cache sweep between launches (the reference's do_bench "
                               "flushes): immaterial for the compute-bound C2 (K/V 16.8 MB) and for C3 / C5, whose 2.1 / 1.1 GB streams dwarf the 256 MB cache",
                "config": {"workload": "C2 prefill: Llama-3-8B shape Hq32/Hk8/D128, 1 seq x 4096 tokens per GPU, causal, paged KV (16-token pages)",
                           "global_batch": n_gpus, "seq_len": 4096, "parallelism": f"batch-sharded x{n_gpus}, no collective", "kernel": pf["kernel"]},
                "roofline": roofline("mfma", pf["w"]["flops"], MFMA_BF16_PEAK_TFLOPS, "TFLOP/s", 1e12, pf["w"], pf, K, {
                    "algorithmic_flops_per_launch": pf["w"]["flops"],
                    "note": "peak = nominal dense bf16 MFMA rate (2.4 GHz); under this kernel the chip holds ~2.1-2.3 GHz (in-kernel clock: 2.22 GHz "
                            "median in the tile loop, tools/pw_clock.py, profiles/r03/pw_clock.log)"}),
            }

        def decode_leg(res, label, cfg):
            val = res["w"]["bytes"] * n_gpus * K / res["wall"] / 1e9
            return {"metric": label, "value": round(val, 1), "unit": "GB/s", "ms_per_step": round(res["wall"] / K * 1e3, 4),
                    "config": {"workload": cfg, "global_batch": res["w"]["B"] * n_gpus, "kernel": res["kernel"]},
                    "roofline": roofline("hbm", res["w"]["bytes"], HBM_PEAK_GBS, "GB/s", 1e9, res["w"], res, K, {
                        "algorithmic_bytes_per_launch": res["w"]["bytes"],
                        "note": "peak = nominal HBM3E rate; a plain streaming read of 2 GiB reaches 7.15 TB/s on this chip with nt loads, "
                                "6.2 TB/s with ordinary ones (profiles/r01/hbm_read_reference_point.log)"})}

        if dc:
            line["decode"] = decode_leg(dc, "KV GB/s (paged decode)", "C3 decode: Hq32/Hk8/D128, batch 64 x kv_len 8192 per GPU, bf16, 16-token pages")
        if d8:
            line["decode_fp8"] = decode_leg(d8, "KV GB/s (paged decode, fp8-e4m3 KV)",
                                            "C5 decode: Llama-3-70B shape Hq64/Hk8/D128, batch 16 x kv_len 32768 per GPU, fp8-e4m3 KV, bf16 Q, 16-token pages")
        if mx:
            mx_val = mx["w"]["flops"] * K / mx["wall"] / 1e12          # ONE global batch: strong scaling
            line["mixed"] = {"metric": "attn fwd TFLOPS (mixed chunked-prefill + decode batch)", "value": round(mx_val, 2), "unit": "TFLOP/s",
                             "ms_per_step": round(mx["wall"] / K * 1e3, 4), "scaling": "strong",
                             "kv_gbs": round(mx["w"]["bytes"] * K / mx["wall"] / 1e9, 1),
                             "config": {"workload": "C4: Granite-3.1-8B shape Hq32/Hk8/D128, 64 sequences (32 decodes ctx 4095, 16 partial prefills 2048+2048, "
                                                    "16 full prefills 4096), 98336 query tokens, ONE batch dealt to the ranks by parallel.shard_batch",
                                        "global_batch": 64, "parallelism": f"batch-sharded x{n_gpus} (LPT by attention cost), no collective",
                                        "rank0_share": mx["w"]["local"], "kernel": mx["kernel"]}}
        if b8:
            b8_val = b8["w"]["flops"] * K / b8["wall"] / 1e12          # ONE global batch: strong scaling
            line["prefill_b8"] = {"metric": "attn fwd TFLOPS (prefill, batch of 8 dealt to the ranks)", "value": round(b8_val, 2), "unit": "TFLOP/s",
                                  "ms_per_step": round(b8["wall"] / K * 1e3, 4), "scaling": "strong",
                                  "config": {"workload": "C2 family: Hq32/Hk8/D128, 8 sequences x 4096 tokens, ONE batch dealt to the ranks by parallel.shard_batch",
                                             "global_batch": 8, "parallelism": f"batch-sharded x{n_gpus}, no collective",
                                             "rank0_share": b8["w"]["local"], "kernel": b8["kernel"]}}
            if "gather" in b8:
                line["prefill_b8"]["gather_check"] = b8["gather"]
        if d64:
            d64_val = d64["w"]["bytes"] * K / d64["wall"] / 1e9       # ONE global batch: strong scaling
            line["decode_b64"] = {"metric": "KV GB/s (paged decode, C3's batch of 64 dealt to the ranks)", "value": round(d64_val, 1), "unit": "GB/s",
                                  "ms_per_step": round(d64["wall"] / K * 1e3, 4), "scaling": "strong",
                                  "config": {"workload": "C3 as ONE batch: Hq32/Hk8/D128, 64 sequences x kv_len 8192, bf16, dealt to the ranks by parallel.shard_batch",
                                             "global_batch": 64, "parallelism": f"batch-sharded x{n_gpus}, no collective",
                                             "rank0_share": d64["w"]["local"], "kernel": d64["kernel"]}}
        if p512:
            line["prefill_512"] = {"metric": "us per launch (one 512-token prompt per GPU: the reference's latency regime)",
                                   "value": round(p512["wall"] / K * 1e6, 2), "unit": "us", "higher_is_better": False,
                                   "tflops_per_gpu": round(p512["w"]["flops"] * K / p512["wall"] / 1e12, 1),
                                   "us_per_launch_events": round(p512["per_launch"] * 1e6, 2),
                                   "config": {"workload": "Hq32/Hk8/D128, 1 seq x 512 tokens per GPU, causal, paged KV (16-token pages), launches back to back",
                                              "kernel": p512["kernel"]}}
        if pf8:
            pf8_val = pf8["w"]["flops"] * n_gpus * K / pf8["wall"] / 1e12
            line["prefill_fp8"] = {"metric": "attn fwd TFLOPS (prefill over an fp8-e4m3 KV cache, bf16 queries)", "value": round(pf8_val, 2), "unit": "TFLOP/s",
                                   "ms_per_step": round(pf8["wall"] / K * 1e3, 4), "frac_of_mfma_peak": round(pf8_val / n_gpus / MFMA_BF16_PEAK_TFLOPS, 4),
                                   "workspace_bytes": pf8.get("workspace_bytes"),
                                   "config": {"workload": "C2's shape over an fp8-e4m3 cache: Hq32/Hk8/D128, 1 seq x 4096 tokens per GPU, causal, paged KV (16-token pages), "
                                                          "scales 1.0; the tiles are widened inside the kernel (reference :434-455), no 16-bit copy of the cache",
                                              "kernel": pf8["kernel"]}}
        if n_gpus == 1 and not args.no_cpu_baseline:
            if pf:
                line["cpu_baseline"] = cpu_baseline(pf["w"], pf["w"]["out"])
            if dc:
                line["decode"]["cpu_baseline"] = cpu_baseline(dc["w"], dc["w"]["out"])
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
