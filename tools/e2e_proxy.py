#!/usr/bin/env python3
"""Attention-only replay of the reference's end-to-end latency protocol.

The reference's headline numbers are vLLM end-to-end latencies of Llama-3.1-8B at batch 1, 500 input tokens and
10 ... 12 800 output tokens (scripts/bench_vllm_latency_range.py:48-50,:98-103 drives vLLM's benchmark_latency.py;
scripts/offline_inference.py:43-87 is the same run by hand; full-graph mode, LIB/backend/triton_attn.py:107,:120-128).
vLLM is not installable here, so this tool replays the ATTENTION part of that protocol, as vLLM would drive this backend:

  * Llama-3.1-8B shape: 32 layers, Hq 32 / Hk 8 / D 128, bf16, 16-token pages, one KV cache per layer;
  * one prefill step of 500 tokens, then one decode step per generated token (kv = 501, 502, ...), every step a call of
    `MI355AttentionImpl.forward` per layer with metadata shaped like the builder's (`backend/attn.py`);
  * each phase is ONE HIP graph of its 32 layer calls (what `full_cudagraph_supported` buys): the decode graph is
    captured once at `max_model_len` and replayed for every generated token with the step's `seq_lens` /
    `slot_mapping` copied into the captured tensors first (H2D, as vLLM's runner does per step).

Reported per output length: attention time of the whole generation (GPU time of the replays, events on the replay
stream), microseconds of attention per generated token, and the share of it that is the graph-node floor (32 nodes of
a one-element kernel replayed the same way). The rest of a vLLM step (GEMMs, norms, sampling, scheduler) is NOT here.
"""
import argparse
import math
import os

os.environ.setdefault("MI355_LAB", "1")      # tools may pin kernels through the library's measurement switches
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vllm-triton-backend_amd")]
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--input-len", type=int, default=500)
    ap.add_argument("--output-lens", type=int, nargs="+", default=[10, 100, 200, 400, 800, 1600, 3200, 6400, 12800])
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--kv-cache-dtype", default="auto", choices=["auto", "fp8"])
    ap.add_argument("--sample-every", type=int, default=1, help="replay every n-th decode step and scale (1 = the full generation)")
    args = ap.parse_args()

    from mi355_attn import _lib
    from mi355_attn.backend import attn

    dev = torch.device("cuda:0")
    Hq, Hk, D, page, L, B = 32, 8, 128, 16, args.layers, args.batch
    dt = torch.bfloat16
    fp8 = args.kv_cache_dtype == "fp8"
    max_out = max(args.output_lens)
    max_model_len = args.input_len + max_out + 1
    pps = (max_model_len + page - 1) // page
    nb = B * pps + 4
    torch.manual_seed(0)
    cache_dt = torch.uint8 if fp8 else dt
    kv_caches = []
    for _ in range(L):
        c = (torch.rand(2, nb, page, Hk, D, device=dev) * 2 - 1)
        kv_caches.append(c.to(torch.float8_e4m3fn).view(torch.uint8) if fp8 else c.to(dt))
        del c
    bt = torch.randperm(nb, device=dev)[: B * pps].to(torch.int32).view(B, pps).contiguous()
    scale = 1.0 / math.sqrt(D)
    impls = [attn.MI355AttentionImpl(Hq, D, scale, Hk, None, None, args.kv_cache_dtype) for _ in range(L)]
    layer = types.SimpleNamespace(_k_scale=torch.ones((), device=dev), _v_scale=torch.ones((), device=dev), _q_scale=torch.ones((), device=dev),
                                  _q_scale_float=1.0)

    def metadata(T, max_q, qsl, seq_lens, slot_mapping, max_seq_len):
        return attn.MI355AttentionMetadata(
            num_actual_tokens=T, max_query_len=max_q, avg_query_len=max_q, avg_seq_len=max_seq_len, query_start_loc=qsl, max_seq_len=max_seq_len,
            seq_lens=seq_lens, block_table=bt, slot_mapping=slot_mapping, use_cascade=False, common_prefix_len=0, cu_prefix_query_lens=None,
            prefix_kv_lens=None, suffix_kv_lens=None)

    def slots_of(pos):                       # positions [B, n] -> flat slot ids, on the device
        return (bt.long().gather(1, pos // page) * page + pos % page).reshape(-1)

    def time_graph(fn, reps_warm=3, reps=10):
        s = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            fn()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                fn()
        return g, s

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    # ---- floor: L one-element kernels in a graph -----------------------------------------------------------------
    x = torch.zeros(1, device=dev)
    g_floor, s_floor = time_graph(lambda: [x.add_(1.0) for _ in range(L)])
    with torch.cuda.stream(s_floor):
        for _ in range(5):
            g_floor.replay()
        e0.record(s_floor)
        for _ in range(50):
            g_floor.replay()
        e1.record(s_floor)
    torch.cuda.synchronize()
    floor_us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"# Llama-3.1-8B shape: {L} layers, Hq {Hq} / Hk {Hk} / D {D}, bf16 Q, KV cache {args.kv_cache_dtype}, batch {B}, "
          f"input {args.input_len}, 16-token pages, max_model_len {max_model_len}")
    print(f"graph-node floor: {L} one-element kernels per replay = {floor_us:.1f} us ({floor_us / L:.2f} us per node)", flush=True)

    # ---- prefill step: B x input_len tokens, 32 layers, one graph ---------------------------------------------
    Tp = B * args.input_len
    qp = (torch.rand(Tp, Hq, D, device=dev) * 2 - 1).to(dt)
    kp = (torch.rand(Tp, Hk, D, device=dev) * 2 - 1).to(dt)
    vp = (torch.rand(Tp, Hk, D, device=dev) * 2 - 1).to(dt)
    outp = torch.empty(Tp, Hq * D, dtype=dt, device=dev)
    qsl_p = (torch.arange(B + 1, dtype=torch.int32, device=dev) * args.input_len)
    sl_p = torch.full((B,), args.input_len, dtype=torch.int32, device=dev)
    slot_p = slots_of(torch.arange(args.input_len, device=dev).repeat(B, 1))
    md_p = metadata(Tp, args.input_len, qsl_p, sl_p, slot_p, args.input_len)

    def prefill_step():
        for li in range(L):
            impls[li].forward(layer, qp, kp, vp, kv_caches[li], md_p, output=outp)

    g_p, s_p = time_graph(prefill_step)
    prefill_kernel = _lib.last_kernel()
    with torch.cuda.stream(s_p):
        for _ in range(3):
            g_p.replay()
        e0.record(s_p)
        for _ in range(20):
            g_p.replay()
        e1.record(s_p)
    torch.cuda.synchronize()
    prefill_us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"prefill step ({args.input_len} tokens x {B}): {prefill_us:.1f} us for {L} layers = {prefill_us / L:.2f} us per layer "
          f"(cache write + attention; attention kernel {prefill_kernel}, the cache write inside its launch where the library serves the step fused: "
          f"MI355_FUSED_PREFILL_WRITE={os.environ.get('MI355_FUSED_PREFILL_WRITE', '1')})", flush=True)

    # ---- decode steps: one graph captured at max_model_len, replayed per generated token ----------------------
    qd = (torch.rand(B, Hq, D, device=dev) * 2 - 1).to(dt)
    kd = (torch.rand(B, Hk, D, device=dev) * 2 - 1).to(dt)
    vd = (torch.rand(B, Hk, D, device=dev) * 2 - 1).to(dt)
    outd = torch.empty(B, Hq * D, dtype=dt, device=dev)
    qsl_d = torch.arange(B + 1, dtype=torch.int32, device=dev)
    sl_d = torch.full((B,), max_model_len, dtype=torch.int32, device=dev)      # capture at max_model_len (triton_attn.py:120-128)
    slot_d = slots_of(torch.full((B, 1), max_model_len - 1, device=dev))
    md_d = metadata(B, 1, qsl_d, sl_d, slot_d, max_model_len)

    def decode_step():
        for li in range(L):
            impls[li].forward(layer, qd, kd, vd, kv_caches[li], md_d, output=outd)

    g_d, s_d = time_graph(decode_step)
    decode_kernel = _lib.last_kernel()
    # the steps' metadata, precomputed on the host like the runner's numpy side, copied per step (pinned -> device)
    steps = max_out
    sl_host = (args.input_len + 1 + torch.arange(steps, dtype=torch.int32)).pin_memory()                       # seq_len of step i
    all_pos = (args.input_len + torch.arange(steps, device=dev)).repeat(B, 1)
    slot_host = slots_of(all_pos).view(B, steps).t().contiguous().cpu().pin_memory()                           # [steps, B]
    results = []
    cum_gpu_us, cum_wall_s, done = 0.0, 0.0, 0
    checkpoints = sorted(args.output_lens)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    with torch.cuda.stream(s_d):
        for target in checkpoints:
            n = target - done
            idx = range(done, target, args.sample_every)
            t0 = time.perf_counter()
            ev[0].record(s_d)
            for i in idx:
                sl_d.copy_(sl_host[i:i + 1].expand(B), non_blocking=True)
                slot_d.copy_(slot_host[i], non_blocking=True)
                g_d.replay()
            ev[1].record(s_d)
            s_d.synchronize()
            wall = time.perf_counter() - t0
            k = n / max(len(idx), 1)
            cum_gpu_us += ev[0].elapsed_time(ev[1]) * 1e3 * k
            cum_wall_s += wall * k
            done = target
            results.append((target, cum_gpu_us, cum_wall_s))
    print(f"decode step: {L} layers per replay, one call per layer with the cache write fused into the decode launch (kernel: {decode_kernel}; "
          f"a plan with more splits than the in-kernel merge takes adds the merge launch), metadata copied H2D per step")
    print(f"{'out':>6} {'kv at end':>9} {'attention total ms':>19} {'incl. prefill ms':>17} {'us / token':>11} {'us / token / layer':>19} {'node floor':>11} {'host wall ms':>13}")
    for target, gpu_us, wall_s in results:
        per_tok = gpu_us / target
        print(f"{target:6d} {args.input_len + target:9d} {gpu_us / 1e3:19.2f} {(gpu_us + prefill_us) / 1e3:17.2f} {per_tok:11.1f} {per_tok / L:19.2f} "
              f"{100.0 * floor_us / per_tok:10.0f}% {wall_s * 1e3:13.1f}", flush=True)


if __name__ == "__main__":
    main()
