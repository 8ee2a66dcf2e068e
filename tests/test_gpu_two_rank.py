"""-m gpu: the N > 1 path with the HIP kernels doing the work. Two rank processes share the one card of the test box
(the driver's scaling run gives each rank its own GPU and RCCL; here the exchange runs over gloo on host copies):
each takes its share of ONE batch with parallel.shard_batch, runs libmi355_attn.so on it, and the gathered output
must match what one process computes for the whole batch. Context parallelism - every rank attends its own key range
of each sequence, ONE exchange of partial outputs + log-sum-exps - is checked the same way."""

import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(dev_inp, scale, lse=None):
    from mi355_attn.kernels import unified_attention

    q = dev_inp["q"]
    out = torch.full_like(q, float("nan"))
    ql = dev_inp["cu_seqlens_q"][1:] - dev_inp["cu_seqlens_q"][:-1]
    unified_attention(q=q, k=dev_inp["k_cache"], v=dev_inp["v_cache"], out=out, cu_seqlens_q=dev_inp["cu_seqlens_q"], max_seqlen_q=int(ql.max()),
                      seqused_k=dev_inp["seqused_k"], max_seqlen_k=int(dev_inp["seqused_k"].max()), avg_seqlen_q=1.0, avg_seqlen_k=1.0,
                      softmax_scale=scale, causal=True, window_size=(-1, -1), block_table=dev_inp["block_table"], softcap=0.0,
                      q_descale=None, k_descale=None, v_descale=None, softmax_lse=lse)
    torch.cuda.synchronize()
    return out


def _worker(rank, world, port, q_out):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "vllm-triton-backend_amd"), os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    # one GPU per rank and RCCL when the box has them (the driver's scaling run), else both ranks on the one card and the
    # exchange over gloo on host copies
    own_gpu = torch.cuda.device_count() >= world
    backend = "nccl" if own_gpu else "gloo"
    dev = torch.device("cuda", rank if own_gpu else 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if own_gpu else {}))
    xdev = dev if own_gpu else torch.device("cpu")          # where the tensors of a collective live
    try:
        from mi355_attn import _lib, parallel
        from oracle import paged_attention_oracle as orc

        # a mixed batch: decode rows, chunked prefills over long contexts, full prefills (one long enough for the
        # 64-rows-per-wave kernel), same seed on both ranks
        query_lens = [1, 1, 300, 1, 2304, 64, 1, 500, 1, 128]
        kv_lens = [700, 33, 300, 2100, 2304, 4000, 1, 1500, 257, 128]
        inp = orc.make_paged_inputs(77, query_lens, kv_lens, 16, 4, 128, 16, torch.bfloat16)
        lb = parallel.shard_batch(rank, world, inp["q"], inp["k_cache"], inp["v_cache"], inp["cu_seqlens_q"], inp["seqused_k"], inp["block_table"])
        local = dict(q=lb.q.to(dev), k_cache=lb.k_cache.to(dev), v_cache=lb.v_cache.to(dev), cu_seqlens_q=lb.cu_seqlens_q.to(dev),
                     seqused_k=lb.seqused_k.to(dev), block_table=lb.block_table.to(dev))
        out_local = _run(local, inp["scale"])
        kernel_local = _lib.last_kernel()
        got = parallel.gather_outputs(out_local.to(xdev), lb, inp["q"].shape[0]).cpu()
        whole = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
        full = _run(whole, inp["scale"]).cpu()
        # the split plans of a share and of the whole batch may differ (rounding of the merges), nothing else may
        batch_err = float((got.float() - full.float()).abs().max())
        nan_free = not bool(torch.isnan(got).any())

        # context parallelism on the decode rows: rank r attends keys [lo_r, hi_r) of every sequence
        dec_lens = [4097, 700, 64, 8200]
        dinp = orc.make_paged_inputs(78, [1] * len(dec_lens), dec_lens, 16, 4, 128, 16, torch.bfloat16)
        ref = _run({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in dinp.items()}, dinp["scale"]).cpu()
        page = 16
        outs, lses = [], []
        for i, n in enumerate(dec_lens):
            lo, hi = parallel.split_key_range(n, page, world)[rank]
            q_i = dinp["q"][i:i + 1].to(dev)
            o_i = torch.zeros_like(q_i)
            lse_i = torch.full((1, 16), float("-inf"), dtype=torch.float32, device=dev)
            if hi > lo:
                bt = dinp["block_table"][i:i + 1, lo // page:].contiguous().to(dev)
                sub = dict(q=q_i, k_cache=whole_kv(dinp, dev)[0], v_cache=whole_kv(dinp, dev)[1], cu_seqlens_q=torch.tensor([0, 1], dtype=torch.int32, device=dev),
                           seqused_k=torch.tensor([hi - lo], dtype=torch.int32, device=dev), block_table=bt)
                o_i = _run(sub, dinp["scale"], lse=lse_i)
            outs.append(o_i.cpu())
            lses.append(lse_i.cpu())
        merged, _ = parallel.all_gather_and_merge(torch.cat(outs).to(xdev), torch.cat(lses).to(xdev))    # nccl: 16-bit partials, merged by the library's kernel
        cp_err = float((merged.float().cpu() - ref.float()).abs().max())
        q_out.put((rank, batch_err, nan_free, cp_err, kernel_local, lb.seq_ids))
    finally:
        dist.destroy_process_group()


_kv_cache = {}


def whole_kv(dinp, dev):
    if "kv" not in _kv_cache:
        _kv_cache["kv"] = (dinp["k_cache"].to(dev), dinp["v_cache"].to(dev))
    return _kv_cache["kv"]


@pytest.mark.timeout(600)
def test_batch_sharding_and_context_parallel_on_the_hip_kernels_world2():
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q_out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q_out.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    owned = []
    for rank, batch_err, nan_free, cp_err, kernel, seq_ids in res:
        assert nan_free, f"rank {rank}: gathered output has unwritten rows"
        assert batch_err <= 2e-2, f"rank {rank}: batch-sharded output differs from the single-process output by {batch_err}"
        assert cp_err <= 2e-2, f"rank {rank}: context-parallel merge differs from the single-process output by {cp_err}"
        assert kernel != "generic", kernel
        owned += seq_ids
    assert sorted(owned) == list(range(10))


def test_bench_two_gpu_path_runs_or_refuses():
    """`bench.py --gpus 2` is the command the driver runs for the scaling curve (one rank per GPU over RCCL: `nccl`
    process group, barrier, MAX over ranks, the strong-scaling legs dealt by parallel.shard_batch, the gather check). With
    two devices visible it runs here as a child, a few steps, and its line must carry every leg asked for and a bitwise
    gather; on a one-GPU box the same command must REFUSE (exit code 2) rather than report a smaller job under that
    label."""
    import json
    import os
    import subprocess
    import sys

    import torch

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--legs", "mixed,prefill_b8,decode_b64", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--verify-gather"]
    env = dict(os.environ, MI355_BENCH_PREWARM_S="0.02", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    if torch.cuda.device_count() < 2:
        assert r.returncode == 2, (r.returncode, r.stderr[-500:])
        assert "refusing" in r.stderr
        return
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    for leg in ("mixed", "prefill_b8", "decode_b64"):
        assert line[leg]["value"] > 0 and line[leg]["scaling"] == "strong", leg
    assert line["prefill_b8"]["gather_check"]["bitwise_equal_to_unsharded"] is True
    assert line["prefill_b8"]["config"]["rank0_share"]["seqs"] == 4


def test_bench_one_gpu_gather_check_and_new_legs():
    """The same legs on ONE GPU (what this box has): the strong-scaling legs' rank-0 share is the whole batch, the gather
    check degenerates to the identity and must still be bitwise, and the latency leg reports microseconds."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--legs", "prefill_b8,decode_b64,prefill_512,prefill_fp8", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--verify-gather"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, MI355_BENCH_PREWARM_S="0.02"), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["prefill_b8"]["gather_check"]["bitwise_equal_to_unsharded"] is True
    assert line["decode_b64"]["unit"] == "GB/s" and line["decode_b64"]["value"] > 1000
    assert line["prefill_512"]["unit"] == "us" and 3 < line["prefill_512"]["value"] < 200
    assert line["prefill_512"]["config"]["kernel"] == "prefill_mfma_lat"
    # C2's shape over an fp8 cache: the prefill kernel's own fp8 form, the workspace is the 256 KiB counter block
    assert line["prefill_fp8"]["config"]["kernel"] == "prefill_mfma_pw_fp8" and line["prefill_fp8"]["workspace_bytes"] == 256 << 10
    assert line["prefill_fp8"]["frac_of_mfma_peak"] > 0.3
