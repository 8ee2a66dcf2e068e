#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own Triton kernels.

Runs only in the build container (needs /root/reference, triton, torch; no GPU): the reference
kernels are executed on CPU under TRITON_INTERPRET=1. Nothing from the reference is copied: the
fixtures hold inputs and the outputs the reference computed for them.

In-memory stubs (sys.modules only) stand in for the reference's un-vendored imports:
  triton_dejavu  -> jitcache = identity; autotune = wrapper injecting a fixed {BLOCK_M, BLOCK_N}
  vllm.platforms -> current_platform with the three predicates the legacy kernels query at import
The reference's latent BLOCK_N < BLOCK_SIZE defect is avoided by always
choosing BLOCK_N >= page size.

    python tests/golden/make_golden.py            # regenerate everything
"""

import importlib.util
import os
import sys
import types

os.environ["TRITON_INTERPRET"] = "1"
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np  # noqa: E402
import torch  # noqa: E402

REF = "/root/reference"
LIBK = f"{REF}/ibm-triton-lib/ibm_triton_lib/kernels"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(OUT, "..", ".."))
from oracle.paged_attention_oracle import make_paged_inputs  # noqa: E402  (input generator only)

TILE = {"BLOCK_M": 16, "BLOCK_N": 16}


def install_stubs():
    dj = types.ModuleType("triton_dejavu")

    class _Lock:
        def lock(self): pass
        def unlock(self): pass
        def __enter__(self): return self
        def __exit__(self, *a): return False

    class _Tuned:
        def __init__(self, fn):
            self.fn = fn

        def __getitem__(self, grid):
            def run(*args, **kwargs):
                kwargs.update(TILE)
                g = grid(kwargs) if callable(grid) else grid
                return self.fn[g](*args, **kwargs)
            return run

    dj.jitcache = lambda **kw: (lambda fn: fn)
    dj.autotune = lambda **kw: (lambda fn: _Tuned(fn))
    dj.ConfigSpace = lambda *a, **kw: None
    dj.global_cache_lock = _Lock()
    sys.modules["triton_dejavu"] = dj

    vllm = types.ModuleType("vllm")
    plat = types.ModuleType("vllm.platforms")

    class _Platform:
        @staticmethod
        def has_device_capability(*a, **k): return True
        @staticmethod
        def is_rocm(): return True
        @staticmethod
        def get_device_capability(*a, **k): return (9, 4)

    plat.current_platform = _Platform()
    vllm.platforms = plat
    sys.modules["vllm"] = vllm
    sys.modules["vllm.platforms"] = plat
    # namespace packages so that the legacy modules' relative imports resolve without running the
    # reference's package __init__ (which touches torch.cuda at import)
    for name, path in (
        ("ibm_triton_lib", f"{REF}/ibm-triton-lib/ibm_triton_lib"),
        ("ibm_triton_lib.utils", f"{REF}/ibm-triton-lib/ibm_triton_lib/utils"),
        ("ibm_triton_lib.kernels", LIBK),
        ("ibm_triton_lib.kernels.legacy", f"{LIBK}/legacy"),
    ):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    torch.cuda.get_device_name = lambda *a, **k: "AMD Instinct MI300X"


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def np_of(t):
    if t.dtype in (torch.float8_e4m3fn, torch.float8_e5m2):
        return t.view(torch.uint8).numpy()
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy()
    return t.numpy()


def save(name, meta, **tensors):
    arrays = {k: np_of(v) for k, v in tensors.items()}
    dtypes = {k: str(v.dtype).replace("torch.", "") for k, v in tensors.items()}
    meta = dict(meta)
    meta["dtypes"] = dtypes
    np.savez_compressed(os.path.join(OUT, name + ".npz"), __meta__=np.array(repr(meta)), **arrays)
    print(f"wrote {name}.npz", {k: tuple(v.shape) for k, v in tensors.items() if k in ("q", "out", "k_cache")})


def run_unified(ua, inp, *, window=0, softcap=0.0, alibi=None, k_scale=None, v_scale=None, force=None, tile=(16, 16)):
    TILE["BLOCK_M"], TILE["BLOCK_N"] = tile
    q = inp["q"]
    out = torch.zeros_like(q)
    ql = (inp["cu_seqlens_q"][1:] - inp["cu_seqlens_q"][:-1])
    ua.unified_attention(
        q=q, k=inp["k_cache"], v=inp["v_cache"], out=out, cu_seqlens_q=inp["cu_seqlens_q"],
        max_seqlen_q=int(ql.max()), seqused_k=inp["seqused_k"], max_seqlen_k=int(inp["seqused_k"].max()),
        avg_seqlen_q=float(ql.float().mean()), avg_seqlen_k=float(inp["seqused_k"].float().mean()),
        softmax_scale=inp["scale"], causal=True, window_size=(window - 1, 0) if window else (-1, -1),
        block_table=inp["block_table"], softcap=softcap, q_descale=None, k_descale=k_scale, v_descale=v_scale,
        alibi_slopes=alibi, force_selection=force,
    )
    return out


def unified_case(ua, name, *, seed, query_lens, kv_lens, hq, hk, d, page, dtype, kv_dtype=None, kv_scale=1.0,
                 window=0, softcap=0.0, use_alibi=False, force=None, tile=(16, 16), num_pages=None, v_scale=None, representable_in=None):
    inp = make_paged_inputs(seed, query_lens, kv_lens, hq, hk, d, page, dtype, kv_dtype=kv_dtype, kv_scale=kv_scale,
                            num_pages=num_pages)
    if representable_in is not None:      # fp32 run on values a 16-bit type holds exactly: the same inputs feed the 16-bit kernels
        for key in ("q", "k_cache", "v_cache"):
            inp[key] = inp[key].to(representable_in).to(dtype)
    alibi = None
    if use_alibi:
        alibi = torch.tensor([2.0 ** (-(i + 1) * 8.0 / hq) for i in range(hq)], dtype=torch.float32)
    ks = vs = None
    if kv_dtype is not None and kv_dtype != dtype:
        ks = torch.tensor([kv_scale], dtype=torch.float32)
        vs = torch.tensor([kv_scale if v_scale is None else v_scale], dtype=torch.float32)   # (the cache bytes are the same either way)
    out = run_unified(ua, inp, window=window, softcap=softcap, alibi=alibi, k_scale=ks, v_scale=vs, force=force, tile=tile)
    path = "3d" if (max(query_lens) == 1 and force != 2) else "2d"
    meta = dict(kind="unified", scale=inp["scale"], window=window, softcap=softcap, kv_scale=kv_scale, path=path,
                tile=tile, query_lens=list(query_lens), kv_lens=list(kv_lens))
    if v_scale is not None:
        meta["v_scale"] = v_scale
    t = dict(q=inp["q"], k_cache=inp["k_cache"], v_cache=inp["v_cache"], cu_seqlens_q=inp["cu_seqlens_q"],
             seqused_k=inp["seqused_k"], block_table=inp["block_table"], out=out)
    if alibi is not None:
        t["alibi_slopes"] = alibi
    if representable_in is not None:      # stored in the narrow type (exact), half the bytes
        meta["stored_as"] = str(representable_in).replace("torch.", "")
        for key in ("q", "k_cache", "v_cache"):
            t[key] = t[key].to(representable_in)
    save(name, meta, **t)


def flash_to_v0(kf, vf, x):
    """[nb, page, Hk, D] -> K [nb, Hk, D/x, page, x], V [nb, Hk, D, page]."""
    nb, page, hk, d = kf.shape
    k = kf.view(nb, page, hk, d // x, x).permute(0, 2, 3, 1, 4).contiguous()
    v = vf.permute(0, 2, 3, 1).contiguous()
    return k, v


def legacy_cases():
    load_by_path("ibm_triton_lib.utils.triton_utils", f"{REF}/ibm-triton-lib/ibm_triton_lib/utils/triton_utils.py")
    p2d = load_by_path("ibm_triton_lib.kernels.legacy.triton_paged_decode_attention_2d", f"{LIBK}/legacy/triton_paged_decode_attention_2d.py")
    p3d = load_by_path("ibm_triton_lib.kernels.legacy.triton_paged_decode_attention_3d", f"{LIBK}/legacy/triton_paged_decode_attention_3d.py")
    ctx = load_by_path("ibm_triton_lib.kernels.legacy.triton_prefix_prefill", f"{LIBK}/legacy/triton_prefix_prefill.py")

    # --- paged decode, legacy layouts -------------------------------------------------------
    for name, fn, x, dtype, use_alibi in (
        ("legacy_paged2d_5d_fp32", p2d.paged_attention_triton_2d, 4, torch.float32, False),
        ("legacy_paged2d_4d_fp16", p2d.paged_attention_triton_2d, 0, torch.float16, False),
        ("legacy_paged2d_5d_alibi_fp32", p2d.paged_attention_triton_2d, 4, torch.float32, True),
        ("legacy_paged3d_5d_fp32", p3d.paged_attention_triton_3d, 4, torch.float32, False),
        ("legacy_paged3d_4d_fp16", p3d.paged_attention_triton_3d, 0, torch.float16, False),
    ):
        kv_lens = [70, 45, 33, 129, 1]
        hq, hk, d, page = 8, 2, 64, 16
        inp = make_paged_inputs(11, [1] * len(kv_lens), kv_lens, hq, hk, d, page, dtype)
        if x:
            k0, v0 = flash_to_v0(inp["k_cache"], inp["v_cache"], x)
        else:
            k0 = inp["k_cache"].permute(0, 2, 3, 1).contiguous()
            v0 = inp["v_cache"].permute(0, 2, 3, 1).contiguous()
        alibi = torch.tensor([2.0 ** (-(i + 1)) for i in range(hq)], dtype=torch.float32) if use_alibi else None
        out = torch.zeros_like(inp["q"])
        one = torch.tensor([1.0], dtype=torch.float32)
        fn(out, inp["q"], k0, v0, inp["scale"], one, one, "auto", inp["block_table"], inp["seqused_k"], alibi,
           page, len(kv_lens), hq, hq // hk, d)
        t = dict(q=inp["q"], k_cache_v0=k0, v_cache_v0=v0, seqused_k=inp["seqused_k"], block_table=inp["block_table"], out=out)
        if alibi is not None:
            t["alibi_slopes"] = alibi
        save(name, dict(kind="legacy_decode", scale=inp["scale"], segments=4 if fn is p3d.paged_attention_triton_3d else 0), **t)

    # --- context_attention_fwd (chunked prefill: cached context + linear new tokens) ---------
    for name, dtype, window in (("legacy_ctxfwd_x8_fp32", torch.float32, 0), ("legacy_ctxfwd_x8_fp16", torch.float16, 0),
                                ("legacy_ctxfwd_x8_sw_fp32", torch.float32, 24)):
        query_lens, ctx_lens = [11, 1, 40, 70], [21, 30, 0, 64]
        kv_lens = [a + b for a, b in zip(query_lens, ctx_lens)]
        hq, hk, d, page, x = 4, 2, 64, 16, 8
        inp = make_paged_inputs(13, query_lens, kv_lens, hq, hk, d, page, dtype)
        g = torch.Generator().manual_seed(14)
        T = sum(query_lens)
        k_new = (torch.rand(T, hk, d, generator=g) * 2 - 1).to(dtype)
        v_new = (torch.rand(T, hk, d, generator=g) * 2 - 1).to(dtype)
        k0, v0 = flash_to_v0(inp["k_cache"], inp["v_cache"], x)
        out = torch.zeros_like(inp["q"])
        one = torch.tensor([1.0], dtype=torch.float32)
        ctx.context_attention_fwd(inp["q"], k_new, v_new, out, "auto", k0, v0, inp["block_table"], inp["cu_seqlens_q"],
                                  inp["seqused_k"], max(query_lens), one, one, sliding_window=window or None)
        save(name, dict(kind="legacy_ctxfwd", scale=inp["scale"], window=window, query_lens=query_lens, ctx_lens=ctx_lens),
             q=inp["q"], k_new=k_new, v_new=v_new, k_cache_v0=k0, v_cache_v0=v0, block_table=inp["block_table"],
             cu_seqlens_q=inp["cu_seqlens_q"], seqused_k=inp["seqused_k"], out=out)


def cache_cases():
    """reshape_and_cache_flash: the reference only has the Python restatement
    ref_reshape_and_cache_flash (scripts/vllm_utils.py:377-401); run THAT on seeded inputs."""
    sys.path.insert(0, f"{REF}/scripts")
    vu = types.ModuleType("vllm.utils")
    vu.get_kv_cache_torch_dtype = lambda *a, **k: None
    sys.modules["vllm.utils"] = vu
    sys.modules["vllm"].utils = vu
    import vllm_utils

    for name, dtype in (("reshape_and_cache_flash_bf16", torch.bfloat16), ("reshape_and_cache_flash_fp32", torch.float32)):
        g = torch.Generator().manual_seed(21)
        T, hk, d, page, nb = 37, 4, 64, 16, 9
        key = (torch.rand(T, hk, d, generator=g) * 2 - 1).to(dtype)
        value = (torch.rand(T, hk, d, generator=g) * 2 - 1).to(dtype)
        slots = torch.randperm(nb * page, generator=g)[:T].to(torch.int64)
        kc = torch.zeros(nb, page, hk, d, dtype=dtype)
        vc = torch.zeros(nb, page, hk, d, dtype=dtype)
        vllm_utils.ref_reshape_and_cache_flash(key, value, kc, vc, slots, page, T)
        save(name, dict(kind="cache_write"), key=key, value=value, slot_mapping=slots, k_cache_out=kc, v_cache_out=vc)


def flash_cases():
    """`prefill_flash_attention` = triton_wrapper_forward_prefill (triton_flash_attention.py:1326-1484, exported at
    kernels/__init__.py:65-67): the non-paged variable-length prefill op, causal with the bottom-right aligned mask.
    Its autotuner is triton_dejavu's (stubbed: fixed tile) and its wrapper asks the device for its name and CU count
    (stand-in values: there is no GPU in the build container); the kernel itself is the reference's, run by the interpreter."""
    global TILE
    saved = TILE
    TILE = {"BLOCK_M": 16, "BLOCK_N": 16, "PRE_LOAD_V": False, "GRID_CU_MULTIP": 2}

    class _Props:
        multi_processor_count = 4

    torch.cuda.get_device_properties = lambda *a, **k: _Props()
    fa = load_by_path("ibm_triton_lib.kernels.triton_flash_attention", f"{LIBK}/triton_flash_attention.py")
    try:
        for name, dtype, hq, hk, d, q_lens, k_lens, seed, causal in (
            ("flash_varlen_causal_gqa2_d64_fp32", torch.float32, 4, 2, 64, [5, 17, 1, 33], [9, 17, 33, 40], 61, True),
            ("flash_varlen_causal_gqa4_d128_fp16", torch.float16, 8, 2, 128, [40, 1, 19, 70], [70, 45, 19, 70], 62, True),
            ("flash_varlen_causal_mha_d64_fp16", torch.float16, 4, 4, 64, [33, 3, 16], [33, 35, 64], 63, True),
            ("flash_varlen_noncausal_gqa2_d64_fp32", torch.float32, 4, 2, 64, [5, 17, 1, 33], [9, 17, 33, 40], 64, False),
            ("flash_varlen_noncausal_gqa4_d128_fp16", torch.float16, 8, 2, 128, [40, 1, 19, 70], [70, 45, 19, 70], 65, False),
        ):
            g = torch.Generator().manual_seed(seed)
            cu_q = torch.tensor([0] + torch.tensor(q_lens).cumsum(0).tolist(), dtype=torch.int32)
            cu_k = torch.tensor([0] + torch.tensor(k_lens).cumsum(0).tolist(), dtype=torch.int32)
            q = (torch.rand(int(cu_q[-1]), hq, d, generator=g) * 2 - 1).to(dtype)
            k = (torch.rand(int(cu_k[-1]), hk, d, generator=g) * 2 - 1).to(dtype)
            v = (torch.rand(int(cu_k[-1]), hk, d, generator=g) * 2 - 1).to(dtype)
            scale = 1.0 / (d ** 0.5)
            out = fa.triton_wrapper_forward_prefill(q, k, v, max(q_lens), max(k_lens), cu_q, cu_k, causal=causal, sm_scale=scale)
            out = out[0] if isinstance(out, tuple) else out
            save(name, dict(kind="flash_varlen", scale=scale, max_seqlen_q=max(q_lens), max_seqlen_k=max(k_lens), causal=causal),
                 q=q, k=k, v=v, cu_seqlens_q=cu_q, cu_seqlens_k=cu_k, out=out)
    finally:
        TILE = saved


def main():
    only = sys.argv[1:]            # optional name prefixes: regenerate just those fixtures
    if only:
        global save
        _save = save

        def save(name, meta, **tensors):      # noqa: F811
            if any(name.startswith(o) for o in only):
                _save(name, meta, **tensors)
    install_stubs()
    ua = load_by_path("ref_unified_attention", f"{LIBK}/triton_unified_attention.py")
    bf, hf, f32 = torch.bfloat16, torch.float16, torch.float32
    e4, e5 = torch.float8_e4m3fn, torch.float8_e5m2

    # BASELINE config C1: 1 head, D=64, seq=128, batch 1, fp32 (scripts/benchmark.py micro)
    unified_case(ua, "c1_micro_fp32", seed=0, query_lens=[128], kv_lens=[128], hq=1, hk=1, d=64, page=16, dtype=f32)
    # mixed chunked-prefill + decode batch, GQA 4, D=128
    mixed = dict(query_lens=[7, 1, 1, 40, 9], kv_lens=[70, 45, 33, 70, 33], hq=8, hk=2, d=128, page=16)
    unified_case(ua, "mixed_gqa4_d128_bs16_fp32", seed=1, dtype=f32, **mixed)
    unified_case(ua, "mixed_gqa4_d128_bs16_fp16", seed=1, dtype=hf, **mixed)
    unified_case(ua, "mixed_gqa4_d128_bs16_fp32_tile64", seed=1, dtype=f32, tile=(64, 64), **mixed)
    # decode-only -> 3D split-KV path (16 segments), multi-page segments
    dec = dict(query_lens=[1] * 5, kv_lens=[300, 17, 256, 1, 129], d=128, page=16)
    unified_case(ua, "decode3d_gqa4_d128_bs16_fp32", seed=2, dtype=f32, hq=8, hk=2, **dec)
    unified_case(ua, "decode3d_gqa4_d128_bs16_fp16", seed=2, dtype=hf, hq=8, hk=2, **dec)
    unified_case(ua, "decode3d_gqa8_d128_bs16_fp32", seed=3, dtype=f32, hq=16, hk=2, **dec)
    unified_case(ua, "decode3d_gqa8_d128_bs16_fp16", seed=3, dtype=hf, hq=16, hk=2, **dec)
    unified_case(ua, "decode2d_gqa4_d128_bs16_fp32", seed=2, dtype=f32, hq=8, hk=2, force=2, **dec)
    # fp8 KV cache, fp16 Q
    unified_case(ua, "decode_fp8e4m3_kv_fp16q_bs16", seed=4, dtype=hf, kv_dtype=e4, kv_scale=0.5, hq=8, hk=2, **dec)
    unified_case(ua, "decode_fp8e4m3_kv_fp16q_bs32", seed=4, dtype=hf, kv_dtype=e4, kv_scale=0.5, hq=8, hk=2,
                 query_lens=[1] * 5, kv_lens=[300, 17, 256, 1, 129], d=128, page=32, tile=(16, 32))
    unified_case(ua, "decode_fp8e5m2_kv_fp16q_bs16", seed=5, dtype=hf, kv_dtype=e5, kv_scale=0.5, hq=8, hk=2, **dec)
    unified_case(ua, "mixed_fp8e4m3_kv_fp16q_bs16", seed=6, dtype=hf, kv_dtype=e4, kv_scale=0.25, **mixed)
    # features
    unified_case(ua, "sw8_mixed_fp32", seed=7, dtype=f32, window=8, **mixed)
    unified_case(ua, "sw8_decode_fp32", seed=7, dtype=f32, window=8, hq=8, hk=2, **dec)
    unified_case(ua, "softcap30_mixed_fp32", seed=8, dtype=f32, softcap=30.0, **mixed)
    unified_case(ua, "alibi_mixed_fp32", seed=9, dtype=f32, use_alibi=True, **mixed)
    unified_case(ua, "alibi_decode_fp32", seed=9, dtype=f32, use_alibi=True, hq=8, hk=2, **dec)
    # head sizes incl. non-powers of two (HEAD_SIZE_PADDED masking :353)
    for d in (64, 80, 96, 256):
        unified_case(ua, f"headsize_{d}_mixed_fp32", seed=10 + d, dtype=f32, query_lens=[5, 1, 19], kv_lens=[37, 50, 19],
                     hq=4, hk=2, d=d, page=16)
    # page size 32 with BLOCK_N = 32
    unified_case(ua, "bs32_blockN32_mixed_fp32", seed=12, dtype=f32, query_lens=[7, 1, 1, 40, 9],
                 kv_lens=[70, 45, 33, 70, 33], hq=8, hk=2, d=128, page=32, tile=(16, 32))
    # MHA (G = 1) and MQA-ish (G = 8, Hk = 1)
    unified_case(ua, "mha_mixed_fp16", seed=13, dtype=hf, query_lens=[3, 1, 33], kv_lens=[35, 64, 33], hq=4, hk=4, d=128, page=16)
    unified_case(ua, "mqa8_mixed_fp16", seed=14, dtype=hf, query_lens=[3, 1, 33], kv_lens=[35, 64, 33], hq=8, hk=1, d=128, page=16)

    # --- round 2: the fast (16-bit, matrix-core) kernels against the reference's own numbers -------------------------
    # features and head sizes in fp16 (the fp32 fixtures above only ever reach the shape-agnostic kernel)
    unified_case(ua, "sw8_mixed_fp16", seed=7, dtype=hf, window=8, **mixed)
    unified_case(ua, "sw8_decode_fp16", seed=7, dtype=hf, window=8, hq=8, hk=2, **dec)
    unified_case(ua, "softcap30_mixed_fp16", seed=8, dtype=hf, softcap=30.0, **mixed)
    unified_case(ua, "alibi_mixed_fp16", seed=9, dtype=hf, use_alibi=True, **mixed)
    unified_case(ua, "alibi_decode_fp16", seed=9, dtype=hf, use_alibi=True, hq=8, hk=2, **dec)
    for d in (64, 80, 96, 120, 256):
        unified_case(ua, f"headsize_{d}_mixed_fp16", seed=10 + d, dtype=hf, query_lens=[5, 1, 19], kv_lens=[37, 50, 19],
                     hq=4, hk=2, d=d, page=16)
    # the reference's own KAT matrix has head size 120 and (40, 40) heads (scripts/test.py:58-63)
    unified_case(ua, "headsize_120_mixed_fp32", seed=130, dtype=f32, query_lens=[5, 1, 19], kv_lens=[37, 50, 19], hq=4, hk=2, d=120, page=16)
    unified_case(ua, "headsize_120_decode_fp16", seed=131, dtype=hf, query_lens=[1, 1, 1], kv_lens=[37, 150, 19], hq=4, hk=2, d=120, page=16)
    unified_case(ua, "heads_40_40_mixed_fp16", seed=15, dtype=hf, query_lens=[3, 1, 9], kv_lens=[19, 24, 9], hq=40, hk=40, d=64, page=16, num_pages=6)
    unified_case(ua, "heads_40_40_decode_fp16", seed=16, dtype=hf, query_lens=[1, 1], kv_lens=[40, 33], hq=40, hk=40, d=64, page=16, num_pages=7)
    # fp8 KV with scales that are no powers of two and differ between K and V (the kernels fold k_scale into the softmax
    # scale and v_scale into the normalisation; the reference rounds (fp8 -> f32) * scale to the Q type first, :434-455)
    unified_case(ua, "decode_fp8e4m3_kv_fp16q_scales", seed=17, dtype=hf, kv_dtype=e4, kv_scale=0.0237, v_scale=0.041, hq=8, hk=2, **dec)
    unified_case(ua, "mixed_fp8e4m3_kv_fp16q_scales", seed=18, dtype=hf, kv_dtype=e4, kv_scale=0.0237, v_scale=0.041, **mixed)
    unified_case(ua, "mixed_fp8e5m2_kv_fp16q_scales", seed=19, dtype=hf, kv_dtype=e5, kv_scale=0.0237, v_scale=0.041, **mixed)

    # --- round 3: the bf16 fast path (prefill_pw_kernel: bf16 only, chosen from 2048 keys on) against the reference's
    # 2D kernel itself. bf16 does not run under the Triton interpreter, so the reference computes in fp32
    # on bf16-REPRESENTABLE inputs, BLOCK_M = BLOCK_N = 64: a 256-token chunk over a 2048-token context (2304 keys)
    unified_case(ua, "long_chunk_gqa4_d128_bf16rep_fp32", seed=20, dtype=f32, query_lens=[256], kv_lens=[2304], hq=4, hk=1, d=128, page=16,
                 tile=(64, 64), representable_in=bf)
    unified_case(ua, "long_chunk_gqa4_d128_f16rep_fp32", seed=21, dtype=f32, query_lens=[256], kv_lens=[2304], hq=4, hk=1, d=128, page=16,
                 tile=(64, 64), representable_in=hf)
    # ... and its sliding-window form (prefill_pw_kernel<.., SW>): a window of 300 keys = a few tiles, both masked ends inside the chunk
    unified_case(ua, "long_chunk_sw300_gqa4_d128_bf16rep_fp32", seed=22, dtype=f32, query_lens=[256], kv_lens=[2304], hq=4, hk=1, d=128, page=16,
                 tile=(64, 64), representable_in=bf, window=300)

    legacy_cases()
    cache_cases()
    flash_cases()


if __name__ == "__main__":
    main()
