"""ctypes binding of libmi355_attn.so (C ABI declared in include/mi355_attn.h).

The library is the product; this module only marshals torch tensors into the plain-pointer
structs of the C ABI. There is NO fallback: if the shared library is missing or a call fails the
error is raised to the caller.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch  # noqa: F401  (must be imported before the library so that both share one HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libmi355_attn.so"
LIB_PATH = os.path.join(_HERE, LIB_NAME)

MI355_OK = 0
MI355_ERR_BAD_ARG = -1
MI355_ERR_UNSUPPORTED = -2
MI355_ERR_HIP = -3
MI355_ERR_WORKSPACE = -4

# mi355_dtype
F32, F16, BF16, FP8_E4M3, FP8_E5M2 = 0, 1, 2, 3, 4
# mi355_kernel_select
SELECT_AUTO, SELECT_2D, SELECT_3D, SELECT_GENERIC = 0, 2, 3, 9

EXPORTS = (
    "mi355_attn_version",
    "mi355_last_error",
    "mi355_last_kernel",
    "mi355_attn_workspace_bytes",
    "mi355_unified_attention",
    "mi355_context_attention_fwd_v0",
    "mi355_paged_attention_v0",
    "mi355_decode_write_fusable",
    "mi355_reshape_and_cache_flash",
    "mi355_merge_attention_partials",
)


class AttnParams(C.Structure):
    """struct mi355_attn_params (include/mi355_attn.h)."""

    _fields_ = [
        ("q", C.c_void_p),
        ("out", C.c_void_p),
        ("k_cache", C.c_void_p),
        ("v_cache", C.c_void_p),
        ("block_table", C.c_void_p),
        ("cu_seqlens_q", C.c_void_p),
        ("seqused_k", C.c_void_p),
        ("alibi_slopes", C.c_void_p),
        ("k_scale", C.c_void_p),
        ("v_scale", C.c_void_p),
        ("k_new", C.c_void_p),
        ("v_new", C.c_void_p),
        ("q_dtype", C.c_int32),
        ("kv_dtype", C.c_int32),
        ("num_tokens", C.c_int32),
        ("num_seqs", C.c_int32),
        ("num_q_heads", C.c_int32),
        ("num_kv_heads", C.c_int32),
        ("head_size", C.c_int32),
        ("page_size", C.c_int32),
        ("max_seqlen_q", C.c_int32),
        ("max_seqlen_k", C.c_int32),
        ("q_stride_token", C.c_int64),
        ("q_stride_head", C.c_int64),
        ("out_stride_token", C.c_int64),
        ("out_stride_head", C.c_int64),
        ("k_stride_page", C.c_int64),
        ("k_stride_slot", C.c_int64),
        ("k_stride_head", C.c_int64),
        ("k_stride_dx", C.c_int64),
        ("k_stride_d", C.c_int64),
        ("k_x", C.c_int32),
        ("reserved0", C.c_int32),
        ("v_stride_page", C.c_int64),
        ("v_stride_slot", C.c_int64),
        ("v_stride_head", C.c_int64),
        ("v_stride_d", C.c_int64),
        ("block_table_stride", C.c_int64),
        ("new_stride_token", C.c_int64),
        ("new_stride_head", C.c_int64),
        ("scale", C.c_float),
        ("softcap", C.c_float),
        ("sliding_window", C.c_int32),
        ("skip_decodes", C.c_int32),
        ("only_decodes", C.c_int32),
        ("kernel_select", C.c_int32),
        ("num_segments", C.c_int32),
        ("reserved1", C.c_int32),
        ("lse", C.c_void_p),
        ("lse_stride_token", C.c_int64),
        ("write_new_kv", C.c_int32),
        ("non_causal", C.c_int32),
        ("slot_mapping", C.c_void_p),
        ("slot_mapping_i32", C.c_void_p),
        ("new_kv_all_rows", C.c_int32),
        ("decode_rows_hint", C.c_int32),
    ]


class CacheParams(C.Structure):
    """struct mi355_cache_params (include/mi355_attn.h)."""

    _fields_ = [
        ("key", C.c_void_p),
        ("value", C.c_void_p),
        ("k_cache", C.c_void_p),
        ("v_cache", C.c_void_p),
        ("slot_mapping", C.c_void_p),
        ("slot_mapping_i32", C.c_void_p),
        ("k_scale", C.c_void_p),
        ("v_scale", C.c_void_p),
        ("src_dtype", C.c_int32),
        ("cache_dtype", C.c_int32),
        ("num_tokens", C.c_int32),
        ("num_kv_heads", C.c_int32),
        ("head_size", C.c_int32),
        ("page_size", C.c_int32),
        ("key_stride_token", C.c_int64),
        ("key_stride_head", C.c_int64),
        ("value_stride_token", C.c_int64),
        ("value_stride_head", C.c_int64),
        ("k_stride_page", C.c_int64),
        ("k_stride_slot", C.c_int64),
        ("k_stride_head", C.c_int64),
        ("v_stride_page", C.c_int64),
        ("v_stride_slot", C.c_int64),
        ("v_stride_head", C.c_int64),
    ]


_lib: Optional[C.CDLL] = None


class MI355AttnLibraryError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen the in-tree library (once). Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MI355AttnLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no fallback path."
        )
    lib = C.CDLL(LIB_PATH)
    lib.mi355_attn_version.restype = C.c_int
    lib.mi355_last_error.restype = C.c_char_p
    lib.mi355_last_kernel.restype = C.c_char_p
    lib.mi355_attn_workspace_bytes.restype = C.c_size_t
    lib.mi355_attn_workspace_bytes.argtypes = [C.POINTER(AttnParams)]
    lib.mi355_unified_attention.restype = C.c_int
    lib.mi355_unified_attention.argtypes = [C.POINTER(AttnParams), C.c_void_p, C.c_size_t, C.c_void_p]
    for legacy in (lib.mi355_context_attention_fwd_v0, lib.mi355_paged_attention_v0):
        legacy.restype = C.c_int
        legacy.argtypes = [C.POINTER(AttnParams), C.c_void_p, C.c_size_t, C.c_void_p]
    lib.mi355_decode_write_fusable.restype = C.c_int
    lib.mi355_decode_write_fusable.argtypes = [C.POINTER(AttnParams)]
    lib.mi355_merge_attention_partials.restype = C.c_int
    lib.mi355_merge_attention_partials.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                   C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
    lib.mi355_reshape_and_cache_flash.restype = C.c_int
    lib.mi355_reshape_and_cache_flash.argtypes = [C.POINTER(CacheParams), C.c_void_p]
    _lib = lib
    return lib


def last_error() -> str:
    return load().mi355_last_error().decode()


def last_kernel() -> str:
    return load().mi355_last_kernel().decode()


def check(rc: int, what: str) -> None:
    """Translate a C-ABI return code into the exception type the reference raises for the same
    condition (asserts / ValueError before launch, NotImplementedError for unsupported modes)."""
    if rc == MI355_OK:
        return
    msg = f"{what}: {last_error()} (rc={rc})"
    if rc == MI355_ERR_BAD_ARG:
        raise ValueError(msg)
    if rc == MI355_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)


_TORCH_DTYPES = {
    torch.float32: F32,
    torch.float16: F16,
    torch.bfloat16: BF16,
    torch.float8_e4m3fn: FP8_E4M3,
    torch.float8_e5m2: FP8_E5M2,
}


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _TORCH_DTYPES[dt]
    except KeyError:
        raise NotImplementedError(f"dtype {dt} is not supported by mi355_attn") from None


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def current_stream_handle(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


# per-(device, stream) scratch, grown on demand; replaces the per-call torch.empty at triton_unified_attention.py:950-971.
# Keyed by the stream as well: two streams of one device may run attention calls concurrently (micro-batch overlap),
# and both would otherwise count arrivals and park partials in the same bytes.
# A buffer that has been handed out is NEVER freed: a HIP graph captured earlier holds its raw address (arrival
# counters, split partials) and replays into it long after a later, larger call made the binding move on to a bigger
# buffer. Growth is geometric, so the retired buffers together stay below the size of the live one.
# A CAPTURING stream gets a workspace of its own per capture (keyed by the capture's id, hipStreamGetCaptureInfo): it is
# allocated inside the capture - from the graph's private pool, like any tensor the captured model code allocates -
# and kept alive for the life of the process. `torch.cuda.graph(g)` without `stream=` captures on a private stream no
# eager call ever ran on (torch's default capture stream; `triton.testing.do_bench_cudagraph` and vLLM's full-graph
# capture do exactly that), and the graph is later replayed on whatever stream the caller picks: with bytes of its own
# a replay can run beside eager calls and beside other graphs' replays on any stream without two calls counting
# arrivals or parking partials in the same place. (Rounds 2-3 lent the capture the device's largest eager workspace,
# which is only safe while the replay stream is the lender's stream.) The counters at the head are zero-filled by a
# memset node of the graph (256 KiB, once per graph, harmless on replay: every call leaves them at zero anyway).
_workspaces: dict = {}
_retired: list = []
_COUNTER_BYTES = 256 << 10
_hip = None


def _capture_id(stream_handle: int) -> int:
    """Id of the capture the stream is in (unique per hipStreamBeginCapture), 0 when it cannot be told."""
    global _hip
    try:
        if _hip is None:
            _hip = C.CDLL("libamdhip64.so")
            _hip.hipStreamGetCaptureInfo.restype = C.c_int
            _hip.hipStreamGetCaptureInfo.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_ulonglong)]
        status, cid = C.c_int(0), C.c_ulonglong(0)
        if _hip.hipStreamGetCaptureInfo(C.c_void_p(stream_handle), C.byref(status), C.byref(cid)) != 0:
            return 0
        return int(cid.value)
    except (OSError, AttributeError):
        return 0


def _capture_workspace(device: torch.device, nbytes: int, stream_handle: int) -> torch.Tensor:
    key = (device.type, device.index, "capture", _capture_id(stream_handle))
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            _retired.append(ws)         # earlier nodes of this capture hold its address
        try:
            ws = torch.empty(max(nbytes, 1 << 20, 2 * (ws.numel() if ws is not None else 0)), dtype=torch.uint8, device=device)
            ws[:_COUNTER_BYTES].zero_()
        except RuntimeError as e:       # a capture torch's allocator does not know of (raw hipStreamBeginCapture)
            raise RuntimeError(
                f"mi355_attn workspace ({nbytes} bytes) cannot be allocated inside this stream capture ({e}); capture with "
                "torch.cuda.graph, or pass a workspace of your own through the C ABI"
            ) from None
        _workspaces[key] = ws
    return ws


def workspace(device: torch.device, nbytes: int) -> Optional[torch.Tensor]:
    if nbytes == 0:
        return None
    stream = current_stream_handle(device)
    if torch.cuda.is_current_stream_capturing():
        return _capture_workspace(device, nbytes, stream)
    key = (device.type, device.index, stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            _retired.append(ws)
        # zero-filled once: the head of the workspace holds the split-merge arrival counters, which
        # every call leaves at zero again (include/mi355_attn.h, mi355_attn_workspace_bytes)
        ws = torch.zeros(max(nbytes, 1 << 20, 2 * (ws.numel() if ws is not None else 0)), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws
