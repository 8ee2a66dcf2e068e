"""vLLM platform that selects the MI355X attention backend (reference: LIB/backend/platform.py:17-91,
ROCm branch :74-91; the CUDA branch has no counterpart here)."""

from __future__ import annotations

import torch

BACKEND_CLS = "mi355_attn.backend.attn.MI355AttentionBackend"

try:  # pragma: no cover - needs vLLM
    import vllm.envs as envs
    from vllm.platforms.rocm import RocmPlatform

    _HAVE_VLLM = True
except ImportError:
    _HAVE_VLLM = False

    class RocmPlatform:  # stand-in so that the module imports (and the hook can be tested) without vLLM
        @classmethod
        def get_attn_backend_cls(cls, *args, **kwargs):
            raise RuntimeError("vLLM is not installed")

    class envs:  # noqa: N801
        VLLM_USE_V1 = True


def _is_gfx950() -> bool:
    if not torch.cuda.is_available():
        return False
    name = getattr(torch.cuda.get_device_properties(0), "gcnArchName", "")
    return name.split(":")[0] == "gfx950"


class MI355Platform(RocmPlatform):
    """RocmPlatform whose attention backend is the hand-written gfx950 one."""

    @classmethod
    def get_attn_backend_cls(cls, selected_backend, head_size, dtype, kv_cache_dtype, block_size, use_v1, use_mla) -> str:
        if not envs.VLLM_USE_V1:
            raise RuntimeError("mi355-attn plugin only supports vLLM V1")  # reference: platform.py:89-90
        if use_mla or not _is_gfx950():
            # not ours: MLA models and other GPUs keep vLLM's own ROCm choice
            return super().get_attn_backend_cls(selected_backend, head_size, dtype, kv_cache_dtype, block_size, use_v1, use_mla)
        from . import attn  # noqa: F401  (loads libmi355_attn.so; fails loudly if it is not built)

        return BACKEND_CLS
