"""Op surface of the reference package `ibm_triton_lib.kernels`
(LIB/kernels/__init__.py:65-71), restricted to the attention hot path."""

from .unified import unified_attention
from .cache import reshape_and_cache_flash
from .flash import prefill_flash_attention

__all__ = ["unified_attention", "reshape_and_cache_flash", "prefill_flash_attention"]
