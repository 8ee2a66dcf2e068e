"""Op surface of the reference package `ibm_triton_lib.kernels`
(LIB/kernels/__init__.py:65-71), restricted to the attention hot path."""

from .unified import unified_attention
from .cache import reshape_and_cache_flash

__all__ = ["unified_attention", "reshape_and_cache_flash"]
