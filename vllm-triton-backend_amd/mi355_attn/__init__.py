"""mi355_attn — MI355X-native paged attention backend for vLLM (hand-written HIP for gfx950 behind
a C ABI). Python here is host-side marshalling and the vLLM plugin surface only."""

from ._version import version_string as _version_string

__version__ = _version_string()      # = MI355_ATTN_VERSION of include/mi355_attn.h, the library's own number
