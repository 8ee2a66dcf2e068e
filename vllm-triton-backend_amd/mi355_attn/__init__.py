"""mi355_attn — MI355X-native paged attention backend for vLLM (hand-written HIP for gfx950 behind
a C ABI). Python here is host-side marshalling and the vLLM plugin surface only."""

__version__ = "0.1.0"
