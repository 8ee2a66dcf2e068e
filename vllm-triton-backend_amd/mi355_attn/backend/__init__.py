"""vLLM platform-plugin entry point (reference: LIB/backend/__init__.py:20-22; entry-point group
`vllm.platform_plugins`, ibm-triton-lib/setup.py:70-72)."""


def register():
    """Register the MI355X attention platform: vLLM calls this and imports the returned class path."""
    return "mi355_attn.backend.platform.MI355Platform"
