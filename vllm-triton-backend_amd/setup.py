"""Packaging of the MI355X attention plugin (reference: ibm-triton-lib/setup.py:70-72 registers the
same entry-point group). The HIP library is built by hipcc (see ../__graft_entry__.py build())."""

from setuptools import find_packages, setup

setup(
    name="mi355-attn",
    version="0.4.0",
    description="MI355X (gfx950) native paged-attention backend for vLLM",
    packages=find_packages(include=["mi355_attn", "mi355_attn.*"]),
    package_data={"mi355_attn": ["libmi355_attn.so"]},
    python_requires=">=3.10",
    entry_points={"vllm.platform_plugins": ["mi355_attn = mi355_attn.backend:register"]},
)
