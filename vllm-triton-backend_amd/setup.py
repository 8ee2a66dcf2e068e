"""Packaging of the MI355X attention plugin (reference: ibm-triton-lib/setup.py:70-72 registers the
same entry-point group). The HIP library is built by hipcc (see ../__graft_entry__.py build())."""

import os
import re

from setuptools import find_packages, setup

_HEADER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "mi355_attn.h")
_N = int(re.search(r"#define\s+MI355_ATTN_VERSION\s+(\d+)", open(_HEADER).read()).group(1))     # the one version source

setup(
    name="mi355-attn",
    version=f"{_N // 10000}.{_N // 100 % 100}.{_N % 100}",
    description="MI355X (gfx950) native paged-attention backend for vLLM",
    packages=find_packages(include=["mi355_attn", "mi355_attn.*"]),
    package_data={"mi355_attn": ["libmi355_attn.so"]},
    python_requires=">=3.10",
    entry_points={"vllm.platform_plugins": ["mi355_attn = mi355_attn.backend:register"]},
)
