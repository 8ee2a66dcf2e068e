import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "vllm-triton-backend_amd")
for p in (ROOT, PKG_DIR):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _cpu_share():
    """CPUs this process may run on (a GPU box hands a one-GPU job 16 of its 256): torch otherwise starts one thread per
    CPU it can SEE, and the oracle's many small host ops then crawl (40 s for one 32768-key row instead of 3)."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]      # the cgroup quota is what the box enforces
        if quota != "max":
            return max(1, int(quota) // int(period))
    except (OSError, ValueError):
        pass
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


def pytest_sessionstart(session):
    """On a GPU box the in-tree library normally arrives with the repo snapshot; if it did not, build it here
    (same image, hipcc present) rather than fail every GPU test on a missing file. Never on a machine without a GPU:
    the CPU suite checks the library that `__graft_entry__.build()` produced."""
    import torch

    torch.set_num_threads(min(torch.get_num_threads(), _cpu_share(), 16))
    lib = os.path.join(PKG_DIR, "mi355_attn", "libmi355_attn.so")
    if torch.cuda.is_available() and not os.path.exists(lib):
        import __graft_entry__ as ge

        ge.build()


def pytest_collection_modifyitems(config, items):
    """GPU tests must never silently pass without a device: skip them cleanly when none is present
    and `-m gpu` was not requested; when `-m gpu` IS requested without a device they fail loudly."""
    import torch

    if torch.cuda.is_available():
        return
    markexpr = config.getoption("-m") or ""
    if "gpu" in markexpr and "not gpu" not in markexpr:
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
