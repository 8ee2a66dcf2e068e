"""Loader for tests/golden/*.npz (written by tests/golden/make_golden.py)."""

import ast
import glob
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

_VIEW = {
    "float8_e4m3fn": (torch.uint8, torch.float8_e4m3fn),
    "float8_e5m2": (torch.uint8, torch.float8_e5m2),
    "bfloat16": (torch.int16, torch.bfloat16),
}


def names(prefix=""):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = ast.literal_eval(str(z["__meta__"]))
    out = {}
    for k in z.files:
        if k == "__meta__":
            continue
        t = torch.from_numpy(z[k])
        dt = meta["dtypes"][k]
        if dt in _VIEW:
            t = t.view(_VIEW[dt][1])
        out[k] = t
    return meta, out


def widen(meta, t):
    """Fixtures whose reference run was fp32 on values a 16-bit type holds exactly store q / k / v in that type
    (`stored_as`): the tensors as the reference saw them (fp32), for comparisons at fp32 tightness."""
    if "stored_as" not in meta:
        return t
    out_dt = t["out"].dtype
    return {k: (v.to(out_dt) if k in ("q", "k_cache", "v_cache") else v) for k, v in t.items()}


def tolerance(q_dtype, kv_dtype=None):
    """Stated tolerances: fp32 1e-5, fp16 1e-3, bf16 2e-2; +1e-2 for an fp8 KV cache
    (reference: scripts/test.py:310-312)."""
    atol = {torch.float32: 1e-5, torch.float16: 1e-3, torch.bfloat16: 2e-2}[q_dtype]
    rtol = {torch.float32: 1e-5, torch.float16: 1e-3, torch.bfloat16: 2e-2}[q_dtype]
    if kv_dtype in (torch.float8_e4m3fn, torch.float8_e5m2):
        atol += 1e-2
    return atol, rtol
