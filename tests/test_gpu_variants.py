"""-m gpu: the kernel variants the dispatcher picks by shape (or that an A/B switch pins) must each pass the parity
tests on the SMALL shapes too, not only where the dispatcher would choose them. The switches are read once per
process, so each variant runs a reduced selection of the parity tests in a child process (one at a time)."""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(env_extra, selection, keyword=None):
    env = dict(os.environ)
    env.update(env_extra)
    env["MI355_LAB"] = "1"                  # the library reads its measurement switches (the pins below) only behind this gate
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", *selection]
    if keyword:
        cmd += ["-k", keyword]
    r = subprocess.run(cmd, cwd=os.path.dirname(HERE), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, f"{env_extra}: {' '.join(selection)}\n{r.stdout[-3000:]}\n{r.stderr[-1000:]}"
    assert " passed" in r.stdout


def test_prefill_wide_workgroup_variant_on_small_shapes():
    """8 waves / 256-row Q blocks / 3 stages / block table in LDS (auto-selected only for >= 4096 keys)."""
    _run({"MI355_PREFILL": "d8"}, ["tests/test_gpu_prefill.py::test_prefill_mixed_batches", "tests/test_gpu_prefill.py::test_prefill_page_sizes",
                                   "tests/test_gpu_prefill.py::test_prefill_strided_q_and_out", "tests/test_gpu_golden.py",
                                   "tests/test_gpu_large_cache.py", "tests/test_gpu_fuzz.py"],
         keyword="(mixed and 128 and (8-2 or 32-1 or 6-2)) or page_sizes or strided or golden or chunked_prefill or prefill_dma or agree")


def test_prefill_narrow_workgroup_variant_at_full_size():
    """4 waves / 128-row Q blocks / 2 stages at the C2 size, where the dispatcher would pick the wide one."""
    _run({"MI355_PREFILL": "d4"}, ["tests/test_gpu_prefill.py::test_prefill_c2_full_size_properties"])


def test_prefill_register_staged_kernel_for_plain_head_size_128():
    _run({"MI355_PREFILL": "v1"}, ["tests/test_gpu_prefill.py::test_prefill_mixed_batches", "tests/test_gpu_prefill.py::test_prefill_c2_full_size_properties"],
         keyword="(mixed and 128 and (8-2 or 32-1)) or c2_full")
    # ... and its fp8-cache form at head size 128, which short fp8 prompts left for the latency kernel's fp8 form in round 4
    _run({"MI355_PREFILL": "v1"}, ["tests/test_gpu_prefill.py::test_prefill_fp8_kv_cache_on_the_mfma_path",
                                   "tests/test_gpu_prefill.py::test_prefill_fp8_kv_stale_nan_bytes_beyond_the_sequence_are_ignored"], keyword="128 or stale")


def test_prefill_64_rows_per_wave_kernel_on_small_shapes():
    """prefill_pw_kernel (auto-selected from 2048 keys on, bf16) pinned on the small parity shapes, and its per-row
    routine for rows whose scores leave the range of its fixed reference, without a key split in front of it."""
    _run({"MI355_PREFILL": "pw", "MI355_PREFILL_KEY_SPLITS": "1"},
         ["tests/test_gpu_prefill.py::test_prefill_mixed_batches", "tests/test_gpu_prefill.py::test_prefill_page_sizes",
          "tests/test_gpu_prefill.py::test_prefill_strided_q_and_out", "tests/test_gpu_prefill.py::test_prefill_rows_whose_scores_leave_the_fixed_reference_range",
          "tests/test_gpu_fuzz.py"],
         keyword="(mixed and 128 and dtype0) or page_sizes or strided or leave or agree")


def test_prefill_64_rows_per_wave_kernel_in_f16():
    """The f16 instantiation (P <= 65504: every row's reference leaves 22 powers of two above its estimate) pinned on
    the small parity shapes, the C2 size, the lse shapes and rows whose scores sit far from zero."""
    _run({"MI355_PREFILL": "pw", "MI355_PREFILL_KEY_SPLITS": "1"},
         ["tests/test_gpu_prefill.py::test_prefill_mixed_batches", "tests/test_gpu_prefill.py::test_prefill_page_sizes", "tests/test_gpu_fuzz.py"],
         keyword="(mixed and 128 and dtype1) or page_sizes or agree")
    _run({"MI355_PREFILL": "pw", "MI355_PW_SLOTS": "2"}, ["tests/test_gpu_prefill.py::test_prefill_mixed_batches", "tests/test_gpu_prefill_ksplit.py"],
         keyword="(mixed and 128 and dtype1) or fp16")


def test_prefill_64_rows_per_wave_kernel_walking_many_items_per_workgroup():
    """Two workgroups per KV head (MI355_PW_SLOTS): every workgroup walks many work items - several sequences, empty Q
    blocks, key splits - with the next item's loads in flight over the current item's output; batches of several
    sequences draw their items from the ticket counters in the workspace (empty items included: the retry path)."""
    _run({"MI355_PREFILL": "pw", "MI355_PW_SLOTS": "2"},
         ["tests/test_gpu_prefill.py::test_prefill_mixed_batches", "tests/test_gpu_prefill.py::test_prefill_page_sizes",
          "tests/test_gpu_prefill.py::test_prefill_strided_q_and_out", "tests/test_gpu_prefill.py::test_prefill_key_split_with_rows_outside_the_fixed_reference_range",
          "tests/test_gpu_prefill_ksplit.py", "tests/test_gpu_fuzz.py"],
         keyword="(mixed and 128 and dtype0) or page_sizes or strided or outside or ksplit or key_split or agree")
    # the same walk with the static deal for every batch (several sequences normally draw their items from ticket counters)
    _run({"MI355_PREFILL": "pw", "MI355_PW_SLOTS": "2", "MI355_PW_TICKETS": "0"},
         ["tests/test_gpu_prefill.py::test_prefill_mixed_batches", "tests/test_gpu_prefill_ksplit.py", "tests/test_gpu_fuzz.py"],
         keyword="(mixed and 128 and dtype0) or ksplit or key_split or agree")
    _run({"MI355_PREFILL": "pw", "MI355_PW_SLOTS": "3", "MI355_PREFILL_KEY_SPLITS": "1"},
         ["tests/test_gpu_prefill.py::test_prefill_mixed_batches", "tests/test_gpu_prefill.py::test_prefill_c2_full_size_properties",
          "tests/test_gpu_prefill.py::test_prefill_rows_whose_scores_leave_the_fixed_reference_range"],
         keyword="(mixed and 128 and dtype0) or c2_full or leave")


def test_prefill_64_rows_per_wave_kernel_on_the_16x16x32_matrix_instruction():
    """The kernel's 16x16x32 instantiation (both contractions on v_mfma_f32_16x16x32_bf16, row sums on the matrix pipe;
    the dispatcher's choice from 4096 keys on) pinned on the small parity shapes: many items per workgroup, the per-row
    routine, the C2 size; and the 32x32x16 instantiation pinned where the dispatcher would pick the other one."""
    _run({"MI355_PREFILL": "pw", "MI355_PW_M16": "1", "MI355_PREFILL_KEY_SPLITS": "1"},
         ["tests/test_gpu_prefill.py::test_prefill_mixed_batches", "tests/test_gpu_prefill.py::test_prefill_page_sizes",
          "tests/test_gpu_prefill.py::test_prefill_strided_q_and_out", "tests/test_gpu_prefill.py::test_prefill_rows_whose_scores_leave_the_fixed_reference_range",
          "tests/test_gpu_prefill.py::test_prefill_c2_full_size_properties", "tests/test_gpu_fuzz.py"],
         keyword="(mixed and 128 and dtype0) or page_sizes or strided or leave or agree or c2_full")
    _run({"MI355_PREFILL": "pw", "MI355_PW_M16": "1", "MI355_PW_SLOTS": "2"},
         ["tests/test_gpu_prefill.py::test_prefill_mixed_batches", "tests/test_gpu_prefill.py::test_prefill_key_split_with_rows_outside_the_fixed_reference_range",
          "tests/test_gpu_prefill_ksplit.py", "tests/test_gpu_fuzz.py"],
         keyword="(mixed and 128 and dtype0) or outside or ksplit or key_split or agree")
    _run({"MI355_PREFILL": "pw", "MI355_PW_M16": "0"},
         ["tests/test_gpu_prefill.py::test_prefill_c2_full_size_properties", "tests/test_gpu_prefill_ksplit.py", "tests/test_gpu_lse.py"])
    # the row sums of the 16x16x32 form come off the matrix pipe: its lse on every shape of the lse tests
    _run({"MI355_PREFILL": "pw", "MI355_PW_M16": "1"}, ["tests/test_gpu_lse.py"])


def test_prefill_8_wave_kernel_with_rows_outside_its_first_reference():
    _run({"MI355_PREFILL": "d8", "MI355_PREFILL_KEY_SPLITS": "1"},
         ["tests/test_gpu_prefill.py::test_prefill_rows_whose_scores_leave_the_fixed_reference_range"])


def test_decode_merge_in_a_launch_of_its_own():
    """The separate merge kernel, also where the in-kernel last-arriver merge would be used."""
    _run({"MI355_DECODE_MERGE_KERNEL": "1"}, ["tests/test_gpu_decode.py::test_decode_heads_and_head_sizes", "tests/test_gpu_decode.py::test_decode_split_counts_agree",
                                              "tests/test_gpu_decode.py::test_decode_fp8_kv_cache_on_the_mfma_path"],
         keyword="(heads and (128-32-8 or 64-6-2 or 256-16-1)) or (counts and (3 or 64)) or (fp8 and 32-8-128 and dtype0)")
