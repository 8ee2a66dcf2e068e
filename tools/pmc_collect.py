#!/usr/bin/env python3
"""Collect rocprofv3 PMC counters for one kernel of a benchmark command, one counter group per pass
(`--pmc` with `--kernel-trace` only, as the pool requires), and print the per-launch averages.
    cd /tmp && python3 $REPO/tools/pmc_collect.py <out_dir> <kernel-substring> -- python3 $REPO/tools/bench_prefill.py --iters 5"""
import csv
import glob
import json
import os

os.environ.setdefault("MI355_LAB", "1")      # tools may pin kernels through the library's measurement switches
import subprocess
import sys

PASSES = [
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAVES"],
    ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"],
    ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM", "SQ_INST_CYCLES_SALU"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_PENDING_STALL_CYCLES_sum"],
    ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_DATA_FIFO_FULL", "SQ_LDS_CMD_FIFO_FULL", "SQ_LDS_ADDR_CONFLICT", "SQ_WAIT_ANY"],
    ["GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE", "TA_BUSY_avr", "TA_BUSY_max"],
]


def main():
    out_dir, kern = sys.argv[1], sys.argv[2]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    os.makedirs(out_dir, exist_ok=True)
    res = {}
    sel = os.environ.get("PMC_PASSES")
    passes = [PASSES[int(x)] for x in sel.split(",")] if sel else PASSES
    for i, group in enumerate(passes):
        d = os.path.join(out_dir, f"pass{i}")
        rc = subprocess.call(["rocprofv3", "--pmc", *group, "--kernel-trace", "--output-format", "csv", "-d", d, "--", *cmd],
                             stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        if rc != 0:
            print(f"pass {i} ({group}) failed rc={rc}", flush=True)
            continue
        vals = {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if kern in r["Kernel_Name"]:
                    vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in vals.items():
            res[k] = sum(v) / len(v)
        print(f"pass {i}: " + ", ".join(f"{k}={res.get(k, float('nan')):.4g}" for k in group), flush=True)
    json.dump(res, open(os.path.join(out_dir, "pmc.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
