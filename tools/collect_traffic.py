#!/usr/bin/env python3
"""Collect HBM traffic of the bench kernels with rocprofv3 PMC counters (separate passes, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass) and
write profiles/<round>/traffic.json, which bench.py reads for `roofline.traffic`.
    cd /tmp && python3 $REPO/tools/collect_traffic.py <out_dir>
hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports exactly half the bytes of a
wide coalesced streaming read (16 B/lane), WRITE_SIZE is exact for 16-B-per-lane stores."""
import csv
import glob
import json
import os

os.environ.setdefault("MI355_LAB", "1")      # tools may pin kernels through the library's measurement switches
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_pass(counter, out, leg):
    subprocess.check_call(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--",
                           "python3", os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "5", "--warmup", "2", "--legs", leg],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    vals = {}
    for f in glob.glob(os.path.join(out, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "mi355::" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in vals.items()}


def main():
    """One pair of passes PER LEG of bench.py: the same kernel symbol serves several legs (prefill_pw_kernel: the C2
    launch and the C4 batch; decode_splitkv_kernel: C3 and the C4 decode rows), and an average over a whole bench.py
    run would mix their launches."""
    out_dir = sys.argv[1]
    os.makedirs(out_dir, exist_ok=True)
    sys.path.insert(0, ROOT)
    import bench
    doc = {"csrc_sha256": bench.kernel_source_digest(), "command": "bench.py --no-cpu-baseline --steps 5 --warmup 2 --legs <leg>", "legs": {}}
    for leg in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["prefill", "decode", "decode_fp8"]):
        fetch = run_pass("FETCH_SIZE", os.path.join(out_dir, f"pmc_fetch_{leg}"), leg)
        write = run_pass("WRITE_SIZE", os.path.join(out_dir, f"pmc_write_{leg}"), leg)
        res = {}
        for k in fetch:
            short = k.split("(")[0].replace("void mi355::", "")
            res[short] = {"FETCH_SIZE_KB": fetch[k], "WRITE_SIZE_KB": write.get(k, 0.0),
                          "hbm_bytes_per_launch": (2 * fetch[k] + write.get(k, 0.0)) * 1024,
                          "note": "(2*FETCH_SIZE + WRITE_SIZE)*1024; FETCH_SIZE doubled per the gfx950 correction for wide coalesced reads"}
        doc["legs"][leg] = res
        print(leg, json.dumps(res, indent=1), flush=True)
    json.dump(doc, open(os.path.join(out_dir, "traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
