#!/bin/bash
export MI355_LAB=1      # the library reads its measurement switches (MI355_PREFILL, ...) only with this set
# The 2D kernel's matrix: sustained TFLOP/s of every prefill variant at 1 x 4096 and 16 x 4096
# (Hq 32 / Hk 8, 16-token pages), kernel names from mi355_last_kernel().   bash tools/sweeps/prefill_matrix.sh [out.log]
cd "$(dirname "$0")/../.."
out=${1:-/dev/stdout}
run() { python tools/bench_prefill.py "$@" 2>&1 | grep -v amdgpu.ids | sed "s/median.*| sustained/sustained/" | awk -v a="$*" '{printf "%-44s %s\n", a, $0}'; }
{
for b in 1 16; do
  run --batch $b
  run --batch $b --dtype f16
  run --batch $b --window 1024
  run --batch $b --window 1024 --dtype f16
  run --batch $b --kvdtype fp8
  run --batch $b --kvdtype fp8 --window 1024
  run --batch $b --softcap 30
  run --batch $b --softcap 50 --window 1024
  run --batch $b --alibi
  run --batch $b --softcap 30 --kvdtype fp8
  run --batch $b --d 64
  run --batch $b --d 80
  run --batch $b --d 256 --hq 16 --hk 8
  run --batch $b --d 96
done
run --batch 1 --seq 16384
run --batch 1 --seq 16384 --dtype f16
run --batch 1 --seq 16384 --window 4096
run --batch 1 --seq 16384 --kvdtype fp8
run --batch 1 --seq 16384 --softcap 50 --window 4096
} > $out
