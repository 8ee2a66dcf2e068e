#!/bin/bash
export MI355_LAB=1      # the library reads its measurement switches (MI355_PREFILL, ...) only with this set
# Same-box A/B of decode libraries on the fp8 shapes: tools/ab_decode_c5.sh <rounds> <lib> [<lib> ...]
# (C5: Hq 64 / Hk 8, batch 16 x 32768 keys, e4m3 KV; and the C3 shape with an fp8 cache)
rounds=$1; shift
for i in $(seq $rounds); do
  for lib in "$@"; do
    a=$(MI355_LIB=$lib timeout -k 10 100 python tools/bench_decode.py --flush none --iters 50 --kvdtype fp8 --batch 16 --kv 32768 --hq 64 --hk 8 2>&1 | tail -1 | sed "s/.*median//;s/min-time.*//")
    b=$(MI355_LIB=$lib timeout -k 10 100 python tools/bench_decode.py --flush none --iters 50 --kvdtype fp8 2>&1 | tail -1 | sed "s/.*median//;s/min-time.*//")
    echo "$lib: C5[$a] C3fp8[$b]"
  done
done
