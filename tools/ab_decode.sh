#!/bin/bash
export MI355_LAB=1      # the library reads its measurement switches (MI355_PREFILL, ...) only with this set
# Same-box A/B of decode libraries: tools/ab_decode.sh <rounds> <lib> [<lib> ...]  (C3 shape: bf16 no flush / read flush, fp8, batch 16)
rounds=$1; shift
for i in $(seq $rounds); do
  for lib in "$@"; do
    a=$(MI355_LIB=$lib timeout -k 10 100 python tools/bench_decode.py --flush none --iters 50 2>&1 | tail -1 | sed "s/.*median//;s/min-time.*//")
    b=$(MI355_LIB=$lib timeout -k 10 100 python tools/bench_decode.py --flush read --iters 30 2>&1 | tail -1 | sed "s/.*median//;s/min-time.*//")
    c=$(MI355_LIB=$lib timeout -k 10 100 python tools/bench_decode.py --flush none --iters 50 --kvdtype fp8 2>&1 | tail -1 | sed "s/.*median//;s/min-time.*//")
    d=$(MI355_LIB=$lib timeout -k 10 100 python tools/bench_decode.py --flush none --iters 50 --batch 16 2>&1 | tail -1 | sed "s/.*median//;s/min-time.*//")
    echo "$lib: none[$a] read[$b] fp8[$c] b16[$d]"
  done
done
