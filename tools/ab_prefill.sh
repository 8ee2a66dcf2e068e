#!/bin/bash
export MI355_LAB=1      # the library reads its measurement switches (MI355_PREFILL, ...) only with this set
# Same-box A/B of prefill libraries: tools/ab_prefill.sh <rounds> <lib> [<lib> ...]   (sustained figure, batch 1 and 4)
rounds=$1; shift
for i in $(seq $rounds); do
  for lib in "$@"; do
    a=$(MI355_LIB=$lib timeout -k 10 100 python tools/bench_prefill.py --batch 1 2>&1 | tail -1 | sed "s/.*| sustained//")
    b=$(MI355_LIB=$lib timeout -k 10 100 python tools/bench_prefill.py --batch 4 2>&1 | tail -1 | sed "s/.*| sustained//")
    echo "$lib: B1 $a | B4 $b"
  done
done
