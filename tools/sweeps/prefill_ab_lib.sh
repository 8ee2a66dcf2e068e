export MI355_LAB=1      # the library reads its measurement switches (MI355_PREFILL, ...) only with this set
# A/B of two library builds on the same box, interleaved: tools/sweeps/prefill_ab_lib.sh <a.so|""> <b.so> [reps]
A=$1; B=$2; N=${3:-3}
for shape in "1 4096" "4 2048" "1 16384"; do
  set -- $shape
  for i in $(seq $N); do
    for lib in "$A" "$B"; do
      r=$(MI355_LIB=$lib MI355_PREFILL=pw timeout -k 10 100 python tools/bench_prefill.py --batch $1 --seq $2 2>&1 | tail -1 | sed "s/.*| sustained//")
      echo "B=$1 L=$2 lib=${lib:-product}: $r"
    done
  done
done
