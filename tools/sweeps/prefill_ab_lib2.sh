export MI355_LAB=1      # the library reads its measurement switches (MI355_PREFILL, ...) only with this set
# as prefill_ab_lib.sh, more shapes, one run each: tools/sweeps/prefill_ab_lib2.sh <a.so|""> <b.so>
A=$1; B=$2
for shape in "1 4096" "2 4096" "4 2048" "16 4096" "1 16384" "8 2048" "3 5000"; do
  set -- $shape
  for lib in "$A" "$B"; do
    r=$(MI355_LIB=$lib MI355_PREFILL=pw timeout -k 10 100 python tools/bench_prefill.py --batch $1 --seq $2 2>&1 | tail -1 | sed "s/.*| sustained//")
    echo "B=$1 L=$2 lib=${lib:-product}: $r"
  done
done
