export MI355_LAB=1      # the library reads its measurement switches (MI355_PREFILL, ...) only with this set
# A/B on one box, interleaved: prefill_pw_kernel on 32x32x16 (product) vs its 16x16x32 instantiation (MI355_PW_M16=1)
N=${1:-3}
for shape in "1 4096" "4 2048" "1 16384" "16 4096"; do
  set -- $shape
  for i in $(seq $N); do
    for m in 0 1; do
      r=$(MI355_PW_M16=$m MI355_PREFILL=pw timeout -k 10 100 python tools/bench_prefill.py --batch $1 --seq $2 2>&1 | tail -1 | sed "s/.*| sustained//")
      echo "B=$1 L=$2 m16=$m: $r"
    done
  done
done
