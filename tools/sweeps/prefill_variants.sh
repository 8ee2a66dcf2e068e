export MI355_LAB=1      # the library reads its measurement switches (MI355_PREFILL, ...) only with this set
for shape in "1 1024" "1 2048" "1 3072" "1 4096" "4 1024" "2 2048" "8 512" "4 4096" "16 4096" "1 8192" "1 16384"; do
  set -- $shape
  for v in pw d8 d4; do
    r=$(MI355_PREFILL=$v timeout -k 10 100 python tools/bench_prefill.py --batch $1 --seq $2 2>&1 | tail -1 | sed "s/.*| sustained//")
    echo "B=$1 L=$2 $v: $r"
  done
done
for shape in "1 8192 7680" "1 8192 7168" "1 32768 32256" "1 32768 30720" "2 4096 3584"; do
  set -- $shape
  for v in pw d8 d4; do
    r=$(MI355_PREFILL=$v timeout -k 10 100 python tools/bench_prefill.py --batch $1 --seq $2 --ctx $3 2>&1 | tail -1 | sed "s/.*kernel=\([a-z_+0-9]*\).*| sustained/\1/")
    echo "B=$1 L=$2 ctx=$3 $v: $r"
  done
done
