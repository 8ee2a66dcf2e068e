export MI355_LAB=1      # the library reads its measurement switches (MI355_PREFILL, ...) only with this set
# where prefill_pw_kernel takes over from the 4-wave kernel: small shapes, both pinned
for shape in "1 512" "1 1024" "1 1536" "1 2048" "4 512" "8 512" "4 1024" "2 1024" "16 256" "2 2048"; do
  set -- $shape
  for v in pw d4; do
    r=$(MI355_PREFILL=$v timeout -k 10 100 python tools/bench_prefill.py --batch $1 --seq $2 2>&1 | tail -1 | sed "s/.*| sustained//")
    echo "B=$1 L=$2 $v: $r"
  done
done
