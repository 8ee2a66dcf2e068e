export MI355_LAB=1      # the library reads its measurement switches (MI355_PREFILL, ...) only with this set
# prefill_pw_kernel: workgroups per KV head (MI355_PW_SLOTS; default = CUs / Hk, "1000" = one item per workgroup)
for shape in "1 4096" "4 2048" "2 4096" "16 4096" "1 16384" "1 2048"; do
  set -- $shape
  for sl in "" 1000 64; do
    r=$(MI355_PW_SLOTS=$sl MI355_PREFILL=pw timeout -k 10 100 python tools/bench_prefill.py --batch $1 --seq $2 2>&1 | tail -1 | sed "s/.*| sustained//")
    echo "B=$1 L=$2 slots=${sl:-default}: $r"
  done
done
