export MI355_LAB=1      # the library reads its measurement switches (MI355_PREFILL, ...) only with this set
# the C2 family - GQA 32/8 at batch 1, 2, 4, 8 (the per-GPU shares of the multi-GPU curve) and the MHA 32/32 variant
for cfg in "1 4096 32 8" "2 4096 32 8" "4 4096 32 8" "8 4096 32 8" "1 4096 32 32" "4 4096 32 32" "1 4096 64 8" "1 8192 32 8" "1 32768 32 8"; do
  set -- $cfg
  r=$(timeout -k 10 100 python tools/bench_prefill.py --batch $1 --seq $2 --hq $3 --hk $4 2>&1 | tail -1 | sed "s/.*kernel=\([a-z_+0-9]*\).*| sustained/\1/")
  echo "B=$1 L=$2 Hq=$3 Hk=$4: $r"
done
