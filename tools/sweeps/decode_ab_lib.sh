export MI355_LAB=1      # the library reads its measurement switches (MI355_PREFILL, ...) only with this set
# A/B of library builds on the decode shapes: tools/sweeps/decode_ab_lib.sh <a.so> <b.so> ...   ("" = the product build)
for args in "--batch 16 --kv 32768 --hq 64 --hk 8 --kvdtype fp8" "--batch 64 --kv 8192" "--batch 64 --kv 8192 --kvdtype fp8" "--batch 16 --kv 32768 --hq 64 --hk 8"; do
  for lib in "$@"; do
    r=$(MI355_LIB=$lib timeout -k 10 200 python tools/bench_decode.py $args 2>&1 | tail -1)
    echo "[$args] lib=${lib:-product}: $r"
  done
done
