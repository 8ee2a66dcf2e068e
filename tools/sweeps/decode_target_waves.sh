export MI355_LAB=1      # the library reads its measurement switches (MI355_PREFILL, ...) only with this set
# MI355_DECODE_TARGET_WAVES sweep behind the split plan of decode_splitkv.hip: bash tools/sweeps/decode_target_waves.sh
for shape in "--batch 1 --kv 8192" "--batch 4 --kv 8192" "--batch 16 --kv 2048" "--batch 16 --kv 8192" "--batch 64 --kv 2048" "--batch 64 --kv 8192" "--batch 128 --kv 8192" "--batch 256 --kv 4096" "--batch 4 --kv 32768" "--batch 16 --kv 32768 --hq 64 --kvdtype fp8" "--batch 64 --kv 8192 --kvdtype fp8"; do
  line="$shape:"
  for t in 512 1024 1536 2048; do
    r=$(MI355_DECODE_TARGET_WAVES=$t timeout -k 10 60 python tools/bench_decode.py --flush none --iters 30 $shape 2>&1 | tail -1 | sed "s/.*median *//;s/ us.*//")
    line="$line  t$t=$r"
  done
  echo "$line"
done
