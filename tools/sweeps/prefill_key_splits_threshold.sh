export MI355_LAB=1      # the library reads its measurement switches (MI355_PREFILL, ...) only with this set
# Where splitting stops paying (workgroup counts around 400-1000): bash tools/sweeps/prefill_key_splits_threshold.sh
for a in "--seq 32768 --ctx 30720" "--seq 8192 --ctx 6144" "--seq 16384 --ctx 14848" "--seq 4096 --ctx 2048 --batch 2" "--seq 8192 --ctx 7680 --batch 3"; do
  for ks in 1 2 3; do
    export MI355_PREFILL_KEY_SPLITS=$ks
    echo -n "$a ks=$ks: "; timeout -k 10 120 python tools/bench_prefill.py $a 2>&1 | tail -1 | sed "s/B=.*kernel=/kernel=/;s/median.*| sustained/sustained/"
  done
done
