export MI355_LAB=1      # the library reads its measurement switches (MI355_PREFILL, ...) only with this set
# Forced key-split counts on chunked-prefill shapes: bash tools/sweeps/prefill_key_splits.sh
for a in "--seq 8192 --ctx 7680" "--seq 8192 --ctx 7168" "--seq 32768 --ctx 32256" "--seq 32768 --ctx 31744" "--seq 4096 --ctx 3584" "--seq 4096 --ctx 3584 --batch 2" "--seq 16384 --ctx 16256"; do
  for ks in 1 0 2 4 8; do
    if [ $ks = 0 ]; then unset MI355_PREFILL_KEY_SPLITS; else export MI355_PREFILL_KEY_SPLITS=$ks; fi
    echo -n "$a ks=$ks: "; timeout -k 10 120 python tools/bench_prefill.py $a 2>&1 | tail -1 | sed "s/B=.*kernel=/kernel=/;s/median.*| sustained/sustained/"
  done
done
