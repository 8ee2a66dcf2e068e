export MI355_LAB=1      # the library reads its measurement switches (MI355_PREFILL, ...) only with this set
# 8-wave (256-row) kernel under a key split vs the 4-wave one: bash tools/sweeps/prefill_wide_with_splits.sh
for a in "--seq 32768 --ctx 30720" "--seq 32768 --ctx 31744" "--seq 8192 --ctx 7168" "--seq 8192 --ctx 7680"; do
  for cfg in "d4 0" "d8 2" "d8 4" "d8 8"; do
    set -- $cfg
    if [ $2 = 0 ]; then unset MI355_PREFILL_KEY_SPLITS; else export MI355_PREFILL_KEY_SPLITS=$2; fi
    echo -n "$a $1 ks=$2: "; MI355_PREFILL=$1 timeout -k 10 120 python tools/bench_prefill.py $a 2>&1 | tail -1 | sed "s/B=.*kernel=/kernel=/;s/median.*| sustained/sustained/"
  done
done
