export MI355_LAB=1      # the library reads its measurement switches (MI355_PREFILL, ...) only with this set
# Prefetch depth 2 for the bf16 decode kernel (needs tools/ab/lib_pf2.so from tools/build_variant_decode.sh lib_pf2 -DMI355_DECODE_PF=2)
for lib in tools/ab/lib_base.so tools/ab/lib_pf2.so; do
 for t in 512 1024 2048; do
  for shape in "--batch 64" "--batch 16" "--batch 128" "--batch 4 --kv 32768"; do
    r=$(MI355_LIB=$lib MI355_DECODE_TARGET_WAVES=$t timeout -k 10 60 python tools/bench_decode.py --flush none --iters 30 $shape 2>&1 | tail -1 | sed "s/.*median *//;s/ us.*//")
    echo -n "$lib t$t [$shape] $r | "
  done; echo
 done
done
