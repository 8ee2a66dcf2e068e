export MI355_LAB=1      # the library reads its measurement switches (MI355_PREFILL, ...) only with this set
# Shapes behind the decode table of DESIGN.md 5 (run from the repo root on a GPU box): bash tools/sweeps/decode_shapes.sh
for shape in "--batch 1" "--batch 4" "--batch 16" "--batch 64" "--batch 128" "--batch 16 --kv 32768 --hq 64 --kvdtype fp8" "--batch 64 --kv 32768 --hq 64 --kvdtype fp8" "--batch 16 --kv 32768 --hq 64" "--batch 64 --kvdtype fp8" "--batch 128 --kvdtype fp8" "--legacy" "--d 64" "--d 96"; do
  for fl in none read; do
    echo -n "$shape flush=$fl: "; timeout -k 10 60 python tools/bench_decode.py --flush $fl --iters 30 $shape 2>&1 | tail -1
  done
done
