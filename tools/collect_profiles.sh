#!/bin/bash
export MI355_LAB=1      # the library reads its measurement switches (MI355_PREFILL, ...) only with this set
# Collect the round's measurement evidence on a GPU box into <out> (default gpurun_out/prof_r04); the summaries are then
# copied by hand into profiles/<round>/. Counter passes are separate runs with --kernel-trace only (pool rule).
#   bash tools/collect_profiles.sh [out]
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$REPO/gpurun_out/prof_r04}
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)                  # (a relative path would be lost with the cd below)
cd /tmp && export TMPDIR=/tmp
echo "== HBM traffic (FETCH_SIZE / WRITE_SIZE passes)"
python3 $REPO/tools/collect_traffic.py $OUT/traffic > $OUT/traffic.log 2>&1; tail -60 $OUT/traffic.log
mkdir -p $REPO/profiles/r04 && cp $OUT/traffic/traffic.json $REPO/profiles/r04/traffic.json   # bench.py reports it when it matches the built sources
echo "== bench.py (default protocol)"
python3 $REPO/bench.py > $OUT/bench.json 2> $OUT/bench.err || echo "bench.py failed"
tail -c 3000 $OUT/bench.json
echo "== rocprofv3 --kernel-trace --stats of bench.py, one leg per run (a kernel symbol serves several legs)"
for leg in prefill decode decode_fp8 mixed prefill_b8 decode_b64 prefill_512 prefill_fp8; do
  rm -rf $OUT/stats_$leg && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$leg -- python3 $REPO/bench.py --no-cpu-baseline --legs $leg > $OUT/stats_bench_$leg.json 2> $OUT/stats_$leg.err
  f=$(find $OUT/stats_$leg -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then (head -1 "$f"; grep "mi355::" "$f") > $OUT/bench_kernel_stats_$leg.csv; cut -c1-230 $OUT/bench_kernel_stats_$leg.csv; fi
  grep -o '"kernel_us": [0-9.]*' $OUT/stats_bench_$leg.json | head -2
  rm -rf $OUT/stats_$leg
done
echo "== short prompts (VERDICT r03 item 1): default dispatch, then the kernels pinned; graph column = GPU side"
for s in "--seq 256" "--seq 512" "--seq 768" "--seq 1024" "--seq 1536" "--seq 512 --batch 2" "--seq 512 --batch 4" "--seq 512 --batch 8" "--seq 128 --batch 16" "--seq 640 --batch 4 --ctx 512"; do
  python3 $REPO/tools/bench_prefill.py $s 2>&1 | grep -v amdgpu.ids | sed "s/^/default: /"
  MI355_PREFILL=lat MI355_LAT_WAVES=8 python3 $REPO/tools/bench_prefill.py $s 2>&1 | grep -v amdgpu.ids | sed "s/^/lat8:    /"
  MI355_PREFILL=lat MI355_LAT_WAVES=4 python3 $REPO/tools/bench_prefill.py $s 2>&1 | grep -v amdgpu.ids | sed "s/^/lat4:    /"
  MI355_PREFILL=d4 python3 $REPO/tools/bench_prefill.py $s 2>&1 | grep -v amdgpu.ids | sed "s/^/dma4:    /"
done > $OUT/short_prompts.log 2>&1
for s in "--seq 256" "--seq 512" "--seq 1024" "--seq 1536" "--seq 640 --batch 4 --ctx 512" "--seq 512 --batch 8"; do      # ... and over an fp8 cache
  python3 $REPO/tools/bench_prefill.py $s --kvdtype fp8 2>&1 | grep -v amdgpu.ids | sed "s/^/fp8 default: /"
  MI355_PREFILL=v1 python3 $REPO/tools/bench_prefill.py $s --kvdtype fp8 2>&1 | grep -v amdgpu.ids | sed "s/^/fp8 staged:  /"
done >> $OUT/short_prompts.log 2>&1
cut -c1-60,170- $OUT/short_prompts.log
rm -rf $OUT/stats_sp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_sp -- python3 $REPO/tools/bench_prefill.py --seq 512 > /dev/null 2>&1
f=$(find $OUT/stats_sp -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then (head -1 "$f"; grep "mi355::" "$f") > $OUT/bench_kernel_stats_short_1x512.csv; cut -c1-200 $OUT/bench_kernel_stats_short_1x512.csv; fi; rm -rf $OUT/stats_sp
echo "== the 2D kernel's matrix: every prefill variant at 1 x 4096 / 16 x 4096, then kernel stats + MFMA busy of the variants on the 64-rows-per-wave kernel"
bash $REPO/tools/sweeps/prefill_matrix.sh $OUT/prefill_matrix.log; cat $OUT/prefill_matrix.log
for v in "f16:--dtype f16" "sw1024:--window 1024" "fp8:--kvdtype fp8" "sc30:--softcap 30" "f16_b16:--dtype f16 --batch 16" "sw1024_b16:--window 1024 --batch 16" "fp8_b16:--kvdtype fp8 --batch 16" "sc30_b16:--softcap 30 --batch 16"; do
  name=${v%%:*}; args=${v#*:}
  rm -rf $OUT/stats_v && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_v -- python3 $REPO/tools/bench_prefill.py --iters 5 $args > $OUT/variant_$name.log 2>&1
  f=$(find $OUT/stats_v -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then (head -1 "$f"; grep "mi355::" "$f") > $OUT/bench_kernel_stats_variant_$name.csv; cut -c1-200 $OUT/bench_kernel_stats_variant_$name.csv; fi
  rm -rf $OUT/stats_v
  PMC_PASSES=1 python3 $REPO/tools/pmc_collect.py $OUT/pmc_v prefill_pw_kernel -- python3 $REPO/tools/bench_prefill.py --iters 5 $args > $OUT/pmc_variant_$name.txt 2>&1; grep -i "mfma\|busy" $OUT/pmc_variant_$name.txt | head -6
  rm -rf $OUT/pmc_v
done
echo "== fp8 cache: the kernel's own fp8 form vs the dequantising scratch vs a bf16 cache (same box), and a vLLM-shaped fp8 step"
for a in "--batch 1 --seq 4096" "--batch 16 --seq 4096" "--batch 1 --seq 16384" "--batch 1 --seq 4096 --ctx 2048"; do
  python3 $REPO/tools/bench_prefill.py $a --kvdtype fp8 2>&1 | grep -v amdgpu.ids | sed "s/^/fp8 direct:  /"
  MI355_FP8_PREFILL_SCRATCH=1 python3 $REPO/tools/bench_prefill.py $a --kvdtype fp8 2>&1 | grep -v amdgpu.ids | sed "s/^/fp8 scratch: /"
  python3 $REPO/tools/bench_prefill.py $a 2>&1 | grep -v amdgpu.ids | sed "s/^/bf16:        /"
done > $OUT/fp8_prefill.log 2>&1
python3 $REPO/tools/bench_fp8_step.py 2>&1 | grep -v amdgpu.ids >> $OUT/fp8_prefill.log
python3 $REPO/tools/bench_fp8_step.py --kvdtype same 2>&1 | grep -v amdgpu.ids >> $OUT/fp8_prefill.log
cut -c1-75,170- $OUT/fp8_prefill.log
echo "== SQ counters, prefill_pw_kernel at C2"
PMC_PASSES=0,1,2,3,4 MI355_PREFILL=pw python3 $REPO/tools/pmc_collect.py $OUT/pmc_prefill prefill_pw_kernel -- python3 $REPO/tools/bench_prefill.py --iters 5 > $OUT/pmc_prefill.txt 2>&1; cat $OUT/pmc_prefill.txt
echo "== SQ counters, fp8 decode at C5"
PMC_PASSES=0,1,2,3,4 python3 $REPO/tools/pmc_collect.py $OUT/pmc_decode_fp8 decode_splitkv_kernel -- python3 $REPO/tools/bench_decode.py --batch 16 --kv 32768 --hq 64 --hk 8 --kvdtype fp8 --iters 5 > $OUT/pmc_decode_fp8.txt 2>&1; cat $OUT/pmc_decode_fp8.txt
echo "== in-kernel clock (diagnostic build)"
if [ -f $REPO/tools/ab/pwstamp.so ]; then
  MI355_LIB=$REPO/tools/ab/pwstamp.so MI355_PREFILL=pw python3 $REPO/tools/pw_clock.py 1 4096 > $OUT/pw_clock.log 2>&1
  MI355_LIB=$REPO/tools/ab/pwstamp.so MI355_PREFILL=pw python3 $REPO/tools/pw_clock.py 1 16384 >> $OUT/pw_clock.log 2>&1
  grep -v amdgpu.ids $OUT/pw_clock.log
fi
echo "== seam stamps (diagnostic build)"
if [ -f $REPO/tools/ab/pwseam.so ]; then
  MI355_LIB=$REPO/tools/ab/pwseam.so MI355_PREFILL=pw MI355_PW_SEAM=1 python3 $REPO/tools/pw_clock.py 1 4096 > $OUT/pw_seam_final.log 2>&1; grep -v amdgpu.ids $OUT/pw_seam_final.log
fi
echo "== end-to-end protocol, attention only (tools/e2e_proxy.py)"
python3 $REPO/tools/e2e_proxy.py > $OUT/e2e_proxy.log 2>&1; grep -v amdgpu.ids $OUT/e2e_proxy.log
python3 $REPO/tools/e2e_proxy.py --kv-cache-dtype fp8 --output-lens 10 100 800 3200 12800 --sample-every 8 > $OUT/e2e_proxy_fp8.log 2>&1; grep -v amdgpu.ids $OUT/e2e_proxy_fp8.log | tail -8
echo "== decode step latency in a graph"
python3 $REPO/tools/decode_latency.py > $OUT/decode_latency.log 2>&1; tail -12 $OUT/decode_latency.log
echo "== decode microbench C3 / C5"
python3 $REPO/tools/bench_decode.py > $OUT/decode_c3.log 2>&1; tail -2 $OUT/decode_c3.log
python3 $REPO/tools/bench_decode.py --batch 16 --kv 32768 --hq 64 --hk 8 --kvdtype fp8 > $OUT/decode_c5.log 2>&1; tail -2 $OUT/decode_c5.log
rm -rf $OUT/traffic/pmc_* $OUT/pmc_prefill/pass* $OUT/pmc_decode_fp8/pass* 2>/dev/null
du -sh $OUT
