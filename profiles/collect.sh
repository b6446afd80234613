#!/bin/bash
# Evidence run for one round tag (e.g. r01_e): bench lines, rocprofv3 kernel stats and the two PMC passes
# for the three workloads.  Run on the GPU box from the repo root:  bash profiles/collect.sh r01_e
# Outputs go to gpurun_out/<tag>/; profiles/summarise.py turns them into the committed summaries.
set -e -o pipefail
TAG=${1:-r01_x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for W in ${WLS:-default prestitch prestitch_fused rrc}; do   # WLS: subset of workloads (a gpurun call is limited to 20 minutes)
  case $W in prestitch_fused) F="--workload prestitch --fused";; *) F="--workload $W";; esac
  echo "== bench $W"; date
  timeout -k 10 700 python3 bench.py $F --steps 10 --warmup 2 --full-record $OUT/bench_${W}_full.json > $OUT/bench_$W.json 2> $OUT/bench_$W.err
  echo "== kernel trace $W"; date
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$W -o trace -- python3 bench.py $F --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-configs --no-cli > $OUT/bench_${W}_under_rocprof.json 2> $OUT/trace_$W.err
  for C in FETCH_SIZE WRITE_SIZE; do
    echo "== pmc $C $W"; date
    timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${C}_$W -o pmc -- python3 bench.py $F --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end --no-configs --no-cli > /dev/null 2> $OUT/pmc_${C}_$W.err
  done
done
if [ -z "$SKIP_EXTRAS" ]; then
echo "== prestitch, fp16-accumulate variant"; timeout -k 10 300 python3 bench.py --workload prestitch --fp16-accumulate --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_prestitch_f16.json 2> $OUT/bench_prestitch_f16.err || true
echo "== prestitch, fused passes, fp16-accumulate variant"; timeout -k 10 300 python3 bench.py --workload prestitch --fused --fp16-accumulate --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_prestitch_fused_f16.json 2> $OUT/bench_prestitch_fused_f16.err || true
echo "== staging probe"; timeout -k 10 300 python3 profiles/experiments/staging_probe.py > $OUT/staging_probe.txt 2>&1 || true
echo "== 12288-wide bench"; timeout -k 10 300 python3 bench.py --width 12288 --steps 10 --warmup 2 --no-end-to-end --no-configs --no-cpu-baseline > $OUT/bench_w12288.json 2> $OUT/bench_w12288.err || true
fi
python3 profiles/summarise.py $TAG
# the raw traces are large; only the summaries travel back
rm -rf $OUT/trace_* $OUT/pmc_*_default $OUT/pmc_*_prestitch $OUT/pmc_*_prestitch_fused $OUT/pmc_*_rrc
tail -n 3 $OUT/*.err | tail -n 40
echo done; date
