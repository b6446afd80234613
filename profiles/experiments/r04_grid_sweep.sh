# round 4: workgroups per CU of the row-block grids (remap_shift8*, align_mss8): is the tail of the last "round" of workgroups visible?
for per in 3 5 6 9 10 12 15 16 20 24 32 48; do
  export OIP_TUNE_WG_PER_CU=$per
  for mode in "" "--fused"; do
    timeout -k 10 300 python bench.py --workload prestitch $mode --steps 6 --warmup 2 --no-cpu-baseline --full-record gpurun_out/r04_ab.json > /dev/null 2> gpurun_out/r04_ab.err || tail -3 gpurun_out/r04_ab.err
    python - <<PY
import json
d=json.load(open('gpurun_out/r04_ab.json')); k=d['kernels']
print('per_cu $per [$mode] ms_per_step %.3f' % d['ms_per_step'], {n: round(v['avg_ms'],4) for n,v in k.items() if n.startswith('remap_shift8')})
PY
  done
  timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end --no-cli --no-configs --full-record gpurun_out/r04_ab.json > /dev/null 2> gpurun_out/r04_ab.err || tail -3 gpurun_out/r04_ab.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r04_ab.json')); k=d['kernels']
print('per_cu $per default', {n: round(v['avg_ms'],4) for n,v in k.items() if n.startswith('align_mss')})
PY
done
