# round 4: lines requested ahead in remap_shift8_rrc_kernel (row table on scalar loads): 4 (new) / 3 (d3) / 2 (base), same box
for i in 1 2; do
  for lib in new d3 base; do
    if [ $lib = new ]; then unset OIP_LIBRARY; else export OIP_LIBRARY=$PWD/profiles/experiments/liboipgpu_$lib.so; fi
    for acc in "" "--fp16-accumulate"; do
      timeout -k 10 300 python bench.py --workload prestitch --fused $acc --steps 10 --warmup 2 --no-cpu-baseline --full-record gpurun_out/r04_pf.json > /dev/null 2>&1
      python - <<PY
import json
d=json.load(open('gpurun_out/r04_pf.json')); k=d['kernels']
print('$lib $i $acc', 'ms_per_step %.3f' % d['ms_per_step'], {n: round(v['avg_ms'],4) for n,v in k.items() if 'remap_shift8' in n or 'window' in n})
PY
    done
  done
done
unset OIP_LIBRARY
timeout -k 10 600 python -m pytest tests/test_gpu_resample.py -q -m gpu -x 2>&1 | tail -2
