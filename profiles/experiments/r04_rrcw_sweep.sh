# round 4: the wave-local RRC-on-load resampling kernel against the LDS form (run on the GPU box from the repo root)
for form in dpp bpermute lds; do
  OIP_REMAP_RRC_FORM=$form timeout -k 10 600 python -m pytest tests/test_gpu_resample.py -q -m gpu -x -k "window or rrc or f16" 2>&1 | tail -2
done
for cfg in "lds 3" "dpp 3" "dpp 4" "bpermute 3" "bpermute 4"; do
  set -- $cfg
  for acc in "" "--fp16-accumulate"; do
    OIP_REMAP_RRC_FORM=$1 OIP_REMAP_RRC_OCC=$2 timeout -k 10 300 python bench.py --workload prestitch --fused $acc --steps 10 --warmup 2 --no-cpu-baseline --full-record gpurun_out/r04_rrcw_$1_$2$acc.json > /dev/null 2>&1
    python - <<PY
import json
d=json.load(open('gpurun_out/r04_rrcw_$1_$2$acc.json'))
k={n:round(v['avg_ms'],3) for n,v in d['kernels'].items() if 'remap' in n or 'rrc' in n}
print('$1 occ$2 $acc', 'ms_per_step', round(d['ms_per_step'],3), k)
PY
  done
done
