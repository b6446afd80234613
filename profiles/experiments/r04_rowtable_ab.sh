# round 4: row tables of remap_shift8_rrc_kernel / align_mss8_kernel through scalar loads (oip_uniform) against the previous
# build (vector loads of the same 16-24 bytes in every lane), same box, ABAB
for i in 1 2; do
  for lib in new base; do
    if [ $lib = base ]; then export OIP_LIBRARY=$PWD/profiles/experiments/liboipgpu_base.so; else unset OIP_LIBRARY; fi
    timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-cli --no-configs --full-record gpurun_out/r04_rowtab_${lib}_$i.json > /dev/null 2>&1
    timeout -k 10 300 python bench.py --workload prestitch --fused --steps 10 --warmup 2 --no-cpu-baseline --full-record gpurun_out/r04_rowtab_fused_${lib}_$i.json > /dev/null 2>&1
    timeout -k 10 300 python bench.py --workload prestitch --fused --fp16-accumulate --steps 10 --warmup 2 --no-cpu-baseline --full-record gpurun_out/r04_rowtab_fused16_${lib}_$i.json > /dev/null 2>&1
    python - <<PY
import json
d=json.load(open('gpurun_out/r04_rowtab_${lib}_$i.json')); k=d['kernels']
print('$lib $i default ms_per_step %.3f' % d['ms_per_step'], {n: round(k[n]['avg_ms'],4) for n in ('align_mss_kernel','mss_split_rrc_kernel') if n in k})
for t in ('fused','fused16'):
    d=json.load(open('gpurun_out/r04_rowtab_%s_${lib}_$i.json' % t)); k=d['kernels']
    print('$lib $i', t, 'ms_per_step %.3f' % d['ms_per_step'], {n: round(v['avg_ms'],4) for n,v in k.items() if 'remap' in n or 'rrc' in n})
PY
  done
done
unset OIP_LIBRARY
timeout -k 10 900 python -m pytest tests/test_gpu_resample.py tests/test_gpu_random_parity.py -q -m gpu -x 2>&1 | tail -2
