# round 4: align_mss8_kernel (packed-pair taps, unclamped loads) against the build before it; default workload, ABAB on one box
KERNELS="${KERNELS:-align_mss_kernel align_fix_kernel mss_split_rrc_kernel}"
for i in 1 2 3; do
  for lib in new base; do
    if [ $lib = base ]; then export OIP_LIBRARY=$PWD/profiles/experiments/liboipgpu_base.so; else unset OIP_LIBRARY; fi
    timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-cli --no-configs --full-record gpurun_out/r04_ab.json > /dev/null 2> gpurun_out/r04_ab.err || tail -3 gpurun_out/r04_ab.err
    KERNELS="$KERNELS" python - <<PY
import json, os
d=json.load(open('gpurun_out/r04_ab.json')); k=d['kernels']
print('$lib $i ms_per_step %.3f' % d['ms_per_step'], {n: round(k[n]['avg_ms'],4) for n in os.environ['KERNELS'].split() if n in k})
PY
  done
done
