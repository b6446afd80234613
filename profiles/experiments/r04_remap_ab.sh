# round 4: the 8-pixel remap kernels with packed-pair taps and unclamped loads against the build before them; ABAB on one box
for i in 1 2; do
  for lib in new base; do
    if [ $lib = base ]; then export OIP_LIBRARY=$PWD/profiles/experiments/liboipgpu_base.so; else unset OIP_LIBRARY; fi
    for mode in "" "--fp16-accumulate"; do
      timeout -k 10 300 python bench.py --workload prestitch $mode --steps 6 --warmup 2 --no-cpu-baseline --full-record gpurun_out/r04_ab.json > /dev/null 2> gpurun_out/r04_ab.err || tail -3 gpurun_out/r04_ab.err
      python - <<PY
import json
d=json.load(open('gpurun_out/r04_ab.json')); k=d['kernels']
print('$lib $i [$mode] ms_per_step %.3f' % d['ms_per_step'], {n: round(v['avg_ms'],4) for n,v in k.items() if n.startswith('remap')})
PY
    done
  done
done
