# round 4: the LZW product of the CLI against the number of encoder threads (OIP_TIFF_THREADS; default min(64, host threads / 2))
nproc
for t in default 32 128 default; do
  if [ $t = default ]; then unset OIP_TIFF_THREADS; else export OIP_TIFF_THREADS=$t; fi
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-configs --full-record gpurun_out/r04_ab.json > gpurun_out/r04_ab_line.json 2> gpurun_out/r04_ab.err || tail -3 gpurun_out/r04_ab.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r04_ab_line.json'))['config']
print('threads $t wall', d.get('cli_wall_ms'), 'pipeline', d.get('cli_pipeline_ms'))
PY
done
