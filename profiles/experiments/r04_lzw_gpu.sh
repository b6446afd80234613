# round 4: the CLI's LZW product with the device strip encoder (default) against the host encoder (OIP_TIFF_GPU_LZW=0)
for m in 1 0 1; do
  export OIP_TIFF_GPU_LZW=$m
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-configs --full-record gpurun_out/r04_ab.json > gpurun_out/r04_ab_line.json 2> gpurun_out/r04_ab.err || tail -3 gpurun_out/r04_ab.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r04_ab_line.json'))['config']
f=json.load(open('gpurun_out/r04_ab.json'))['cli']['runs']['lzw']
print('gpu_lzw $m wall', d.get('cli_wall_ms'), 'pipeline', d.get('cli_pipeline_ms'), 'lzw log', {k: f['log_seconds'].get(k) for k in ('aligned','products_written')}, f.get('product_bytes'), f.get('tiff_lzw_seconds'))
PY
done
