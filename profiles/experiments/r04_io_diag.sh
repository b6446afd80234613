# round-4 diagnostics of the file -> HBM -> file path (run on the GPU box from the repo root)
for i in 1 2 3; do /usr/bin/time -f "oip -v: %e s wall" ./opticalimageprocessor_amd/lib/oip -v > /dev/null; done
for t in 16 32 64; do OIP_TIFF_THREADS=$t LZW_H=8000 ./profiles/experiments/lzw_bench; done
rm -f /dev/shm/lzwbench.tiff
for fe in 1 0; do
  OIP_FAST_EXIT=$fe OIP_STAGE_TRACE=1 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-configs --no-cpu-baseline --no-end-to-end --full-record gpurun_out/r04_d_bench_fe$fe.json > /dev/null 2> gpurun_out/r04_d_bench_fe$fe.err || exit 1
  python - <<PY
import json
d=json.load(open('gpurun_out/r04_d_bench_fe$fe.json'))['cli']
for k,v in d['runs'].items():
    ls=v.get('log_seconds',{})
    print('fast_exit=$fe',k,round(v['wall_ms']),{a:ls.get(a) for a in ('setup','prepared','read_done','correlation_done','aligned','products_written','since_process_start','reader_in_pread','reader_waiting_for_slot','device_create')}, v.get('correlation_calls_start_plus_ms'), v.get('stderr_tail'))
PY
done
