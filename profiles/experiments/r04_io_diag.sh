# round-4 diagnostics of the file -> HBM -> file path (run on the GPU box from the repo root)
for pin in thp malloc; do
  OIP_STAGE_PIN=$pin OIP_STAGE_TRACE=1 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-configs --no-cpu-baseline --no-end-to-end --full-record gpurun_out/r04_e_bench_$pin.json > /dev/null 2> gpurun_out/r04_e_bench_$pin.err || exit 1
  python - <<PY
import json
d=json.load(open('gpurun_out/r04_e_bench_$pin.json'))['cli']
print('version_only_ms', d.get('version_only_ms'))
for k,v in d['runs'].items():
    ls=v.get('log_seconds',{})
    print('pin=$pin',k,round(v['wall_ms']),{a:ls.get(a) for a in ('prepared','read_done','correlation_done','aligned','products_written','since_process_start','reader_in_pread','reader_waiting_for_slot','device_create')}, v.get('correlation_calls_start_plus_ms'), v.get('stderr_tail'))
PY
done
