# round-4 diagnostics of the file -> HBM -> file path (run on the GPU box from the repo root): the CLI leg of bench.py alone
for v in "OIP_STAGE_PIN=thp" "OIP_STAGE_PIN=malloc"; do
  env $v OIP_STAGE_TRACE=1 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-configs --no-cpu-baseline --no-end-to-end --full-record gpurun_out/r04_i_bench_$v.json > /dev/null 2> gpurun_out/r04_i_bench_$v.err || exit 1
  python - <<PY
import json
d=json.load(open('gpurun_out/r04_i_bench_$v.json'))['cli']
for k,v in d['runs'].items():
    ls=v.get('log_seconds',{})
    print('$v',k,round(v['wall_ms']),{a:ls.get(a) for a in ('prepared','read_done','correlation_done','aligned','products_written','since_process_start','reader_in_pread','reader_waiting_for_slot','device_create')}, v.get('correlation_calls_start_plus_ms'), v.get('stderr_tail'))
PY
done
