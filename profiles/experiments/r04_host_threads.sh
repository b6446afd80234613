# round 4: host pool sizes against the box's CPU quota (cpu.max = 16 CPUs): reader pread threads / copy threads
cat /sys/fs/cgroup/cpu.max
for cfg in "32 16" "16 16" "16 8" "12 12" "32 16" "16 16"; do
  set -- $cfg
  export OIP_HOST_READ_THREADS=$1 OIP_HOST_COPY_THREADS=$2
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-configs --full-record gpurun_out/r04_ab.json > gpurun_out/r04_ab_line.json 2> gpurun_out/r04_ab.err || tail -3 gpurun_out/r04_ab.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r04_ab_line.json'))['config']
f=json.load(open('gpurun_out/r04_ab.json'))['cli']['runs']
print('read/copy threads $cfg wall', {k: round(v) for k,v in d.get('cli_wall_ms').items()}, 'pipeline', {k: round(v) for k,v in d.get('cli_pipeline_ms').items()}, 'read_done', [f[k]['log_seconds'].get('read_done') for k in ('raw','raw_again','lzw')])
PY
done
grep -h throttled /sys/fs/cgroup/cpu.stat
