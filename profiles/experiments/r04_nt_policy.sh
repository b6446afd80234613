# round 4: non-temporal loads / stores in rrc_u16_flat_kernel (OIP_RRC_POLICY: bit 0 loads, bit 1 stores), ABAB on one box
for i in 1 2 3; do
  for pol in 0 1 2 3; do
    OIP_RRC_POLICY=$pol timeout -k 10 200 python bench.py --workload rrc --steps 20 --warmup 3 --no-cpu-baseline --full-record gpurun_out/r04_ab.json > /dev/null 2> gpurun_out/r04_ab.err || tail -3 gpurun_out/r04_ab.err
    python - <<PY
import json
d=json.load(open('gpurun_out/r04_ab.json'))
print('policy $pol run $i ms_per_step %.4f' % d['ms_per_step'], d.get('rrc_kernel',{}).get('frac_of_8TBs'))
PY
  done
done
