# round 4: corr_rows_v_kernel, the shipped build against the build before the scalar row look-ups (12288-wide strips), ABAB
for i in 1 2 3; do
  for lib in new base; do
    if [ $lib = base ]; then export OIP_LIBRARY=$PWD/profiles/experiments/liboipgpu_base.so; else unset OIP_LIBRARY; fi
    timeout -k 10 300 python bench.py --width 12288 --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-cli --no-configs --full-record gpurun_out/r04_rl.json > /dev/null 2>&1
    python - <<PY
import json
d=json.load(open('gpurun_out/r04_rl.json')); k=d['kernels']
print('$lib $i ms_per_step %.3f' % d['ms_per_step'], {n: round(k[n]['avg_ms'],4) for n in ('corr_rows_v_kernel','hpack_bands_kernel','fft_pass_ct_kernel_F125') if n in k})
PY
  done
done
