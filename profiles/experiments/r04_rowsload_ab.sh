# round 4: the per-iteration row look-ups of corr_rows_up_kernel / corr_rows_v_kernel through scalar loads (oip_sload_i32)
# against the previous build (vector loads + s_waitcnt vmcnt(0) in front of the prefetch), same box, ABAB; 30000- and 12288-wide
for i in 1 2; do
  for lib in new base; do
    if [ $lib = base ]; then export OIP_LIBRARY=$PWD/profiles/experiments/liboipgpu_base.so; else unset OIP_LIBRARY; fi
    for w in 30000 12288; do
      timeout -k 10 300 python bench.py --width $w --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-cli --no-configs --full-record gpurun_out/r04_rl.json > /dev/null 2>&1
      python - <<PY
import json
d=json.load(open('gpurun_out/r04_rl.json')); k=d['kernels']
print('$lib $i w=$w ms_per_step %.3f' % d['ms_per_step'], {n: round(k[n]['avg_ms'],4) for n in ('corr_rows_up_kernel','corr_rows_v_kernel','fft_pass_ct_kernel_F128_peak','fft_pass_ct_kernel_F128_pack','fft_pass_ct_kernel_F125') if n in k})
PY
    done
  done
done
unset OIP_LIBRARY
timeout -k 10 900 python -m pytest tests/test_gpu_correlation.py tests/test_gpu_overflow.py -q -m gpu -x 2>&1 | tail -2
