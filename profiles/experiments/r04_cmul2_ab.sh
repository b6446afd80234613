# round 4, row stage: paired complex products (cmul2) against the library built with -DOIP_FFT_NO_CMUL2, same box, ABAB
for i in 1 2; do
  for lib in new base; do
    if [ $lib = base ]; then export OIP_LIBRARY=$PWD/profiles/experiments/liboipgpu_no_cmul2.so; else unset OIP_LIBRARY; fi
    timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-cli --no-configs --full-record gpurun_out/r04_cmul2_${lib}_$i.json > /dev/null 2>&1
    python - <<PY
import json
d=json.load(open('gpurun_out/r04_cmul2_${lib}_$i.json'))
k=d['kernels']
print('$lib $i ms_per_step %.3f' % d['ms_per_step'], {n: round(k[n]['avg_ms'],4) for n in ('corr_rows_up_kernel','fft_pass_ct_kernel_F125','fft_pass_ct_kernel_F128_pack','fft_pass_ct_kernel_F128_peak','mss_split_rrc_kernel','align_mss_kernel','rrc_u16_flat_kernel') if n in k})
PY
  done
done
unset OIP_LIBRARY
timeout -k 10 600 python -m pytest tests/test_gpu_correlation.py -q -m gpu -x 2>&1 | tail -2
