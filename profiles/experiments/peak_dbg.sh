# phase masks of fft_col128_peak_kernel (see peak_dbg_probe.py).  The masks 16/32/64 existed in the kernel only while
# r03_peak_dbg_result.txt was taken; with the product as committed every run prints the full pass.
for d in 0 16 32 64 48 96; do OIP_PACK_DBG=$d timeout -k 10 120 python profiles/experiments/peak_dbg_probe.py || exit 1; done
