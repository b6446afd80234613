"""peak_dbg_probe.py -- round-3 experiment: where does the last inverse column pass (fft_col128_peak_kernel) spend its time?
Runs interband_correlate_units on 4 synthetic 16000 x 3000 units (results ignored) and prints the library's per-kernel times.
OIP_PACK_DBG masks inside the kernel: 16 no global loads, 32 loads + first stage only, 64 no wave reduction / atomics.
Run on the box:  for d in 0 16 32 64 48; do OIP_PACK_DBG=$d python profiles/experiments/peak_dbg_probe.py; done
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import opticalimageprocessor_amd as oip  # noqa: E402

ctx = oip.Context(0)
rows, cols, n = 16000, 3000, 4
g = torch.Generator(device="cuda").manual_seed(3)
pan = [torch.randint(0, 4096, (rows, cols), device="cuda", generator=g, dtype=torch.int32).to(torch.int16).view(torch.uint16) for _ in range(n)]
bands = [[torch.randint(0, 4096, (rows // 4, cols // 4), device="cuda", generator=g, dtype=torch.int32).to(torch.int16).view(torch.uint16)
          for _ in range(4)] for _ in range(n)]


def run():
    try:
        ctx.interband_correlate_units(pan, [cols] * n, bands, [cols // 4] * n, rows, cols)
    except oip.OipError:
        pass


run()
ctx.profile_reset()
ctx.profile_enable(True)
for _ in range(3):
    run()
ctx.sync()
ctx.profile_enable(False)
p = ctx.profile()
keys = [k for k in p if "peak" in k or "again" in k or "F125" == k[-4:] or "rows" in k or "pack" in k]
print("OIP_PACK_DBG=%s  " % os.environ.get("OIP_PACK_DBG", "0") + "  ".join("%s %.4f ms" % (k, p[k][0] / max(p[k][1], 1)) for k in sorted(keys)))
