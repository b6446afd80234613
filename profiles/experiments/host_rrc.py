"""PCIe-inclusive RRC: oip_rrc_u16_host on a pageable host raster (30000 x 32768), in place like IMO::InplaceRRC."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401  (HIP runtime first)
import opticalimageprocessor_amd as oip
from opticalimageprocessor_amd import synth

W, H = 30000, 32768
ctx = oip.Context(0)
img = np.random.default_rng(0).integers(0, 4096, (H, W), dtype=np.uint16)
kb = synth.lut(W)
best = 1e9
for rep in range(3):
    t = time.time()
    ctx.rrc_u16_host(img, kb)
    best = min(best, time.time() - t)
print(json.dumps({"what": "oip_rrc_u16_host, pageable host buffer in place, H2D + kernel + D2H overlapped",
                  "width": W, "lines": H, "seconds": best, "Mpix_s": W * H / best / 1e6,
                  "GB_s_each_way": W * H * 2 / best / 1e9}))
