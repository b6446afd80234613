"""round 4: device LZW strip encoder on a product-sized image (7500 x 25000 x 4, sensor-like 12-bit data): wall time of the call
and the library's own kernel times"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import opticalimageprocessor_amd as oip
c = oip.Context(0)
W, H, S = 7500, int(sys.argv[1]) if len(sys.argv) > 1 else 25000, 4
g = torch.Generator(device="cuda"); g.manual_seed(1)
img = (torch.randn(H, W * S, device="cuda", generator=g) * 300 + 1800).clamp(64, 4095).to(torch.int16)
cap = c.tiff_lzw_worst_bytes(H, W, S, 1)
pay = torch.empty(cap, dtype=torch.uint8, device="cuda")
scratch = torch.empty(c.tiff_lzw_scratch_bytes(H, W, S, 1), dtype=torch.uint8, device="cuda")     # prepared once, as the pipelined CLI does
torch.cuda.synchronize()
for rep in range(3):
    if hasattr(c, "profile_reset"): c.profile_reset()
    t0 = time.time()
    off, ln, total = c.tiff_lzw_strips(img, H, W, S, 1, pay, scratch)
    dt = time.time() - t0
    print("rows %d: %.1f ms for %.2f GB -> %.2f GB (%.3f)" % (H, dt * 1e3, H * W * S * 2 / 1e9, total / 1e9, total / (H * W * S * 2)))
back = torch.zeros_like(img)
for rep in range(3):
    t0 = time.time()
    c.tiff_lzw_decode(pay, off, ln, H, W, S, 1, 2, back)
    dt = time.time() - t0
    print("decode rows %d: %.1f ms" % (H, dt * 1e3))
print("round trip equal:", bool(torch.equal(back, img)))
