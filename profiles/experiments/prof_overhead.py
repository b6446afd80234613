"""Wall time of one interband correlation batch (50 units of 4 bands) with and without the per-kernel HIP events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import opticalimageprocessor_amd as oip

W, L = 30000, 100000
ctx = oip.Context(0)
s = torch.cuda.Stream()
torch.cuda.set_stream(s)
ctx.set_stream(s.cuda_stream)
g = torch.Generator(device="cuda").manual_seed(1)
pan = torch.randint(0, 4096, (L, W), dtype=torch.int16, device="cuda", generator=g)
planes = torch.randint(0, 4096, (4, L // 4, W // 4), dtype=torch.int16, device="cuda", generator=g)
for on in (True, False, True, False):
    ctx.profile_enable(on)
    ctx.profile_reset()
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize()
        t = time.time()
        ctx.interband_correlate(pan, L, 0, L, planes, (L // 4) * (W // 4), 0, L // 4, W, 10, 5, 16000)
        torch.cuda.synchronize()
        best = min(best, time.time() - t)
    print("profiling %s: %.2f ms" % (on, best * 1e3), flush=True)
