"""Do two correlation batches on two streams overlap?  Wall time of 2 concurrent contexts vs one after the other."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import opticalimageprocessor_amd as oip

W, L = 30000, 48000
g = torch.Generator(device="cuda").manual_seed(1)
pan = torch.randint(0, 4096, (L, W), dtype=torch.int16, device="cuda", generator=g)
planes = torch.randint(0, 4096, (4, L // 4, W // 4), dtype=torch.int16, device="cuda", generator=g)
ctxs = [oip.Context(0), oip.Context(0)]
torch.cuda.synchronize()

def run(c):
    c.interband_correlate(pan, L, 0, L, planes, (L // 4) * (W // 4), 0, L // 4, W, 10, 2, 16000)
    c.sync()

for c in ctxs:
    run(c)            # warm-up, workspaces
for rep in range(2):
    t = time.time(); run(ctxs[0]); run(ctxs[1]); seq = time.time() - t
    t = time.time()
    th = [threading.Thread(target=run, args=(c,)) for c in ctxs]
    [x.start() for x in th]; [x.join() for x in th]
    par = time.time() - t
    print("sequential %.1f ms   concurrent %.1f ms" % (seq * 1e3, par * 1e3), flush=True)
