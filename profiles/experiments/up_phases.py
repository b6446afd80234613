"""Phase breakdown of corr_rows_up_kernel by skipping phases (OIP_ROWS_DBG mask; results are wrong, times are the point).
bits: 1 fwd stages, 2 cross-power, 4 inverse stages, 8 global loads after the first pair, 16 stores, 32 vertical expansion"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import opticalimageprocessor_amd as oip
W, L = 24000, 16000
ctx = oip.Context(0)
s = torch.cuda.Stream(); torch.cuda.set_stream(s); ctx.set_stream(s.cuda_stream)
g = torch.Generator(device="cuda").manual_seed(1)
pan = torch.randint(0, 4096, (L, W), dtype=torch.int16, device="cuda", generator=g)
planes = torch.randint(0, 4096, (4, L // 4, W // 4), dtype=torch.int16, device="cuda", generator=g)
masks = [0, 1, 2, 4, 8, 16, 1 | 4, 1 | 2 | 4, 1 | 2 | 4 | 8, 1 | 2 | 4 | 16, 31]
if len(sys.argv) > 1:
    masks = [int(a) for a in sys.argv[1:]]
for thr in os.environ.get("PROBE_THREADS", "768").split(","):
    os.environ["OIP_UP_THREADS"] = thr
    for m in masks:
        os.environ["OIP_ROWS_DBG"] = str(m)
        for rep in range(2):
            ctx.profile_enable(True); ctx.profile_reset()
            ctx.interband_correlate(pan, L, 0, L, planes, (L // 4) * (W // 4), 0, L // 4, W, 8, 1, 16000)
            torch.cuda.synchronize()
        pr = ctx.profile()
        ms, cnt = pr["corr_rows_up_kernel"]
        print("threads %s mask %2d: corr_rows_up_kernel %.4f ms x%d" % (thr, m, ms / cnt, cnt), flush=True)
