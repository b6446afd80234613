"""One 16000 x 30000 correlation section (10 units = 5 pairs, the BASELINE geometry) on random 12-bit data: wall time of
the section and the library's per-kernel HIP-event averages.  Experiment knobs are environment variables the library
reads once per process, so a sweep runs this file once per setting (see r03_sweep.sh).  `--cols N` changes the strip
width (12288: the reference's native geometry, 1228-column units padded to 1250)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import opticalimageprocessor_amd as oip

W = int(sys.argv[sys.argv.index("--cols") + 1]) if "--cols" in sys.argv else 30000
tag = sys.argv[sys.argv.index("--tag") + 1] if "--tag" in sys.argv else "-"
L = 16000
ctx = oip.Context(0)
s = torch.cuda.Stream(); torch.cuda.set_stream(s); ctx.set_stream(s.cuda_stream)
g = torch.Generator(device="cuda").manual_seed(1)
pan = torch.randint(0, 4096, (L, W), dtype=torch.int16, device="cuda", generator=g)
planes = torch.randint(0, 4096, (4, L // 4, W // 4), dtype=torch.int16, device="cuda", generator=g)

def run():
    return ctx.interband_correlate(pan, L, 0, L, planes, (L // 4) * (W // 4), 0, L // 4, W, 10, 1, 16000)

run(); torch.cuda.synchronize()
best, res = 1e9, None
for rep in range(3):
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    ev0.record(); res = run(); ev1.record(); torch.cuda.synchronize()
    best = min(best, ev0.elapsed_time(ev1))
ctx.profile_enable(True); ctx.profile_reset()
for rep in range(2):
    run()
torch.cuda.synchronize()
pr = ctx.profile()
ctx.profile_enable(False)
kern = {k: (round(ms / max(cnt, 1), 4), cnt // 2) for k, (ms, cnt) in pr.items()}
print(json.dumps({"tag": tag, "W": W, "section_ms": round(best, 3), "checksum": float(np.nansum(np.asarray(res))),
                  "kernels_avg_ms_x_launches": kern}), flush=True)
