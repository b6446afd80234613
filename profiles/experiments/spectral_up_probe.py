"""Spectral up-sampling route (corr_rows_up_kernel) against the image-domain route (OIP_SPECTRAL_UP=0):
shifts / responses of the same units by both, and the per-kernel times of one 16000 x 30000 section."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import opticalimageprocessor_amd as oip
from opticalimageprocessor_amd import synth

ctx = oip.Context(0)
s = torch.cuda.Stream(); torch.cuda.set_stream(s); ctx.set_stream(s.cuda_stream)

def run(pan, planes, L, W, slices, sections, lines):
    return ctx.interband_correlate(pan, L, 0, L, planes, (L // 4) * (W // 4), 0, L // 4, W, slices, sections, lines)

for (L, W, slices, lines) in ((64, 24000, 8, 64), (400, 24000, 8, 400), (16000, 24000, 8, 16000)):
    scene = synth.default_strip(W, L, seed=7) if hasattr(synth, "default_strip") else None
    if scene is None:
        g = torch.Generator(device="cuda").manual_seed(3)
        base = torch.rand((L // 4 + 8, W // 4 + 8), device="cuda", generator=g)
        import torch.nn.functional as Fn
        up = Fn.interpolate(base[None, None], scale_factor=4, mode="bicubic", align_corners=False)[0, 0]
        pan = (up[16:16 + L, 16:16 + W] * 3000 + 500 + torch.randn((L, W), device="cuda", generator=g) * 10).clamp(64, 4095).to(torch.int16)
        planes = torch.stack([(base[4 + (b & 1):4 + (b & 1) + L // 4, 4:4 + W // 4] * 3000 + 500
                               + torch.randn((L // 4, W // 4), device="cuda", generator=g) * 10).clamp(64, 4095) for b in range(4)]).to(torch.int16).contiguous()
    res = {}
    for mode in ("0", "1", "2"):
        os.environ["OIP_SPECTRAL_UP"] = mode
        res[mode] = np.array(run(pan, planes, L, W, slices, 1, lines))
    if L == 64:       # three explicit units: a pair and a single one
        for mode in ("0", "1", "2"):
            os.environ["OIP_SPECTRAL_UP"] = mode
            pp = [pan[:, 3000 * u:] for u in (0, 1, 2)]
            bp = [[planes[b][:, 750 * u:] for b in range(4)] for u in (0, 1, 2)]
            r3 = ctx.interband_correlate_units(pp, [W] * 3, bp, [W // 4] * 3, 64, 3000)
            print("units route %s:" % mode, np.round(r3[2], 5).tolist(), flush=True)
    if "2" in res:
        d2 = np.abs(res["2"] - res["0"])
        print("L=%d both axes: max |d shift| %.3g px, max |d response| %.3g" % (L, d2[..., :2].max(), d2[..., 2].max()), flush=True)
    d = np.abs(res["1"] - res["0"])
    print("L=%d W=%d: max |d shift| %.3g px, max |d response| %.3g;  first unit old %s new %s" % (
        L, W, d[..., :2].max(), d[..., 2].max(), np.round(res["0"].reshape(-1, res["0"].shape[-1])[0], 5), np.round(res["1"].reshape(-1, res["1"].shape[-1])[0], 5)), flush=True)

L, W = 16000, 30000
g = torch.Generator(device="cuda").manual_seed(1)
pan = torch.randint(0, 4096, (L, W), dtype=torch.int16, device="cuda", generator=g)
planes = torch.randint(0, 4096, (4, L // 4, W // 4), dtype=torch.int16, device="cuda", generator=g)
for mode in ("0", "1", "2"):
    os.environ["OIP_SPECTRAL_UP"] = mode
    for thr in (("512", "768") if mode == "1" else ("-",)):
        os.environ["OIP_UP_THREADS"] = thr
        for rep in range(2):
            ctx.profile_enable(True); ctx.profile_reset()
            ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
            ev0.record(); run(pan, planes, L, W, 10, 1, 16000); ev1.record(); torch.cuda.synchronize()
        pr = ctx.profile()
        print("route %s threads %s: section %.2f ms | " % (mode, thr, ev0.elapsed_time(ev1)) +
              "  ".join("%s %.4f x%d" % (k.replace("fft_pass_ct_kernel_", ""), ms / max(cnt, 1), cnt) for k, (ms, cnt) in pr.items()), flush=True)
