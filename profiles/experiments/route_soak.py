import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import opticalimageprocessor_amd as oip
import _synth
ctx = oip.Context(0)
s = torch.cuda.Stream(); torch.cuda.set_stream(s); ctx.set_stream(s.cuda_stream)
worst = 0
for seed in range(int(os.environ.get("SOAK_SEEDS", "4"))):
    L, W = 16000, 24000
    rng = np.random.default_rng(seed)
    shifts = [(int(rng.integers(-6, 7)), int(rng.integers(-6, 7))) for _ in range(4)]
    pan, bands = _synth.pan_mss(L, W, shifts, seed=seed + 40)
    dpan = torch.from_numpy(pan).cuda(); planes = torch.from_numpy(np.stack(bands, 0)).cuda()
    res = {}
    for mode in ("0", "2"):
        os.environ["OIP_SPECTRAL_UP"] = mode
        res[mode] = np.array(ctx.interband_correlate(dpan, L, 0, L, planes, bands[0].size, 0, L // 4, W, 8, 1, 16000))
    d = np.abs(res["2"] - res["0"])
    worst = max(worst, d[..., :3].max())
    print(seed, shifts, "max diff", d[..., :2].max(), d[..., 2].max(), "resp range", res["2"][..., 2].min().round(3), res["2"][..., 2].max().round(3),
          "band0 unit0", np.round(res["2"][0, 0, :3], 4), flush=True)
print("worst", worst)
