"""python tests/soak_bicubic.py [N] -- round 4: soak of the re-written 8-pixel bicubic kernels (packed-pair taps, unclamped loads, quads of lines) on random geometries
larger than the unit tests use: plain remap and MSS alignment against the CPU oracle, the fused RRC-on-load window call (f32 and
fp16 accumulate) against RRC + plain window call.  Prints one line per case and a summary; exits 1 on any difference."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opticalimageprocessor_amd as oip
import oracle as orc                      # the CPU checker: this file lives under tests/ for that reason (not collected by pytest)

orc.lib()
c = oip.Context(0)
bad = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cu = lambda a: torch.from_numpy(a.view(np.int16)).cuda().view(torch.uint16) if a.dtype == np.uint16 else torch.from_numpy(a).cuda()

def scene(rng, L, W):
    x = np.arange(W)[None, :]; y = np.arange(L)[:, None]
    s = 1800 + 900 * np.sin(x / 37.0 + y / 91.0) + 500 * np.cos(x / 11.0 - y / 23.0) + rng.normal(0, 40, (L, W))
    return np.clip(s, 0, 4095).astype(np.uint16)

for i in range(N):
    rng = np.random.default_rng(7000 + i)
    # ---- plain remap vs oracle
    W = int(rng.choice([2040, 2048, 4096, 6000]))
    sr = int(rng.integers(300, 2500)); guard = sr + int(rng.integers(0, 40)); L = guard + 1 + int(rng.integers(sr, 3 * sr))
    dx = float(rng.uniform(-40, 40)); dy = float(rng.uniform(-25, 25))
    src = scene(rng, L, W)
    want, _ = orc.prestitch(src, dx, dy, sr, guard)
    dst = torch.zeros(L, W, dtype=torch.uint16, device="cuda")
    c.remap_shift_bicubic_u16(cu(src), dst, W, L, dx, dy, sr, guard); c.sync()
    ok1 = np.array_equal(dst.cpu().numpy(), want)
    # ---- fused window (RRC on load) vs RRC + plain window, both accumulate modes
    fold = int(rng.choice([0, 50, 100, 104]))
    kb = np.stack([1.0 + rng.integers(-3, 4, W) / 64.0, rng.integers(-8, 9, W) / 4.0], 1)
    d_kb = c.upload_kb(kb)
    raw = cu(src)
    corrected = torch.empty_like(raw); c.rrc_u16(raw, corrected, W, L, d_kb)
    ok2 = True
    if W % 8 == 0:
        for f16 in (False, True):
            a = torch.zeros(L, 2 * (W - fold), dtype=torch.uint16, device="cuda"); b = torch.zeros_like(a)
            c.remap_shift_bicubic_u16_window(corrected, a, 2 * (W - fold), fold, W - fold, W, L, dx, dy, sr, guard, f16acc=f16)
            c.remap_shift_rrc_bicubic_u16_window(raw, d_kb, b, 2 * (W - fold), fold, W - fold, W, L, dx, dy, sr, guard, f16acc=f16)
            c.sync()
            ok2 = ok2 and bool(torch.equal(a.view(torch.int16), b.view(torch.int16)))
    # ---- alignment vs oracle
    Wb = int(rng.choice([512, 750, 1024, 1876])); lps = int(rng.integers(400, 1500)); ovl = int(rng.integers(0, min(lps // 2, 200) + 1))
    minl = int(rng.integers(ovl + 1, lps + 1)); off = int(rng.integers(0, 30)); Lm = off + minl + int(rng.integers(lps, 3 * lps))
    bands = [scene(rng, Lm, Wb) for _ in range(4)]
    W4 = 4 * Wb
    cx = np.stack([rng.uniform(-8, 8, 4), rng.uniform(-3e-3, 3e-3, 4)], 1)
    cy = np.stack([rng.uniform(-12, 12, 4), rng.uniform(-6, 6, 4) / W4, rng.uniform(-10, 10, 4) / (W4 * W4)], 1)
    wantA, nvalid = orc.align_mss(bands, cx, cy, lps, off, ovl, bool(i % 2), minl)
    dstA = torch.full(wantA.shape, 9, dtype=torch.uint16, device="cuda")
    gv = c.align_mss_bicubic_u16x4(cu(np.stack(bands, 0)), Wb * Lm, dstA, Wb, Lm, cx, cy, lps, off, ovl, bool(i % 2), minl); c.sync()
    ok3 = gv == nvalid and np.array_equal(dstA.cpu().numpy(), wantA)
    print("case %2d: remap W=%d L=%d sr=%d dx=%.3f dy=%.3f %s | fused fold=%d %s | align Wb=%d Lm=%d lps=%d %s"
          % (i, W, L, sr, dx, dy, "ok" if ok1 else "DIFF", fold, "ok" if ok2 else "DIFF", Wb, Lm, lps, "ok" if ok3 else "DIFF"), flush=True)
    bad += (not ok1) + (not ok2) + (not ok3)
print("%d cases, %d differences" % (N, bad))
sys.exit(1 if bad else 0)
