"""GPU parity, randomised: seeded sweeps over shapes / shifts / section geometry for the kernels
that must match the oracle bit for bit.  Each case is small enough for the CPU oracle to finish
in well under a second; the seeds are fixed so failures reproduce."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cuda(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _scene(rng, L, W):
    img = rng.integers(0, 4096, (L, W)).astype(np.uint16)
    img[rng.integers(0, L, 3)] = 65535
    img[rng.integers(0, L, 3)] = 0
    return img


@pytest.mark.parametrize("seed", range(12))
def test_random_rrc(ctx, oracle_mod, seed):
    import torch
    rng = np.random.default_rng(1000 + seed)
    w = int(rng.choice([8, 16, 24, 40, 100, 1000, 1001, 2048, 7500, 7504, 12288, 30000]))
    h = int(rng.integers(1, 70))
    img = rng.integers(0, 65536, (h, w), dtype=np.uint16)
    kb = np.stack([rng.uniform(-2, 3, w), rng.uniform(-70000, 70000, w)], 1)
    kb[rng.integers(0, w, 4), 0] = [np.nan, 1e12, -1e12, 0.0]
    off = int(rng.choice([0, 8, 24, 1, 3]))                       # element offset of the raster in its allocation
    base = torch.zeros(off + h * w + 8, dtype=torch.uint16, device="cuda")
    view = base[off:off + h * w]
    view.copy_(_cuda(img).reshape(-1))
    ctx.rrc_u16(view, view, w, h, ctx.upload_kb(kb))
    ctx.sync()
    assert np.array_equal(view.cpu().numpy().reshape(h, w), oracle_mod.rrc(img, kb))


@pytest.mark.parametrize("seed", range(16))
def test_random_remap_shift(ctx, oracle_mod, seed):
    import torch
    rng = np.random.default_rng(2000 + seed)
    W = int(rng.choice([8, 24, 33, 64, 96, 200, 257]))
    sr = int(rng.integers(40, 400))
    guard = sr + int(rng.integers(0, 30))
    L = guard + 1 + int(rng.integers(0, 3 * sr))
    dx = float(rng.uniform(-12, 12)) if seed % 4 else float(rng.integers(-5, 6))
    dy = float(rng.uniform(-9, 9)) if seed % 3 else float(rng.integers(-4, 5))
    if seed == 7:
        dx, dy = 2.0 - 2 ** -20, -(1.0 - 2 ** -20)                 # phases that round across a pixel
    src = _scene(rng, L, W)
    want, _ = oracle_mod.prestitch(src, dx, dy, sr, guard)
    dst = torch.zeros(L, W, dtype=torch.uint16, device="cuda")
    ctx.remap_shift_bicubic_u16(_cuda(src), dst, W, L, dx, dy, sr, guard)
    ctx.sync()
    got = dst.cpu().numpy()
    assert np.array_equal(got, want), (W, L, sr, guard, dx, dy, np.argwhere(got != want)[:4])


@pytest.mark.parametrize("seed", range(10))
def test_random_align(ctx, oracle_mod, seed):
    import torch
    rng = np.random.default_rng(3000 + seed)
    Wb = int(rng.choice([16, 40, 75, 128, 333]))
    lps = int(rng.integers(120, 500))
    ovl = int(rng.integers(0, lps // 2 + 1))
    minl = int(rng.integers(ovl + 1, lps + 1))
    off = int(rng.integers(0, 30))
    Lm = off + minl + int(rng.integers(0, 3 * lps))
    keep = bool(seed % 2)
    bands = [_scene(rng, Lm, Wb) for _ in range(4)]
    W = 4 * Wb
    cx = np.stack([rng.uniform(-8, 8, 4), rng.uniform(-3e-3, 3e-3, 4)], 1)
    cy = np.stack([rng.uniform(-12, 12, 4), rng.uniform(-6, 6, 4) / W, rng.uniform(-10, 10, 4) / (W * W)], 1)
    want, nvalid = oracle_mod.align_mss(bands, cx, cy, lps, off, ovl, keep, minl)
    dst = torch.full(want.shape, 9, dtype=torch.uint16, device="cuda")
    got_valid = ctx.align_mss_bicubic_u16x4(_cuda(np.stack(bands, 0)), Wb * Lm, dst, Wb, Lm, cx, cy, lps, off, ovl, keep, minl)
    ctx.sync()
    assert got_valid == nvalid
    got = dst.cpu().numpy()
    assert np.array_equal(got, want), (Wb, Lm, lps, off, ovl, keep, minl, np.argwhere(got != want)[:4])


@pytest.mark.parametrize("seed", range(8))
def test_random_stitch_and_split(ctx, oracle_mod, seed):
    import torch
    rng = np.random.default_rng(4000 + seed)
    W = int(rng.choice([16, 64, 200, 999, 1000, 4096]))
    L = int(rng.integers(1, 40))
    fold = int(rng.integers(0, W // 2))
    left = rng.integers(0, 65536, (L, W), dtype=np.uint16)
    right = rng.integers(0, 65536, (L, W), dtype=np.uint16)
    out = torch.zeros(L, 2 * (W - fold), dtype=torch.uint16, device="cuda")
    ctx.stitch_rows_u16(_cuda(left), _cuda(right), out, W, L, fold)
    ctx.sync()
    assert np.array_equal(out.cpu().numpy(), oracle_mod.stitch_raw(left, right, fold))
    if W % 4 == 0:
        bw = W // 4
        planes = torch.zeros(4, L, bw, dtype=torch.uint16, device="cuda")
        kb = np.stack([rng.uniform(0.5, 1.5, W), rng.uniform(-100, 100, W)], 1)
        ctx.mss_split_rrc_u16(_cuda(left), planes, L * bw, W, L, ctx.upload_kb(kb))
        ctx.sync()
        bands = oracle_mod.split_mss(left)
        for b in range(4):
            assert np.array_equal(planes[b].cpu().numpy(), oracle_mod.rrc(bands[b], kb[b * bw:(b + 1) * bw]))


@pytest.mark.parametrize("seed", range(6))
def test_random_resize(ctx, oracle_mod, seed):
    import torch
    rng = np.random.default_rng(5000 + seed)
    sw, sh = int(rng.integers(4, 120)), int(rng.integers(4, 90))
    if seed % 2:
        dw, dh = 4 * sw, 4 * sh                                    # the x4 kernel
    else:
        dw, dh = int(sw * rng.uniform(1.0, 5.0)) + 1, int(sh * rng.uniform(1.0, 5.0)) + 1
    src = rng.uniform(0, 65535, (sh, sw)).astype(np.float32)
    dst = torch.zeros(dh, dw, dtype=torch.float32, device="cuda")
    ctx.resize_cubic_f32(_cuda(src), sw, sh, dst, dw, dh)
    ctx.sync()
    want = oracle_mod.resize_cubic(src, dw, dh)
    assert np.array_equal(dst.cpu().numpy().view(np.uint32), want.view(np.uint32)), (sw, sh, dw, dh)
