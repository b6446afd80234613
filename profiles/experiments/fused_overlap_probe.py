"""fused_overlap_probe.py -- round-3 experiment: do the two kernels of the fused config-5 step overlap on two streams?

The fused prestitch -> stitch step is  rrc_u16_window (CCD 1 -> left half, HBM-bound, ~5.3 TB/s)  followed by
remap_shift8_lds_kernel<RRC> (CCD 2 raw -> right half, vector-issue bound, ~2.9 TB/s).  Two contexts on two streams
run them side by side; reported: one after the other on one stream, side by side, and each alone.

Run on the box:  python profiles/experiments/fused_overlap_probe.py
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import opticalimageprocessor_amd as oip  # noqa: E402

W, L, fold = 30000, 100000, 200
dx, dy = 2.37, -1.62
rng = np.random.default_rng(5)
a = oip.Context(0)
b = oip.Context(0)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
a.set_stream(sa)
b.set_stream(sb)
pan1 = torch.randint(0, 4096, (L, W), dtype=torch.int16, device="cuda").view(torch.uint16)
pan2 = torch.randint(0, 4096, (L, W), dtype=torch.int16, device="cuda").view(torch.uint16)
kb = np.stack([1.0 + rng.integers(-3, 4, W) / 64.0, rng.integers(-8, 9, W) / 4.0], 1)
kb1, kb2 = a.upload_kb(kb), a.upload_kb(kb[::-1].copy())
out = torch.zeros(L, 2 * (W - fold), dtype=torch.uint16, device="cuda")
P = 2 * (W - fold)


def left(c):
    c.rrc_u16_window(pan1, W, out, P, W - fold, L, kb1)


def right(c):
    c.remap_shift_rrc_bicubic_u16_window(pan2, kb2, out, P, fold, W - fold, W, L, dx, dy)


def timed(fn, n=8):
    fn()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3


def serial():
    left(a)
    right(a)


def side_by_side():
    left(b)
    right(a)


def side_by_side_rev():
    right(a)
    left(b)


print("left alone      %.3f ms" % timed(lambda: left(a)))
print("right alone     %.3f ms" % timed(lambda: right(a)))
print("one stream      %.3f ms" % timed(serial))
print("two streams     %.3f ms (left enqueued first)" % timed(side_by_side))
print("two streams     %.3f ms (right enqueued first)" % timed(side_by_side_rev))
ref = out.clone()
serial()
torch.cuda.synchronize()
print("bits equal after the two-stream runs:", bool(torch.equal(ref.view(torch.int16), out.view(torch.int16))))
