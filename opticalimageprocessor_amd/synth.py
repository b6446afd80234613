"""Deterministic synthetic pushbroom strips, generated on the GPU with torch (bench/test data
only -- nothing here is on the product path).

SURVEY 8d: 12-bit textured scene (band-limited Gaussian noise, sigma ~1.5 px, so phase
correlation has signal) on a slowly varying pedestal, per-column fixed-pattern response that
the RRC LUT removes (k ~ U(0.9,1.1), b ~ U(-8,8), rounded like a CSV), MSS bands cut from the
same scene at known PAN-pixel offsets (4x4 box mean), second CCD displaced by a known shift.
Seed 0x0A11CE (+ config index / rank); a line's content depends only on its GLOBAL line index
so that row-block shards of a strip are consistent across ranks.
"""
from __future__ import annotations

import math

import numpy as np
import torch

SEED = 0x0A11CE
BAND_SHIFTS = ((2, -1), (1, 1), (-1, -2), (-2, 1))   # (sx, sy) of MSS band b in PAN pixels
# Band-limited texture.  One octave at sigma 1.5 px: un-windowed phase correlation (the reference
# passes noArray() as window) degrades quickly when coarser octaves make the image borders
# dominate the spectrum, and real inter-band offsets are a few PAN pixels at most.
OCTAVES = ((1.5, 1.0),)
CCD_SHIFT = (3, -2)                                  # (sx, sy) of CCD-2's overlap vs CCD-1


def lut(w: int, seed: int = 0) -> np.ndarray:
    rng = np.random.default_rng(SEED + 1000 + seed)
    return np.stack([np.round(rng.uniform(0.9, 1.1, w), 6), np.round(rng.uniform(-8, 8, w), 4)], 1)


def _noise_rows(row0: int, rows: int, cols: int, seed: int, device) -> torch.Tensor:
    """white N(0,1) field for global lines [row0, row0+rows): generated in fixed blocks of 1024
    lines keyed by the global block index, so any window of the strip sees the same values."""
    blk = 1024
    out = torch.empty(rows, cols, dtype=torch.float32, device=device)
    g = torch.Generator(device=device)
    b0, b1 = row0 // blk, (row0 + rows - 1) // blk
    for b in range(b0, b1 + 1):
        g.manual_seed((SEED + seed) * 1_000_003 + (b + 4096))
        block = torch.randn(blk, cols, dtype=torch.float32, device=device, generator=g)
        lo, hi = max(row0, b * blk), min(row0 + rows, (b + 1) * blk)
        out[lo - row0:hi - row0] = block[lo - b * blk:hi - b * blk]
    return out


def _gauss_kernel(sigma: float, device):
    r = max(1, int(math.ceil(3 * sigma)))
    x = torch.arange(-r, r + 1, dtype=torch.float32, device=device)
    k = torch.exp(-0.5 * (x / sigma) ** 2)
    return (k / k.sum()), r


def scene_rows(row0: int, rows: int, cols: int, field_cols: int, seed: int = 0, device="cuda",
               amp: float = 420.0, base: float = 1800.0, x0: int = 0) -> torch.Tensor:
    """f32 scene for global lines [row0, row0+rows) and scene columns [x0, x0+cols).
    field_cols is the width of the ground scene; every caller that looks at the same ground
    must pass the same value (the white-noise field is generated field_cols+128 wide).
    Texture = the white field blurred with the Gaussians listed in OCTAVES (sigma, weight)."""
    octaves = OCTAVES
    R = max(_gauss_kernel(sg, device)[1] for sg, _ in octaves)
    assert -64 + R <= x0 and x0 + cols + R <= field_cols + 64 and row0 + 64 - R >= 0
    # the noise field has a fixed margin of 64 lines/columns around the scene
    n = _noise_rows(row0 + 64 - R, rows + 2 * R, field_cols + 128, seed, device)[:, x0 + 64 - R:x0 + 64 + cols + R]
    out = torch.zeros(rows, cols, dtype=torch.float32, device=device)
    step = 4096
    for sigma, weight in octaves:
        k, r = _gauss_kernel(sigma, device)
        kl = [float(v) for v in k.tolist()]
        gain = amp * weight / float((k * k).sum())       # unit variance after the separable blur
        for y in range(0, rows, step):
            m = min(step, rows - y)
            t = n[y + R - r:y + m + R + r, R - r:R + cols + r]
            # separable Gaussian as explicit shifted adds (no MIOpen: its conv search is slow)
            h = kl[0] * t[:, 0:cols]
            for j in range(1, 2 * r + 1):
                h = h + kl[j] * t[:, j:j + cols]
            v = kl[0] * h[0:m]
            for j in range(1, 2 * r + 1):
                v = v + kl[j] * h[j:j + m]
            out[y:y + m] += gain * v
    gy = torch.arange(row0, row0 + rows, dtype=torch.float32, device=device)
    out += (base + 200.0 * torch.sin(gy / 30000.0)).unsqueeze(1)
    return out


def _raw_u16(scene: torch.Tensor, kb: np.ndarray) -> torch.Tensor:
    """sensor counts whose RRC (k*raw + b) restores the scene: raw = (scene - b)/k, 12 bit."""
    dev = scene.device
    k = torch.from_numpy(kb[:, 0].astype(np.float32)).to(dev)
    b = torch.from_numpy(kb[:, 1].astype(np.float32)).to(dev)
    raw = torch.clamp(torch.round((scene - b) / k), 64, 4095)
    return raw.to(torch.int16).view(torch.uint16)


def pan_strip(row0: int, rows: int, W: int, kb: np.ndarray, seed: int = 0, device="cuda", chunk: int = 8192):
    """raw PAN lines [row0, row0+rows) as u16 (rows x W)."""
    out = torch.empty(rows, W, dtype=torch.uint16, device=device)
    for y in range(0, rows, chunk):
        m = min(chunk, rows - y)
        out[y:y + m] = _raw_u16(scene_rows(row0 + y, m, W, W, seed, device), kb)
    return out


def mss_strip(mrow0: int, mrows: int, W: int, kb4: np.ndarray, seed: int = 0, device="cuda", chunk: int = 2048,
              band_shifts=BAND_SHIFTS):
    """raw BIL MSS lines [mrow0, mrow0+mrows) as u16 (mrows x W): 4 bands x W/4 px per line,
    band b = 4x4 box mean of the scene displaced by band_shifts[b] PAN pixels."""
    bw = W // 4
    out = torch.empty(mrows, W, dtype=torch.uint16, device=device)
    pad = 8
    for y in range(0, mrows, chunk):
        m = min(chunk, mrows - y)
        # scene window with a margin for the displacements (columns clamp at the strip edge)
        sc = scene_rows((mrow0 + y) * 4 - pad, m * 4 + 2 * pad, W + 2 * pad, W, seed, device, x0=-pad)
        for b, (sx, sy) in enumerate(band_shifts):
            sub = sc[pad - sy:pad - sy + 4 * m, pad - sx:pad - sx + W]
            box = sub.reshape(m, 4, bw, 4).mean(dim=(1, 3))
            out[y:y + m, b * bw:(b + 1) * bw] = _raw_u16(box, kb4[b * bw:(b + 1) * bw])
    return out


def ccd_pair(row0: int, rows: int, W: int, overlap: int, kb1: np.ndarray, kb2: np.ndarray, seed: int = 0,
             device="cuda", chunk: int = 8192, shift=CCD_SHIFT):
    """two CCD segments whose `overlap` columns see the same ground displaced by `shift`."""
    sx, sy = shift
    pad = 16
    p1 = torch.empty(rows, W, dtype=torch.uint16, device=device)
    p2 = torch.empty(rows, W, dtype=torch.uint16, device=device)
    for y in range(0, rows, chunk):
        m = min(chunk, rows - y)
        sc = scene_rows(row0 + y - pad, m + 2 * pad, 2 * W - overlap + 2 * pad, 2 * W - overlap, seed, device, x0=-pad)
        p1[y:y + m] = _raw_u16(sc[pad:pad + m, pad:pad + W], kb1)
        x2 = pad + W - overlap - sx
        p2[y:y + m] = _raw_u16(sc[pad - sy:pad - sy + m, x2:x2 + W], kb2)
    return p1, p2
