"""oracle.phasecorr -- numpy restatement of cv::phaseCorrelate and of the reference's
correlation drivers.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED: cv::phaseCorrelate / cv::resize live in OpenCV (un-vendored, unpinned,
CMakeLists.txt:8) and Poly1d::fit in NumCpp (CMakeLists.txt:9); neither library is in the
reference tree nor in this image and the reference has no tests.  What follows restates
OpenCV 4.x modules/imgproc/src/phasecorr.cpp step by step; the FFT itself is numpy's
(float64 pocketfft) rounded to float32 storage, so results agree with any correct float32
implementation to rounding level only -- tests state the tolerance.

Reference call sites: stitcher.h:148-201 (CalcSttParameters), preproc.h:224-347
(CalcInterBandCorrelation), preproc.h:492-550 (filter + polynomial fit).
"""
from __future__ import annotations

import numpy as np

from . import resize_cubic, window_u16_to_f32

FLT_EPSILON = np.float32(1.1920929e-07)
DBL_EPSILON = 2.220446049250313e-16


def optimal_dft_size(n: int) -> int:
    """cv::getOptimalDFTSize: smallest 2^a*3^b*5^c >= n."""
    best = None
    p5 = 1
    while p5 < 2 * n + 1:
        p35 = p5
        while p35 < 2 * n + 1:
            v = p35
            while v < n:
                v *= 2
            if best is None or v < best:
                best = v
            p35 *= 3
        p5 *= 5
    return best


def _cross_power_ccs(F1: np.ndarray, F2: np.ndarray, M: int, N: int) -> np.ndarray:
    """mulSpectrums(conjB) -> magSpectrums -> divSpectrums on the half spectrum
    (ky in [0,M), kx in [0,N/2]), following the CCS-packed special cases:
      * real-only bins (ky in {0,M/2}, kx in {0,N/2}): "mag" is re*re and C = A/(A*A+eps)
      * first/last column: double-precision formula
      * everything else: float formula with denom = mag*mag + eps in f32.
    """
    # |P|^2 overflows f32 for large windows (16000 x 3000: the DC column; a strong horizontal ramp: interior bins too) -- inf and
    # inf/inf = NaN are the reference's own f32 arithmetic there (divSpectrums), not an accident of this restatement
    with np.errstate(over="ignore", invalid="ignore"):
        F1 = F1.astype(np.complex64)
        F2 = F2.astype(np.complex64)
        ar, ai = F1.real, F1.imag
        br, bi = F2.real, F2.imag
        # mulSpectrums, conjB=true (f32 arithmetic)
        pr = (ar * br + ai * bi).astype(np.float32)
        pi = (ai * br - ar * bi).astype(np.float32)
        mag = np.sqrt(pr.astype(np.float64) ** 2 + pi.astype(np.float64) ** 2).astype(np.float32)
        eps = FLT_EPSILON
        # generic bins: float formula (B = (mag, 0))
        denom = (mag * mag + np.float32(0) + eps).astype(np.float32).astype(np.float64)
        cr = ((pr * mag).astype(np.float32).astype(np.float64) / denom).astype(np.float32)
        ci = ((pi * mag).astype(np.float32).astype(np.float64) / denom).astype(np.float32)
        C = (cr + 1j * ci).astype(np.complex64)
        # first (kx=0) and, for even N, last (kx=N/2) column: double formula
        cols = [0] + ([N // 2] if N % 2 == 0 else [])
        for kx in cols:
            m = mag[:, kx].astype(np.float64)
            d = m * m + float(eps)
            C[:, kx] = ((pr[:, kx].astype(np.float64) * m / d).astype(np.float32)
                        + 1j * (pi[:, kx].astype(np.float64) * m / d).astype(np.float32))
            # real-only bins: product is a*b, "magnitude" is its square
            rows = [0] + ([M // 2] if M % 2 == 0 else [])
            for ky in rows:
                a = np.float32(ar[ky, kx] * br[ky, kx])
                C[ky, kx] = np.float32(a / np.float32(np.float32(a * a) + eps))
            # the mirrored half of these columns is implied by Hermitian symmetry in CCS
            for ky in range(M // 2 + 1, M):
                C[ky, kx] = np.conj(C[M - ky, kx])
    return C


def phase_correlate(a: np.ndarray, b: np.ndarray, fft: str = "f64"):
    """cv::phaseCorrelate(src1, src2, noArray(), &response) -> ((dx, dy), response).

    fft="f64": the transforms run in float64 (numpy) and are rounded to float32 storage -- the tightest statement of
    the algorithm.  fft="f32": scipy.fft on float32 arrays, single precision throughout like OpenCV's own dft() -- a
    second, independent float32 implementation next to the GPU's: the spread between the two float32 results and the
    float64 one is what "any correct float32 FFT" costs, and the tests report GPU-vs-f32 and f32-vs-f64 side by side."""
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    assert a.shape == b.shape and a.ndim == 2
    rows, cols = a.shape
    M, N = optimal_dft_size(rows), optimal_dft_size(cols)
    if (M, N) != (rows, cols):
        pa = np.zeros((M, N), np.float32); pa[:rows, :cols] = a
        pb = np.zeros((M, N), np.float32); pb[:rows, :cols] = b
    else:
        pa, pb = a, b
    if fft == "f32":
        import scipy.fft as sfft
        F1 = sfft.rfft2(np.ascontiguousarray(pa, np.float32))
        F2 = sfft.rfft2(np.ascontiguousarray(pb, np.float32))
        assert F1.dtype == np.complex64
        C = _cross_power_ccs(F1, F2, M, N)
        c = sfft.irfft2(C.astype(np.complex64), s=(M, N))
        assert c.dtype == np.float32
        c = (c * np.float32(M * N)).astype(np.float32)
    else:
        F1 = np.fft.rfft2(pa.astype(np.float64))
        F2 = np.fft.rfft2(pb.astype(np.float64))
        C = _cross_power_ccs(F1, F2, M, N)
        c = (np.fft.irfft2(C.astype(np.complex128), s=(M, N)) * (M * N)).astype(np.float32)
    c = np.roll(c, (M >> 1, N >> 1), axis=(0, 1))            # fftShift
    peak = int(np.argmax(c))                                 # minMaxLoc: first maximum
    py, px = divmod(peak, N)
    # weightedCentroid(C, peak, Size(5,5))
    minr, maxr = max(py - 2, 0), min(py + 2, M - 1)
    minc, maxc = max(px - 2, 0), min(px + 2, N - 1)
    sx = sy = si = 0.0
    for y in range(minr, maxr + 1):
        for x in range(minc, maxc + 1):
            v = float(c[y, x])
            sx += x * v
            sy += y * v
            si += v
    response = si
    si += DBL_EPSILON
    cx, cy = sx / si, sy / si
    response /= (M * N)
    return (N / 2.0 - cx, M / 2.0 - cy), response


# --------------------------------------------------------------------------------------
def calc_stt_parameters(pan1: np.ndarray, pan2: np.ndarray, sections=10, lines_per_section=16000,
                        overlap_cols=200, threshold=0.4, max_delta_y=0.0, edge_cols=0):
    """Stitcher::CalcSttParameters, stitcher.h:148-201.  Returns (rows, (dx, dy, resp)) where
    rows is the per-section table [(line_offset, dx, dy, resp, valid)]."""
    L, W = pan1.shape
    if L < sections * lines_per_section:
        raise ValueError("PAN line count less than sections times line-per-section, "
                         "use smaller -s and/or -l value(s)")
    gap = (L - sections * lines_per_section) // (sections + 1)
    step = gap + lines_per_section
    sdx = sdy = sr = 0.0
    valid = 0
    table = []
    for i in range(sections):
        off = gap + i * step
        s1 = window_u16_to_f32(pan1, off, W - overlap_cols, lines_per_section, overlap_cols - edge_cols)
        s2 = window_u16_to_f32(pan2, off, edge_cols, lines_per_section, overlap_cols - edge_cols)
        (dx, dy), resp = phase_correlate(s1, s2)
        ok = resp >= threshold and (max_delta_y <= 0.0 or abs(dy) <= max_delta_y)
        if ok:
            sdx += dx; sdy += dy; sr += resp; valid += 1
        table.append((off, dx, dy, resp, ok))
    if valid == 0:
        raise RuntimeError("No valid delta value found for stitching parameter calculating")
    return table, (sdx / valid, sdy / valid, sr / valid)


def calc_interband_correlation(pan: np.ndarray, bands, slices=10, sections=5, corr_lines=16000, fft="f64"):
    """PreProcessor::CalcInterBandCorrelation, preproc.h:224-329.  Returns shifts[b][sec*slices+i]
    = (dx, dy, rs, cx)."""
    Lp, W = pan.shape
    if slices < 8:
        raise ValueError("CalcInterBandCorrelation: at lease 8 slice needed")
    if sections <= 0:
        raise ValueError("CalcInterBandCorrelation: section count should be a positive integer")
    if sections > 1 and sections * corr_lines > Lp:
        raise ValueError("CalcInterBandCorrelation: too many sections")
    base_rows = min(Lp, corr_lines)
    base_gap = (Lp - base_rows * sections) // (sections + 1)
    base_cols = W // slices
    band_rows, band_gap, band_cols = base_rows // 4, base_gap // 4, base_cols // 4
    out = np.zeros((4, slices * sections, 4), np.float64)
    for sec in range(sections):
        for i in range(slices):
            r0 = base_gap + sec * (base_rows + base_gap)
            base = window_u16_to_f32(pan, r0, i * base_cols, base_rows, base_cols)
            for b in range(4):
                br0 = band_gap + sec * (band_rows + band_gap)
                bs = window_u16_to_f32(bands[b], br0, i * band_cols, band_rows, band_cols)
                up = resize_cubic(bs, base_cols, base_rows)
                (dx, dy), rs = phase_correlate(base, up, fft)
                out[b, sec * slices + i] = (dx, dy, rs, i * base_cols + base_cols // 2)
    return out


def filter_and_fit(shifts: np.ndarray, threshold=0.4, min_count=5, method="numcpp"):
    """FilterInterBandShiftValues + DoCorrelationPolynomialFitting, preproc.h:492-550.
    Returns (cx[4][2], cy[4][3]) ascending coefficients.

    method="numcpp" (the product's default, OIP_FIT_REFERENCE): NumCpp Poly1d::fit as the
    reference calls it (preproc.h:535-536), restated operation by operation -- inv(A^T A) A^T y
    on the raw Vandermonde matrix with NumCpp's own Gauss-Jordan inv().  At W=12288..30000
    cond(A^T A) ~ 1e16..1e18, so the low digits of the result belong to that operation order.
    method="lstsq": the exact least-squares solution (SVD on a centred/scaled abscissa,
    coefficients mapped back) -- the product's `--fit lstsq`.
    method="normal": the same normal equations through numpy.linalg.inv (LAPACK) -- a third
    opinion showing how far inverses of this matrix differ.
    NumCpp is un-vendored and unpinned (CMakeLists.txt:9): PARITY UNPINNED.
    """
    cxs = np.zeros((4, 2)); cys = np.zeros((4, 3))
    for b in range(4):
        # preproc.h:498-503 counts a unit as valid unless `rs < threshold` -- a NaN response (f32 overflow of the cross-power
        # spectrum, see _cross_power_ccs) is not below the threshold and IS counted; preproc.h:527 takes a unit into the
        # fit when `rs >= threshold` -- the NaN unit is NOT taken.  Two different tests, restated as they are.
        with np.errstate(invalid="ignore"):
            fc = int((~(shifts[b, :, 2] < threshold)).sum())
            ok = shifts[b, :, 2] >= threshold
        if fc < min_count:
            raise RuntimeError("Not enough valid correlation values for band#%d: %d valid values "
                               "found, %d expected at least" % (b + 1, fc, min_count))
        x = shifts[b, ok, 3]; dx = shifts[b, ok, 0]; dy = shifts[b, ok, 1]
        cxs[b] = polyfit(x, dx, 1, method)
        cys[b] = polyfit(x, dy, 2, method)
    return cxs, cys


def _numcpp_inv(G):
    """nc::linalg::inv as recalled: one sweep per diagonal element, no pivoting (the zero-diagonal
    row swap never triggers for A^T A of distinct abscissae).  Plain Python floats: the same IEEE
    double operations in the same order as the product's C++."""
    m = len(G)
    G = [list(map(float, r)) for r in G]
    R = [[0.0] * m for _ in range(m)]
    for k in range(m):
        if G[k][k] == 0.0:
            raise ZeroDivisionError("singular normal equations")
        R[k][k] = -1.0 / G[k][k]
        for i in range(m):
            for j in range(m):
                if i != k and j != k:
                    R[i][j] = G[i][j] + G[k][j] * G[i][k] * R[k][k]
                elif i != k and j == k:
                    R[i][k] = G[i][k] * R[k][k]
                elif i == k and j != k:
                    R[k][j] = G[k][j] * R[k][k]
        G = [r[:] for r in R]
    return [[v * -1.0 for v in r] for r in R]


def polyfit_numcpp(x, y, deg):
    """nc::polynomial::Poly1d<double>::fit (NumCpp, as recalled; call sites preproc.h:535-536)"""
    x = [float(v) for v in x]; y = [float(v) for v in y]
    n, m = len(x), deg + 1
    A = []
    for xi in x:
        row = []
        for j in range(m):
            v = 1.0
            if j > 0:
                v = xi
                for _ in range(1, j):
                    v *= xi
            row.append(v)
        A.append(row)
    G = [[0.0] * m for _ in range(m)]
    for i in range(m):
        for j in range(m):
            acc = 0.0
            for k in range(n):
                acc = acc + A[k][i] * A[k][j]
            G[i][j] = acc
    R = _numcpp_inv(G)
    P = [[0.0] * n for _ in range(m)]
    for i in range(m):
        for c in range(n):
            acc = 0.0
            for k in range(m):
                acc = acc + R[i][k] * A[c][k]
            P[i][c] = acc
    out = []
    for i in range(m):
        acc = 0.0
        for c in range(n):
            acc = acc + P[i][c] * y[c]
        out.append(acc)
    return np.array(out)


def polyfit(x, y, deg, method="numcpp"):
    x = np.asarray(x, np.float64); y = np.asarray(y, np.float64)
    if method == "numcpp":
        return polyfit_numcpp(x, y, deg)
    if method == "normal":
        A = np.vander(x, deg + 1, increasing=True)
        return np.linalg.inv(A.T @ A) @ A.T @ y
    mu, sc = x.mean(), max(np.abs(x - x.mean()).max(), 1e-300)
    t = (x - mu) / sc
    c = np.linalg.lstsq(np.vander(t, deg + 1, increasing=True), y, rcond=None)[0]
    # expand sum c_j ((x-mu)/sc)^j into ascending powers of x
    p = np.zeros(deg + 1)
    for j, cj in enumerate(c):
        # ((x-mu)/sc)^j = sum_i C(j,i) x^i (-mu)^(j-i) / sc^j
        from math import comb
        for i in range(j + 1):
            p[i] += cj * comb(j, i) * (-mu) ** (j - i) / sc ** j
    return p


# ---- timing helpers for bench.py's cpu_baseline leg (each call is one worker's share of a parallel sample) ----------
def bench_unit_band(seed: int, rows: int, cols: int, repeat: int = 1) -> float:
    """seconds for `repeat` (window conversion + x4 cubic up-sampling + phaseCorrelate) of one band of one unit"""
    import time
    rng = np.random.default_rng(seed)
    pan = rng.integers(64, 4096, (rows, cols), dtype=np.uint16)
    band = rng.integers(64, 4096, (rows // 4, cols // 4), dtype=np.uint16)
    t = time.perf_counter()
    for _ in range(repeat):
        a = window_u16_to_f32(pan, 0, 0, rows, cols)
        b = resize_cubic(window_u16_to_f32(band, 0, 0, rows // 4, cols // 4), cols, rows)
        phase_correlate(a, b)
    return time.perf_counter() - t
