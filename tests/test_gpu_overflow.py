"""GPU: the f32-overflow edge of cv::phaseCorrelate at BASELINE sizes (Weak 2).

divSpectrums works in f32: a bin whose |P|^2 = |F1 conj(F2)|^2 exceeds FLT_MAX gives mag*mag = inf, p*mag = inf and C = inf/inf = NaN;
one NaN bin makes the whole inverse transform NaN, minMaxLoc finds nothing, the weighted centroid sums NaN and the response is NaN.
The reference then REJECTS the unit: `resp >= threshold` is false (stitcher.h:181, preproc.h:527) -- although
FilterInterBandShiftValues, which tests `rs < threshold`, still counts it towards IBCV_MIN_COUNT (preproc.h:498-503).
On 16000 x 3000 windows of 12-bit data only the DC column overflows in ordinary scenes (and that column takes the double-precision
formula); a horizontal illumination ramp of +-1000 DN puts |F(0,1)| ~ 1.5e10, |P| ~ 2e20, |P|^2 > FLT_MAX in an INTERIOR bin.  What
must hold: the GPU reports the same NaN response -- never a finite one that could pass the threshold -- through
oip_interband_correlate + oip_filter_and_fit and oip_stt_correlate + oip_stt_mean, and the units next to it are untouched."""
import numpy as np
import pytest

import _synth

pytestmark = pytest.mark.gpu


def _cuda(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_interband_unit_with_an_overflowing_interior_bin_is_rejected_like_the_reference(ctx, oracle_mod):
    import opticalimageprocessor_amd as oip
    from oracle import phasecorr as pc
    Lp, W, slices = 16000, 24000, 8                                  # 3000-column units: the spectral route of the BASELINE shape
    pan, bands = _synth.pan_mss(Lp, W, [(2, -1), (1, 1), (-1, -2), (-2, 1)], seed=31)
    bc, sc, u = W // slices, W // slices // 4, 2
    ramp = np.linspace(-1000.0, 1000.0, bc)
    pan = pan.copy()
    pan[:, u * bc:(u + 1) * bc] = np.clip(np.rint(pan[:, u * bc:(u + 1) * bc] + ramp), 0, 65535).astype(np.uint16)
    small = ramp.reshape(sc, 4).mean(axis=1)
    bands = [b.copy() for b in bands]
    for b in bands:
        b[:, u * sc:(u + 1) * sc] = np.clip(np.rint(b[:, u * sc:(u + 1) * sc] + small), 0, 65535).astype(np.uint16)
    planes = _cuda(np.stack(bands, 0))
    got = ctx.interband_correlate(_cuda(pan), Lp, 0, Lp, planes, bands[0].size, 0, Lp // 4, W, slices, 1, 16000)
    # the oracle on the ramp unit: NaN response for every band (its cross-power spectrum has NaN bins)
    a = oracle_mod.window_u16_to_f32(pan, 0, u * bc, Lp, bc)
    for b in (0, 3):
        s = oracle_mod.window_u16_to_f32(bands[b], 0, u * sc, Lp // 4, sc)
        (wdx, wdy), wr = pc.phase_correlate(a, oracle_mod.resize_cubic(s, bc, Lp))
        assert np.isnan(wr), (b, wr)
    assert np.isnan(got[:, u, 2]).all(), got[:, u]                   # same outcome on the GPU: not a finite response
    others = [i for i in range(slices) if i != u]
    assert np.isfinite(got[:, others, :3]).all() and (got[:, others, 2] >= 0.4).all(), got[..., 2]
    # the reference's validity decision: the NaN unit is NOT fitted (preproc.h:527) ...
    cx, cy = oip.filter_and_fit(got, 0.4, 5)
    clean = got[:, others]
    cx2, cy2 = oip.filter_and_fit(clean, 0.4, 5)
    assert np.array_equal(cx, cx2) and np.array_equal(cy, cy2)
    wcx, wcy = pc.filter_and_fit(got, 0.4, 5)
    assert np.array_equal(cx, wcx) and np.array_equal(cy, wcy)      # bit-equal to the restatement of NumCpp's fit
    # ... but it IS counted by FilterInterBandShiftValues' `rs < threshold` test (preproc.h:498-503): 8 "valid", not 7
    oip.filter_and_fit(got, 0.4, 8)
    pc.filter_and_fit(got, 0.4, 8)
    with pytest.raises(RuntimeError, match="8 valid values found, 9 expected"):
        oip.filter_and_fit(got, 0.4, 9)
    with pytest.raises(RuntimeError, match="8 valid values found, 9 expected"):
        pc.filter_and_fit(got, 0.4, 9)


def test_ccd_section_with_an_overflowing_interior_bin_is_rejected_like_the_reference(ctx, oracle_mod):
    """16000 x 200 CCD windows: 12-bit data cannot overflow there; a 16-bit-range ramp (+-30000 DN) does.  Section 0 carries the
    ramp, section 1 does not: CalcSttParameters must average section 1 alone (stitcher.h:181-198), and fail when only the ramp
    section exists."""
    import opticalimageprocessor_amd as oip
    from oracle import phasecorr as pc
    L, W, ov, lines = 32100, 512, 200, 16000
    pan1, pan2 = _synth.ccd_pair(L, W, ov, (3, -2), seed=33)
    gap = (L - 2 * lines) // 3
    ramp = np.linspace(-30000.0, 30000.0, ov) + 30000.0
    r0 = gap
    pan1 = pan1.copy(); pan2 = pan2.copy()
    pan1[r0:r0 + lines, W - ov:] = np.clip(np.rint(pan1[r0:r0 + lines, W - ov:] + ramp), 0, 65535).astype(np.uint16)
    pan2[r0:r0 + lines, :ov] = np.clip(np.rint(pan2[r0:r0 + lines, :ov] + ramp), 0, 65535).astype(np.uint16)
    got = ctx.stt_correlate(_cuda(pan1), _cuda(pan2), W, L, 0, L, 2, lines, ov, 0)
    table, mean = pc.calc_stt_parameters(pan1, pan2, sections=2, lines_per_section=lines, overlap_cols=ov)
    assert np.isnan(table[0][3]) and not table[0][4] and table[1][4]
    assert np.isnan(got[0, 2]) and np.isfinite(got[1]).all(), got
    dx, dy, resp, valid = oip.stt_mean(got, 0.4)
    assert valid == 1 and (dx, dy, resp) == (got[1, 0], got[1, 1], got[1, 2])
    assert abs(dx - mean[0]) < 2e-4 and abs(dy - mean[1]) < 2e-4 and abs(resp - mean[2]) < 1e-4
    # the ramp section alone: "No valid delta value found for stitching parameter calculating"
    one = ctx.stt_correlate(_cuda(pan1[r0:r0 + lines]), _cuda(pan2[r0:r0 + lines]), W, lines, 0, lines, 1, lines, ov, 0)
    assert np.isnan(one[0, 2])
    with pytest.raises(RuntimeError, match="No valid delta value"):
        oip.stt_mean(one, 0.4)
    with pytest.raises(RuntimeError, match="No valid delta value"):
        pc.calc_stt_parameters(pan1[r0:r0 + lines], pan2[r0:r0 + lines], sections=1, lines_per_section=lines, overlap_cols=ov)
