"""BASELINE config 3 composed at its own geometry on one MI355X: PAN 30000 x 65536 + MSS 4 x (7500 x 16384), the
reference's default action with --do-rrc4pan and --ibc-sections 4 (5 x 16000 > 65536, preproc.h:234) through
opticalimageprocessor_amd.dist.default_action_step on the HIP backend -- RRC of five planes, 4 x 10 x 4 phase
correlations on 16000 x 3000 units (spectral up-sampling route), filter + polynomial fit, bicubic alignment to 16UC4.

The components have their own parity tests; this one checks the COMPOSITION against the oracle where the oracle can
follow at this size: corrected PAN and band lines bit for bit on sampled line blocks, one whole correlation unit against
the oracle's phaseCorrelate, the fitted polynomials against the oracle's fit of the same table, and the whole aligned
image against oracle.align_mss evaluated on the fitted polynomials (bit for bit)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHIFT_TOL, RESP_TOL = 2e-4, 1e-4          # tests/test_gpu_correlation.py


def test_config3_composed_default_action_at_30000x65536_with_4_sections(ctx, oracle_mod, parity_log):
    import torch
    import opticalimageprocessor_amd as oip
    from oracle import phasecorr as pc
    from opticalimageprocessor_amd import synth
    from opticalimageprocessor_amd.dist import HipBackend, ShardBuffers, StripPlan, default_action_step
    W, Lp, slices, sections = 30000, 65536, 10, 4
    kb = synth.lut(W)
    kb4 = np.concatenate([synth.lut(W // 4, 10 + b) for b in range(4)], 0)
    dev = torch.device("cuda", 0)
    plan = StripPlan(W, Lp, 1, slices, sections)
    assert plan.base_rows == 16000 and plan.base_cols == 3000 and plan.n_units == 40
    bufs = ShardBuffers(plan, 0, dev)
    raw_pan = synth.pan_strip(0, Lp, W, kb, device=dev)
    raw_mss = synth.mss_strip(0, plan.mb, W, kb4, device=dev)
    out = torch.zeros(plan.out_rows, W // 4, 4, dtype=torch.uint16, device=dev)
    cx, cy, (o0, o1) = default_action_step(HipBackend(ctx, plan), plan, bufs, raw_pan, raw_mss, ctx.upload_kb(kb), ctx.upload_kb(kb4),
                                           out, 0, threshold=0.4)
    ctx.sync()
    assert (o0, o1) == (0, plan.out_rows) and np.isfinite(cx).all() and np.isfinite(cy).all()
    print("\nfitted cx", cx.tolist(), "cy", cy.tolist())

    # ---- RRC of PAN and of the four bands: sampled line blocks, bit for bit
    for r0 in (0, 30000, Lp - 1024):
        want = oracle_mod.rrc(raw_pan[r0:r0 + 1024].cpu().numpy(), kb, threads=8)
        assert np.array_equal(bufs.pan[r0:r0 + 1024].cpu().numpy(), want), r0
    bil = raw_mss.cpu().numpy()
    bands = [oracle_mod.rrc(bnd, kb4[b * (W // 4):(b + 1) * (W // 4)], threads=8) for b, bnd in enumerate(oracle_mod.split_mss(bil))]
    del bil
    planes = bufs.planes[:, :plan.Lm].cpu().numpy()
    for b in range(4):
        assert np.array_equal(planes[b], bands[b]), b

    # ---- one whole unit (section 2, slice 7) against the oracle's phaseCorrelate on the corrected rasters
    sec, sl = 2, 7
    p0, p1, m0, m1 = plan.section(sec)
    pan_win = bufs.pan[p0:p1, sl * 3000:(sl + 1) * 3000].cpu().numpy()
    units = [bufs.unit_windows(u) for u in (sec * slices + sl - 1, sec * slices + sl)]        # its pair (6, 7), as the step computes it
    got = ctx.interband_correlate_units([u[0].data_ptr() for u in units], [u[0].stride(0) for u in units],
                                        [[x.data_ptr() for x in u[1]] for u in units], [u[1][0].stride(0) for u in units], 16000, 3000)[1]
    a = oracle_mod.window_u16_to_f32(pan_win, 0, 0, 16000, 3000)
    worst_s = worst_r = 0.0
    for b in range(4):
        small = oracle_mod.window_u16_to_f32(bands[b], m0, sl * 750, 4000, 750)
        (wdx, wdy), wr = pc.phase_correlate(a, oracle_mod.resize_cubic(small, 3000, 16000))
        gdx, gdy, gr = got[b]
        worst_s, worst_r = max(worst_s, abs(gdx - wdx), abs(gdy - wdy)), max(worst_r, abs(gr - wr))
        assert wr >= 0.4, "the synthetic scene clears the reference's threshold"
        assert abs(gdx - wdx) < SHIFT_TOL and abs(gdy - wdy) < SHIFT_TOL and abs(gr - wr) < RESP_TOL, (b, got[b], (wdx, wdy, wr))
    parity_log(shift_px=worst_s, response=worst_r, units=4, masked_out=0, shift_bar=SHIFT_TOL, response_bar=RESP_TOL)

    # ---- the fit: the oracle's restatement of filter + Poly1d::fit on the table the step used gives the same polynomials
    shifts = ctx.interband_correlate(bufs.pan, Lp, 0, Lp, bufs.planes, bufs.plane_stride, 0, plan.Lm, W, slices, sections, 16000)
    wcx, wcy = pc.filter_and_fit(shifts, 0.4, 5)
    assert np.array_equal(np.asarray(wcx), cx) and np.array_equal(np.asarray(wcy), cy)
    assert (shifts[..., 2] >= 0.4).sum(1).min() >= 5                      # preproc.h:492-512: enough valid units in every band

    # ---- the aligned image: oracle.align_mss on the fitted polynomials, the whole 15864 x 7500 x 4 image, bit for bit
    want, n = oracle_mod.align_mss(bands, cx, cy, plan.lps, plan.line_offset, plan.overlap, plan.keep, plan.min_lines)
    got_img = out.cpu().numpy()
    assert got_img.shape == want.shape == (plan.out_rows, W // 4, 4)
    assert np.array_equal(got_img, want)
