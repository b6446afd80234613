"""CPU stand-in for the HIP backend of opticalimageprocessor_amd.dist, built on the oracle.
TESTS ONLY: lets the gloo world_size>1 tests exercise the row planning, halo exchange and
all-gather of the sharded default action without a GPU."""
import numpy as np
import torch

import opticalimageprocessor_amd as oip
import oracle
from oracle import phasecorr as pc


class OracleBackend:
    def __init__(self, plan):
        self.plan = plan

    def sync(self):
        pass

    def rrc(self, src, dst, w, h, kb):
        dst.copy_(torch.from_numpy(oracle.rrc(src.numpy(), kb)))

    def mss_split_rrc(self, bil, planes, elem_offset, plane_stride, w, lines, kb4):
        bw = w // 4
        r0 = elem_offset // bw
        bands = oracle.split_mss(bil.numpy())
        for b in range(4):
            planes[b, r0:r0 + lines] = torch.from_numpy(oracle.rrc(bands[b], kb4[b * bw:(b + 1) * bw]))

    def interband(self, pan, Lp, p0, pn, planes, plane_stride, m_first, mv0, mv1, W, slices, sections, corr):
        p = self.plan
        out = np.full((4, slices * sections, 4), np.nan)
        base_cols = W // slices
        band_cols = base_cols // 4
        pan_np = pan.numpy()
        for sec in range(sections):
            a0, a1, b0, b1 = p.section(sec)
            for i in range(slices):
                out[:, sec * slices + i, 3] = i * base_cols + base_cols // 2
            if a0 < p0 or a1 > p0 + pn or b0 < mv0 or b1 > mv1:
                continue
            for i in range(slices):
                base = oracle.window_u16_to_f32(pan_np, a0 - p0, i * base_cols, a1 - a0, base_cols)
                for b in range(4):
                    bs = oracle.window_u16_to_f32(planes[b].numpy(), b0 - m_first, i * band_cols, b1 - b0, band_cols)
                    up = oracle.resize_cubic(bs, base_cols, a1 - a0)
                    (dx, dy), rs = pc.phase_correlate(base, up)
                    out[b, sec * slices + i, :3] = (dx, dy, rs)
        return out

    def filter_and_fit(self, shifts, threshold, min_count):
        return oip.filter_and_fit(shifts, threshold, min_count)     # product host code (no GPU needed)

    def align_src_range(self, o0, n, cy):
        p = self.plan
        return oip.align_mss_src_range(o0, n, p.Lm, cy, p.W // 4, p.lps, p.line_offset, p.overlap, p.keep, p.min_lines)

    def align(self, planes, plane_stride, m_first, mv0, mv1, out, o0, n, cx, cy):
        p = self.plan
        Wb = p.W // 4
        full = [np.zeros((p.Lm, Wb), np.uint16) for _ in range(4)]
        for b in range(4):
            full[b][mv0:mv1] = planes[b, mv0 - m_first:mv1 - m_first].numpy()
        res, _ = oracle.align_mss(full, cx, cy, p.lps, p.line_offset, p.overlap, p.keep, p.min_lines)
        out[:n] = torch.from_numpy(res[o0:o0 + n])
