"""CPU stand-in for the HIP backend of opticalimageprocessor_amd.dist, built on the oracle.
TESTS ONLY: lets the gloo world_size>1 tests exercise the row planning, window / halo exchange and
all-gather of the sharded work-flows without a GPU."""
import numpy as np
import torch

import opticalimageprocessor_amd as oip
import oracle
from oracle import phasecorr as pc


def _np16(t):
    """a (possibly strided) uint16 tensor view as a contiguous numpy array"""
    return np.ascontiguousarray(t.contiguous().numpy())


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

    def interband_units(self, pan_wins, band_wins):
        p = self.plan
        out = np.zeros((len(pan_wins), 4, 3))
        for j, (pw, bws) in enumerate(zip(pan_wins, band_wins)):
            base = _np16(pw).astype(np.float32)
            for b in range(4):
                up = oracle.resize_cubic(_np16(bws[b]).astype(np.float32), p.base_cols, p.base_rows)
                (dx, dy), rs = pc.phase_correlate(base, up)
                out[j, b] = (dx, dy, rs)
        return out

    def stt_windows(self, a_wins, b_wins):
        out = np.zeros((len(a_wins), 3))
        for j, (a, b) in enumerate(zip(a_wins, b_wins)):
            (dx, dy), rs = pc.phase_correlate(_np16(a).astype(np.float32), _np16(b).astype(np.float32))
            out[j] = (dx, dy, rs)
        return out

    def stt_mean(self, table, threshold, max_dy):
        return oip.stt_mean(table, threshold, max_dy)                   # product host code (no GPU needed)

    def filter_and_fit(self, shifts, threshold, min_count, fit="reference"):
        return oip.filter_and_fit(shifts, threshold, min_count, fit)    # product host code (no GPU needed)

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

    def remap_src_range(self, out_row0, out_rows, dy):
        p = self.plan
        return oip.remap_shift_src_range(out_row0, out_rows, p.L, dy, p.section_rows)

    def remap(self, src, src_row0, src_rows, dst, out_row0, out_rows, dx, dy, f16acc):
        # the whole-strip restatement on a raster that holds ONLY the lines this rank has: if the halo
        # range were too small, the missing lines (zeros) would change the rank's output rows
        p = self.plan
        full = np.zeros((p.L, p.W), np.uint16)
        full[src_row0:src_row0 + src_rows] = src.numpy()
        res, _ = oracle.prestitch(full, dx, dy, p.section_rows, p.row_guard)
        dst[:out_rows] = torch.from_numpy(res[out_row0:out_row0 + out_rows])

    def stitch(self, left, right, out, rows):
        out.copy_(torch.from_numpy(oracle.stitch_raw(left.numpy(), right.numpy(), self.plan.fold)))

    # fused prestitch -> stitch (dist.prestitch_stitch_step(fused=True)): the same oracle calls, written into the halves
    def rrc_window(self, src, src_pitch, dst, dst_pitch, w, h, kb):
        res = oracle.rrc(np.ascontiguousarray(src.numpy()[:h, :w]), np.asarray(kb)[:w])
        dst[:h, :w] = torch.from_numpy(res)

    def remap_window(self, src, src_row0, src_rows, dst, dst_pitch, dst_col0, dst_col_off, out_row0, out_rows, dx, dy, f16acc):
        p = self.plan
        full = np.zeros((p.L, p.W), np.uint16)
        full[src_row0:src_row0 + src_rows] = src.numpy()
        res, _ = oracle.prestitch(full, dx, dy, p.section_rows, p.row_guard)
        dst[:out_rows, dst_col_off:dst_col_off + p.W - dst_col0] = torch.from_numpy(res[out_row0:out_row0 + out_rows, dst_col0:])

    def remap_rrc_window(self, src_raw, src_row0, src_rows, kb, dst, dst_pitch, dst_col0, dst_col_off, out_row0, out_rows, dx, dy, f16acc=False):
        corrected = torch.from_numpy(oracle.rrc(np.ascontiguousarray(src_raw.numpy()), np.asarray(kb)))
        self.remap_window(corrected, src_row0, src_rows, dst, dst_pitch, dst_col0, dst_col_off, out_row0, out_rows, dx, dy, f16acc)
