"""Scan-line-block sharding of one 4-band strip over the GPUs of a node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI, "gloo" on CPU for
tests).  Rank r owns PAN lines [r*pb, (r+1)*pb) and MSS lines [r*pb/4, (r+1)*pb/4) of every
plane.  The path has exactly three exchange steps, none of them a reduction over pixels:

  1. correlation windows: a 16000-line correlation section (preproc.h:245-259) that straddles a
     block boundary is computed by the rank owning its first line; the lines it lacks are sent
     point-to-point by the neighbour(s) (ncclSend/ncclRecv, one direct xGMI link each);
  2. an all-gather of the per-(section, slice, band) results (<= 200 x 4 doubles), after which
     every rank runs the identical fixed-order filter + polynomial fit, so the maps are
     bit-identical on all ranks and to the 1-GPU run;
  3. align halo: the few MSS lines above/below a block that the bicubic taps of its output
     lines reach (oip_align_mss_src_range), again point-to-point.

Section seams of the reference (20000-line align sections, preproc.h:379-408) are computed
from GLOBAL line indices on every rank, so sharded output == unsharded output bit for bit.
Everything numerical is delegated to a backend (the HIP context, or the CPU oracle in the
gloo tests); this module only plans rows and moves them.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch
import torch.distributed as dist


@dataclass
class Transfer:
    src: int          # sending rank
    dst: int          # receiving rank
    kind: str         # "pan" | "mss"
    row0: int         # global first line
    rows: int


class StripPlan:
    """Row bookkeeping for the default action (RRC -> inter-band correlation -> align)."""

    def __init__(self, W, Lp_total, world, slices=10, sections=5, corr_lines=16000, lines_per_section=20000,
                 line_offset=0, overlap=520, keep_leading=False, min_lines=1500, halo_cap=64):
        if Lp_total % (4 * world):
            raise ValueError("PAN line count must be a multiple of 4*world")
        self.W, self.Lp, self.world = W, Lp_total, world
        self.Lm = Lp_total // 4
        self.pb = Lp_total // world
        self.mb = self.pb // 4
        self.slices, self.sections, self.corr_lines = slices, sections, corr_lines
        self.lps, self.line_offset, self.overlap = lines_per_section, line_offset, overlap
        self.keep, self.min_lines, self.halo_cap = keep_leading, min_lines, halo_cap
        # preproc.h:234-237, :245-247, :274-276
        if sections > 1 and sections * corr_lines > Lp_total:
            raise ValueError("CalcInterBandCorrelation: too many sections")
        self.base_rows = min(Lp_total, corr_lines)
        self.base_gap = (Lp_total - self.base_rows * sections) // (sections + 1)
        self.band_rows = self.base_rows // 4
        self.band_gap = self.base_gap // 4
        self.out_rows = self.Lm - line_offset - (0 if keep_leading else overlap)

    # -- blocks
    def pan_block(self, r):
        return r * self.pb, (r + 1) * self.pb

    def mss_block(self, r):
        return r * self.mb, (r + 1) * self.mb

    # -- correlation sections
    def section(self, sec):
        p0 = self.base_gap + sec * (self.base_rows + self.base_gap)
        m0 = self.band_gap + sec * (self.band_rows + self.band_gap)
        return p0, p0 + self.base_rows, m0, m0 + self.band_rows

    def owner(self, sec):
        return min(self.section(sec)[0] // self.pb, self.world - 1)

    def pan_tail(self, r):
        """PAN lines beyond rank r's block that its owned sections need"""
        e = self.pan_block(r)[1]
        return max([0] + [self.section(s)[1] - e for s in range(self.sections) if self.owner(s) == r])

    def mss_head_tail_capacity(self, r):
        b0, b1 = self.mss_block(r)
        tail = max([0] + [self.section(s)[3] - b1 for s in range(self.sections) if self.owner(s) == r])
        head = max([0] + [b0 - self.section(s)[2] for s in range(self.sections) if self.owner(s) == r])
        return max(head, self.halo_cap), max(tail, self.halo_cap)

    def correlation_transfers(self):
        """lines each section's owner lacks, cut at block boundaries (identical on all ranks)"""
        out = []
        for s in range(self.sections):
            o = self.owner(s)
            p0, p1, m0, m1 = self.section(s)
            for kind, a, b, blk in (("pan", p0, p1, self.pb), ("mss", m0, m1, self.mb)):
                for r in range(self.world):
                    if r == o:
                        continue
                    lo, hi = max(a, r * blk), min(b, (r + 1) * blk)
                    if lo < hi:
                        out.append(Transfer(r, o, kind, lo, hi - lo))
        return out

    # -- align
    def align_out_rows(self, r):
        """output lines of the aligned image rank r produces: those whose nominal source line
        (o + line_offset [+ overlap]) falls in its MSS block"""
        shift = self.line_offset + (0 if self.keep else self.overlap)
        b0, b1 = self.mss_block(r)
        o0 = 0 if r == 0 else min(max(b0 - shift, 0), self.out_rows)
        o1 = self.out_rows if r == self.world - 1 else min(max(b1 - shift, 0), self.out_rows)
        return o0, max(o1, o0)

    def align_transfers(self, src_range_fn):
        """src_range_fn(o0, n) -> (first, last) MSS lines (oip_align_mss_src_range)"""
        out = []
        need = []
        for r in range(self.world):
            o0, o1 = self.align_out_rows(r)
            f, l = src_range_fn(o0, o1 - o0) if o1 > o0 else (0, 0)
            need.append((f, l))
            for q in range(self.world):
                if q == r or l <= f:
                    continue
                lo, hi = max(f, q * self.mb), min(l, (q + 1) * self.mb)
                if lo < hi:
                    out.append(Transfer(q, r, "mss", lo, hi - lo))
        return out, need


class ShardBuffers:
    """Per-rank device buffers with room for received lines.

    pan:    (pb + pan_tail) x W u16, holds global PAN lines [p_first, ...)
    planes: 4 x (head + mb + tail) x Wb u16, holds global MSS lines [m_first, ...)
    """

    def __init__(self, plan: StripPlan, rank: int, device):
        self.plan, self.rank = plan, rank
        W, Wb = plan.W, plan.W // 4
        self.p_first = plan.pan_block(rank)[0]
        self.pan_rows_cap = plan.pb + plan.pan_tail(rank)
        self.pan = torch.zeros(self.pan_rows_cap, W, dtype=torch.uint16, device=device)
        head, tail = plan.mss_head_tail_capacity(rank)
        b0, b1 = plan.mss_block(rank)
        head = min(head, b0)
        tail = min(tail, plan.Lm - b1)
        self.m_first = b0 - head
        self.m_rows_cap = head + plan.mb + tail
        self.planes = torch.zeros(4, self.m_rows_cap, Wb, dtype=torch.uint16, device=device)
        self.plane_stride = self.m_rows_cap * Wb
        # valid extents (global lines), grown as halos arrive
        self.p_valid = [self.p_first, self.p_first + plan.pb]
        self.m_valid = [b0, b1]

    def pan_view(self, row0, rows):
        a = row0 - self.p_first
        assert 0 <= a and a + rows <= self.pan_rows_cap, "PAN halo exceeds buffer capacity"
        return self.pan[a:a + rows]

    def mss_view(self, band, row0, rows):
        a = row0 - self.m_first
        assert 0 <= a and a + rows <= self.m_rows_cap, "MSS halo exceeds buffer capacity"
        return self.planes[band, a:a + rows]

    def own_planes_offset(self):
        """element offset of the rank's own first MSS line inside each plane"""
        return (self.plan.mss_block(self.rank)[0] - self.m_first) * (self.plan.W // 4)


def run_transfers(transfers, bufs: ShardBuffers, rank: int, group=None):
    """execute the point-to-point line transfers (grouped: one launch on RCCL)"""
    # gloo has no device point-to-point: device lines are staged through host memory (used only
    # to rehearse the N-rank path on a 1-GPU box; the real runs use RCCL and move HBM to HBM)
    stage = dist.get_backend(group) == "gloo" and bufs.pan.is_cuda
    staged = []
    ops = []
    for t in transfers:
        if t.kind == "pan":
            views = [bufs.pan_view(t.row0, t.rows)] if rank in (t.src, t.dst) else []
        else:
            views = [bufs.mss_view(b, t.row0, t.rows) for b in range(4)] if rank in (t.src, t.dst) else []
        for v in views:
            # carry the lines as bytes: RCCL/NCCL has no 16-bit integer type, every backend has uint8
            v16 = v.view(torch.uint8)
            if stage:
                h = v16.cpu() if rank == t.src else torch.empty(v16.shape, dtype=torch.uint8)
                if rank == t.dst:
                    staged.append((v16, h))
                v16 = h
            if rank == t.src:
                ops.append(dist.P2POp(dist.isend, v16, t.dst, group=group))
            elif rank == t.dst:
                ops.append(dist.P2POp(dist.irecv, v16, t.src, group=group))
        if rank == t.dst:
            ext = bufs.p_valid if t.kind == "pan" else bufs.m_valid
            ext[0] = min(ext[0], t.row0)
            ext[1] = max(ext[1], t.row0 + t.rows)
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for dev, host in staged:
        dev.copy_(host)


def gather_shifts(local: np.ndarray, device, group=None) -> np.ndarray:
    """all-gather the per-rank correlation tables (NaN = not mine) and merge them in rank
    order: every rank ends with the same complete table."""
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo":
        device = "cpu"
    t = torch.from_numpy(np.ascontiguousarray(local)).to(device)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t, group=group)
    out = np.full(local.shape, np.nan)
    for p in parts:
        a = p.cpu().numpy()
        m = ~np.isnan(a[..., 0])
        out[m] = a[m]
    return out


def default_action_step(backend, plan: StripPlan, bufs: ShardBuffers, raw_pan, raw_mss_bil, kb_pan, kb_mss4, out,
                        rank: int, threshold=0.4, min_count=5, group=None):
    """One pass of the sharded default action on this rank.

    backend provides (all on device memory, line windows in GLOBAL indices):
      rrc(src, dst, w, h, kb); mss_split_rrc(bil, planes_ptr_offset, plane_stride, w, lines, kb4)
      interband(pan, Lp, prow0, pn, planes, plane_stride, mrow0, mn, W, slices, sections, corr)
      filter_and_fit(shifts, threshold, min_count); align_src_range(o0, n, cy)
      align(planes, plane_stride, src_row0, src_rows, dst, out_row0, out_rows, cx, cy)
    Returns (cx, cy, (o0, o1)).
    """
    W = plan.W
    multi = plan.world > 1
    # RRC of the rank's own lines, written straight into the halo-capable buffers
    backend.rrc(raw_pan, bufs.pan_view(bufs.p_first, plan.pb), W, plan.pb, kb_pan)
    backend.mss_split_rrc(raw_mss_bil, bufs.planes, bufs.own_planes_offset(), bufs.plane_stride, W, plan.mb, kb_mss4)
    bufs.p_valid = [bufs.p_first, bufs.p_first + plan.pb]
    bufs.m_valid = list(plan.mss_block(rank))
    if multi:
        backend.sync()
        run_transfers(plan.correlation_transfers(), bufs, rank, group)
    shifts = backend.interband(bufs.pan, plan.Lp, bufs.p_valid[0], bufs.p_valid[1] - bufs.p_valid[0],
                               bufs.planes, bufs.plane_stride, bufs.m_first, bufs.m_valid[0], bufs.m_valid[1],
                               W, plan.slices, plan.sections, plan.corr_lines)
    if multi:
        shifts = gather_shifts(shifts, bufs.pan.device if bufs.pan.is_cuda else "cpu", group)
    cx, cy = backend.filter_and_fit(shifts, threshold, min_count)
    o0, o1 = plan.align_out_rows(rank)
    if multi:
        transfers, _ = plan.align_transfers(lambda a, n: backend.align_src_range(a, n, cy))
        run_transfers(transfers, bufs, rank, group)
    if o1 > o0:
        backend.align(bufs.planes, bufs.plane_stride, bufs.m_first, bufs.m_valid[0], bufs.m_valid[1], out, o0,
                      o1 - o0, cx, cy)
    return cx, cy, (o0, o1)


class HipBackend:
    """adapter from the plan's calls to the C ABI (opticalimageprocessor_amd.Context)"""

    def __init__(self, ctx, plan: StripPlan):
        self.ctx, self.plan = ctx, plan

    def sync(self):
        self.ctx.sync()

    def rrc(self, src, dst, w, h, kb):
        self.ctx.rrc_u16(src, dst, w, h, kb)

    def mss_split_rrc(self, bil, planes, elem_offset, plane_stride, w, lines, kb4):
        self.ctx.mss_split_rrc_u16(bil, planes.data_ptr() + 2 * elem_offset, plane_stride, w, lines, kb4)

    def interband(self, pan, Lp, p0, pn, planes, plane_stride, m_first, mv0, mv1, W, slices, sections, corr):
        Wb = W // 4
        base = planes.data_ptr() + 2 * (mv0 - m_first) * Wb
        pan_base = pan.data_ptr()      # buffer row 0 == global line p0 (own block start)
        return self.ctx.interband_correlate(pan_base, Lp, p0, pn, base, plane_stride, mv0, mv1 - mv0, W, slices,
                                            sections, corr)

    def filter_and_fit(self, shifts, threshold, min_count):
        from .capi import filter_and_fit
        return filter_and_fit(shifts, threshold, min_count)

    def align_src_range(self, o0, n, cy):
        from .capi import align_mss_src_range
        p = self.plan
        return align_mss_src_range(o0, n, p.Lm, cy, p.W // 4, p.lps, p.line_offset, p.overlap, p.keep, p.min_lines)

    def align(self, planes, plane_stride, m_first, mv0, mv1, out, o0, n, cx, cy):
        p = self.plan
        Wb = p.W // 4
        base = planes.data_ptr() + 2 * (mv0 - m_first) * Wb
        self.ctx.align_mss_bicubic_u16x4(base, plane_stride, out, Wb, p.Lm, cx, cy, p.lps, p.line_offset, p.overlap,
                                         p.keep, p.min_lines, src_row0=mv0, src_rows=mv1 - mv0, out_row0=o0,
                                         out_rows=n)
