"""Scan-line-block sharding of the two strip work-flows over the GPUs of a node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI, "gloo" on CPU for
tests).  Rank r owns lines [r*pb, (r+1)*pb) of every raster (MSS planes: a quarter of that).

  default action (BASELINE config 4; preproc.h:224-468)      -> StripPlan / default_action_step
  prestitch + stitch (BASELINE config 5; stitcher.h:83-201)   -> CcdPlan / prestitch_stitch_step

Both have the same three exchange steps, none of them a reduction over pixels:

  1. correlation windows.  The reference correlates a FIXED number of windows per strip (5 sections
     x 10 slices x 4 bands, or 10 CCD sections), however long the strip is -- so on N GPUs the
     windows, not the lines, are the unit of work.  Every (section, slice) unit -- a 16000 x 3000
     PAN window plus four 4000 x 750 band windows, 120 MB -- is placed on a rank by predicted cost
     (`assign_groups_by_cost`: a pair of units moves away from the rank that holds its lines only when
     bytes / link bandwidth beats computing it at home); lines a unit's rank lacks arrive point-to-point
     as compact windows (ncclSend/ncclRecv, one batch per pair, posted before the resident pairs are
     computed and waited for pair by pair).  The kernels read windows through (pointer, pitch), so
     resident windows are used in place.
  2. an all-gather of the per-unit results (<= 200 x 4 doubles), after which every rank runs the
     identical fixed-order host step (filter + polynomial fit, or the CCD shift mean), so the
     maps are bit-identical on all ranks and to the 1-GPU run;
  3. resampling halo: the few source lines above/below a block that the bicubic taps of its
     output lines reach (oip_align_mss_src_range / oip_remap_shift_src_range), point-to-point.

Section seams of the reference (20000-line align sections, 30000-row remap sections) are computed
from GLOBAL line indices on every rank, so sharded output == unsharded output bit for bit.
Everything numerical is delegated to a backend (the HIP context, or the CPU oracle in the gloo
tests); this module only plans rows and moves them.
"""
from __future__ import annotations

import os

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


@dataclass
class Piece:
    """lines [row0, row0+rows) x columns [col0, col0+cols) of raster `kind`, held by rank `src`, wanted by
    rank `dst` as rows [dst_row, ...) of window `kind` of its unit `unit`"""
    src: int
    dst: int
    kind: str         # "pan" | "mss" (all four planes) | "pan1" | "pan2"
    unit: int         # global unit index
    row0: int
    rows: int
    col0: int
    cols: int
    dst_row: int


# ---- cost-aware placement ---------------------------------------------------------------------------------------
# A group (a pair of inter-band units, or one CCD section) may be computed on any rank; what it costs there is the time
# to compute it plus the time to bring in the window bytes that rank does not hold.  The model, in integer microseconds
# so that the Python and the C++ host (csrc/oip_multigpu.hpp) plan identically:
#   * a rank computes its fully resident groups first, then the others in index order;
#   * the bytes a rank lacks arrive over ONE xGMI link at `bytes_per_us`, one group after the other from time 0
#     (the senders pack and post right after their own RRC; the exchange runs on a communication stream under the
#     home groups' kernels -- default_action_step);
#   * a group starts when its bytes are in and the previous group is done:  t = max(t, arrival) + compute_us.
# Placement: every group starts at the rank that holds most of its bytes; then single groups move, best move first, as long
# as a move lowers the ranks' finish times sorted from the latest down (the critical path first, then the next rank, ...).
# Constants (round 2, one MI355X): a pair of 16000 x 3000 units computes in 2.5 ms; a pair moved whole is 240 MB;
# LINK_GBS is what one xGMI link is assumed to sustain for a grouped send/recv (spec peak 153 GB/s per direction;
# 50 is deliberately conservative and, like every number here, UNMEASURED on hardware: no multi-GPU node was
# available to the builder).
# OIP_LINK_GBS overrides it for both hosts (this module and csrc/oip_multigpu.hpp read the same variable): the first run on
# a multi-GPU node prints predicted against measured times (bench.py `multi_gpu`), the variable is how the result is fed back.
def _link_gbs_default():
    try:
        v = int(os.environ.get("OIP_LINK_GBS", "50"))
    except ValueError:
        v = 50
    return v if v > 0 else 50


LINK_GBS = _link_gbs_default()
PAIR_US_16000x3000 = 2500
CCD_SECTION_US_16000x200 = 150


def rank_finish_us(costs, compute_us, bytes_per_us):
    """finish time of a rank whose groups lack `costs` bytes each (index order): resident groups first"""
    t = compute_us * sum(1 for c in costs if c == 0)
    arrived = 0
    for c in costs:
        if c:
            arrived += -(-c // bytes_per_us)
            t = max(t, arrived) + compute_us
    return t


def assign_groups_by_cost(missing, world, compute_us, bytes_per_us):
    """missing[g][q]: bytes of group g's windows that rank q does not hold.  Returns (rank of every group, predicted
    finish time of every rank in microseconds).  Deterministic; identical on every rank and in the C++ host."""
    ng = len(missing)
    where = [min(range(world), key=lambda q: (missing[g][q], q)) for g in range(ng)]

    def finish(r, w):
        return rank_finish_us([missing[g][r] for g in range(ng) if w[g] == r], compute_us, bytes_per_us)
    fin = [finish(r, where) for r in range(world)]
    # objective: the finish times sorted from the latest down, compared lexicographically (the critical path first, then
    # the next-latest rank, ...): a move is taken only if it lowers that vector, the best such move first
    for _ in range(4 * ng * world):
        cur = sorted(fin, reverse=True)
        best = None
        for g in range(ng - 1, -1, -1):
            r = where[g]
            for q in range(world):
                if q == r:
                    continue
                w2 = list(where)
                w2[g] = q
                f2 = list(fin)
                f2[r], f2[q] = finish(r, w2), finish(q, w2)
                key = sorted(f2, reverse=True)
                if key < cur and (best is None or key < best[0]):
                    best = (key, g, q, f2)
        if best is None:
            break
        _, g, q, fin = best
        where[g] = q
    return where, fin


def _missing_bytes(windows, block_of, world):
    """windows: list of (kind, row0, rows, cols, planes); block_of[kind] = lines per rank of that raster.
    bytes (u16) of those windows that each rank does not hold"""
    out = [0] * world
    for kind, row0, rows, cols, planes in windows:
        blk = block_of[kind]
        for q in range(world):
            held = max(0, min(row0 + rows, (q + 1) * blk) - max(row0, q * blk))
            out[q] += (rows - held) * cols * planes * 2
    return out


def _window_pieces(kind, unit, dst, row0, rows, col0, cols, block, world):
    """cut the line range of one window at block boundaries"""
    out = []
    for r in range(world):
        lo, hi = max(row0, r * block), min(row0 + rows, (r + 1) * block)
        if lo < hi:
            out.append(Piece(r, dst, kind, unit, lo, hi - lo, col0, cols, lo - row0))
    return out


class StripPlan:
    """Row bookkeeping for the default action (RRC -> inter-band correlation -> align)."""

    def __init__(self, W, Lp_total, world, slices=10, sections=5, corr_lines=16000, lines_per_section=20000,
                 line_offset=0, overlap=520, keep_leading=False, min_lines=1500, halo_cap=64, link_gbs=LINK_GBS):
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
        self.base_cols = W // slices
        self.band_cols = self.base_cols // 4
        self.out_rows = self.Lm - line_offset - (0 if keep_leading else overlap)
        self.n_units = sections * slices
        # placement by predicted cost (pairs of units stay together: the kernels process two units per launch)
        self.link_gbs = link_gbs
        self.compute_us = max(1, PAIR_US_16000x3000 * self.base_rows * self.base_cols // (16000 * 3000))
        groups = [list(range(g, min(g + 2, self.n_units))) for g in range(0, self.n_units, 2)]
        missing = []
        for us in groups:
            wins = []
            for u in us:
                p0, p1, m0, m1 = self.section(u // slices)
                wins += [("pan", p0, p1 - p0, self.base_cols, 1), ("mss", m0, m1 - m0, self.band_cols, 4)]
            missing.append(_missing_bytes(wins, {"pan": self.pb, "mss": self.mb}, world))
        where, self.predicted_finish_us = assign_groups_by_cost(missing, world, self.compute_us, link_gbs * 1000)
        self.assign = [where[u // 2] for u in range(self.n_units)]

    # -- blocks
    def pan_block(self, r):
        return r * self.pb, (r + 1) * self.pb

    def mss_block(self, r):
        return r * self.mb, (r + 1) * self.mb

    # -- correlation sections and units
    def section(self, sec):
        p0 = self.base_gap + sec * (self.base_rows + self.base_gap)
        m0 = self.band_gap + sec * (self.band_rows + self.band_gap)
        return p0, p0 + self.base_rows, m0, m0 + self.band_rows

    def owner(self, sec):
        """rank holding the section's first PAN line (its `home`)"""
        return min(self.section(sec)[0] // self.pb, self.world - 1)

    def units_of(self, r):
        return [u for u in range(self.n_units) if self.assign[u] == r]

    def unit_pieces(self, u):
        """what unit u's rank needs: its PAN window and its band windows, cut at block boundaries"""
        sec, i = divmod(u, self.slices)
        p0, p1, m0, m1 = self.section(sec)
        dst = self.assign[u]
        return (_window_pieces("pan", u, dst, p0, p1 - p0, i * self.base_cols, self.base_cols, self.pb, self.world) +
                _window_pieces("mss", u, dst, m0, m1 - m0, i * self.band_cols, self.band_cols, self.mb, self.world))

    def unit_is_local(self, u):
        return all(p.src == p.dst for p in self.unit_pieces(u))

    def correlation_pieces(self):
        """every piece of every unit that is not entirely on its rank (identical on all ranks)"""
        out = []
        for u in range(self.n_units):
            if not self.unit_is_local(u):
                out += self.unit_pieces(u)
        return out

    def exchange_groups(self):
        """the same pieces pair by pair, in unit order: [(units of the pair, their pieces)] for every pair that is not
        entirely on its rank -- the posting order of the overlapped exchange (identical on all ranks)"""
        out = []
        for g in range(0, self.n_units, 2):
            us = [u for u in range(g, min(g + 2, self.n_units)) if not self.unit_is_local(u)]
            if us:
                out.append((us, [p for u in us for p in self.unit_pieces(u)]))
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
    """Per-rank device buffers of the default action.

    pan:    pb x W u16, the rank's own PAN lines
    planes: 4 x (head + mb + tail) x Wb u16, global MSS lines [m_first, ...): own lines plus align halo
    win:    compact windows of the units this rank computes but does not hold entirely
    """

    def __init__(self, plan: StripPlan, rank: int, device):
        self.plan, self.rank = plan, rank
        W, Wb = plan.W, plan.W // 4
        self.p_first = plan.pan_block(rank)[0]
        self.pan = torch.zeros(plan.pb, W, dtype=torch.uint16, device=device)
        b0, b1 = plan.mss_block(rank)
        head = min(plan.halo_cap, b0)
        tail = min(plan.halo_cap, plan.Lm - b1)
        self.m_first = b0 - head
        self.m_rows_cap = head + plan.mb + tail
        self.planes = torch.zeros(4, self.m_rows_cap, Wb, dtype=torch.uint16, device=device)
        self.plane_stride = self.m_rows_cap * Wb
        self.m_valid = [b0, b1]
        self.win = {}
        for u in plan.units_of(rank):
            if not plan.unit_is_local(u):
                self.win[u] = {"pan": torch.zeros(plan.base_rows, plan.base_cols, dtype=torch.uint16, device=device),
                               "mss": torch.zeros(4, plan.band_rows, plan.band_cols, dtype=torch.uint16, device=device)}

    def pan_rows(self, row0, rows):
        a = row0 - self.p_first
        assert 0 <= a and a + rows <= self.plan.pb, "PAN lines outside the rank's block"
        return self.pan[a:a + rows]

    def mss_view(self, band, row0, rows):
        a = row0 - self.m_first
        assert 0 <= a and a + rows <= self.m_rows_cap, "MSS halo exceeds buffer capacity"
        return self.planes[band, a:a + rows]

    def own_planes_offset(self):
        """element offset of the rank's own first MSS line inside each plane"""
        return (self.plan.mss_block(self.rank)[0] - self.m_first) * (self.plan.W // 4)

    # -- views the piece exchange works on
    def source_views(self, p: Piece):
        """the 2-D views (lines x columns) piece p covers in this rank's rasters"""
        if p.kind == "pan":
            return [self.pan_rows(p.row0, p.rows)[:, p.col0:p.col0 + p.cols]]
        return [self.mss_view(b, p.row0, p.rows)[:, p.col0:p.col0 + p.cols] for b in range(4)]

    def window_views(self, p: Piece):
        w = self.win[p.unit]
        if p.kind == "pan":
            return [w["pan"][p.dst_row:p.dst_row + p.rows]]
        return [w["mss"][b, p.dst_row:p.dst_row + p.rows] for b in range(4)]

    def unit_windows(self, u):
        """(PAN window view, [4 band window views]) of unit u: views of the rasters when the unit is local,
        the compact windows otherwise"""
        plan = self.plan
        if u in self.win:
            w = self.win[u]
            return w["pan"], [w["mss"][b] for b in range(4)]
        sec, i = divmod(u, plan.slices)
        p0, p1, m0, m1 = plan.section(sec)
        pan = self.pan_rows(p0, p1 - p0)[:, i * plan.base_cols:(i + 1) * plan.base_cols]
        bands = [self.mss_view(b, m0, m1 - m0)[:, i * plan.band_cols:(i + 1) * plan.band_cols] for b in range(4)]
        return pan, bands


def _as_bytes(t):
    """RCCL/NCCL has no 16-bit integer type; every backend has uint8"""
    return t.view(torch.uint8)


def run_pieces(pieces, bufs, rank: int, group=None):
    """execute window-piece exchanges: the holder packs the (lines x columns) sub-block contiguously and
    sends it, the unit's rank receives straight into the rows of its compact window; pieces a rank holds
    itself are plain device copies.  One grouped launch on RCCL."""
    stage = dist.is_initialized() and dist.get_backend(group) == "gloo" and bufs_is_cuda(bufs)
    ops, keep, staged = [], [], []
    for p in pieces:
        if rank == p.src and rank == p.dst:
            for s, d in zip(bufs.source_views(p), bufs.window_views(p)):
                d.copy_(s)
        elif rank == p.src:
            for s in bufs.source_views(p):
                c = s.contiguous()
                c8 = _as_bytes(c)
                if stage:
                    c8 = c8.cpu()
                keep.append(c8)
                ops.append(dist.P2POp(dist.isend, c8, p.dst, group=group))
        elif rank == p.dst:
            for d in bufs.window_views(p):
                d8 = _as_bytes(d)
                if stage:
                    h = torch.empty(d8.shape, dtype=torch.uint8)
                    staged.append((d8, h))
                    d8 = h
                ops.append(dist.P2POp(dist.irecv, d8, p.src, group=group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for dev, host in staged:
        dev.copy_(host)


def post_pieces(groups, bufs, rank: int, group=None):
    """Asynchronous form of run_pieces.  `groups`: list of (units, pieces) in a global order every rank iterates
    identically (one entry per group of units that is not entirely on its rank).  Every group is posted as ONE batch
    of sends and receives (a grouped launch on RCCL's own stream: it runs beside the compute stream's kernels) and
    the list of (units, wait) handles is returned in posting order: wait() makes the current stream (RCCL) or the host
    (gloo) wait for that group's bytes only -- so a rank computes its resident groups first and each received group
    as soon as it is in.  Every handle must be waited on before the buffers change (senders included)."""
    stage = dist.is_initialized() and dist.get_backend(group) == "gloo" and bufs_is_cuda(bufs)
    out = []
    for units, pieces in groups:
        ops, keep, staged = [], [], []
        for p in pieces:
            if rank == p.src and rank == p.dst:
                for s, d in zip(bufs.source_views(p), bufs.window_views(p)):
                    d.copy_(s)
            elif rank == p.src:
                for s in bufs.source_views(p):
                    c8 = _as_bytes(s.contiguous())
                    if stage:
                        c8 = c8.cpu()
                    keep.append(c8)
                    ops.append(dist.P2POp(dist.isend, c8, p.dst, group=group))
            elif rank == p.dst:
                for d in bufs.window_views(p):
                    d8 = _as_bytes(d)
                    if stage:
                        h = torch.empty(d8.shape, dtype=torch.uint8)
                        staged.append((d8, h))
                        d8 = h
                    ops.append(dist.P2POp(dist.irecv, d8, p.src, group=group))
        works = dist.batch_isend_irecv(ops) if ops else []

        def wait(works=works, staged=staged, keep=keep):
            for w in works:
                w.wait()
            for dev, host in staged:
                dev.copy_(host)
            keep.clear()
        out.append((units, wait))
    return out


def bufs_is_cuda(bufs):
    t = getattr(bufs, "pan", None)
    if t is None:
        t = bufs.pan1
    return t.is_cuda


def run_transfers(transfers, bufs, rank: int, group=None):
    """execute whole-line transfers into the halo rows of a raster (grouped: one launch on RCCL).
    bufs provides line_views(kind, row0, rows) -> list of contiguous row-block tensors and note_valid()."""
    # gloo has no device point-to-point: device lines are staged through host memory (used only
    # to rehearse the N-rank path on a 1-GPU box; the real runs use RCCL and move HBM to HBM)
    stage = dist.is_initialized() and dist.get_backend(group) == "gloo" and bufs_is_cuda(bufs)
    staged = []
    ops = []
    for t in transfers:
        views = bufs.line_views(t.kind, t.row0, t.rows) if rank in (t.src, t.dst) else []
        for v in views:
            v8 = _as_bytes(v)
            if stage:
                h = v8.cpu() if rank == t.src else torch.empty(v8.shape, dtype=torch.uint8)
                if rank == t.dst:
                    staged.append((v8, h))
                v8 = h
            if rank == t.src:
                ops.append(dist.P2POp(dist.isend, v8, t.dst, group=group))
            elif rank == t.dst:
                ops.append(dist.P2POp(dist.irecv, v8, t.src, group=group))
        if rank == t.dst:
            bufs.note_valid(t.kind, t.row0, t.rows)
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for dev, host in staged:
        dev.copy_(host)


def _shard_line_views(self, kind, row0, rows):
    return [self.mss_view(b, row0, rows) for b in range(4)]


def _shard_note_valid(self, kind, row0, rows):
    self.m_valid[0] = min(self.m_valid[0], row0)
    self.m_valid[1] = max(self.m_valid[1], row0 + rows)


ShardBuffers.line_views = _shard_line_views
ShardBuffers.note_valid = _shard_note_valid


def gather_table(local: np.ndarray, device, group=None) -> np.ndarray:
    """all-gather the per-rank result tables (NaN = not mine) and merge them in rank order: every rank
    ends with the same complete table."""
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


gather_shifts = gather_table


class StepTimer:
    """Stage times of ONE instrumented N-rank step on this rank: the placement model (`predicted_finish_us`)
    has never met hardware, so the first run on a node must show where a step's time goes.  tick(name) drains the backend's
    stream and books the host time since the previous tick to `name`; list-valued stages (one entry per received pair) keep
    every entry.  An instrumented step is slower than a plain one (the drains) -- bench.py runs it once, outside the timed region.
    `correlation_finish_ms`: from the end of the RRC (the model's time 0) to this rank's last correlation."""

    def __init__(self, backend):
        import time as _t
        self._now = _t.perf_counter
        self.backend = backend
        self.ms = {}
        self.backend.sync()
        self.t0 = self.last = self._now()
        self.t_rrc = None

    def tick(self, name, listed=False):
        self.backend.sync()
        now = self._now()
        d = (now - self.last) * 1e3
        if listed:
            self.ms.setdefault(name, []).append(d)
        else:
            self.ms[name] = self.ms.get(name, 0.0) + d
        self.last = now
        if name == "rrc_ms":
            self.t_rrc = now
        if name.startswith("correlate"):
            self.ms["correlation_finish_ms"] = (now - (self.t_rrc or self.t0)) * 1e3

    def result(self):
        self.ms["step_ms"] = (self.last - self.t0) * 1e3
        return self.ms


def default_action_step(backend, plan: StripPlan, bufs: ShardBuffers, raw_pan, raw_mss_bil, kb_pan, kb_mss4, out,
                        rank: int, threshold=0.4, min_count=5, group=None, fit="reference", timer: "StepTimer | None" = None):
    """One pass of the sharded default action on this rank.

    backend provides (all on device memory):
      rrc(src, dst, w, h, kb); mss_split_rrc(bil, planes, elem_offset, plane_stride, w, lines, kb4)
      interband_units(pan_windows, band_windows) -> (n, 4, 3)   [2-D views: .data_ptr(), .stride(0)]
      filter_and_fit(shifts, threshold, min_count, fit); align_src_range(o0, n, cy)
      align(planes, plane_stride, m_first, mv0, mv1, dst, out_row0, out_rows, cx, cy)
    Returns (cx, cy, (o0, o1)).
    """
    W = plan.W
    multi = plan.world > 1
    # RRC of the rank's own lines, written straight into the halo-capable buffers
    backend.rrc(raw_pan, bufs.pan, W, plan.pb, kb_pan)
    backend.mss_split_rrc(raw_mss_bil, bufs.planes, bufs.own_planes_offset(), bufs.plane_stride, W, plan.mb, kb_mss4)
    bufs.m_valid = list(plan.mss_block(rank))
    tick = timer.tick if timer is not None else (lambda *a, **k: None)
    tick("rrc_ms")
    shifts = np.full((4, plan.n_units, 4), np.nan)
    for u in range(plan.n_units):
        shifts[:, u, 3] = (u % plan.slices) * plan.base_cols + plan.base_cols // 2        # preproc.h:326

    def correlate(units):
        if not units:
            return
        wins = [bufs.unit_windows(u) for u in units]
        res = backend.interband_units([w[0] for w in wins], [w[1] for w in wins])
        for j, u in enumerate(units):
            shifts[:, u, :3] = res[j]
    mine = plan.units_of(rank)
    # Window exchange and correlation overlap: the pairs that need lines of another rank are posted first (one batch per
    # pair, on RCCL's stream), the rank's resident pairs are computed meanwhile, then every received pair as soon as its
    # bytes are in.  Pairs (2k, 2k+1) stay together and are resident or not together, so the pairing -- hence every bit of
    # the result -- is that of the single-GPU order.
    pending = []
    if multi:
        if bufs_is_cuda(bufs) and hasattr(backend, "check_stream_contract"):
            backend.check_stream_contract()
        if bufs_is_cuda(bufs) and dist.get_backend(group) == "gloo":
            backend.sync()                      # host-staged rehearsal: the RRC output is read through .cpu()
        pending = post_pieces(plan.exchange_groups(), bufs, rank, group)
        tick("exchange_post_ms")
    # (with an odd slice count a pair spans two sections and one unit can be local while its partner is not: the pair then
    # waits for its exchange group and is computed whole -- a local unit computed alone would shift every later pairing)
    pair_of = lambda u: [v for v in (u - u % 2, u - u % 2 + 1) if v < plan.n_units]
    pair_local = lambda u: all(plan.unit_is_local(v) for v in pair_of(u))
    correlate([u for u in mine if pair_local(u)])
    tick("correlate_resident_ms")
    for units, wait in pending:
        wait()
        here = [u for u in pair_of(units[0]) if plan.assign[u] == rank]
        if here:
            tick("exchange_wait_ms", listed=True)          # what this rank waited for the pair's bytes beyond its own work
        correlate(here)
        if here:
            tick("correlate_received_ms", listed=True)
    tick("exchange_drain_ms")                              # pairs sent to other ranks: their handles are waited for too
    if multi:
        shifts = gather_table(shifts, bufs.pan.device if bufs.pan.is_cuda else "cpu", group)
        tick("allgather_ms")
    cx, cy = backend.filter_and_fit(shifts, threshold, min_count, fit)
    tick("fit_ms")
    o0, o1 = plan.align_out_rows(rank)
    if multi:
        transfers, _ = plan.align_transfers(lambda a, n: backend.align_src_range(a, n, cy))
        run_transfers(transfers, bufs, rank, group)
        tick("halo_ms")
    if o1 > o0:
        backend.align(bufs.planes, bufs.plane_stride, bufs.m_first, bufs.m_valid[0], bufs.m_valid[1], out, o0,
                      o1 - o0, cx, cy)
    tick("align_ms")
    if timer is not None:
        timer.ms["units_resident"] = len([u for u in mine if pair_local(u)])
        timer.ms["units_received"] = len(mine) - timer.ms["units_resident"]
    return cx, cy, (o0, o1)


# ------------------------------------------------------------------------------------------------------
# cross-CCD path: prestitch (CalcSttParameters -> DoRRC -> PreStitch) + RAW stitch, BASELINE config 5
# ------------------------------------------------------------------------------------------------------
class CcdPlan:
    """Row bookkeeping for `prestitch` followed by `stitch` of two CCD segments (main.cpp:270-286, :177-190)."""

    def __init__(self, W, L, world, sections=10, lines_per_section=16000, overlap_cols=200, edge_cols=0,
                 section_rows=30000, row_guard=32767, fold=None, link_gbs=LINK_GBS):
        if L % world:
            raise ValueError("line count must be a multiple of world")
        # stitcher.h:75-77
        if L < sections * lines_per_section:
            raise ValueError("PAN line count less than sections times line-per-section, use smaller -s and/or -l value(s)")
        self.W, self.L, self.world = W, L, world
        self.pb = L // world
        self.sections, self.lps = sections, lines_per_section
        self.ov, self.edge = overlap_cols, edge_cols
        self.cols = overlap_cols - edge_cols
        self.section_rows, self.row_guard = section_rows, row_guard
        self.fold = overlap_cols // 2 if fold is None else fold          # main.cpp:189: --fold-cols is halved
        # stitcher.h:151-152, :167
        self.gap = (L - sections * lines_per_section) // (sections + 1)
        self.step = self.gap + lines_per_section
        self.n_units = sections
        self.link_gbs = link_gbs
        self.compute_us = max(1, CCD_SECTION_US_16000x200 * lines_per_section * self.cols // (16000 * 200))
        missing = []
        for u in range(sections):
            a, b = self.section(u)
            missing.append(_missing_bytes([("pan", a, b - a, self.cols, 2)], {"pan": self.pb}, world))
        self.assign, self.predicted_finish_us = assign_groups_by_cost(missing, world, self.compute_us, link_gbs * 1000)

    def block(self, r):
        return r * self.pb, (r + 1) * self.pb

    def section(self, s):
        off = self.gap + s * self.step
        return off, off + self.lps

    def units_of(self, r):
        return [u for u in range(self.n_units) if self.assign[u] == r]

    def unit_pieces(self, u):
        a, b = self.section(u)
        dst = self.assign[u]
        # stitcher.h:175-176: PAN1 cols [W-ov, W-edge), PAN2 cols [edge, ov)
        return (_window_pieces("pan1", u, dst, a, b - a, self.W - self.ov, self.cols, self.pb, self.world) +
                _window_pieces("pan2", u, dst, a, b - a, self.edge, self.cols, self.pb, self.world))

    def unit_is_local(self, u):
        return all(p.src == p.dst for p in self.unit_pieces(u))

    def correlation_pieces(self):
        out = []
        for u in range(self.n_units):
            if not self.unit_is_local(u):
                out += self.unit_pieces(u)
        return out

    def exchange_groups(self):
        """[(section, its pieces)] for every section that is not entirely on its rank, in section order"""
        return [([u], self.unit_pieces(u)) for u in range(self.n_units) if not self.unit_is_local(u)]

    def remap_transfers(self, src_range_fn):
        """src_range_fn(out_row0, out_rows) -> (first, last) source lines (oip_remap_shift_src_range);
        returns the line transfers and every rank's needed range"""
        out, need = [], []
        for r in range(self.world):
            b0, b1 = self.block(r)
            f, l = src_range_fn(b0, b1 - b0)
            need.append((f, l))
            for q in range(self.world):
                if q == r:
                    continue
                lo, hi = max(f, q * self.pb), min(l, (q + 1) * self.pb)
                if lo < hi:
                    out.append(Transfer(q, r, "pan2", lo, hi - lo))
        return out, need


class CcdBuffers:
    """raw1/raw2: the rank's own raw lines (inputs, not owned); rrc1: own corrected CCD-1 lines;
    rrc2: corrected CCD-2 lines [first, ...) incl. the remap halo -- allocated once the shift is known;
    win: compact windows of sections this rank computes but does not hold entirely"""

    def __init__(self, plan: CcdPlan, rank: int, raw1, raw2):
        self.plan, self.rank = plan, rank
        self.pan1, self.pan2 = raw1, raw2
        dev = raw1.device
        self.rrc1 = torch.empty_like(raw1)
        self.rrc2 = None
        self.r2_first = plan.block(rank)[0]
        self.win = {}
        for u in plan.units_of(rank):
            if not plan.unit_is_local(u):
                self.win[u] = {"pan1": torch.zeros(plan.lps, plan.cols, dtype=torch.uint16, device=dev),
                               "pan2": torch.zeros(plan.lps, plan.cols, dtype=torch.uint16, device=dev)}

    def source_views(self, p: Piece):
        b0 = self.plan.block(self.rank)[0]
        t = self.pan1 if p.kind == "pan1" else self.pan2
        return [t[p.row0 - b0:p.row0 - b0 + p.rows, p.col0:p.col0 + p.cols]]

    def window_views(self, p: Piece):
        return [self.win[p.unit][p.kind][p.dst_row:p.dst_row + p.rows]]

    def unit_windows(self, u):
        plan = self.plan
        if u in self.win:
            return self.win[u]["pan1"], self.win[u]["pan2"]
        a, b = plan.section(u)
        b0 = plan.block(self.rank)[0]
        return (self.pan1[a - b0:b - b0, plan.W - plan.ov:plan.W - plan.edge],
                self.pan2[a - b0:b - b0, plan.edge:plan.ov])

    def alloc_rrc2(self, first, last):
        b0, b1 = self.plan.block(self.rank)
        first, last = min(first, b0), max(last, b1)
        if self.rrc2 is None or self.r2_first != first or self.rrc2.shape[0] != last - first:
            self.rrc2 = torch.zeros(last - first, self.plan.W, dtype=torch.uint16, device=self.pan1.device)
        self.r2_first = first

    def line_views(self, kind, row0, rows):
        a = row0 - self.r2_first
        assert 0 <= a and a + rows <= self.rrc2.shape[0], "remap halo exceeds buffer"
        return [self.rrc2[a:a + rows]]

    def note_valid(self, kind, row0, rows):
        pass


def prestitch_stitch_step(backend, plan: CcdPlan, bufs: CcdBuffers, kb1, kb2, prestt, stitched, rank: int,
                          threshold=0.4, max_delta_y=0.0, f16acc=False, group=None, fused=False, timer: "StepTimer | None" = None):
    """One pass of the sharded cross-CCD path on this rank: CalcSttParameters on the RAW lines (App. B-1) ->
    RRC of both CCDs -> constant-shift bicubic remap of CCD 2 with row halo -> RAW stitch.

    backend provides:
      stt_windows(a_windows, b_windows) -> (n, 3);  stt_mean(table, threshold, max_dy) -> (dx, dy, resp, valid)
      rrc(src, dst, w, h, kb);  remap_src_range(out_row0, out_rows, dy)
      remap(src, src_row0, src_rows, dst, out_row0, out_rows, dx, dy, f16acc);  stitch(left, right, out, rows)
    prestt: pb x W (the rank's block of .RRC.PRESTT.RAW); stitched: pb x 2(W - fold).  Returns (dx, dy, table).

    fused=True (the single-pass form of `oip task`, when .RRC.RAW / .RRC.PRESTT.RAW are not requested products): the RRC of
    CCD 1 writes straight into the left half of `stitched` (backend.rrc_window) and the resampled CCD-2 lines straight into
    its right half (backend.remap_window) -- three full passes over the strips become one each; `prestt` and bufs.rrc1 are
    not touched and `stitched` holds the same bits.
    """
    W = plan.W
    multi = plan.world > 1
    b0, b1 = plan.block(rank)
    table = np.full((plan.sections, 3), np.nan)
    tick = timer.tick if timer is not None else (lambda *a, **k: None)
    tick("rrc_ms")                                          # (nothing yet: CalcSttParameters reads the RAW lines -- the model's time 0)

    def correlate(units):
        if not units:
            return
        wins = [bufs.unit_windows(u) for u in units]
        res = backend.stt_windows([w[0] for w in wins], [w[1] for w in wins])
        for j, u in enumerate(units):
            table[u] = res[j]
    mine = plan.units_of(rank)
    pending = []
    if multi:
        if bufs_is_cuda(bufs) and hasattr(backend, "check_stream_contract"):
            backend.check_stream_contract()
        if bufs_is_cuda(bufs) and dist.get_backend(group) == "gloo":
            backend.sync()
        pending = post_pieces(plan.exchange_groups(), bufs, rank, group)     # posted first, computed as they arrive
        tick("exchange_post_ms")
    correlate([u for u in mine if plan.unit_is_local(u)])
    tick("correlate_resident_ms")
    for units, wait in pending:
        wait()
        here = [u for u in units if plan.assign[u] == rank]
        if here:
            tick("exchange_wait_ms", listed=True)
        correlate(here)
        if here:
            tick("correlate_received_ms", listed=True)
    tick("exchange_drain_ms")
    if multi:
        table = gather_table(table, bufs.pan1.device if bufs.pan1.is_cuda else "cpu", group)
        tick("allgather_ms")
    dx, dy, _, _ = backend.stt_mean(table, threshold, max_delta_y)       # identical on every rank
    # DoRRC (stitcher.h:141-146): own lines of both CCDs; CCD 2 lands in the halo-capable buffer
    if fused:
        backend.rrc_window(bufs.pan1, W, stitched, 2 * (W - plan.fold), W - plan.fold, plan.pb, kb1)
    else:
        backend.rrc(bufs.pan1, bufs.rrc1, W, plan.pb, kb1)
    transfers, need = plan.remap_transfers(lambda a, n: backend.remap_src_range(a, n, dy))
    f, l = need[rank]
    if fused and not transfers:
        # no halo line moves anywhere (one rank) -- a decision every rank takes alike: the resampling kernel corrects the raw
        # CCD-2 samples on load -- DoRRC of CCD 2, PreStitch and the right half of the stitch are one pass, nothing in between
        backend.remap_rrc_window(bufs.pan2, b0, plan.pb, kb2, stitched, 2 * (W - plan.fold), plan.fold, W - plan.fold, b0, plan.pb, dx, dy,
                                 f16acc)
        tick("rrc_remap_stitch_ms")
        return dx, dy, table
    bufs.alloc_rrc2(f, l)
    backend.rrc(bufs.pan2, bufs.rrc2[b0 - bufs.r2_first:b1 - bufs.r2_first], W, plan.pb, kb2)
    tick("rrc_x2_ms")
    if multi:
        backend.sync()
        run_transfers(transfers, bufs, rank, group)
        tick("halo_ms")
    if fused:
        backend.remap_window(bufs.rrc2, bufs.r2_first, bufs.rrc2.shape[0], stitched, 2 * (W - plan.fold), plan.fold, W - plan.fold, b0,
                             plan.pb, dx, dy, f16acc)
        tick("remap_stitch_ms")
        return dx, dy, table
    backend.remap(bufs.rrc2, bufs.r2_first, bufs.rrc2.shape[0], prestt, b0, plan.pb, dx, dy, f16acc)
    backend.stitch(bufs.rrc1, prestt, stitched, plan.pb)
    tick("remap_stitch_ms")
    return dx, dy, table


class HipBackend:
    """adapter from the plans' calls to the C ABI (opticalimageprocessor_amd.Context)"""

    def __init__(self, ctx, plan):
        self.ctx, self.plan = ctx, plan

    def sync(self):
        self.ctx.sync()

    def check_stream_contract(self):
        """The N-rank steps order RRC output -> RCCL send, RCCL receive -> work.wait() -> correlation kernels through ONE stream:
        torch's current stream (the one RCCL synchronises against) must be the context's compute stream (a caller that
        kept the context's own stream would send windows before their RRC has run, silently)."""
        import torch
        if torch.cuda.is_available() and self.ctx.get_stream() != torch.cuda.current_stream().cuda_stream:
            raise RuntimeError("the oip context's stream is not torch's current stream: call ctx.set_stream(torch.cuda.current_stream()) "
                               "before an N-rank step (RCCL orders its transfers against torch's current stream)")

    def rrc(self, src, dst, w, h, kb):
        self.ctx.rrc_u16(src, dst, w, h, kb)

    def mss_split_rrc(self, bil, planes, elem_offset, plane_stride, w, lines, kb4):
        self.ctx.mss_split_rrc_u16(bil, planes.data_ptr() + 2 * elem_offset, plane_stride, w, lines, kb4)

    def interband_units(self, pan_wins, band_wins):
        p = self.plan
        return self.ctx.interband_correlate_units([w.data_ptr() for w in pan_wins], [w.stride(0) for w in pan_wins],
                                                  [[b.data_ptr() for b in u] for u in band_wins],
                                                  [u[0].stride(0) for u in band_wins], p.base_rows, p.base_cols)

    def stt_windows(self, a_wins, b_wins):
        p = self.plan
        return self.ctx.stt_correlate_windows([w.data_ptr() for w in a_wins], [w.stride(0) for w in a_wins],
                                              [w.data_ptr() for w in b_wins], [w.stride(0) for w in b_wins],
                                              p.lps, p.cols)

    def stt_mean(self, table, threshold, max_dy):
        from .capi import stt_mean
        return stt_mean(table, threshold, max_dy)

    def filter_and_fit(self, shifts, threshold, min_count, fit="reference"):
        from .capi import filter_and_fit
        return filter_and_fit(shifts, threshold, min_count, fit)

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

    def remap_src_range(self, out_row0, out_rows, dy):
        from .capi import remap_shift_src_range
        p = self.plan
        if p.L <= p.row_guard:
            raise ValueError("too few data rows, please use cv::remap()")        # imageop.h:242-244
        return remap_shift_src_range(out_row0, out_rows, p.L, dy, p.section_rows)

    def remap(self, src, src_row0, src_rows, dst, out_row0, out_rows, dx, dy, f16acc):
        p = self.plan
        self.ctx.remap_shift_bicubic_u16(src, dst, p.W, p.L, dx, dy, p.section_rows, p.row_guard, src_row0=src_row0,
                                         src_rows=src_rows, out_row0=out_row0, out_rows=out_rows, f16acc=f16acc)

    def stitch(self, left, right, out, rows):
        self.ctx.stitch_rows_u16(left, right, out, self.plan.W, rows, self.plan.fold)

    def rrc_window(self, src, src_pitch, dst, dst_pitch, w, h, kb):
        self.ctx.rrc_u16_window(src, src_pitch, dst, dst_pitch, w, h, kb)

    def remap_rrc_window(self, src_raw, src_row0, src_rows, kb, dst, dst_pitch, dst_col0, dst_col_off, out_row0, out_rows, dx, dy, f16acc=False):
        p = self.plan
        self.ctx.remap_shift_rrc_bicubic_u16_window(src_raw, kb, dst, dst_pitch, dst_col0, dst_col_off, p.W, p.L, dx, dy, p.section_rows,
                                                    p.row_guard, src_row0=src_row0, src_rows=src_rows, out_row0=out_row0, out_rows=out_rows,
                                                    f16acc=f16acc)

    def remap_window(self, src, src_row0, src_rows, dst, dst_pitch, dst_col0, dst_col_off, out_row0, out_rows, dx, dy, f16acc):
        p = self.plan
        self.ctx.remap_shift_bicubic_u16_window(src, dst, dst_pitch, dst_col0, dst_col_off, p.W, p.L, dx, dy, p.section_rows,
                                                p.row_guard, src_row0=src_row0, src_rows=src_rows, out_row0=out_row0,
                                                out_rows=out_rows, f16acc=f16acc)
