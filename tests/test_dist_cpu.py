"""CPU suite, part 3: the N>1 path on gloo, world_size 2.

Two processes each own half of a small 4-band strip and run the sharded default action of
opticalimageprocessor_amd.dist (row planning, point-to-point halo exchange, all-gather of the
correlation table, identical fit on every rank) with the oracle standing in for the GPU.
Concatenated output must equal the single-process result bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

W, LP, SLICES, SECTIONS, CORR = 640, 4800, 8, 3, 1200
ALIGN = dict(lines_per_section=700, line_offset=0, overlap=60, keep_leading=False, min_lines=100)
THR = -1.0      # tiny windows: accept every correlation, the test is about the plumbing


def _inputs(W=W):
    from opticalimageprocessor_amd import synth
    kb = synth.lut(W)
    kb4 = np.concatenate([synth.lut(W // 4, 10 + b) for b in range(4)], 0)
    return kb, kb4


def _run_rank(rank, world, port, tmp, slices=SLICES, W=W, timed=False):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from opticalimageprocessor_amd import synth
    from opticalimageprocessor_amd.dist import ShardBuffers, StripPlan, default_action_step
    from _oracle_backend import OracleBackend
    kb, kb4 = _inputs(W)
    plan = StripPlan(W, LP, world, slices, SECTIONS, CORR, halo_cap=16, **ALIGN)
    bufs = ShardBuffers(plan, rank, "cpu")
    # which units every correlation call was handed (the GPU kernels process the list two at a time: calls must be whole pairs)
    calls, seen = [], []
    orig_windows = bufs.unit_windows
    bufs.unit_windows = lambda u: (seen.append(u), orig_windows(u))[1]
    backend = OracleBackend(plan)
    orig_units = backend.interband_units

    def logged_units(pw, bw):
        calls.append(list(seen[-len(pw):]))
        return orig_units(pw, bw)
    backend.interband_units = logged_units
    raw_pan = synth.pan_strip(64 + rank * plan.pb, plan.pb, W, kb, device="cpu")
    raw_mss = synth.mss_strip(16 + rank * plan.mb, plan.mb, W, kb4, device="cpu")
    o0, o1 = plan.align_out_rows(rank)
    out = torch.zeros(o1 - o0, W // 4, 4, dtype=torch.uint16)
    from opticalimageprocessor_amd.dist import StepTimer
    timer = StepTimer(backend) if timed else None
    cx, cy, rows = default_action_step(backend, plan, bufs, raw_pan, raw_mss, kb, kb4, out, rank,
                                       threshold=THR, timer=timer)
    import json
    np.savez(os.path.join(tmp, "w%d_r%d.npz" % (world, rank)), out=out.numpy(), cx=cx, cy=cy, rows=np.array(rows),
             stages=np.array(json.dumps(timer.result() if timer else {})),
             calls=np.array(sum([c + [-1] for c in calls], []), dtype=np.int64),
             remote=np.array([u for u in plan.units_of(rank) if not plan.unit_is_local(u)]),
             mine=np.array(plan.units_of(rank)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_two_rank_shards_equal_single_process(tmp_path):
    tmp = str(tmp_path)
    _run_rank(0, 1, _free_port(), tmp)                                  # reference: one process, no exchange
    mp.spawn(_run_rank, args=(2, _free_port(), tmp), nprocs=2, join=True)
    one = np.load(os.path.join(tmp, "w1_r0.npz"))
    parts = [np.load(os.path.join(tmp, "w2_r%d.npz" % r)) for r in range(2)]
    # a correlation section straddles the block boundary: its units were computed from received windows,
    # and both ranks computed units
    assert sum(len(p["remote"]) for p in parts) > 0
    assert all(len(p["mine"]) > 0 for p in parts)
    # every rank fitted the same polynomials as the single process, bit for bit
    for p in parts:
        assert np.array_equal(p["cx"], one["cx"]) and np.array_equal(p["cy"], one["cy"])
    assert tuple(parts[0]["rows"])[0] == 0 and parts[0]["rows"][1] == parts[1]["rows"][0]
    whole = np.concatenate([p["out"] for p in parts], 0)
    assert whole.shape == one["out"].shape
    assert np.array_equal(whole, one["out"])


def test_four_rank_shards_equal_single_process(tmp_path):
    """the same strip over FOUR ranks: sections straddle more than one block boundary, the cost model moves pairs between
    ranks, pairs are posted one batch each and computed as they arrive -- still the single-process bits"""
    tmp = str(tmp_path)
    _run_rank(0, 1, _free_port(), tmp)
    mp.spawn(_run_rank, args=(4, _free_port(), tmp, SLICES, W, True), nprocs=4, join=True)
    one = np.load(os.path.join(tmp, "w1_r0.npz"))
    parts = [np.load(os.path.join(tmp, "w4_r%d.npz" % r)) for r in range(4)]
    # the instrumented form of the step (dist.StepTimer; what bench.py --gpus N prints as `multi_gpu`): every stage is booked on
    # every rank, a rank that received pairs lists a wait and a correlation time per pair -- and the results are still the bits
    import json
    for p in parts:
        st = json.loads(str(p["stages"]))
        for k in ("rrc_ms", "exchange_post_ms", "correlate_resident_ms", "exchange_drain_ms", "allgather_ms", "fit_ms", "halo_ms", "align_ms",
                  "step_ms", "correlation_finish_ms", "units_resident", "units_received"):
            assert k in st, (k, st)
        assert st["units_resident"] + st["units_received"] == len(p["mine"])
        assert len(st.get("exchange_wait_ms", [])) == len(st.get("correlate_received_ms", [])) == (st["units_received"] + 1) // 2
        assert abs(st["step_ms"] - sum(v if not isinstance(v, list) else sum(v) for k, v in st.items()
                                       if k.endswith("_ms") and k not in ("step_ms", "correlation_finish_ms"))) < 1e-6
    assert sum(len(p["remote"]) for p in parts) > 0
    assert sorted(np.concatenate([p["mine"] for p in parts]).tolist()) == list(range(SLICES * SECTIONS))
    for p in parts:
        assert np.array_equal(p["cx"], one["cx"]) and np.array_equal(p["cy"], one["cy"])
    assert np.array_equal(np.concatenate([p["out"] for p in parts], 0), one["out"])


def test_odd_slice_count_keeps_the_unit_pairs_whole(tmp_path):
    """with an odd slice count a pair (2k, 2k+1) of correlation units spans two sections, so one unit can be resident
    on its rank while its partner needs lines of another rank.  The GPU kernels process a call's units two at a time and a unit's
    last digits depend on its partner: every call must consist of whole pairs of the single-GPU order, on every rank -- and the
    result must still be the single process's."""
    tmp = str(tmp_path)
    slices, width = 9, 720                                         # 80-column units, as in the 8-slice tests
    _run_rank(0, 1, _free_port(), tmp, slices, width)
    mp.spawn(_run_rank, args=(2, _free_port(), tmp, slices, width), nprocs=2, join=True)
    one = np.load(os.path.join(tmp, "w1_r0.npz"))
    parts = [np.load(os.path.join(tmp, "w2_r%d.npz" % r)) for r in range(2)]
    n_units = slices * SECTIONS
    done = []
    mixed = 0
    for p in parts:
        flat = p["calls"].tolist()
        call = []
        for v in flat:
            if v >= 0:
                call.append(v)
                continue
            assert len(call) > 0
            for j in range(0, len(call), 2):                       # consecutive entries form the kernel's pairs
                a = call[j]
                assert a % 2 == 0, call
                if a + 1 < n_units:
                    assert j + 1 < len(call) and call[j + 1] == a + 1, call
                else:
                    assert j + 1 == len(call), call
            done += call
            call = []
        remote = set(p["remote"].tolist())
        mixed += sum(1 for u in p["mine"].tolist() if u % 2 == 0 and u + 1 < n_units and ((u in remote) != (u + 1 in remote)))
    assert sorted(done) == list(range(n_units))
    assert mixed > 0                                               # the case the test is about does occur in this geometry
    for p in parts:
        assert np.array_equal(p["cx"], one["cx"]) and np.array_equal(p["cy"], one["cy"])
    assert np.array_equal(np.concatenate([p["out"] for p in parts], 0), one["out"])


def test_cost_model_placement():
    """assign_groups_by_cost (dist.py; mirrored in csrc/oip_multigpu.hpp): a pair moves only when that shortens the
    predicted critical path -- bytes / link bandwidth against 2.5 ms of computing per pair."""
    from opticalimageprocessor_amd.dist import StripPlan, assign_groups_by_cost, rank_finish_us
    # a rank's finish time: resident groups first, then the others as they arrive over one link
    assert rank_finish_us([0, 0, 0], 2500, 50000) == 7500
    assert rank_finish_us([240_000_000], 2500, 50000) == 4800 + 2500
    assert rank_finish_us([0, 240_000_000, 240_000_000], 2500, 50000) == max(max(2500, 4800) + 2500, 9600) + 2500
    # BASELINE config 4 (30000 x 524288 on 8 ranks): sections 0, 1, 3, 4 lie inside the blocks of ranks 1, 2, 5, 6 (five
    # pairs each), section 2 straddles the blocks of ranks 3 and 4 (its pairs need lines from the other rank wherever they run).
    # A link too slow to be worth it: no pair of a resident section moves
    slow = StripPlan(30000, 524288, 8, link_gbs=1)
    for sec, home in ((0, 1), (1, 2), (3, 5), (4, 6)):
        assert all(slow.assign[u] == home and slow.unit_is_local(u) for u in range(10 * sec, 10 * sec + 10))
    assert all(not slow.unit_is_local(u) for u in range(20, 30))
    assert {p.unit // 10 for p in slow.correlation_pieces() if p.src != p.dst} == {2}
    # the default link figure (50 GB/s: 4.8 ms per pair moved): pairs move to the idle ranks while that pays
    plan = StripPlan(30000, 524288, 8)
    load = [len(plan.units_of(r)) for r in range(8)]
    assert sum(load) == 50 and all(l % 2 == 0 for l in load) and min(load) >= 2
    assert max(plan.predicted_finish_us) < 5 * 2500
    moved = sorted({p.unit // 2 for p in plan.correlation_pieces() if p.src != p.dst})
    assert 0 < len(moved) <= 12
    # a link that costs nothing: the pairs spread evenly (25 pairs over 8 ranks: three or four each)
    fast = StripPlan(30000, 524288, 8, link_gbs=100000)
    assert sorted(len(fast.units_of(r)) for r in range(8)) == [6, 6, 6, 6, 6, 6, 6, 8]
    # the objective itself: a move is only taken when the sorted finish vector drops
    where, fin = assign_groups_by_cost([[0, 10**12], [0, 10**12], [0, 10**12]], 2, 1000, 1000)
    assert where == [0, 0, 0] and fin == [3000, 0]
    where, fin = assign_groups_by_cost([[0, 1000], [0, 1000], [0, 1000], [0, 1000]], 2, 1000, 1000)
    assert sorted(where) == [0, 0, 1, 1] and max(fin) == 2001


# ---- cross-CCD path (BASELINE config 5): prestitch + stitch over 2 ranks ----------------------------
CW, CL, COV, CSEC, CLPS = 256, 4800, 40, 3, 1200
CSR, CGUARD = 700, 800          # remap section rows / row guard scaled down with the strip


def _run_ccd_rank(rank, world, port, tmp, shift):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from opticalimageprocessor_amd import synth
    from opticalimageprocessor_amd.dist import CcdBuffers, CcdPlan, prestitch_stitch_step
    from _oracle_backend import OracleBackend
    kb1, kb2 = synth.lut(CW, 1), synth.lut(CW, 2)
    plan = CcdPlan(CW, CL, world, CSEC, CLPS, COV, 0, CSR, CGUARD)
    b0, b1 = plan.block(rank)
    raw1, raw2 = synth.ccd_pair(64 + b0, plan.pb, CW, COV, kb1, kb2, device="cpu", shift=shift)
    bufs = CcdBuffers(plan, rank, raw1, raw2)
    prestt = torch.zeros(plan.pb, CW, dtype=torch.uint16)
    stitched = torch.zeros(plan.pb, 2 * (CW - plan.fold), dtype=torch.uint16)
    dx, dy, table = prestitch_stitch_step(OracleBackend(plan), plan, bufs, kb1, kb2, prestt, stitched, rank,
                                          threshold=-1.0, group=None)
    # the fused single-pass form: RRC of CCD 1 and the resampled CCD-2 lines written straight into the stitched raster
    fused = torch.zeros_like(stitched)
    prestitch_stitch_step(OracleBackend(plan), plan, bufs, kb1, kb2, None, fused, rank, threshold=-1.0, group=None, fused=True)
    assert torch.equal(fused, stitched), "fused prestitch -> stitch differs from the three-pass flow"
    np.savez(os.path.join(tmp, "c%d_r%d.npz" % (world, rank)), prestt=prestt.numpy(), stitched=stitched.numpy(),
             shift=np.array([dx, dy]), table=table, halo=np.array([bufs.r2_first, bufs.rrc2.shape[0]]),
             remote=np.array([u for u in plan.units_of(rank) if not plan.unit_is_local(u)]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("shift", [(3, -2), (-2, 3)])
def test_two_rank_prestitch_stitch_equals_single_process(tmp_path, shift):
    """stitcher.h:83-201 + imageop.h:340-351 sharded by scan-line block: section windows gathered across the
    block boundary, all-gathered table -> identical in-order mean, remap with row halo (both signs of dy: the
    halo is above or below the block), stitch.  Bit-equal to one process."""
    tmp = str(tmp_path)
    _run_ccd_rank(0, 1, _free_port(), tmp, shift)
    mp.spawn(_run_ccd_rank, args=(2, _free_port(), tmp, shift), nprocs=2, join=True)
    one = np.load(os.path.join(tmp, "c1_r0.npz"))
    parts = [np.load(os.path.join(tmp, "c2_r%d.npz" % r)) for r in range(2)]
    assert sum(len(p["remote"]) for p in parts) == 1          # the middle section straddles the boundary
    for p in parts:
        assert np.array_equal(p["table"], one["table"]) and np.array_equal(p["shift"], one["shift"])
    # the shift was found (synthetic truth) and at least one rank needed halo lines of the other
    assert abs(one["shift"][0] - shift[0]) < 0.3 and abs(one["shift"][1] - shift[1]) < 0.3
    assert any(int(p["halo"][1]) > CL // 2 for p in parts)
    assert np.array_equal(np.concatenate([p["prestt"] for p in parts], 0), one["prestt"])
    assert np.array_equal(np.concatenate([p["stitched"] for p in parts], 0), one["stitched"])


def test_unit_assignment_balances_and_prefers_local():
    from opticalimageprocessor_amd.dist import CcdPlan, StripPlan
    # BASELINE config 4: 30000 x 524288 on 8 ranks, the reference's 5 sections for the WHOLE strip
    plan = StripPlan(30000, 524288, 8)
    assert plan.pb == 65536 and plan.n_units == 50
    load = [len(plan.units_of(r)) for r in range(8)]
    assert sum(load) == 50 and max(load) == 8 and min(load) >= 2           # 25 pairs over 8 ranks: at most 4 pairs
    assert all(len(plan.units_of(r)) % 2 == 0 for r in range(8))           # pairs stay together
    homes = [plan.owner(s) for s in range(5)]
    local = sum(plan.unit_is_local(u) for u in range(50))
    assert local >= 26                                                     # most units never move (test_cost_model_placement)
    for u in range(50):
        if plan.unit_is_local(u):
            assert plan.assign[u] == homes[u // 10]
    moved = [p for p in plan.correlation_pieces() if p.src != p.dst]
    assert moved and all(p.rows > 0 and p.cols in (3000, 750) for p in moved)
    # every unit's pieces tile its windows exactly
    for u in range(50):
        pan = sorted((p.dst_row, p.rows) for p in plan.unit_pieces(u) if p.kind == "pan")
        assert pan[0][0] == 0 and sum(r for _, r in pan) == 16000
        assert all(pan[i][0] + pan[i][1] == pan[i + 1][0] for i in range(len(pan) - 1))
    # BASELINE config 5: two 30000 x 262144 segments on 8 ranks, 10 sections of 16000 lines
    ccd = CcdPlan(30000, 262144, 8)
    assert ccd.pb == 32768 and ccd.gap == (262144 - 160000) // 11
    load = [len(ccd.units_of(r)) for r in range(8)]
    assert sum(load) == 10 and max(load) == 2
    assert sum(not ccd.unit_is_local(u) for u in range(10)) >= 3           # sections straddle the 32768-line blocks
    for u in range(10):
        a = sorted((p.dst_row, p.rows) for p in ccd.unit_pieces(u) if p.kind == "pan1")
        assert a[0][0] == 0 and sum(r for _, r in a) == 16000
    # weak-scaling variant (5 N sections): everything is local, nothing moves
    for world in (2, 4, 8):
        w = StripPlan(30000, 100000 * world, world, 10, 5 * world)
        assert all(w.unit_is_local(u) for u in range(w.n_units)) and w.correlation_pieces() == []
        assert [len(w.units_of(r)) for r in range(world)] == [50] * world


def test_remap_halo_plan_at_config5():
    """rows each rank must receive for the constant-shift remap at 30000 x 262144 / 8 ranks (host arithmetic)"""
    from opticalimageprocessor_amd.dist import CcdPlan
    plan = CcdPlan(30000, 262144, 8)
    for dy in (-1.62, 2.4):
        tr, need = plan.remap_transfers(lambda a, n: oip_range(a, n, plan.L, dy))
        for r, (f, l) in enumerate(need):
            b0, b1 = plan.block(r)
            assert f <= b0 + 8 and l >= b1 - 8 and f >= 0 and l <= plan.L
            if dy < 0 and r > 0:
                assert f < b0                      # taps reach above the block
        assert all(t.src != t.dst and t.rows > 0 for t in tr)
        # interior ranks exchange only a handful of lines with a neighbour
        assert all(t.rows <= 8 for t in tr if abs(t.src - t.dst) == 1 and t.dst < 7)


def oip_range(a, n, L, dy):
    import opticalimageprocessor_amd as oip
    return oip.remap_shift_src_range(a, n, L, dy, 30000)


def test_plan_geometry():
    from opticalimageprocessor_amd.dist import StripPlan
    plan = StripPlan(30000, 800000, 8)
    # preproc.h:245-247 at 8 x 100000 lines: gap = (800000 - 5*16000)/6
    assert plan.base_gap == 120000 and plan.section(0)[:2] == (120000, 136000)
    owners = [plan.owner(s) for s in range(5)]
    assert owners == [1, 2, 3, 5, 6]
    # ... but the 50 units are spread: no rank computes more than 4 pairs, none idles
    assert sorted(len(plan.units_of(r)) for r in range(8)) == [2, 4, 4, 8, 8, 8, 8, 8] or \
        max(len(plan.units_of(r)) for r in range(8)) == 8
    covered = []
    for r in range(8):
        o0, o1 = plan.align_out_rows(r)
        covered.append((o0, o1))
    assert covered[0][0] == 0 and covered[-1][1] == plan.out_rows
    assert all(covered[i][1] == covered[i + 1][0] for i in range(7))
    with pytest.raises(ValueError):
        StripPlan(30000, 100001, 2)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_weak_scaling_plan_keeps_per_rank_work_fixed(world):
    """bench.py's N-GPU workload: an N x 100000-line strip with 5 N correlation sections.  Every rank must own
    exactly 5 sections (the per-GPU work of the single-GPU run), and at these sizes no section straddles a block
    boundary, so the only exchanges are the align halos and the all-gather of the table."""
    from opticalimageprocessor_amd.dist import StripPlan
    plan = StripPlan(30000, 100000 * world, world, 10, 5 * world)
    owners = [plan.owner(s) for s in range(plan.sections)]
    assert [owners.count(r) for r in range(world)] == [5] * world
    assert plan.correlation_pieces() == []
    assert [len(plan.units_of(r)) for r in range(world)] == [50] * world
    # the aligned image is covered once, in order
    rows = [plan.align_out_rows(r) for r in range(world)]
    assert rows[0][0] == 0 and rows[-1][1] == plan.out_rows
    assert all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))


# ---- the C++ multi-GPU host (csrc/oip_multigpu.hpp, `oip --gpus N`) plans exactly like dist.py -----------------------
OIP = os.path.join(ROOT, "opticalimageprocessor_amd", "lib", "oip")


def _plan_json(args):
    import json
    import subprocess
    r = subprocess.run([OIP, "plan"] + [str(a) for a in args], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return json.loads(r.stdout)


@pytest.mark.parametrize("W,L,world,slices,sections", [(30000, 524288, 8, 10, 5), (30000, 800000, 8, 10, 5), (30000, 200000, 2, 10, 5),
                                                       (12288, 262144, 4, 10, 5), (640, 4800, 2, 8, 3), (30000, 400000, 4, 10, 20)])
def test_cpp_strip_plan_equals_python_plan(W, L, world, slices, sections):
    import opticalimageprocessor_amd as oip
    from opticalimageprocessor_amd.dist import StripPlan
    corr = 16000 if L >= 100000 else 1200
    lps, ovl, minl = (20000, 520, 1500) if L >= 100000 else (700, 60, 100)
    plan = StripPlan(W, L, world, slices, sections, corr, lps, 0, ovl, False, minl)
    cy = np.tile([-1.3, 2e-5, -3e-10], (4, 1))
    got = _plan_json(["strip", "--width", W, "--lines", L, "--gpus", world, "--slices", slices, "--ibc-sections", sections, "--corr-lines", corr,
                      "--lines-section", lps, "--overlap-lines", ovl, "--min-lines", minl, "--cy=-1.3,2e-5,-3e-10"])
    assert got["assign"] == plan.assign
    kinds = {"pan": 0, "mss": 1}
    want = [[p.src, p.dst, kinds[p.kind], p.unit, p.row0, p.rows, p.col0, p.cols, p.dst_row] for p in plan.correlation_pieces()]
    assert got["pieces"] == want
    assert got["align_rows"] == [list(plan.align_out_rows(r)) for r in range(world)]
    tr, _ = plan.align_transfers(lambda a, n: oip.align_mss_src_range(a, n, plan.Lm, cy, W // 4, lps, 0, ovl, False, minl))
    assert got["align_transfers"] == [[t.src, t.dst, t.row0, t.rows] for t in tr]


@pytest.mark.parametrize("W,L,world,sections,lps,ov,dy", [(30000, 262144, 8, 10, 16000, 200, -1.62), (30000, 262144, 8, 10, 16000, 200, 2.4),
                                                          (12288, 400000, 4, 10, 16000, 200, 0.3), (30000, 200000, 2, 10, 16000, 200, -0.2)])
def test_cpp_ccd_plan_equals_python_plan(W, L, world, sections, lps, ov, dy):
    import opticalimageprocessor_amd as oip
    from opticalimageprocessor_amd.dist import CcdPlan
    plan = CcdPlan(W, L, world, sections, lps, ov, 0)
    got = _plan_json(["ccd", "--width", W, "--lines", L, "--gpus", world, "--sections", sections, "--section-lines", lps, "--stitch-overlap", ov,
                      "--dy=%r" % dy])
    assert got["assign"] == plan.assign
    kinds = {"pan1": 0, "pan2": 1}
    want = [[p.src, p.dst, kinds[p.kind], p.unit, p.row0, p.rows, p.col0, p.cols, p.dst_row] for p in plan.correlation_pieces()]
    assert got["pieces"] == want
    tr, need = plan.remap_transfers(lambda a, n: oip.remap_shift_src_range(a, n, L, dy, 30000))
    assert got["remap_transfers"] == [[t.src, t.dst, t.row0, t.rows] for t in tr]
    assert got["need"] == [list(x) for x in need]
