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


def _inputs():
    from opticalimageprocessor_amd import synth
    kb = synth.lut(W)
    kb4 = np.concatenate([synth.lut(W // 4, 10 + b) for b in range(4)], 0)
    return kb, kb4


def _run_rank(rank, world, port, tmp):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from opticalimageprocessor_amd import synth
    from opticalimageprocessor_amd.dist import ShardBuffers, StripPlan, default_action_step
    from _oracle_backend import OracleBackend
    kb, kb4 = _inputs()
    plan = StripPlan(W, LP, world, SLICES, SECTIONS, CORR, halo_cap=16, **ALIGN)
    bufs = ShardBuffers(plan, rank, "cpu")
    raw_pan = synth.pan_strip(64 + rank * plan.pb, plan.pb, W, kb, device="cpu")
    raw_mss = synth.mss_strip(16 + rank * plan.mb, plan.mb, W, kb4, device="cpu")
    o0, o1 = plan.align_out_rows(rank)
    out = torch.zeros(o1 - o0, W // 4, 4, dtype=torch.uint16)
    cx, cy, rows = default_action_step(OracleBackend(plan), plan, bufs, raw_pan, raw_mss, kb, kb4, out, rank,
                                       threshold=THR)
    np.savez(os.path.join(tmp, "w%d_r%d.npz" % (world, rank)), out=out.numpy(), cx=cx, cy=cy, rows=np.array(rows),
             tail=plan.pan_tail(rank))
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
    # a correlation section straddles the block boundary: rank 0 had to receive PAN lines
    assert int(parts[0]["tail"]) > 0
    # every rank fitted the same polynomials as the single process, bit for bit
    for p in parts:
        assert np.array_equal(p["cx"], one["cx"]) and np.array_equal(p["cy"], one["cy"])
    assert tuple(parts[0]["rows"])[0] == 0 and parts[0]["rows"][1] == parts[1]["rows"][0]
    whole = np.concatenate([p["out"] for p in parts], 0)
    assert whole.shape == one["out"].shape
    assert np.array_equal(whole, one["out"])


def test_plan_geometry():
    from opticalimageprocessor_amd.dist import StripPlan
    plan = StripPlan(30000, 800000, 8)
    # preproc.h:245-247 at 8 x 100000 lines: gap = (800000 - 5*16000)/6
    assert plan.base_gap == 120000 and plan.section(0)[:2] == (120000, 136000)
    owners = [plan.owner(s) for s in range(5)]
    assert owners == [1, 2, 3, 5, 6]
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
    assert plan.correlation_transfers() == []
    assert all(plan.pan_tail(r) == 0 for r in range(world))
    # the aligned image is covered once, in order
    rows = [plan.align_out_rows(r) for r in range(world)]
    assert rows[0][0] == 0 and rows[-1][1] == plan.out_rows
    assert all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))
    # the results of all sections fit the library's result buffer (64 KiB of doubles, 12 per unit)
    assert plan.slices * plan.sections * 12 * 8 <= 65536
