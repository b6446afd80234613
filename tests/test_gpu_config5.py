"""GPU tests for the pieces BASELINE configs 1, 4 and 5 add: the 4096 x 8192 RRC shape (config 1) against the
reference's own compiled loop, the explicit-window correlation entry points the multi-GPU host uses, the
fp16-accumulate resampling variant with its stated tolerance, and the sharded work-flows of
opticalimageprocessor_amd.dist run by two ranks that share the one GPU of the box (gloo, host-staged
transfers) against the single-rank HIP result, bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

# |fp16-accumulate result - fp32 result| bound stated in include/oip_c.h
F16_ABS_DN = 6          # measured on the MI355X: max 5, mean 0.25 DN
F16_REL = 1.0 / 64


def _cuda(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_config1_rrc_4096x8192_matches_the_reference_loop(ctx, oracle_mod):
    """BASELINE config 1: 4096-col x 8192-line single-band strip, RRC only.  The checker here is oracle/_ref --
    the reference's own InplaceRRC lines (imageop.h:129-138) compiled in the build container -- when present,
    the restatement otherwise (the two are proven equal in tests/test_oracle_cpu.py)."""
    import torch
    from opticalimageprocessor_amd import synth
    W, L = 4096, 8192
    kb = synth.lut(W)
    rng = np.random.default_rng(41)
    img = rng.integers(0, 65536, (L, W), dtype=np.uint16)
    want = oracle_mod.rrc_reference(img, kb) if oracle_mod.ref_lib() is not None else oracle_mod.rrc(img, kb)
    src = _cuda(img)
    dst = torch.empty_like(src)
    ctx.rrc_u16(src, dst, W, L, ctx.upload_kb(kb))
    ctx.sync()
    assert np.array_equal(dst.cpu().numpy(), want)
    host = img.copy()
    ctx.rrc_u16_host(host, kb)              # DoRRC4RAW's heap buffer, staged through pinned blocks
    assert np.array_equal(host, want)


@pytest.mark.parametrize("W,L,dx,dy", [(4096, 3000, 3.37, -1.62), (30000, 1500, -2.21, 2.4), (1000, 2100, 0.5, 0.5)])
def test_remap_f16acc_tolerance(ctx, W, L, dx, dy):
    """fp16-accumulate variant vs the fp32 (parity) kernel on 12-bit pushbroom data: |delta| <= 4 DN; the measured
    maximum is printed and recorded in DESIGN.md.  Border lines / irregular columns are computed in fp32 by both."""
    import torch
    from opticalimageprocessor_amd import synth
    kb = np.stack([np.ones(W), np.zeros(W)], 1)
    src = synth.pan_strip(64, L, W, kb, device="cuda")
    a, b = torch.zeros_like(src), torch.zeros_like(src)
    ctx.remap_shift_bicubic_u16(src, a, W, L, dx, dy, 700, 800)
    ctx.remap_shift_bicubic_u16(src, b, W, L, dx, dy, 700, 800, f16acc=True)
    ctx.sync()
    d = np.abs(a.cpu().numpy().astype(np.int32) - b.cpu().numpy().astype(np.int32))
    print("\nf16acc vs fp32 at %dx%d: max |delta| %d DN, mean %.3f DN, %.1f %% of pixels differ" %
          (W, L, d.max(), d.mean(), 100.0 * (d > 0).mean()))
    assert d.max() <= F16_ABS_DN
    assert (d > 0).any()                    # it really is another arithmetic


def test_remap_f16acc_general_bound_and_fallback(ctx):
    """15-bit white noise (the worst case for cancellation): |delta| <= 6 + max|sample - 2048| / 64; a width that is
    not a multiple of 8 falls back to fp32 exactly"""
    import torch
    rng = np.random.default_rng(5)
    W, L = 2048, 1700
    img = rng.integers(0, 32768, (L, W), dtype=np.uint16)
    src = _cuda(img)
    a, b = torch.zeros_like(src), torch.zeros_like(src)
    ctx.remap_shift_bicubic_u16(src, a, W, L, 1.3, -0.7, 700, 800)
    ctx.remap_shift_bicubic_u16(src, b, W, L, 1.3, -0.7, 700, 800, f16acc=True)
    ctx.sync()
    d = np.abs(a.cpu().numpy().astype(np.int32) - b.cpu().numpy().astype(np.int32))
    print("\nf16acc vs fp32 on 15-bit white noise: max |delta| %d DN, mean %.2f DN" % (d.max(), d.mean()))
    assert d.max() <= F16_ABS_DN + F16_REL * 32768, d.max()
    W2 = 1001
    src2 = _cuda(img[:, :W2])
    a2, b2 = torch.zeros_like(src2), torch.zeros_like(src2)
    ctx.remap_shift_bicubic_u16(src2, a2, W2, L, 1.3, -0.7, 700, 800)
    ctx.remap_shift_bicubic_u16(src2, b2, W2, L, 1.3, -0.7, 700, 800, f16acc=True)
    ctx.sync()
    assert np.array_equal(a2.cpu().numpy(), b2.cpu().numpy())


def test_unit_window_entries_equal_the_raster_entries(ctx):
    """oip_interband_correlate_units / oip_stt_correlate_windows on compact copies of the windows (another pitch,
    another order) give the bits of oip_interband_correlate / oip_stt_correlate on the resident rasters."""
    import torch
    from opticalimageprocessor_amd import synth
    W, Lp, slices, sections, corr = 2560, 6400, 8, 2, 2400
    kb = np.stack([np.ones(W), np.zeros(W)], 1)
    pan = synth.pan_strip(64, Lp, W, kb, device="cuda")
    bil = synth.mss_strip(16, Lp // 4, W, kb, device="cuda")
    Wb, Lm = W // 4, Lp // 4
    planes = torch.zeros(4, Lm, Wb, dtype=torch.uint16, device="cuda")
    ctx.mss_split_rrc_u16(bil, planes, Lm * Wb, W, Lm, None)
    whole = ctx.interband_correlate(pan, Lp, 0, Lp, planes, Lm * Wb, 0, Lm, W, slices, sections, corr)
    gap = (Lp - corr * sections) // (sections + 1)
    bc, brows, bcols = W // slices, corr // 4, W // slices // 4
    # Units ride two at a time through shared transforms (the fourth bands of a pair share one complex FFT), so a
    # unit's last digits depend on its partner: bit-equality holds for the SAME pairs -- which the multi-GPU plan
    # keeps (assign_groups_by_cost places whole pairs) -- here (0,0)+(0,1) and (1,6)+(1,7) as in the raster call; the fifth unit
    # runs alone: its third and fourth band (one inverse transform carries both surfaces) may differ in the last digits.
    order = [(0, 0), (0, 1), (1, 6), (1, 7), (0, 4)]
    pans, bands = [], []
    for k, (sec, i) in enumerate(order):
        p0 = gap + sec * (corr + gap)
        m0 = gap // 4 + sec * (brows + gap // 4)
        if k % 2:       # compact copies
            pans.append(pan[p0:p0 + corr, i * bc:(i + 1) * bc].contiguous())
            bands.append([planes[b, m0:m0 + brows, i * bcols:(i + 1) * bcols].contiguous() for b in range(4)])
        else:           # views of the rasters
            pans.append(pan[p0:p0 + corr, i * bc:(i + 1) * bc])
            bands.append([planes[b, m0:m0 + brows, i * bcols:(i + 1) * bcols] for b in range(4)])
    got = ctx.interband_correlate_units([t.data_ptr() for t in pans], [t.stride(0) for t in pans],
                                        [[t.data_ptr() for t in u] for u in bands], [u[0].stride(0) for u in bands],
                                        corr, bc)
    for k, (sec, i) in enumerate(order):
        want = whole[:, sec * slices + i, :3]
        if k < 4:
            assert np.array_equal(got[k], want), (k, got[k], want)
        else:
            assert np.array_equal(got[k][:2], want[:2]) and np.abs(got[k][2:] - want[2:]).max() < 1e-5, (got[k], want)
    # CCD windows
    OV, nsec, lps = 200, 3, 1600
    kb1, kb2 = synth.lut(W, 1), synth.lut(W, 2)
    p1, p2 = synth.ccd_pair(64, Lp, W, OV, kb1, kb2, device="cuda")
    t = ctx.stt_correlate(p1, p2, W, Lp, 0, Lp, nsec, lps, OV, 4)
    g = (Lp - nsec * lps) // (nsec + 1)
    aw = [p1[g + s * (g + lps):g + s * (g + lps) + lps, W - OV:W - 4] for s in range(nsec)]
    bw = [p2[g + s * (g + lps):g + s * (g + lps) + lps, 4:OV] for s in range(nsec)]
    aw[1], bw[1] = aw[1].contiguous(), bw[1].contiguous()
    got = ctx.stt_correlate_windows([x.data_ptr() for x in aw], [x.stride(0) for x in aw],
                                    [x.data_ptr() for x in bw], [x.stride(0) for x in bw], lps, OV - 4)
    assert np.array_equal(got, t)


def test_many_units_exceed_the_old_result_buffer(ctx):
    """more than 64 KiB of results in one call (the result scratch grows)"""
    import torch
    rng = np.random.default_rng(9)
    rows, cols, n = 64, 64, 700                     # 700 x 12 doubles = 67 KB
    pan = _cuda(rng.integers(64, 4096, (rows, cols), dtype=np.uint16))
    band = _cuda(rng.integers(64, 4096, (rows // 4, cols // 4), dtype=np.uint16))
    got = ctx.interband_correlate_units([pan.data_ptr()] * n, [cols] * n, [[band.data_ptr()] * 4] * n, [cols // 4] * n,
                                        rows, cols)
    assert got.shape == (n, 4, 3) and np.isfinite(got).all()
    # same unit, same role in its pair (first / second) -> same bits
    assert np.array_equal(got[0], got[-2]) and np.array_equal(got[0], got[n // 2])
    assert np.array_equal(got[1], got[-1]) and np.abs(got[0] - got[1]).max() < 1e-4


# ---- the sharded work-flows, two ranks on the one GPU ----------------------------------------------
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


W4, LP4, SL4, SEC4, CORR4 = 2560, 19200, 8, 3, 4000
ALIGN4 = dict(lines_per_section=1400, line_offset=0, overlap=120, keep_leading=False, min_lines=200)
CW, CL, COV, CSEC, CLPS, CSR, CGUARD = 2048, 9600, 200, 3, 2400, 1400, 1600


def _rank_main(rank, world, port, tmp, what):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    import opticalimageprocessor_amd as oip
    from opticalimageprocessor_amd import synth
    from opticalimageprocessor_amd.dist import (CcdBuffers, CcdPlan, HipBackend, ShardBuffers, StripPlan,
                                                default_action_step, prestitch_stitch_step)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    ctx = oip.Context(0)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream)
    dev = torch.device("cuda", 0)
    if what == "default":
        kb = synth.lut(W4)
        kb4 = np.concatenate([synth.lut(W4 // 4, 10 + b) for b in range(4)], 0)
        plan = StripPlan(W4, LP4, world, SL4, SEC4, CORR4, halo_cap=16, **ALIGN4)
        bufs = ShardBuffers(plan, rank, dev)
        raw_pan = synth.pan_strip(64 + rank * plan.pb, plan.pb, W4, kb, device=dev)
        raw_mss = synth.mss_strip(16 + rank * plan.mb, plan.mb, W4, kb4, device=dev)
        o0, o1 = plan.align_out_rows(rank)
        out = torch.zeros(o1 - o0, W4 // 4, 4, dtype=torch.uint16, device=dev)
        cx, cy, rows = default_action_step(HipBackend(ctx, plan), plan, bufs, raw_pan, raw_mss, ctx.upload_kb(kb),
                                           ctx.upload_kb(kb4), out, rank, threshold=0.05)
        ctx.sync()
        np.savez(os.path.join(tmp, "d%d_r%d.npz" % (world, rank)), out=out.cpu().numpy(), cx=cx, cy=cy,
                 remote=np.array([u for u in plan.units_of(rank) if not plan.unit_is_local(u)]))
    else:
        kb1, kb2 = synth.lut(CW, 1), synth.lut(CW, 2)
        plan = CcdPlan(CW, CL, world, CSEC, CLPS, COV, 0, CSR, CGUARD)
        b0, _ = plan.block(rank)
        raw1, raw2 = synth.ccd_pair(64 + b0, plan.pb, CW, COV, kb1, kb2, device=dev)
        bufs = CcdBuffers(plan, rank, raw1, raw2)
        prestt = torch.zeros(plan.pb, CW, dtype=torch.uint16, device=dev)
        stitched = torch.zeros(plan.pb, 2 * (CW - plan.fold), dtype=torch.uint16, device=dev)
        dx, dy, table = prestitch_stitch_step(HipBackend(ctx, plan), plan, bufs, ctx.upload_kb(kb1), ctx.upload_kb(kb2),
                                              prestt, stitched, rank, threshold=0.05, f16acc=(what == "ccd16"))
        ctx.sync()
        # the fused single-pass form (oip_rrc_u16_window + oip_remap_shift_bicubic_u16_window): same bits in `stitched`
        fused = torch.zeros_like(stitched)
        prestitch_stitch_step(HipBackend(ctx, plan), plan, bufs, ctx.upload_kb(kb1), ctx.upload_kb(kb2), None, fused, rank,
                              threshold=0.05, f16acc=(what == "ccd16"), fused=True)
        ctx.sync()
        assert torch.equal(fused, stitched), "fused prestitch -> stitch differs from the three-pass flow"
        np.savez(os.path.join(tmp, "%s%d_r%d.npz" % (what, world, rank)), prestt=prestt.cpu().numpy(),
                 stitched=stitched.cpu().numpy(), shift=np.array([dx, dy]), table=table)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def _spawn(world, tmp, what):
    import torch.multiprocessing as mp
    mp.spawn(_rank_main, args=(world, _free_port(), tmp, what), nprocs=world, join=True)


def test_two_ranks_default_action_equals_one_rank_on_the_gpu(tmp_path):
    tmp = str(tmp_path)
    _spawn(1, tmp, "default")
    _spawn(2, tmp, "default")
    one = np.load(os.path.join(tmp, "d1_r0.npz"))
    parts = [np.load(os.path.join(tmp, "d2_r%d.npz" % r)) for r in range(2)]
    assert sum(len(p["remote"]) for p in parts) > 0           # windows did cross the block boundary
    for p in parts:
        assert np.array_equal(p["cx"], one["cx"]) and np.array_equal(p["cy"], one["cy"])
    assert np.array_equal(np.concatenate([p["out"] for p in parts], 0), one["out"])


@pytest.mark.parametrize("what", ["ccd", "ccd16"])
def test_two_ranks_prestitch_stitch_equals_one_rank_on_the_gpu(tmp_path, what):
    tmp = str(tmp_path)
    _spawn(1, tmp, what)
    _spawn(2, tmp, what)
    one = np.load(os.path.join(tmp, "%s1_r0.npz" % what))
    parts = [np.load(os.path.join(tmp, "%s2_r%d.npz" % (what, r))) for r in range(2)]
    for p in parts:
        assert np.array_equal(p["table"], one["table"]) and np.array_equal(p["shift"], one["shift"])
    assert abs(one["shift"][0] - 3) < 0.3 and abs(one["shift"][1] + 2) < 0.3
    assert np.array_equal(np.concatenate([p["prestt"] for p in parts], 0), one["prestt"])
    assert np.array_equal(np.concatenate([p["stitched"] for p in parts], 0), one["stitched"])


def test_bench_starts_its_own_ranks_and_reports_stage_times():
    """`python bench.py --gpus 2` as the driver calls it -- no launcher, WORLD_SIZE unset: the parent (which never touches the GPU)
    starts the two ranks itself and relays rank 0's line.  On this one-GPU box the ranks share the card and
    gloo stages the transfers through the host (OIP_BENCH_BACKEND=gloo): a rehearsal of the plumbing, not a measurement -- the
    line must say so, carry the per-rank stage times of one instrumented step next to the placement model's prediction
    (`config.multi_gpu`, item 6) and stay under 8 KB."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OIP_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--lines", "65536", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 8192, (len(lines), [len(x) for x in lines])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    mg = line["config"]["multi_gpu"]
    assert "rehearsal" in mg["mode"] and len(mg["predicted_correlation_finish_us"]) == 2 and len(mg["measured_correlation_finish_us"]) == 2
    for k in ("rrc_ms", "correlate_resident_ms", "allgather_ms", "halo_ms", "align_ms", "step_ms"):
        assert len(mg["measured_ms"][k]) == 2, k
    assert all(v > 0 for v in mg["measured_ms"]["step_ms"])
