#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X OpticalImageProcessor hot path.

Metric (BASELINE.json): Mpix/s end-to-end RRC + align/stitch on a 30000 x 100000 x 4 strip,
with the dominant kernel's fraction of the HBM roofline.

One "step" = one pass of the reference's default action (main.cpp:288-317, with --do-rrc4pan)
over one synthetic 4-band strip already resident in HBM:
    RRC of PAN (30000 x L)                          IMO::InplaceRRC           imageop.h:129
    BIL split + RRC of the 4 MSS bands (7500 x L/4) LoadMSS + DoRRC4MSS       preproc.h:56,202
    inter-band phase correlation, 5 x 10 x 4 units  CalcInterBandCorrelation  preproc.h:224
    shift filter + polynomial fit (host)            preproc.h:492-550
    4-band bicubic alignment -> 16UC4               DoInterBandAlignment      preproc.h:351
pixels per step = PAN pixels + MSS pixels of all bands = 1.25 * W * L.

N GPUs (`--gpus N`, one rank per GPU under torch.distributed.run): every rank owns a `--lines`-line block
(default 100000, the metric's strip; `--lines 65536 --gpus 8` is BASELINE config 4 exactly) of an N x lines
strip (scan-line blocks).  The strip carries the reference's 5 correlation sections for the WHOLE
strip (preproc.h:245-247: the number of windows does not grow with the strip), their 50 (section, slice)
units are dealt to the ranks and lines a unit's rank lacks move point-to-point over RCCL; results are
all-gathered; align halos move point-to-point.  Pixels per GPU are fixed ("scaling": "weak"); the
correlation stage is a fixed-size job that the ranks share.  `--workload weak5n` keeps the PER-GPU
correlation work fixed instead (5 N sections; round-1's variant).  `--workload prestitch --gpus N` is the
cross-CCD path (BASELINE config 5; `--fp16-accumulate` for its resampling variant).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=30000)
    ap.add_argument("--lines", type=int, default=100000, help="PAN lines per GPU")
    ap.add_argument("--slices", type=int, default=10)
    ap.add_argument("--sections", type=int, default=5)
    ap.add_argument("--ibc-threshold", type=float, default=0.4)
    ap.add_argument("--workload", choices=["default", "weak5n", "rrc", "prestitch"], default="default",
                    help="default: the full 4-band path (5 sections for the whole strip); weak5n: the same with 5 N "
                         "sections (per-GPU correlation work fixed); rrc: BASELINE config 2 (RRC kernel only, "
                         "30000x65536); prestitch: cross-CCD path (BASELINE config 5): CalcSttParameters + RRC x2 + "
                         "PreStitch remap + RAW stitch of two CCD segments")
    ap.add_argument("--fp16-accumulate", action="store_true",
                    help="prestitch: the fp16-accumulate resampling variant (not the parity mode)")
    ap.add_argument("--fused", action="store_true",
                    help="prestitch: RRC of CCD 1 and the resampled CCD-2 lines written straight into the stitched raster "
                         "(oip_rrc_u16_window + oip_remap_shift_bicubic_u16_window; same stitched bits, .RRC.RAW of CCD 1 and "
                         ".RRC.PRESTT.RAW not materialised)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the PCIe-inclusive pass from pageable host buffers")
    ap.add_argument("--no-cli", action="store_true", help="skip the timed run of the `oip` executable on files in tmpfs (the `cli` object)")
    ap.add_argument("--full-record", default=None,
                    help="where the verbose record goes (default gpurun_out/bench_full.json); stdout carries the compact line")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the short legs for the other single-GPU BASELINE configs (2, 3, 5 at N=1, 12288-wide) that "
                         "the default one-GPU run attaches as `configs`")
    return ap.parse_args()


# algorithmic HBM bytes per launch of each kernel, keyed by the name the
# library's profiler and rocprofv3 both report
def algorithmic_bytes(W, pb, M, N, base_rows, base_cols, out_rows_local, rows_arrays=5.0):
    mb, Wb = pb // 4, W // 4
    MN = M * N
    win = base_rows * base_cols
    d = {
        "rrc_u16_kernel": 4.0 * W * pb,                       # 2 B read + 2 B written per pixel
        "rrc_u16_flat_kernel": 4.0 * W * pb,
        "mss_split_rrc_kernel": 4.0 * W * mb,
        "cross_power_kernel": 8.0 * MN * 3.5,                 # 2 or 3 spectra in, 1 out
        "xpower_rows_kernel": 8.0 * MN * 3.5,                 # same traffic, the inverse row FFT rides along
        # whole row stage (forward rows + cross-power + inverse rows): every spectrum of the job read
        # once, every output written once -- 3 in + 2 out for a PAN-vs-4-bands unit, 1 + 1 for a CCD pair
        "corr_rows_kernel": 8.0 * MN * rows_arrays,
        # row stage of a pair of units with the up-sampling applied to the band spectra: PAN_A + i PAN_B in (1 array),
        # every line of the four band transforms read once per frequency line it serves (4 x 1/4 array), four outputs
        "corr_rows_up_kernel": 8.0 * MN * 6.0,
        # row stage of ONE unit of the 1250-point geometry (12288-wide strips), vertical up-sampling on the spectra: the PAN array,
        # two quarter-height band arrays in, two outputs
        "corr_rows_v_kernel": 2 * 8.0 * MN * 3.5,                 # (the profiler scope spans the two launches of a pair of units)
        # horizontal taps of that geometry: the four u16 band windows of a unit in, two bands per complex value out ((M/4) x 4N)
        "hpack_bands_kernel": 4 * 2.0 * win / 16.0 + 8.0 * (M / 4.0) * (4.0 * N),
        # column passes of one quarter-width band array (two vertically up-sampled f32 images in): OIP_SPECTRAL_UP=1
        "fft_pass_ct_kernel_F128_pack_quarter": 8.0 * MN / 4.0 + 2 * 4.0 * win / 4.0,
        "fft_pass_ct_kernel_F125_quarter": 16.0 * MN / 4.0,
        # the eight band windows of a pair of units as one complex array (u16 in, float2 out) and its two column passes
        "pack_bands_kernel": 8 * 2.0 * win / 16.0 + 8.0 * MN / 4.0,
        "fft_pass_ct_kernel_F32_band": 16.0 * MN / 4.0,
        "fft_pass_ct_kernel_F125_band": 16.0 * MN / 4.0,
        # vertical half of the x4 up-sampling, the eight bands of a pair of units per launch: u16 band
        # window in (win/16 px), f32 rows x4 out (win/4 px)
        "resize_cubic_v_kernel": 8 * (4.0 * win / 4.0 + 2.0 * win / 16.0),
        # first forward pass of the two PAN windows of a pair of units (u16 in, one complex array out)
        "fft_pass_ct_kernel_F128_pack": 8.0 * MN + 2 * 2.0 * win,
        # last inverse pass: reads the array, keeps only per-tile maxima
        "fft_pass_ct_kernel_F128_peak": 8.0 * MN,
        "resize_cubic_kernel": 4.0 * win + 2.0 * win / 16.0,  # u16 window in, x4 f32 out
        "align_mss_kernel": 16.0 * Wb * out_rows_local,       # 4 x 2 B read + 8 B written per pixel
        "remap_shift_kernel": 4.0 * W * pb,                   # 2 B read + 2 B written per pixel
        "remap_shift8_kernel": 4.0 * W * pb,
        "remap_shift8_f16_kernel": 4.0 * W * pb,
        "stitch_rows_kernel": 4.0 * 2 * (W - 100) * pb,       # 2 B read + 2 B written per output pixel
        "rrc_u16_window_kernel": 4.0 * (W - 100) * pb,        # the left half of the stitched raster straight from raw CCD 1
        # raw CCD 2 in (corrected on load), the right half of the stitched raster out: 2 B read per source pixel of the columns
        # that are stored + 2 B written
        "remap_shift8_rrc_kernel": 4.0 * (W - 100) * pb,
        "remap_shift8_rrc_f16_kernel": 4.0 * (W - 100) * pb,
    }
    return d


def fft_pass_bytes(name, M, N, base_rows, base_cols):
    """algorithmic bytes of one FFT pass launch: 8 B read + 8 B written per point.  The first
    forward pass reads the real windows instead, the last inverse pass stores nothing; both
    are launched under the same kernel name, so the per-name figure is the plain 16 B/point
    (an upper bound on what the fused launches move)."""
    if name.startswith("fft_pass"):
        return 16.0 * M * N
    return None


def optimal_dft_size(n):
    best = None
    p5 = 1
    while p5 < 2 * n + 1:
        p35 = p5
        while p35 < 2 * n + 1:
            v = p35
            while v < n:
                v *= 2
            best = v if best is None or v < best else best
            p35 *= 3
        p5 *= 5
    return best


def _cores():
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, min(cores, 64))              # one socket's worth at most


_RATES = {}


def cpu_rates(W, base_rows, base_cols):
    """Per-unit costs of the oracle (CPU restatement; RRC leg through the reference's own loop when oracle/_ref is
    present) on this host, each measured once per (W, unit shape) on a bounded sample: seconds per PAN pixel of RRC,
    per BIL line of split + RRC, per (unit, band) phase correlation, per aligned MSS line -- on ONE core and on ALL the
    cores this process may use (one socket's worth at most).  Every leg is embarrassingly parallel over
    lines or units, so the all-core figure runs independent samples on independent workers: threads where the work is C
    code that drops the GIL, spawned processes for the numpy FFT correlation."""
    key = (W, base_rows, base_cols)
    if key in _RATES:
        return _RATES[key]
    import concurrent.futures as cf
    import multiprocessing as mp
    import oracle
    from oracle import phasecorr as pc
    from opticalimageprocessor_amd import synth
    t_all = time.time()
    cores = _cores()
    rng = np.random.default_rng(1)
    kb = synth.lut(W)
    r = {"cores": cores}
    # 1. PAN RRC on W x hs: 1 thread (the reference as shipped), then row-parallel
    hs = max(4096, min(16384, (1 << 29) // W // 4096 * 4096))
    img = rng.integers(64, 4096, (hs, W), dtype=np.uint16)
    use_ref = oracle.ref_lib() is not None
    f = oracle.rrc_reference if use_ref else oracle.rrc
    best = 1e9
    for _ in range(3):
        t = time.time(); f(img, kb); best = min(best, time.time() - t)
    best_mt = 1e9
    for _ in range(3):
        t = time.time(); oracle.rrc(img, kb, threads=cores); best_mt = min(best_mt, time.time() - t)
    r["rrc_s_per_px"] = (best / (W * hs), best_mt / (W * hs))
    r["rrc_via"] = "oracle/_ref (the reference's own InplaceRRC)" if use_ref else "the restatement"
    r["rrc_sample"] = "%dx%d" % (W, hs)
    # 2. MSS split + RRC on W x 4096 BIL lines; all cores: one 4096-line chunk per worker thread
    ms = 4096
    bil = rng.integers(64, 4096, (ms, W), dtype=np.uint16)
    kb4 = [synth.lut(W // 4, 10 + b) for b in range(4)]

    def mss_chunk(_):
        bands = oracle.split_mss(bil)
        return [oracle.rrc(b, k) for b, k in zip(bands, kb4)]
    t = time.time(); mss_chunk(0); m1 = (time.time() - t) / ms
    with cf.ThreadPoolExecutor(cores) as ex:
        t = time.time(); list(ex.map(mss_chunk, range(cores))); mn = (time.time() - t) / cores / ms
    r["mss_s_per_line"] = (m1, mn)
    # 3. correlation: one (section, slice, band) at full size per worker, spawned processes (numpy FFT)
    t = time.time(); pc.bench_unit_band(7, base_rows, base_cols); t_ub = time.time() - t
    c1 = cn = t_ub
    per_worker_gb = 2.5 * base_rows * base_cols / 48e6          # ~2.5 GB per worker at 16000 x 3000
    workers = max(1, min(cores, 32, int(160 / max(per_worker_gb, 0.1))))
    if workers > 1:
        ctx_mp = mp.get_context("spawn")
        env_path = os.environ.get("PYTHONPATH", "")
        os.environ["PYTHONPATH"] = ROOT + (os.pathsep + env_path if env_path else "")
        try:
            with cf.ProcessPoolExecutor(workers, mp_context=ctx_mp) as ex:
                list(ex.map(pc.bench_unit_band, range(workers), [64] * workers, [64] * workers))       # start the workers
                t = time.time()
                list(ex.map(pc.bench_unit_band, range(workers), [base_rows] * workers, [base_cols] * workers))
                cn = (time.time() - t) / workers
        finally:
            os.environ["PYTHONPATH"] = env_path
    r["corr_s_per_unit_band"] = (c1, cn)
    r["corr_workers"] = workers
    # 4. align on 4096 MSS lines; all cores: one chunk per worker thread (the C restatement drops the GIL)
    al = 4096
    bands = [rng.integers(64, 4096, (al, W // 4), dtype=np.uint16) for _ in range(4)]
    cx = np.tile([2.0, 1e-5], (4, 1)); cy = np.tile([-1.0, 1e-5, -1e-10], (4, 1))

    def align_chunk(_):
        return oracle.align_mss(bands, cx, cy, 20000, 0, 520, False, 1500)[1]
    t = time.time(); align_chunk(0); a1 = (time.time() - t) / (al - 520)
    with cf.ThreadPoolExecutor(cores) as ex:
        t = time.time(); list(ex.map(align_chunk, range(cores))); an = (time.time() - t) / cores / (al - 520)
    r["align_s_per_line"] = (a1, an)
    r["wall_s"] = time.time() - t_all
    r["unit"] = (base_rows, base_cols)
    _RATES[key] = r
    return r


def cpu_baseline(W, L, slices, sections):
    """The oracle's rates (cpu_rates) scaled to one step of the default action on a W x L strip."""
    base_rows, base_cols = min(L, 16000), W // slices
    r = cpu_rates(W, base_rows, base_cols)
    nub = slices * sections * 4
    tot = []
    stages = []
    for k in (0, 1):
        st = {"rrc_pan": r["rrc_s_per_px"][k] * W * L, "mss_split_rrc": r["mss_s_per_line"][k] * (L // 4),
              "correlation": r["corr_s_per_unit_band"][k] * nub, "align": r["align_s_per_line"][k] * (L // 4 - 520)}
        stages.append(st)
        tot.append(sum(st.values()))
    mpix = 1.25 * W * L / 1e6
    return {
        "value": mpix / tot[1], "unit": "Mpix/s", "cores": r["cores"], "kind": "port",
        "sample": ("oracle (CPU restatement) on all %d cores this process may use, timed per stage on a bounded sample and scaled "
                   "to the step: PAN RRC %s via %s (row-parallel), MSS split+RRC and align one 4096-line chunk per thread, "
                   "correlation one full-size (%dx%d) unit-band per worker process x %d workers (numpy float64 FFT); "
                   "%.1f s of wall time" % (r["cores"], r["rrc_sample"], r["rrc_via"], base_rows, base_cols, r["corr_workers"], r["wall_s"])),
        "stage_seconds_full_step": stages[1],
        "one_core": {"value": mpix / tot[0], "unit": "Mpix/s", "stage_seconds_full_step": stages[0]},
        "rrc_reference_1thread_Mpix_s": 1e-6 / r["rrc_s_per_px"][0],
        "rrc_all_cores_Mpix_s": 1e-6 / r["rrc_s_per_px"][1], "rrc_all_cores": r["cores"],
    }


def cpu_baseline_prestitch(W, L, sections, overlap=200):
    """The oracle on the cross-CCD path (CalcSttParameters on 16000 x overlap windows, RRC of both CCDs, SectionaryRemap
    bicubic, StitchBigRaw), timed on bounded samples and scaled to one step over two W x L segments; one core and all
    cores (one sample per thread: every stage is line-parallel, the C restatement drops the GIL)."""
    import concurrent.futures as cf
    import oracle
    from oracle import phasecorr as pc
    from opticalimageprocessor_amd import synth
    t_all = time.time()
    cores = _cores()
    rng = np.random.default_rng(4)
    base = cpu_rates(W, min(L, 16000), W // 10) if (W, min(L, 16000), W // 10) in _RATES else None
    if base is None:
        kb = synth.lut(W)
        img = rng.integers(64, 4096, (4096, W), dtype=np.uint16)
        f = oracle.rrc_reference if oracle.ref_lib() is not None else oracle.rrc
        t = time.time(); f(img, kb); r1 = (time.time() - t) / img.size
        t = time.time(); oracle.rrc(img, kb, threads=cores); rn = (time.time() - t) / img.size
        rrc = (r1, rn)
    else:
        rrc = base["rrc_s_per_px"]
    # correlation of one section: 16000 x overlap, float64 numpy FFT (single process: the sections are few and short)
    a = rng.integers(64, 4096, (16000, overlap)).astype(np.float32)
    b = np.roll(a, (2, 3), (0, 1))
    t = time.time(); pc.phase_correlate(a, b); c1 = time.time() - t
    # remap: SectionaryRemap needs more than 32767 lines (imageop.h:242-244); a narrower strip keeps the sample bounded
    ws, ls = 2048, 32768
    src = rng.integers(64, 4096, (ls, ws), dtype=np.uint16)
    t = time.time(); oracle.prestitch(src, 2.37, -1.62); m1 = (time.time() - t) / src.size
    nthr = min(cores, 16)
    with cf.ThreadPoolExecutor(nthr) as ex:
        t = time.time(); list(ex.map(lambda _: oracle.prestitch(src, 2.37, -1.62), range(nthr))); mn = (time.time() - t) / nthr / src.size
    mn *= nthr / cores if cores > nthr else 1.0               # perfect line-parallel scaling beyond the threads tried: generous to the CPU
    left = rng.integers(64, 4096, (8192, 4096), dtype=np.uint16)
    t = time.time(); oracle.stitch_raw(left, left, overlap // 2); s1 = (time.time() - t) / (2 * (4096 - overlap // 2) * 8192)
    with cf.ThreadPoolExecutor(nthr) as ex:
        t = time.time(); list(ex.map(lambda _: oracle.stitch_raw(left, left, overlap // 2), range(nthr)))
        sn = (time.time() - t) / nthr / (2 * (4096 - overlap // 2) * 8192)
    sn *= nthr / cores if cores > nthr else 1.0
    out_px = 2 * (W - overlap // 2) * L
    st1 = {"correlation": c1 * sections, "rrc_x2": rrc[0] * 2 * W * L, "remap": m1 * W * L, "stitch": s1 * out_px}
    stn = {"correlation": c1 * sections / min(cores, sections), "rrc_x2": rrc[1] * 2 * W * L, "remap": mn * W * L, "stitch": sn * out_px}
    mpix = 2.0 * W * L / 1e6
    return {"value": mpix / sum(stn.values()), "unit": "Mpix/s", "cores": cores, "kind": "port",
            "sample": ("oracle (CPU restatement) per stage on bounded samples, scaled to two %dx%d segments: %d x (16000x%d) "
                       "phase correlations (numpy float64 FFT), RRC via the rates of the default leg, SectionaryRemap bicubic "
                       "on %dx%d per thread x %d threads, StitchBigRaw on 2 x 4096x8192 per thread; %.1f s of wall time" %
                       (W, L, sections, overlap, ws, ls, nthr, time.time() - t_all)),
            "stage_seconds_full_step": stn,
            "one_core": {"value": mpix / sum(st1.values()), "unit": "Mpix/s", "stage_seconds_full_step": st1}}


def end_to_end_default(ctx, plan, bufs, raw_pan, raw_mss, d_kb_pan, d_kb_mss, out, threshold, reps=3):
    """PCIe-inclusive pass of the default action on one GPU: the raw rasters start in PAGEABLE host memory (the
    reference's heap buffers, imageop.h:52-82) and the aligned image ends in pageable host memory; file I/O is not
    included.  Uploads run on a second host thread through the staging ring (oip_upload_staged), section by section (its MSS
    lines, then its PAN lines in blocks); each block is corrected as it lands, and a correlation section is computed as soon as its
    lines are resident -- so H2D, the RRC kernels and the FFT correlation overlap; the fit, the align kernel and the
    staged download of the aligned image follow.  Results are the bits of the HBM-resident step (same unit pairs)."""
    import queue
    import threading
    import torch
    import opticalimageprocessor_amd as oip
    W, pb, mb = plan.W, plan.pb, plan.mb
    host_pan = np.empty((pb, W), np.uint16); host_mss = np.empty((mb, W), np.uint16)
    ctx.download_staged(host_pan, raw_pan); ctx.download_staged(host_mss, raw_mss)
    host_out = np.empty(tuple(out.shape), np.uint16)
    # PAN line blocks in upload order: the lines of the correlation sections first (about three blocks per section),
    # then the lines between the sections -- so the last section's correlation runs under the rest of the upload
    sec_rows = [plan.section(s)[:2] for s in range(plan.sections)]
    blocks, covered = [], []
    for a, b in sec_rows:
        nb = max(1, -(-(b - a) // 6400))
        cuts = [a + (b - a) * i // nb for i in range(nb + 1)]
        blocks += [(cuts[i], cuts[i + 1]) for i in range(nb)]
        covered.append((a, b))
    n_sec_blocks = len(blocks)
    prev = 0
    for a, b in covered + [(pb, pb)]:
        while prev < a:
            e = min(a, prev + 6400)
            blocks.append((prev, e))
            prev = e
        prev = max(prev, b)
    nblk = len(blocks)
    # upload order: the MSS lines of a section go right before its PAN lines (a section's correlation needs only its own band
    # windows), the MSS lines between the sections right after the last section -- they are needed by the align kernel only -- and
    # the PAN lines between the sections last.  (Uploading the whole MSS strip first delayed the last section by 5 ms.)
    order, mss_done = [], 0
    blk_of_sec = [[i for i, (a, b) in enumerate(blocks[:n_sec_blocks]) if sa <= a and b <= sb] for sa, sb in sec_rows]
    mss_ranges = [plan.section(s)[2:] for s in range(plan.sections)]
    # Tried and switched off (OIP_E2E_SPLIT=1 turns it on): the LAST section in two groups of column slices (a unit is a column slice
    # of all 16000 lines; oip_upload_staged_2d), so that the first group's correlation runs under the upload of the second.  The 2-D
    # copies into the ring (24-36 KB rows instead of one 32 MiB run) cost the upload lane 21 ms more than the 7 ms the overlap
    # returns: 164 ms against 154.
    split = (plan.slices * 6 // 10) * plan.base_cols if plan.slices >= 4 and os.environ.get("OIP_E2E_SPLIT") == "1" else 0
    mss_first = os.environ.get("OIP_E2E_MSS_FIRST") == "1"       # A/B switch: the whole MSS strip first (the order before round 3)
    if mss_first:
        order.append(("mss", 0, mb))
        mss_ranges = [(0, mb)]
    for s in range(plan.sections):
        if not mss_first:
            order.append(("mss",) + tuple(mss_ranges[s]))
        if s == plan.sections - 1 and split:
            order += [("pan2d", i, 0, split) for i in blk_of_sec[s]] + [("units", s, 0, split)]
            order += [("pan2d", i, split, W) for i in blk_of_sec[s]] + [("units", s, split, W)]
        else:
            order += [("pan", i) for i in blk_of_sec[s]] + [("units", s, 0, W)]
    prev = 0
    for m0, m1 in sorted(mss_ranges) + [(mb, mb)]:
        if prev < m0:
            order.append(("mss", prev, m0))
        prev = max(prev, m1)
    n_before_fit = len(order)                      # every section and every MSS line is in after these
    order += [("pan", i) for i in range(n_sec_blocks, nblk)]
    bw = W // 4
    times, lane = [], []
    for rep in range(reps):
        raw_pan.zero_(); raw_mss.zero_(); out.zero_()
        torch.cuda.synchronize()
        ctx.stage_stats(reset=True)
        t0 = time.perf_counter()
        q = queue.Queue()

        def uploader():
            for it in order:
                if it[0] == "mss":
                    q.put(ctx.upload_staged(raw_mss, host_mss[it[1]:it[2]], want_ticket=True, byte_offset=it[1] * W * 2))
                elif it[0] == "pan":
                    a, b = blocks[it[1]]
                    q.put(ctx.upload_staged(raw_pan, host_pan[a:b], want_ticket=True, byte_offset=a * W * 2))
                elif it[0] == "pan2d":
                    a, b = blocks[it[1]]
                    q.put(ctx.upload_staged_2d(raw_pan, W * 2, host_pan[a:b, it[2]:it[3]], want_ticket=True, byte_offset=(a * W + it[2]) * 2))
        th = threading.Thread(target=uploader)
        th.start()
        shifts = np.full((4, plan.n_units, 4), np.nan)
        for u in range(plan.n_units):
            shifts[:, u, 3] = (u % plan.slices) * plan.base_cols + plan.base_cols // 2
        cx = cy = None
        for n, it in enumerate(order):
            if it[0] == "units":
                # the lines (and columns) of these units are resident: their blocks were waited for above
                sec, c0, c1 = it[1], it[2], it[3]
                units = [u for u in range(sec * plan.slices, (sec + 1) * plan.slices)
                         if c0 <= (u % plan.slices) * plan.base_cols and ((u % plan.slices) + 1) * plan.base_cols <= c1]
                wins = [bufs.unit_windows(u) for u in units]
                res = ctx.interband_correlate_units([w[0].data_ptr() for w in wins], [w[0].stride(0) for w in wins],
                                                    [[x.data_ptr() for x in w[1]] for w in wins], [w[1][0].stride(0) for w in wins],
                                                    plan.base_rows, plan.base_cols)
                for j, u in enumerate(units):
                    shifts[:, u, :3] = res[j]
            else:
                ctx.stage_wait(q.get())
            if it[0] == "mss":
                m0, m1 = it[1], it[2]
                ctx.mss_split_rrc_u16(raw_mss.data_ptr() + m0 * W * 2, bufs.planes.data_ptr() + 2 * (bufs.own_planes_offset() + m0 * bw),
                                      bufs.plane_stride, W, m1 - m0, d_kb_mss)
            elif it[0] == "pan":
                a, b = blocks[it[1]]
                ctx.rrc_u16(raw_pan.data_ptr() + a * W * 2, bufs.pan.data_ptr() + a * W * 2, W, b - a, d_kb_pan)
            elif it[0] == "pan2d":
                a, b = blocks[it[1]]
                c0, c1 = it[2], it[3]
                ctx.rrc_u16_window(raw_pan.data_ptr() + (a * W + c0) * 2, W, bufs.pan.data_ptr() + (a * W + c0) * 2, W, c1 - c0, b - a,
                                   d_kb_pan[c0:])
            if n == n_before_fit - 1:
                # every section and the whole MSS strip are in: fit, align and bring the aligned image down while the uploader
                # thread is still sending the PAN lines between the sections (the download lane of the staging layer is independent)
                cx, cy = oip.filter_and_fit(shifts, threshold, 5)
                ctx.align_mss_bicubic_u16x4(bufs.planes.data_ptr() + 2 * bufs.own_planes_offset(), bufs.plane_stride, out, W // 4, plan.Lm, cx, cy,
                                            plan.lps, plan.line_offset, plan.overlap, plan.keep, plan.min_lines)
                ctx.download_staged(host_out, out)
        th.join()
        ctx.sync()
        times.append(time.perf_counter() - t0)
        lane.append(ctx.stage_stats())
    best = min(times)
    cs, ws, nb, _ = lane[times.index(best)]
    pix = 1.25 * W * pb
    return {"value": pix / best / 1e6, "unit": "Mpix/s", "ms_per_pass": best * 1e3, "passes": reps,
            "bytes_up": int(host_pan.nbytes + host_mss.nbytes), "bytes_down": int(host_out.nbytes),
            "host_copy_threads": oip.load_library().oip_stage_threads(),
            "upload_lane": {"seconds_in_pageable_to_pinned_copies": cs, "seconds_waiting_for_a_ring_slot": ws, "GB": nb / 1e9,
                            "note": "the uploader thread's own time (oip_stage_stats): copies into the pinned ring vs waiting for a "
                                    "slot whose DMA has not finished -- the second is the link's share of the pass"},
            "what": "pageable host rasters -> pinned staging ring -> H2D (on a second thread: per section its MSS lines, then its PAN "
                    "lines in blocks -- %d PAN blocks in all; the lines between the sections last) || RRC per block || correlation per section "
                    "as its lines land -> fit -> align -> staged D2H of the aligned image into pageable memory, under the rest of "
                    "the upload; file I/O excluded; best of %d" % (nblk, reps)}, (cx, cy)


def cli_default_action(env, d, kb_mss, threshold):
    """The PRODUCT timed, file I/O included: `lib/oip --pan P.RAW --mss M.RAW --do-rrc4pan ...` (the C++ host's pipelined default
    action, csrc/oip_host.hpp PreProcessor::RunPipelined) on the headline strip written to files in tmpfs by this harness.  Wall time
    of the whole process and the log's own stage times (the TIMING line: seconds since the action started), for the aligned product
    uncompressed (`raw`), with <pan>.RRC.RAW as well (`raw_rrcpan`), in the reference's LZW + predictor encoding (`lzw`) and for the
    step-by-step flow (OIP_PIPELINE=0, the reference's order of work).  The uncompressed product is compared with the resident
    step's aligned image."""
    import shutil
    import subprocess
    import tempfile
    ctx, torch = env.ctx, env.torch
    W, pb, mb = d.W, d.pb, d.plan.mb
    exe = os.path.join(ROOT, "opticalimageprocessor_amd", "lib", "oip")
    if not os.path.exists(exe):
        return {"error": "lib/oip not built"}
    base = os.environ.get("OIP_BENCH_TMP", "/dev/shm")
    if not (os.path.isdir(base) and os.access(base, os.W_OK)):
        base = tempfile.gettempdir()
    need = 2.2 * (pb + mb) * W * 2                      # inputs + the largest set of products, with some room
    if shutil.disk_usage(base).free < need:
        return {"error": "not enough room under %s for the strip files (%.0f GB wanted)" % (base, need / 1e9)}
    tmp = tempfile.mkdtemp(prefix="oip_bench_", dir=base)
    try:
        t0 = time.time()
        host = np.empty((pb, W), np.uint16)
        ctx.download_staged(host, d.raw_pan)
        host.tofile(os.path.join(tmp, "B_PAN.RAW"))
        host = np.empty((mb, W), np.uint16)
        ctx.download_staged(host, d.raw_mss)
        host.tofile(os.path.join(tmp, "B_MSS.RAW"))
        del host

        def csv(name, kb):
            with open(os.path.join(tmp, name), "w") as f:
                f.write("1\n%d\n0\n" % len(kb))
                f.write("".join("%.6f , %.4f\n" % (k, b) for k, b in kb))
        csv("PAN.csv", d.kb_pan)
        bw = W // 4
        for b in range(4):
            csv("MSS.B%d.csv" % (b + 1), kb_mss[b * bw:(b + 1) * bw])
        want = d.out.cpu().numpy()[..., [2, 1, 0, 3]]                     # the product's payload: cv::imwrite's sample order
        t_inputs = time.time() - t0
        args = [exe, "--width", str(W), "--pan", "B_PAN.RAW", "--mss", "B_MSS.RAW", "--do-rrc4pan", "--rrc-pan", "PAN.csv",
                "--slices", str(d.plan.slices), "--ibc-sections", str(d.plan.sections), "--ibc-threshold", repr(threshold),
                "--lines-section", str(d.plan.lps), "--line-offset", str(d.plan.line_offset), "--overlap-lines", str(d.plan.overlap)]
        for b in range(4):
            args += ["--rrc-msb%d" % (b + 1), "MSS.B%d.csv" % (b + 1)]
        runs = {}
        same = None
        # what a process pays before and after its work: `oip --version` (exec, dynamic linking, static initialisers, exit -- no GPU)
        t1 = time.perf_counter()
        subprocess.run([exe, "--version"], cwd=tmp, capture_output=True)
        startup_ms = (time.perf_counter() - t1) * 1e3
        plan = [("raw", {"OIP_TIFF_COMPRESS": "none"}, []), ("raw_again", {"OIP_TIFF_COMPRESS": "none"}, []),
                ("raw_rrcpan", {"OIP_TIFF_COMPRESS": "none"}, ["--write-rrcpan"]),
                ("steps_raw", {"OIP_TIFF_COMPRESS": "none", "OIP_PIPELINE": "0"}, []), ("lzw", {}, [])]
        for name, extra_env, extra_args in plan:
            for f in ("B_MSS.ALIGNED.TIFF", "B_PAN.RRC.RAW"):
                if os.path.exists(os.path.join(tmp, f)):
                    os.remove(os.path.join(tmp, f))
            e = dict(os.environ, LOGFILE=os.path.join(tmp, "oip.log"))
            e.pop("OIP_TIFF_COMPRESS", None)
            e.update(extra_env)
            t1 = time.perf_counter()
            r = subprocess.run(args + extra_args, cwd=tmp, env=e, capture_output=True, text=True)
            wall = time.perf_counter() - t1
            rec = {"wall_ms": wall * 1e3, "exit": r.returncode}
            for ln in r.stdout.splitlines():
                if ln.startswith("TIMING default_action"):
                    kv = dict(x.split("=", 1) for x in ln.split()[2:])
                    rec["correlation_calls_start_plus_ms"] = kv.pop("correlation_calls_ms", None)
                    rec["log_seconds"] = {k: float(v) for k, v in kv.items()}
                if ln.startswith("TIMING tiff_lzw"):                       # the device strip encoder's own split (csrc/oip_host.hpp)
                    rec["tiff_lzw_seconds"] = {k: float(v) for k, v in (x.split("=", 1) for x in ln.split()[2:])}
            if r.returncode != 0:
                rec["tail"] = (r.stdout + r.stderr)[-400:]
            if r.stderr.strip():
                rec["stderr_tail"] = r.stderr.strip()[-300:]
            prod = os.path.join(tmp, "B_MSS.ALIGNED.TIFF")
            if os.path.exists(prod):
                rec["product_bytes"] = os.path.getsize(prod)
            if name == "raw" and r.returncode == 0:
                with open(prod, "rb") as f:
                    big = f.read(4)[2] == 43
                got = np.fromfile(prod, np.uint16, count=want.size, offset=16 if big else 8).reshape(want.shape)
                same = bool(np.array_equal(got, want))
                del got
            runs[name] = rec
        # the reference's step 5 on that LZW product (imageop.h:365-457: cv::imread x 2, concat, cv::imwrite): `oip stitch` of the
        # file with itself -- strips decoded and encoded on the device (csrc/tifflzw.hip)
        prod = os.path.join(tmp, "B_MSS.ALIGNED.TIFF")
        if runs.get("lzw", {}).get("exit") == 0 and os.path.exists(prod):
            e = dict(os.environ, LOGFILE=os.path.join(tmp, "oip.log"))
            e.pop("OIP_TIFF_COMPRESS", None)
            t1 = time.perf_counter()
            r = subprocess.run([exe, "stitch", "--image1", "B_MSS.ALIGNED.TIFF", "--image2", "B_MSS.ALIGNED.TIFF", "--fold-cols", "50", "-o",
                                "B_STITCHED.TIFF"], cwd=tmp, env=e, capture_output=True, text=True)
            rec = {"wall_ms": (time.perf_counter() - t1) * 1e3, "exit": r.returncode}
            if r.returncode != 0:
                rec["tail"] = (r.stdout + r.stderr)[-400:]
            out_path = os.path.join(tmp, "B_STITCHED.TIFF")
            if os.path.exists(out_path):
                rec["product_bytes"] = os.path.getsize(out_path)
                os.remove(out_path)
            runs["stitch_tiff_lzw"] = rec
        best = min((runs[k] for k in ("raw", "raw_again") if runs[k]["exit"] == 0), key=lambda r_: r_["wall_ms"], default=None)
        out = {"runs": runs, "version_only_ms": startup_ms, "aligned_product_equals_resident_step": same, "inputs_written_s": t_inputs, "tmp": base,
               "bytes_in": int((pb + mb) * W * 2), "what": ("wall time of the `oip` executable (process start, HIP initialisation, file reads, "
               "kernels, product writes) on %dx%d PAN + MSS files in tmpfs; log_seconds = the log's TIMING line, seconds since the action "
               "started (products_written: everything on disk)" % (W, pb))}
        if best:
            ls = best.get("log_seconds", {})
            out.update({"wall_ms": best["wall_ms"], "pipeline_ms": ls.get("products_written", 0.0) * 1e3,
                        "read_GBs": out["bytes_in"] / ls["read_done"] / 1e9 if ls.get("read_done") else None,
                        "Mpix_s_wall": 1.25 * W * pb / best["wall_ms"] / 1e3,
                        "Mpix_s_pipeline": 1.25 * W * pb / ls["products_written"] / 1e6 if ls.get("products_written") else None})
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


class Params:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def build_workload(env, p):
    """Allocate the synthetic rasters of one workload on this rank and return (step, description dict).  `env` carries the
    context, the device, rank/world and the imported modules; `p` names the workload (Params)."""
    torch, synth, ctx, dev, rank, world = env.torch, env.synth, env.ctx, env.dev, env.rank, env.world
    from opticalimageprocessor_amd.dist import (CcdBuffers, CcdPlan, HipBackend, ShardBuffers, StripPlan,
                                                default_action_step, prestitch_stitch_step)
    W = p.width
    pb = 65536 if p.workload == "rrc" else p.lines
    Lp = pb * world
    kb_pan = synth.lut(W)
    kb_mss = np.concatenate([synth.lut(W // 4, 10 + b) for b in range(4)], 0)
    d_kb_pan, d_kb_mss = ctx.upload_kb(kb_pan), ctx.upload_kb(kb_mss)
    info = {}
    d = Params(W=W, pb=pb, info=info, kb_pan=kb_pan, d_kb_pan=d_kb_pan, d_kb_mss=d_kb_mss, plan=None, rows_arrays=5.0)
    if p.workload == "prestitch":
        OV = 200
        kb2 = synth.lut(W, 5)
        d_kb2 = ctx.upload_kb(kb2)
        nsec = min(10, Lp // 16000)
        cplan = CcdPlan(W, Lp, world, nsec, 16000, OV, 0)
        pan1, pan2 = synth.ccd_pair(rank * pb, pb, W, OV, kb_pan, kb2, device=dev)
        cbufs = CcdBuffers(cplan, rank, pan1, pan2)
        prestt = torch.empty_like(pan1)
        stitched = torch.empty(pb, 2 * (W - cplan.fold), dtype=torch.uint16, device=dev)
        backend = HipBackend(ctx, cplan)

        def step(timer=None):
            # main.cpp:270-286 then :177-190: correlation on the raw strips, RRC of both, remap of CCD 2, stitch
            dx, dy, _ = prestitch_stitch_step(backend, cplan, cbufs, d_kb_pan, d_kb2, prestt, stitched, rank,
                                              threshold=p.threshold, f16acc=p.fp16, fused=getattr(p, "fused", False), timer=timer)
            info["dx"], info["dy"] = dx, dy
        d.__dict__.update(step=step, pix_per_rank=2 * W * pb, base_rows=16000, base_cols=OV, M=16000, N=OV, out_local=0,
                          rows_arrays=2.0, prestt=prestt, stitched=stitched, sections=nsec, plan=cplan, backend=backend,
                          workload=("prestitch + stitch: 2 CCD segments %dx%d%s, %d x (16000x%d) phase correlations, RRC x2, "
                                    "constant-shift bicubic remap (30000-row sections, %s accumulate), RAW stitch fold %d%s" %
                                    (W, Lp, " in %d scan-line blocks" % world if world > 1 else "", nsec, OV,
                                     "fp16" if p.fp16 else "fp32", cplan.fold,
                                     "; FUSED: RRC of CCD 1 and the resampled CCD-2 lines written straight into the stitched raster"
                                     if getattr(p, "fused", False) else "")))
    elif p.workload == "rrc":
        raw_pan = synth.pan_strip(rank * pb, pb, W, kb_pan, device=dev)
        dst = torch.empty_like(raw_pan)

        def step():
            ctx.rrc_u16(raw_pan, dst, W, pb, d_kb_pan)
        d.__dict__.update(step=step, pix_per_rank=W * pb, base_rows=1, base_cols=1, M=1, N=1, out_local=0,
                          workload="RRC kernel only, %dx%d u16 per GPU (BASELINE config 2)" % (W, pb))
    else:
        # default: the reference's sections for the WHOLE strip, whatever its length (BASELINE config 4);
        # weak5n: `world` times the sections, so that every rank keeps the single-GPU run's correlation work
        raw_pan = synth.pan_strip(rank * pb, pb, W, kb_pan, device=dev)
        sections_total = p.sections * (world if p.workload == "weak5n" else 1)
        plan = StripPlan(W, Lp, world, p.slices, sections_total)
        bufs = ShardBuffers(plan, rank, dev)
        raw_mss = synth.mss_strip(rank * plan.mb, plan.mb, W, kb_mss, device=dev)
        o0, o1 = plan.align_out_rows(rank)
        out = torch.zeros(max(o1 - o0, 1), W // 4, 4, dtype=torch.uint16, device=dev)
        backend = HipBackend(ctx, plan)

        def step(timer=None):
            cx, cy, _ = default_action_step(backend, plan, bufs, raw_pan, raw_mss, d_kb_pan, d_kb_mss, out, rank,
                                            threshold=p.threshold, timer=timer)
            info["cx"], info["cy"] = cx, cy
        base_rows, base_cols = plan.base_rows, W // p.slices
        M, N = optimal_dft_size(base_rows), optimal_dft_size(base_cols)
        workload = ("default action (--do-rrc4pan): PAN %dx%d + MSS 4x(%dx%d) per GPU; RRC + %dx%dx4 inter-band phase "
                    "correlations (%dx%d FFT) + polyfit + bicubic align to 16UC4" %
                    (W, pb, W // 4, plan.mb, p.sections, p.slices, M, N))
        if world > 1:
            workload += ("; strip of %d x %d lines with %d sections in total, %d..%d of its %d units per rank" %
                         (world, pb, sections_total, min(len(plan.units_of(r)) for r in range(world)),
                          max(len(plan.units_of(r)) for r in range(world)), plan.n_units))
        d.__dict__.update(step=step, pix_per_rank=W * pb + W * plan.mb, base_rows=base_rows, base_cols=base_cols, M=M, N=N,
                          out_local=o1 - o0, workload=workload, plan=plan, bufs=bufs, raw_pan=raw_pan, raw_mss=raw_mss, out=out,
                          backend=backend)
    # a synthetic scene that does not clear --ibc-threshold is an error here, not a reason to change the workload
    d.step()
    ctx.sync()
    return d


def measure(env, d, steps, warmup):
    """W untimed warm-up steps, then exactly K timed steps between barriers; max over ranks.

    Per-kernel table: ONE untimed step with every kernel bracketed by HIP events (the last warm-up step, or an extra
    step when --warmup 0), recorded by the library on the stream the kernels run on.  The events cost about 2 % of a
    step, so the timed region times only the dominant kernel found there -- which is the kernel the roofline object is
    about."""
    ctx, torch, dist = env.ctx, env.torch, env.dist
    for _ in range(max(warmup - 1, 0)):
        d.step()
    ctx.profile_filter(None)
    ctx.profile_reset()
    ctx.profile_enable(True)
    d.step()
    ctx.sync()
    ctx.profile_enable(False)
    prof_all = ctx.profile()
    dom = max(prof_all.items(), key=lambda kv: kv[1][0])[0] if prof_all else None
    env.barrier()
    ctx.profile_reset()
    ctx.profile_filter(dom)
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        d.step()
    ctx.sync()
    env.barrier()
    elapsed = time.perf_counter() - t0
    ctx.profile_enable(False)
    ctx.profile_filter(None)
    prof = ctx.profile()                     # the dominant kernel, over the timed region
    t = torch.tensor([elapsed], dtype=torch.float64, device=env.dev if env.dist_backend == "nccl" else "cpu")
    if env.world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item()), prof_all, prof, dom


def summarise(env, d, p, elapsed, steps, prof_all, prof, dom):
    ms_per_step = elapsed / steps * 1e3
    value = d.pix_per_rank * env.world * steps / elapsed / 1e6
    ab = algorithmic_bytes(d.W, d.pb, d.M, d.N, d.base_rows, d.base_cols, d.out_local, d.rows_arrays)
    kernels = {}
    for name, (ms, n) in prof_all.items():
        avg = ms / max(n, 1)
        e = {"launches_per_step": float(n), "avg_ms": avg, "total_ms_per_step": ms}
        nbytes = ab.get(name) or fft_pass_bytes(name, d.M, d.N, d.base_rows, d.base_cols)
        if nbytes:
            e["algorithmic_GBs"] = nbytes / (avg * 1e-3) / 1e9
            ab[name] = nbytes
        kernels[name] = e
    roof = None
    if dom and dom in prof and prof[dom][1] > 0:
        # the dominant kernel's row comes from the timed region
        avg = prof[dom][0] / prof[dom][1]
        kernels[dom].update({"avg_ms_untimed_step": kernels[dom]["avg_ms"], "avg_ms": avg,
                             "launches_per_step": prof[dom][1] / steps, "total_ms_per_step": prof[dom][0] / steps})
        if ab.get(dom):
            kernels[dom]["algorithmic_GBs"] = ab[dom] / (avg * 1e-3) / 1e9
            avg_s = avg * 1e-3
            achieved = ab[dom] / avg_s / 1e9
            traffic = tsrc = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            tkey = "default" if p.workload == "weak5n" else p.workload
            if p.workload == "prestitch" and getattr(p, "fused", False):
                tkey = "prestitch_fused"
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                traffic = tj.get(tkey, {}).get(dom)
                tsrc = tj.get("_source", {}).get(tkey)
            roof = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "traffic_source": (("profiles/traffic.json (%s): rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of this "
                                        "command taken when the profile set was committed, (2*FETCH_SIZE + WRITE_SIZE)*1024 B per launch; "
                                        "NOT re-measured in this run" % (tsrc or "profiles/collect.sh")) if traffic is not None else None),
                    "algorithmic_bytes_per_launch": ab[dom], "avg_launch_ms": avg_s * 1e3}
            if dom == "corr_rows_up_kernel":
                roof["note"] = ("row stage of a pair of units: six 3000-point row transforms, eight cross-powers per bin pair and "
                                "the x4 up-sampling operator in one pass over the data; it moves its algorithmic bytes once "
                                "(traffic / algorithmic = 1.01) and is bound by vector-instruction issue, not by HBM "
                                "(DESIGN.md 4.3) -- the HBM-bound passes of the step run at 3.6-5.8 TB/s")
                gf = row_stage_gflop(d.M, d.N)
                roof["vector_f32"] = {"achieved": gf / avg_s / 1e3, "peak": VECTOR_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": gf / avg_s / 1e3 / VECTOR_F32_PEAK_TFLOPS, "algorithmic_gflop_per_launch": gf,
                                      "model": ROW_STAGE_FLOP_MODEL}
    return ms_per_step, value, kernels, roof


VECTOR_F32_PEAK_TFLOPS = 157.3          # MI355X vector fp32 (packed FMA; /opt/skills/guides/MI355X_MICROARCH.md)
ROW_STAGE_FLOP_MODEL = ("nominal flops of one corr_rows_up_kernel launch (a pair of M x N units, spectral x4 up-sampling): complex "
                        "n-point transforms at 5 n log2 n -- 5 M of length N (1 forward + 4 inverse) and 4 M of length N/4 (the band lines); "
                        "4 M N output bins x 2 cross-powers x 12 flops (complex product, magnitude, division); horizontal expansion "
                        "H z + 4 G e per band bin: 4 M N x 38; vertical expansion on the way into LDS: M N x 38")


def row_stage_gflop(M, N):
    """ROW_STAGE_FLOP_MODEL in numbers: 29.9 GFLOP for M = 16000, N = 3000"""
    fft = lambda n: 5.0 * n * np.log2(n)
    return (5 * M * fft(N) + 4 * M * fft(N / 4) + 4.0 * M * N * 24 + 4.0 * M * N * 38 + 1.0 * M * N * 38) / 1e9


def rrc_line(kernels):
    rk = "rrc_u16_flat_kernel" if "rrc_u16_flat_kernel" in kernels else "rrc_u16_kernel"
    if rk in kernels and "algorithmic_GBs" in kernels[rk]:
        g = kernels[rk]["algorithmic_GBs"]
        return {"GBs_read_plus_write": g, "frac_of_8TBs": g / HBM_PEAK_GBS, "GBs_read_only": g / 2,
                "frac_of_8TBs_read_only": g / 2 / HBM_PEAK_GBS, "Mpix_s": g / 4 * 1e3}
    return None


def config_legs(env, args, line):
    """The other single-GPU BASELINE configurations, run after the headline workload with a few steps each and attached to
    the line as `configs`: 2 (RRC 30000x65536), 3 (65536 lines, 4 sections), the N = 1 form of 5 (cross-CCD, fp32 and
    fp16 accumulate with the measured max |delta| in DN) and the reference's native 12288-wide strips."""
    torch = env.torch
    out = {}

    def leg(name, p, steps=5, warmup=1, cpu=None, extra=None):
        t0 = time.time()
        torch.cuda.empty_cache()
        d = build_workload(env, p)
        elapsed, prof_all, prof, dom = measure(env, d, steps, warmup)
        ms, value, kernels, roof = summarise(env, d, p, elapsed, steps, prof_all, prof, dom)
        e = {"workload": d.workload, "steps": steps, "warmup": warmup, "ms_per_step": ms, "Mpix_s": value,
             "dominant_kernel": ({k: roof[k] for k in ("kernel", "achieved", "unit", "frac", "avg_launch_ms", "algorithmic_bytes_per_launch")}
                                 if roof else None),
             "kernels_ms_per_step": {k: round(v["total_ms_per_step"], 4) for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["total_ms_per_step"])}}
        r = rrc_line(kernels)
        if r:
            e["rrc_kernel"] = r
        if extra:
            e.update(extra(d))
        if cpu and not args.no_cpu_baseline:
            e["cpu_baseline"] = cpu()
        e["wall_s"] = round(time.time() - t0, 1)
        out[name] = e
        return d

    W = 30000
    base = line.get("cpu_baseline")

    def cpu_rrc():
        return {"value": base["rrc_all_cores_Mpix_s"], "unit": "Mpix/s", "cores": base["rrc_all_cores"],
                "kind": "reference" if "oracle/_ref" in base["sample"] else "port",
                "sample": "InplaceRRC row-parallel on all cores (restatement, bit-equal to oracle/_ref); the reference's own loop on "
                          "one thread: %.0f Mpix/s" % base["rrc_reference_1thread_Mpix_s"]}
    leg("config2_rrc_30000x65536", Params(workload="rrc", width=W, lines=65536, slices=10, sections=5, threshold=args.ibc_threshold, fp16=False),
        steps=20, cpu=cpu_rrc if base else None)
    leg("config3_30000x65536_4sections", Params(workload="default", width=W, lines=65536, slices=10, sections=4, threshold=args.ibc_threshold, fp16=False),
        cpu=lambda: cpu_baseline(W, 65536, 10, 4))
    keep = {}

    def grab(d):
        keep["prestt32"] = d.prestt.clone()
        keep["stitched32"] = d.stitched.clone()
        return {"shift": {"dx": d.info.get("dx"), "dy": d.info.get("dy"), "truth_px": list(env.synth.CCD_SHIFT)}}
    pl = 100000
    leg("config5_n1_prestitch_2x30000x100000_fp32", Params(workload="prestitch", width=W, lines=pl, slices=10, sections=5, threshold=args.ibc_threshold, fp16=False),
        cpu=lambda: cpu_baseline_prestitch(W, pl, min(10, pl // 16000)), extra=grab)

    def delta(d):
        a, b = keep.pop("prestt32"), d.prestt
        worst, diff, tot = 0, 0, 0
        for y in range(0, a.shape[0], 8192):
            x = (a[y:y + 8192].to(torch.int32) - b[y:y + 8192].to(torch.int32)).abs()
            worst = max(worst, int(x.max().item())); diff += int((x > 0).sum().item()); tot += int(x.sum().item())
        n = a.numel()
        return {"fp16_vs_fp32": {"max_abs_delta_DN": worst, "mean_abs_delta_DN": tot / n, "pixels_differing_frac": diff / n,
                                 "tolerance_asserted_DN": 6, "where": "tests/test_gpu_resample.py::test_remap_f16acc_tolerance"}}
    leg("config5_n1_prestitch_2x30000x100000_fp16acc", Params(workload="prestitch", width=W, lines=pl, slices=10, sections=5, threshold=args.ibc_threshold, fp16=True),
        extra=delta)
    def same(d):
        ok = bool(torch.equal(keep["stitched32"].view(torch.int16), d.stitched.view(torch.int16)))
        keep.clear()
        return {"stitched_equals_the_three_pass_flow": ok}
    leg("config5_n1_prestitch_2x30000x100000_fp32_fused", Params(workload="prestitch", width=W, lines=pl, slices=10, sections=5, threshold=args.ibc_threshold,
                                                                 fp16=False, fused=True), extra=same)
    keep.clear()
    leg("reference_geometry_12288x100000", Params(workload="default", width=12288, lines=100000, slices=10, sections=5, threshold=args.ibc_threshold, fp16=False),
        cpu=lambda: cpu_baseline(12288, 100000, 10, 5))
    return out


def multi_gpu_stage_times(env, d):
    """ONE instrumented step after the timed region (dist.StepTimer: the stream is drained at every stage boundary, so the step is
    slower than a timed one): per rank the stage times, next to what the placement model predicted for the correlation stage --
    the model's constants (dist.LINK_GBS, PAIR_US_16000x3000) have never met a multi-GPU node; this is what the first run on one
    has to confirm or refute.  On the gloo rehearsal (ranks sharing GPUs, host-staged transfers) the numbers only show that the
    fields work."""
    from opticalimageprocessor_amd.dist import LINK_GBS, StepTimer
    env.barrier()
    t = StepTimer(d.backend)
    d.step(timer=t)
    mine = t.result()
    allr = [None] * env.world
    env.dist.all_gather_object(allr, mine)
    if env.rank != 0:
        return None
    rnd = lambda v: [round(x, 3) for x in v] if isinstance(v, list) else round(v, 3)
    keys = sorted({k for r in allr for k in r})
    return {"mode": "rccl over xGMI" if env.dist_backend == "nccl" else "rehearsal (%s, ranks share GPUs, transfers staged through the host)" % env.dist_backend,
            "link_GBs_assumed": LINK_GBS,
            "predicted_correlation_finish_us": [int(v) for v in d.plan.predicted_finish_us],
            "measured_correlation_finish_us": [int(r.get("correlation_finish_ms", 0.0) * 1e3) for r in allr],
            "measured_ms": {k: [rnd(r.get(k, 0.0)) for r in allr] for k in keys if k != "correlation_finish_ms"},
            "note": "per rank; list entries = one per received pair of units; measured on one instrumented step outside the timed region"}


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: this process -- which never touches the GPU -- starts the N ranks as child
    processes of the same command line (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, rendezvous on 127.0.0.1),
    relays rank 0's JSON line and exits with the worst exit code.  Under torch.distributed.run the ranks arrive with WORLD_SIZE
    set and this is not used."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        e = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [q.wait() for q in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(c) for c in codes)


def compact_line(line, full_path):
    """What goes to stdout: the contract's keys, `roofline`, `cpu_baseline`, and a summary of everything else INSIDE `config` (the
    driver's record keeps `config` whole) -- the verbose record (per-kernel tables, every leg's kernels, the CPU samples) goes to
    `full_path`.  Kept under 8 KB."""
    c = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                              "vs_baseline", "dtype", "data") if k in line}
    cfg = dict(line["config"])
    e2e = line.get("end_to_end")
    if e2e:
        cfg["end_to_end_ms"] = round(e2e["ms_per_pass"], 2)
        cfg["end_to_end_Mpix_s"] = round(e2e["value"], 1)
        cfg["end_to_end_same_fit_as_resident_step"] = e2e.get("same_fit_as_resident_step")
        cfg["rrc_host_buffer_Gpix_s"] = round(e2e.get("rrc_host_buffer_Gpix_s", 0.0), 2)
    cli = line.get("cli")
    if cli and "runs" in cli:
        cfg["cli_wall_ms"] = {k: round(v["wall_ms"], 1) for k, v in cli["runs"].items()}
        cfg["cli_pipeline_ms"] = {k: round(v["log_seconds"]["products_written"] * 1e3, 1) for k, v in cli["runs"].items() if "log_seconds" in v}
        cfg["cli_read_GBs"] = round(cli["read_GBs"], 1) if cli.get("read_GBs") else None
        cfg["cli_product_equals_resident_step"] = cli.get("aligned_product_equals_resident_step")
        cfg["cli_note"] = ("wall of the `oip` executable on files in tmpfs, ms per run: raw = uncompressed aligned product (two runs), "
                           "raw_rrcpan = + <pan>.RRC.RAW, lzw = the reference's LZW product (strips encoded on the device), steps_raw = the step-by-step "
                           "flow, stitch_tiff_lzw = `oip stitch` of that LZW product with itself (device decode + encode); "
                           "cli_pipeline_ms = first byte read to products on disk (the log's own clock)")
    elif cli:
        cfg["cli_error"] = cli.get("error")
    legs = {}
    for name, e in (line.get("configs") or {}).items():
        dk = e.get("dominant_kernel") or {}
        cb = e.get("cpu_baseline") or {}
        legs[name] = [round(e["ms_per_step"], 3), dk.get("kernel"), round(dk["frac"], 4) if dk.get("frac") is not None else None,
                      round(cb["value"]) if cb.get("value") else None]
        for k in ("fp16_vs_fp32", "stitched_equals_the_three_pass_flow"):
            if k in e:
                legs[name].append({k: e[k]["max_abs_delta_DN"] if isinstance(e[k], dict) else e[k]})
    if legs:
        cfg["legs"] = legs
        cfg["legs_fields"] = "ms_per_step, dominant kernel, its fraction of the 8 TB/s HBM peak, CPU baseline Mpix/s (all cores)"
    if line.get("multi_gpu"):
        cfg["multi_gpu"] = line["multi_gpu"]
    c["config"] = cfg
    roof = line.get("roofline")
    if roof:
        c["roofline"] = {k: roof[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                                              "avg_launch_ms") if k in roof}
        if "vector_f32" in roof:
            c["roofline"]["vector_f32_frac"] = round(roof["vector_f32"]["frac"], 4)
        if roof.get("traffic") is not None:
            c["roofline"]["traffic_source"] = "profiles/traffic.json (rocprofv3 --pmc passes of this command; not re-measured in this run)"
    else:
        c["roofline"] = None
    cb = line.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind") if k in cb}
        c["cpu_baseline"]["sample"] = cb["sample"][:420]
        for k in ("rrc_reference_1thread_Mpix_s", "rrc_all_cores_Mpix_s"):
            if k in cb:
                c["cpu_baseline"][k] = round(cb[k], 1)
    if line.get("rrc_kernel"):
        c["rrc_kernel"] = {k: round(v, 4) for k, v in line["rrc_kernel"].items()}
    if line.get("shift"):
        c["shift"] = line["shift"]
    c["value_note"] = ("`value`: rasters resident in HBM when the timed region starts; PCIe-inclusive = config.end_to_end_*, file-inclusive "
                       "(the `oip` executable) = config.cli_*")
    c["full_record"] = full_path
    return c


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    import torch
    import torch.distributed as dist
    import opticalimageprocessor_amd as oip
    from opticalimageprocessor_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world                    # under a launcher the launcher's world size is the truth
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    # OIP_BENCH_BACKEND=gloo rehearses the N-rank path on a box with fewer GPUs than ranks
    # (ranks share devices, transfers are staged through the host); real runs use RCCL ("nccl")
    dist_backend = os.environ.get("OIP_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if dist_backend == "nccl" and world > ndev:
        sys.exit("bench.py: %d ranks but %d GPUs visible" % (world, ndev))
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(dist_backend)
    ctx = oip.Context(local_rank)
    # one stream for everything: the library's kernels, torch's copies and RCCL's ordering
    # (RCCL synchronises against torch's CURRENT stream, so make that the context's stream)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    env = Params(torch=torch, dist=dist, synth=synth, ctx=ctx, dev=dev, rank=rank, world=world, barrier=barrier,
                 dist_backend=dist_backend)
    p = Params(workload=args.workload, width=args.width, lines=args.lines, slices=args.slices, sections=args.sections,
               threshold=args.ibc_threshold, fp16=args.fp16_accumulate, fused=args.fused)
    d = build_workload(env, p)
    elapsed, prof_all, prof, dom = measure(env, d, args.steps, args.warmup)
    multi_gpu = multi_gpu_stage_times(env, d) if world > 1 and args.workload != "rrc" else None

    if rank == 0:
        W, pb = d.W, d.pb
        ms_per_step, value, kernels, roof = summarise(env, d, p, elapsed, args.steps, prof_all, prof, dom)
        line = {
            "metric": "Mpix/s end-to-end RRC+stitch on 30000x100000x4 strip; % HBM roofline",
            "value": value, "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16 rasters; fp64 RRC/maps, f32 bicubic and FFT", "data": "synthetic",
            "config": {"workload": d.workload, "width": W, "pan_lines_per_gpu": pb, "parallelism": "rowblock%d" % world,
                       "backend": "rccl" if dist_backend == "nccl" else dist_backend + " (rehearsal, host-staged)",
                       "ibc_threshold": p.threshold, "inputs": "resident in HBM"},
            "value_note": ("`value` is the whole step with the rasters already resident in HBM when the timed region starts (the bench "
                           "contract); the PCIe-inclusive rate of the same step -- pageable host rasters in, aligned image out, file "
                           "I/O excluded -- is the `end_to_end` object of this line and is never `value`"),
            "roofline": roof,
            "kernels": kernels,
            "kernels_note": "per-kernel HIP-event times of one untimed step, except the dominant kernel: timed over the timed region (where only it carries events)",
        }
        r = rrc_line(kernels)
        if r:
            line["rrc_kernel"] = r
        if multi_gpu:
            line["multi_gpu"] = multi_gpu
        if args.workload == "prestitch":
            line["shift"] = {"dx": d.info.get("dx"), "dy": d.info.get("dy"), "truth_px": list(synth.CCD_SHIFT)}
        if world == 1 and not args.no_cpu_baseline:
            if args.workload == "prestitch":
                line["cpu_baseline"] = cpu_baseline_prestitch(W, pb, d.sections)
            elif args.workload in ("default", "weak5n"):
                line["cpu_baseline"] = cpu_baseline(W, pb, args.slices, args.sections)
            else:
                import oracle
                img = np.random.default_rng(1).integers(64, 4096, (4096, W), dtype=np.uint16)
                use_ref = oracle.ref_lib() is not None
                f = oracle.rrc_reference if use_ref else oracle.rrc
                best = 1e9
                for _ in range(5):
                    t1 = time.time(); f(img, d.kb_pan); best = min(best, time.time() - t1)
                line["cpu_baseline"] = {"value": W * 4096 / best / 1e6, "unit": "Mpix/s", "cores": 1,
                                        "kind": "reference" if use_ref else "port",
                                        "sample": "InplaceRRC on %dx4096 u16, best of 5" % W}
        if world == 1 and args.workload == "default" and not args.no_end_to_end:
            e2e, (ecx, ecy) = end_to_end_default(ctx, d.plan, d.bufs, d.raw_pan, d.raw_mss, d.d_kb_pan, d.d_kb_mss, d.out, p.threshold)
            e2e["same_fit_as_resident_step"] = bool(np.array_equal(ecx, d.info["cx"]) and np.array_equal(ecy, d.info["cy"]))
            # the host-buffer form of the RRC seam alone (IMO::InplaceRRC on DoRRC4RAW's heap buffer), in place
            hb = np.random.default_rng(2).integers(64, 4096, (32768, W), dtype=np.uint16)
            ctx.rrc_u16_host(hb[:2048], d.kb_pan)
            t1 = time.perf_counter(); ctx.rrc_u16_host(hb, d.kb_pan); dt = time.perf_counter() - t1
            e2e["rrc_host_buffer_Gpix_s"] = hb.size / dt / 1e9
            del hb
            line["end_to_end"] = e2e
        if world == 1 and args.workload == "default" and not args.no_cli:
            kb_mss = np.concatenate([synth.lut(W // 4, 10 + b) for b in range(4)], 0)
            try:
                line["cli"] = cli_default_action(env, d, kb_mss, p.threshold)
            except Exception as e:                                  # noqa: BLE001 -- the leg must not take the line with it
                line["cli"] = {"error": repr(e)[:300]}
        if world == 1 and args.workload == "default" and not args.no_configs and W == 30000 and pb == 100000:
            d = None                        # release the headline workload's rasters before the other configurations
            line["configs"] = config_legs(env, args, line)
        full = args.full_record or os.path.join(ROOT, "gpurun_out", "bench_full.json")
        try:
            os.makedirs(os.path.dirname(full), exist_ok=True)
            with open(full, "w") as f:
                json.dump(line, f)
        except OSError:
            full = None
        print(json.dumps(compact_line(line, full and os.path.relpath(full, ROOT))))
    if world > 1:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
