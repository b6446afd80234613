"""GPU: the `oip` CLI end to end (the reference's 4-step task file, DOC/sample-task.sh, steps 1-3
on RAW files) against the oracle.  Shifts / polynomials are re-derived in-process with the same
library calls the CLI makes (bit-identical), so the resampled files can be compared bit for bit."""
import os
import subprocess

import numpy as np
import pytest

import _synth
import _tiff

pytestmark = pytest.mark.gpu
OIP = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "opticalimageprocessor_amd", "lib", "oip")


def _csv(path, kb):
    with open(path, "w") as f:
        f.write("1\n%d\n0\n" % len(kb))
        for k, b in kb:
            f.write("%.6f , %.4f\n" % (k, b))


def _cuda(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_prestitch_stitch_and_default_action(ctx, oracle_mod, tmp_path):
    import opticalimageprocessor_amd as oip
    W, L, OV = 1024, 33024, 64
    d = str(tmp_path)
    # the strip-sized TIFF products are written uncompressed here so that the independent Python reader can check their
    # pixels (its LZW decoder is slow); the reference's LZW products are exercised at the end of this test
    env = dict(os.environ, LOGFILE=os.path.join(d, "oip.log"), OIP_TIFF_COMPRESS="none")
    pan1, pan2 = _synth.ccd_pair(L, W, OV, (3, -2), seed=5)
    kb1, kb2 = _synth.lut(W, 1), _synth.lut(W, 2)
    pan1.tofile(os.path.join(d, "S_PAN-1.RAW")); pan2.tofile(os.path.join(d, "S_PAN-2.RAW"))
    _csv(os.path.join(d, "PAN-1.csv"), kb1); _csv(os.path.join(d, "PAN-2.csv"), kb2)

    # ---- step 1: prestitch (stitcher.h: CalcSttParameters on the raw files, DoRRC, PreStitch)
    r = subprocess.run([OIP, "prestitch", "--width", str(W), "--pan1", "S_PAN-1.RAW", "--pan2", "S_PAN-2.RAW", "--rrc1", "PAN-1.csv",
                        "--rrc2", "PAN-2.csv", "-s", "3", "-l", "1600", "--stitch-overlap", str(OV), "--stt-threshold", "0.05"],
                       cwd=d, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    rrc1 = np.fromfile(os.path.join(d, "S_PAN-1.RRC.RAW"), np.uint16).reshape(L, W)
    rrc2 = np.fromfile(os.path.join(d, "S_PAN-2.RRC.RAW"), np.uint16).reshape(L, W)
    assert np.array_equal(rrc1, oracle_mod.rrc(pan1, kb1)) and np.array_equal(rrc2, oracle_mod.rrc(pan2, kb2))
    t = ctx.stt_correlate(_cuda(pan1), _cuda(pan2), W, L, 0, L, 3, 1600, OV, 0)
    ok = t[:, 2] >= 0.05
    dx = dy = 0.0
    for i in range(3):                       # same accumulation order as stitcher.h:181-198
        if ok[i]:
            dx += t[i, 0]; dy += t[i, 1]
    dx /= ok.sum(); dy /= ok.sum()
    assert ("dx: %.5f, dy: %.5f" % (dx, dy)) in r.stdout
    want, _ = oracle_mod.prestitch(rrc2, dx, dy)
    got = np.fromfile(os.path.join(d, "S_PAN-2.RRC.PRESTT.RAW"), np.uint16).reshape(L, W)
    assert np.array_equal(got, want)

    # ---- step 2: stitch the two PAN strips (RAW out); --fold-cols is halved (main.cpp:189)
    r = subprocess.run([OIP, "stitch", "--width", str(W), "--image1", "S_PAN-1.RRC.RAW", "--image2", "S_PAN-2.RRC.PRESTT.RAW",
                        "--fold-cols", "40", "-o", "stitched-PAN.RAW"], cwd=d, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    st = np.fromfile(os.path.join(d, "stitched-PAN.RAW"), np.uint16).reshape(L, 2 * (W - 20))
    assert np.array_equal(st, oracle_mod.stitch_raw(rrc1, got, 20))
    # ... and as the reference's default output type: a 1-band 16-bit TIFF (imageop.h:299-328)
    r = subprocess.run([OIP, "stitch", "--width", str(W), "--image1", "S_PAN-1.RRC.RAW", "--image2", "S_PAN-2.RRC.PRESTT.RAW",
                        "--fold-cols", "40"], cwd=d, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    tif, tags, big = _tiff.read_tiff_u16(os.path.join(d, "stitched_%dn16b.TIFF" % (2 * (W - 20))))
    assert not big and tags[262] == [1] and np.array_equal(tif, st)
    from PIL import Image
    Image.MAX_IMAGE_PIXELS = None
    with Image.open(os.path.join(d, "stitched_%dn16b.TIFF" % (2 * (W - 20)))) as im:      # independent reader
        assert im.size == (2 * (W - 20), L) and np.array_equal(np.asarray(im), st)

    # ---- step 3: default action on (RRC'd PAN, raw MSS)
    shifts_true = [(2, -1), (1, 1), (-1, -2), (-2, 1)]
    pan, bands = _synth.pan_mss(L, W, shifts_true, seed=6)
    bil = np.concatenate(bands, axis=1)
    pan.tofile(os.path.join(d, "T_PAN.RAW")); bil.tofile(os.path.join(d, "T_MSS.RAW"))
    kbs = [_synth.lut(W // 4, 20 + b) for b in range(4)]
    for b in range(4):
        _csv(os.path.join(d, "MSS.B%d.csv" % (b + 1)), kbs[b])
    args = [OIP, "--width", str(W), "--pan", "T_PAN.RAW", "--mss", "T_MSS.RAW", "--slices", "8", "--ibc-sections", "1",
            "--ibc-threshold", "0", "--lines-section", "3000", "--overlap-lines", "100"]
    for b in range(4):
        args += ["--rrc-msb%d" % (b + 1), "MSS.B%d.csv" % (b + 1)]
    r = subprocess.run(args, cwd=d, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    cbands = [oracle_mod.rrc(bands[b], kbs[b]) for b in range(4)]
    Lm, Wb = L // 4, W // 4
    planes = _cuda(np.stack(cbands, 0))
    sh = ctx.interband_correlate(_cuda(pan), L, 0, L, planes, Lm * Wb, 0, Lm, W, 8, 1, 16000)
    cx, cy = oip.filter_and_fit(sh, 0.0, 5)
    want, nvalid = oracle_mod.align_mss(cbands, cx, cy, 3000, 0, 100, False, 1500)
    tif, tags, _ = _tiff.read_tiff_u16(os.path.join(d, "T_MSS.ALIGNED.TIFF"))
    assert tags[262] == [2] and tags[338] == [2]            # RGB + unassociated alpha, as cv::imwrite(16UC4)
    got = tif[:, :, [2, 1, 0, 3]]                           # stored in OpenCV's on-disk order (BGRA -> RGBA)
    assert np.array_equal(got, want)
    assert "%d lines valid" % nvalid in r.stdout

    # ---- step 4: stitch two aligned 4-channel TIFFs (imageop.h:365-457; -g/-m: :460-567)
    rolled = np.roll(tif, 7, axis=0)                         # a second, different image in file order
    _tiff.write_tiff_u16(os.path.join(d, "T2_MSS.ALIGNED.TIFF"), rolled)
    r = subprocess.run([OIP, "stitch", "--image1", "T_MSS.ALIGNED.TIFF", "--image2", "T2_MSS.ALIGNED.TIFF", "--fold-cols", "12",
                        "-o", "stitched-MSS.TIFF"], cwd=d, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    out, _, _ = _tiff.read_tiff_u16(os.path.join(d, "stitched-MSS.TIFF"))
    fold = 6
    assert np.array_equal(out, np.concatenate([tif[:, :Wb - fold], rolled[:, fold:]], axis=1))
    # GDAL flavour with a band map: band b <- Mat channel map[b]-1, the Mat being (c0..c3) = file samples (2,1,0,3)
    r = subprocess.run([OIP, "stitch", "--image1", "T_MSS.ALIGNED.TIFF", "--image2", "T2_MSS.ALIGNED.TIFF", "--fold-cols", "12",
                        "-g", "-m", "3,2,1,4", "-o", "stitched-MSS-g.TIFF"], cwd=d, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    outg, _, _ = _tiff.read_tiff_u16(os.path.join(d, "stitched-MSS-g.TIFF"))
    mat = np.concatenate([tif[:, :Wb - fold], rolled[:, fold:]], axis=1)[:, :, [2, 1, 0, 3]]
    assert np.array_equal(outg, mat[:, :, [2, 1, 0, 3]])
    r = subprocess.run([OIP, "stitch", "--image1", "T_MSS.ALIGNED.TIFF", "--image2", "T2_MSS.ALIGNED.TIFF", "-g", "-m", "3,2,9,4",
                        "--fold-cols", "12", "-o", "x.TIFF"], cwd=d, env=env, capture_output=True, text=True)
    assert r.returncode == 105
    # ---- the reference's own product format: LZW + horizontal predictor (cv::imwrite, GDAL COMPRESS=LZW PREDICTOR=2).
    # The same stitch with the default policy writes LZW; feeding LZW files back in gives the same pixels.
    env_ref = dict(env)
    del env_ref["OIP_TIFF_COMPRESS"]
    r = subprocess.run([OIP, "stitch", "--image1", "T_MSS.ALIGNED.TIFF", "--image2", "T2_MSS.ALIGNED.TIFF", "--fold-cols", "12",
                        "-o", "stitched-MSS-lzw.TIFF"], cwd=d, env=env_ref, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    tg = _tiff.read_tags(os.path.join(d, "stitched-MSS-lzw.TIFF"))
    assert tg[259] == [5] and tg[317] == [2] and _tiff.read_tags(os.path.join(d, "stitched-MSS.TIFF"))[259] == [1]
    small = os.path.join(d, "lzw-a.TIFF")
    r = subprocess.run([OIP, "stitch", "--image1", "stitched-MSS-lzw.TIFF", "--image2", "stitched-MSS-lzw.TIFF", "--fold-cols", "2",
                        "--tiff-compress", "none", "-o", "lzw-a.TIFF"], cwd=d, env=env_ref, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    twice, _, _ = _tiff.read_tiff_u16(small)
    assert np.array_equal(twice, np.concatenate([out[:, :out.shape[1] - 1], out[:, 1:]], axis=1))


def test_pipelined_default_action_equals_the_step_by_step_one(tmp_path):
    """`oip --pan --mss` runs the default action as a pipeline (reader thread || compute thread || product writers,
    PreProcessor::RunPipelined); OIP_PIPELINE=0 runs the reference's steps one after the other (preproc.h:51-80, :188-222,
    :224-347, :351-425).  Same product files, byte for byte -- with an ODD slice count over two sections, so that a pair of
    correlation units spans the sections (the pipeline has to keep the one-call pairing), with --write-rrcpan (the corrected
    PAN strip goes out block by block behind the RRC kernels) and in both TIFF encodings."""
    W, L = 1024, 33024
    base = str(tmp_path)
    pan, bands = _synth.pan_mss(L, W, [(2, -1), (1, 1), (-1, -2), (-2, 1)], seed=21)
    bil = np.concatenate(bands, axis=1)
    runs = {}
    for comp in ("lzw", "none"):
        for name, pl in (("steps", "0"), ("pipe", "1")):
            d = os.path.join(base, comp + "-" + name)
            os.makedirs(d)
            pan.tofile(os.path.join(d, "T_PAN.RAW")); bil.tofile(os.path.join(d, "T_MSS.RAW"))
            _csv(os.path.join(d, "PAN.csv"), _synth.lut(W, 3))
            for b in range(4):
                _csv(os.path.join(d, "MSS.B%d.csv" % (b + 1)), _synth.lut(W // 4, 20 + b))
            # (OIP_TIFF_CHUNK_MB: the LZW product comes down in 1 MiB blocks, so that the download || encode || write overlap of
            # tiff_rows_from_device runs over many blocks on this small image)
            # The step-by-step run encodes its LZW strips on the host's threads (OIP_TIFF_GPU_LZW=0), the pipelined one on the
            # device (csrc/tifflzw.hip): the byte comparison below is also the device encoder against the host encoder.
            # OIP_TIFF_FORCE_BIG: the LZW products are BigTIFF files (64-bit offsets, as products past 4 GiB are), read back below.
            env = dict(os.environ, LOGFILE=os.path.join(d, "oip.log"), OIP_PIPELINE=pl, OIP_TIFF_COMPRESS=comp, OIP_TIFF_CHUNK_MB="1",
                       OIP_TIFF_GPU_LZW=pl, OIP_TIFF_FORCE_BIG="1" if comp == "lzw" else "0")
            args = [OIP, "--width", str(W), "--pan", "T_PAN.RAW", "--mss", "T_MSS.RAW", "--do-rrc4pan", "--rrc-pan", "PAN.csv", "--write-rrcpan",
                    "--slices", "9", "--ibc-sections", "2", "--ibc-threshold", "0", "--lines-section", "3000", "--overlap-lines", "100"]
            for b in range(4):
                args += ["--rrc-msb%d" % (b + 1), "MSS.B%d.csv" % (b + 1)]
            r = subprocess.run(args, cwd=d, env=env, capture_output=True, text=True)
            assert r.returncode == 0, r.stdout + r.stderr
            assert ("TIMING default_action pipelined=1" in r.stdout) == (pl == "1")
            runs[(comp, name)] = (d, r.stdout)
        for f in ("T_PAN.RRC.RAW", "T_MSS.ALIGNED.TIFF"):
            a = open(os.path.join(runs[(comp, "steps")][0], f), "rb").read()
            b = open(os.path.join(runs[(comp, "pipe")][0], f), "rb").read()
            assert len(a) > 100000 and a == b, (comp, f)
        # the same correlation table and polynomials in the two logs
        pick = lambda out: [ln for ln in out.splitlines() if ln.startswith("|") or "coeff:" in ln]
        ta, tb = pick(runs[(comp, "steps")][1]), pick(runs[(comp, "pipe")][1])
        strip = lambda lines: [ln.split(" ", 2)[-1] if "coeff:" in ln else ln for ln in lines]
        assert len(ta) > 20 and strip(ta) == strip(tb)
    # the LZW product (strips encoded block by block as the lines come down) decodes to the uncompressed product's pixels: it is
    # read back by `oip stitch` (the product's decoder, pinned against the Python decoder and Pillow in tests/test_cli_cpu.py) and
    # re-written uncompressed, which the independent reader of tests/_tiff.py compares (its own LZW decoder needs minutes per MB)
    d = runs[("lzw", "pipe")][0]
    assert _tiff.read_tags(os.path.join(d, "T_MSS.ALIGNED.TIFF"))[259] == [5] and _tiff.read_tags(os.path.join(d, "T_MSS.ALIGNED.TIFF"))[317] == [2]
    with open(os.path.join(d, "T_MSS.ALIGNED.TIFF"), "rb") as f:
        assert f.read(4) == b"II+\x00"                                  # BigTIFF
    r = subprocess.run([OIP, "stitch", "--image1", "T_MSS.ALIGNED.TIFF", "--image2", "T_MSS.ALIGNED.TIFF", "--fold-cols", "2", "--tiff-compress", "none",
                        "-o", "roundtrip.TIFF"], cwd=d, env=dict(os.environ, LOGFILE=os.path.join(d, "oip.log")), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    twice, _, _ = _tiff.read_tiff_u16(os.path.join(d, "roundtrip.TIFF"))
    un, _, _ = _tiff.read_tiff_u16(os.path.join(runs[("none", "pipe")][0], "T_MSS.ALIGNED.TIFF"))
    assert un.any() and np.array_equal(twice, np.concatenate([un[:, :un.shape[1] - 1], un[:, 1:]], axis=1))
    # errors keep their exit codes in the pipeline: a truncated MSS file (size check, preproc.h:552-572) and too many sections
    d = runs[("none", "pipe")][0]
    env = dict(os.environ, LOGFILE=os.path.join(d, "oip.log"))
    r = subprocess.run([OIP, "--width", str(W), "--pan", "T_PAN.RAW", "--mss", "T_MSS.RAW", "--no-rrc4mss", "--ibc-sections", "3"], cwd=d, env=env,
                       capture_output=True, text=True)
    assert r.returncode == 2 and "too many sections" in r.stdout
    with open(os.path.join(d, "T_MSS.RAW"), "r+b") as f:
        f.truncate(bil.nbytes - W * 2 * 4)
    r = subprocess.run([OIP, "--width", str(W), "--pan", "T_PAN.RAW", "--mss", "T_MSS.RAW", "--no-rrc4mss", "--ibc-sections", "1"], cwd=d, env=env,
                       capture_output=True, text=True)
    assert r.returncode == 2 and "PAN file size does not match MSS file size" in r.stdout


def test_default_action_at_the_30000_wide_geometry(ctx, oracle_mod, tmp_path):
    """The C++ host on a strip of the BASELINE width: 30000 columns, 10 slices of 3000 -- the geometry whose inter-band
    correlation runs on the band spectra -- one correlation section of 16000 lines.  The polynomials
    are re-derived in-process through the same C ABI (bit-identical), the aligned image is compared with the oracle."""
    import opticalimageprocessor_amd as oip
    W, L = 30000, 16000
    d = str(tmp_path)
    env = dict(os.environ, LOGFILE=os.path.join(d, "oip.log"), OIP_TIFF_COMPRESS="none")
    pan, bands = _synth.pan_mss(L, W, [(2, -1), (1, 1), (-1, -2), (-2, 1)], seed=8)
    np.concatenate(bands, axis=1).tofile(os.path.join(d, "W_MSS.RAW"))
    pan.tofile(os.path.join(d, "W_PAN.RAW"))
    args = [OIP, "--width", str(W), "--pan", "W_PAN.RAW", "--mss", "W_MSS.RAW", "--ibc-sections", "1", "--no-rrc4mss",
            "--lines-section", "3000", "--overlap-lines", "100"]
    r = subprocess.run(args, cwd=d, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    Lm, Wb = L // 4, W // 4
    planes = _cuda(np.stack(bands, 0))
    sh = ctx.interband_correlate(_cuda(pan), L, 0, L, planes, Lm * Wb, 0, Lm, W, 10, 1, 16000)
    assert (sh[..., 2] >= 0.4).all()                          # every unit clears the reference's threshold
    cx, cy = oip.filter_and_fit(sh, 0.4, 5)
    want, nvalid = oracle_mod.align_mss(bands, cx, cy, 3000, 0, 100, False, 1500)
    tif, tags, _ = _tiff.read_tiff_u16(os.path.join(d, "W_MSS.ALIGNED.TIFF"))
    assert np.array_equal(tif[:, :, [2, 1, 0, 3]], want)
    assert "%d lines valid" % nvalid in r.stdout


def test_fused_task_equals_the_five_command_flow(ctx, tmp_path):
    """`oip task` (everything resident on the GPU, four RAW files in, two TIFFs out) must
    produce exactly the two products of DOC/sample-task.sh's five commands run through the file system."""
    import time
    W, L, OV = 1024, 33024, 64
    d = str(tmp_path)
    env = dict(os.environ, LOGFILE=os.path.join(d, "oip.log"))
    pan1, pan2 = _synth.ccd_pair(L, W, OV, (3, -2), seed=11)
    rng = np.random.default_rng(4)

    def mss_of(pan, shifts):
        # four bands = 4x4 box means of the PAN strip, each displaced by a small shift, + a little noise
        small = pan.astype(np.float64).reshape(L // 4, 4, W // 4, 4).mean(axis=(1, 3))
        bands = [np.roll(small, (sy, sx), (0, 1)) + rng.normal(0, 2.0, small.shape) for sx, sy in shifts]
        return np.concatenate([np.clip(np.rint(b), 0, 65535).astype(np.uint16) for b in bands], axis=1)      # BIL

    mss1 = mss_of(pan1, [(1, 0), (0, 1), (-1, 0), (0, -1)])
    mss2 = mss_of(pan2, [(0, 1), (1, 0), (0, -1), (-1, 0)])
    for name, a in (("A_PAN-1.RAW", pan1), ("A_PAN-2.RAW", pan2), ("A_MSS-1.RAW", mss1), ("A_MSS-2.RAW", mss2)):
        a.tofile(os.path.join(d, name))
    _csv(os.path.join(d, "P1.csv"), _synth.lut(W, 1)); _csv(os.path.join(d, "P2.csv"), _synth.lut(W, 2))
    for c in (1, 2):
        for b in range(4):
            _csv(os.path.join(d, "M%dB%d.csv" % (c, b + 1)), _synth.lut(W // 4, 30 + 4 * c + b))
    stt = ["-s", "3", "-l", "1600", "--stitch-overlap", str(OV), "--stt-threshold", "0.05"]
    ibc = ["--slices", "8", "--ibc-sections", "1", "--ibc-threshold", "0", "--lines-section", "3000", "--overlap-lines", "100"]

    def run(args):
        r = subprocess.run([OIP] + args, cwd=d, env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        return r

    # ---- the reference's flow: five commands, six intermediate files
    t0 = time.time()
    run(["prestitch", "--width", str(W), "--pan1", "A_PAN-1.RAW", "--pan2", "A_PAN-2.RAW", "--rrc1", "P1.csv", "--rrc2", "P2.csv"] + stt)
    # (the ALIGNED.TIFF intermediates carry the reference's LZW + predictor encoding and are read back by `stitch`; the final
    # products are written uncompressed so that the independent Python reader can compare them quickly)
    plain = ["--tiff-compress", "none"]
    run(["stitch", "--width", str(W), "--image1", "A_PAN-1.RRC.RAW", "--image2", "A_PAN-2.RRC.PRESTT.RAW", "--fold-cols", "40", "-o", "ref-PAN.TIFF"] + plain)
    for c, s1 in ((1, "A_PAN-1.RRC.RAW"), (2, "A_PAN-2.RRC.PRESTT.RAW")):
        run(["--width", str(W), "--pan", s1, "--mss", "A_MSS-%d.RAW" % c] + ibc + sum([["--rrc-msb%d" % (b + 1), "M%dB%d.csv" % (c, b + 1)] for b in range(4)], []))
    run(["stitch", "--image1", "A_MSS-1.ALIGNED.TIFF", "--image2", "A_MSS-2.ALIGNED.TIFF", "--fold-cols", "12", "-o", "ref-MSS.TIFF"] + plain)
    run(["stitch", "--image1", "A_MSS-1.ALIGNED.TIFF", "--image2", "A_MSS-2.ALIGNED.TIFF", "--fold-cols", "12", "-g", "-m", "3,2,1,4", "-o", "ref-MSS-g.TIFF"] + plain)
    t_ref = time.time() - t0

    # ---- the fused task
    task = ["task", "--width", str(W), "--pan1", "A_PAN-1.RAW", "--pan2", "A_PAN-2.RAW", "--rrc1", "P1.csv", "--rrc2", "P2.csv",
            "--mss1", "A_MSS-1.RAW", "--mss2", "A_MSS-2.RAW", "--fold-cols-pan", "40", "--fold-cols-mss", "12"] + stt + ibc
    for c in (1, 2):
        for b in range(4):
            task += ["--rrc-mss%d-b%d" % (c, b + 1), "M%dB%d.csv" % (c, b + 1)]
    t0 = time.time()
    run(task + ["--out-pan", "fused-PAN.TIFF", "--out-mss", "fused-MSS.TIFF"] + plain)
    t_fused = time.time() - t0
    run(task + ["--out-pan", "fused-PAN2.TIFF", "--out-mss", "fused-MSS-g.TIFF", "-g", "-m", "3,2,1,4"] + plain)
    for a, b in (("ref-PAN.TIFF", "fused-PAN.TIFF"), ("ref-MSS.TIFF", "fused-MSS.TIFF"), ("ref-MSS-g.TIFF", "fused-MSS-g.TIFF")):
        ia, ta, _ = _tiff.read_tiff_u16(os.path.join(d, a))
        ib, tb, _ = _tiff.read_tiff_u16(os.path.join(d, b))
        assert ia.shape == ib.shape and np.array_equal(ia, ib), (a, b)
        assert ta[262] == tb[262] and ta[277] == tb[277]
    assert ia.any()                                    # not trivially empty
    print("five commands %.2f s, fused task %.2f s" % (t_ref, t_fused))
    # --pan-only: the stitched PAN product alone, RRC of CCD 1 and the resampled CCD-2 lines written straight into the
    # stitched raster (oip_rrc_u16_window + oip_remap_shift_bicubic_u16_window): the same file, byte for byte
    pan_only = ["task", "--pan-only", "--width", str(W), "--pan1", "A_PAN-1.RAW", "--pan2", "A_PAN-2.RAW", "--rrc1", "P1.csv", "--rrc2", "P2.csv",
                "--fold-cols-pan", "40", "--out-pan", "fused-PAN-only.TIFF"] + stt + plain
    run(pan_only)
    assert open(os.path.join(d, "fused-PAN-only.TIFF"), "rb").read() == open(os.path.join(d, "fused-PAN.TIFF"), "rb").read()
    # the same pair in the fp16-accumulate mode (RRC on load in remap_shift8_rrc_kernel<true> against RRC + the fp16 resampling
    # kernel + stitch of the full task): one file again, and not the fp32 one
    run(task + ["--fp16-accumulate", "--out-pan", "fused16-PAN.TIFF", "--out-mss", "fused16-MSS.TIFF"] + plain)
    run(pan_only[:-len(stt + plain) - 1] + ["fused16-PAN-only.TIFF", "--fp16-accumulate"] + stt + plain)
    f16 = open(os.path.join(d, "fused16-PAN.TIFF"), "rb").read()
    assert open(os.path.join(d, "fused16-PAN-only.TIFF"), "rb").read() == f16
    assert f16 != open(os.path.join(d, "fused-PAN.TIFF"), "rb").read()
    # argument errors keep the CLI's codes
    r = subprocess.run([OIP, "task", "--pan1", "A_PAN-1.RAW"], cwd=d, env=env, capture_output=True, text=True)
    assert r.returncode == 106


def test_multigpu_host_on_one_gpu_equals_the_plain_cli(tmp_path):
    """`oip --gpus N` / `oip prestitch --gpus N` (csrc/oip_multigpu.hpp: one context, host thread and RCCL communicator per
    GPU, block-offset file reads, grouped ncclSend/ncclRecv, ncclAllGather).  This box has one GPU, so N = 1 is what can
    run here: the whole RCCL host path with a single rank must reproduce the plain CLI's product files byte for byte.
    (The N > 1 plan is checked against dist.py's in tests/test_dist_cpu.py, dist.py itself on gloo.)"""
    import shutil
    W, L, OV = 1024, 33024, 64
    base = str(tmp_path)
    pan1, pan2 = _synth.ccd_pair(L, W, OV, (3, -2), seed=15)
    pan, bands = _synth.pan_mss(L, W, [(2, -1), (1, 1), (-1, -2), (-2, 1)], seed=16)
    dirs = {}
    for name in ("plain", "node"):
        d = os.path.join(base, name)
        os.makedirs(d)
        dirs[name] = d
        pan1.tofile(os.path.join(d, "S_PAN-1.RAW")); pan2.tofile(os.path.join(d, "S_PAN-2.RAW"))
        pan.tofile(os.path.join(d, "T_PAN.RAW")); np.concatenate(bands, axis=1).tofile(os.path.join(d, "T_MSS.RAW"))
        _csv(os.path.join(d, "PAN-1.csv"), _synth.lut(W, 1)); _csv(os.path.join(d, "PAN-2.csv"), _synth.lut(W, 2))
        for b in range(4):
            _csv(os.path.join(d, "MSS.B%d.csv" % (b + 1)), _synth.lut(W // 4, 20 + b))
    for name, extra in (("plain", []), ("node", ["--gpus", "1"])):
        d = dirs[name]
        env = dict(os.environ, LOGFILE=os.path.join(d, "oip.log"), OIP_TIFF_COMPRESS="none")
        r = subprocess.run([OIP, "prestitch", "--width", str(W), "--pan1", "S_PAN-1.RAW", "--pan2", "S_PAN-2.RAW", "--rrc1", "PAN-1.csv",
                            "--rrc2", "PAN-2.csv", "-s", "3", "-l", "1600", "--stitch-overlap", str(OV), "--stt-threshold", "0.05"] + extra,
                           cwd=d, env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        args = [OIP, "--width", str(W), "--pan", "T_PAN.RAW", "--mss", "T_MSS.RAW", "--slices", "8", "--ibc-sections", "1",
                "--ibc-threshold", "0", "--lines-section", "3000", "--overlap-lines", "100"] + extra
        for b in range(4):
            args += ["--rrc-msb%d" % (b + 1), "MSS.B%d.csv" % (b + 1)]
        r = subprocess.run(args, cwd=d, env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
    for f in ("S_PAN-1.RRC.RAW", "S_PAN-2.RRC.RAW", "S_PAN-2.RRC.PRESTT.RAW", "T_MSS.ALIGNED.TIFF"):
        a = open(os.path.join(dirs["plain"], f), "rb").read()
        b = open(os.path.join(dirs["node"], f), "rb").read()
        assert len(a) > 1000 and a == b, f
    # a rank that fails inside the communication phase (OIP_FAULT_INJECT, oip_multigpu.hpp): the communicators are aborted
    # under the guard, the process leaves with that rank's message and exit code 2 -- no hang, no crash
    d = dirs["node"]
    env = dict(os.environ, LOGFILE=os.path.join(d, "oip.log"), OIP_TIFF_COMPRESS="none", OIP_FAULT_INJECT="0:allgather")
    r = subprocess.run(args, cwd=d, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "injected failure on GPU 0 at allgather" in r.stdout, r.stdout + r.stderr
