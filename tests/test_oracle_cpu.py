"""CPU suite, part 1: the oracle against the golden vectors and analytic known answers.

The reference ships no tests or fixtures.  What pins the oracle:
  * tests/golden/rrc_reference.npz -- outputs of the reference's own InplaceRRC lines
    (oracle/_ref, built by oracle/Makefile from /root/reference where it lies);
  * oracle/_ref itself when present (this container), on fresh random inputs;
  * analytic known answers for the OpenCV-restating parts (tests/golden/known_answers.json).
"""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ka():
    with open(os.path.join(GOLD, "known_answers.json")) as f:
        return json.load(f)


def test_rrc_restatement_matches_reference_golden(oracle_mod):
    g = np.load(os.path.join(GOLD, "rrc_reference.npz"))
    assert np.array_equal(oracle_mod.rrc(g["src"], g["kb"]), g["dst"])
    assert np.array_equal(oracle_mod.rrc(g["src"], g["kb"], threads=3), g["dst"])


def test_rrc_restatement_matches_reference_build(oracle_mod):
    if oracle_mod.ref_lib() is None:
        pytest.skip("oracle/_ref not built here (no /root/reference): golden fixture test covers it")
    rng = np.random.default_rng(11)
    for w, h in [(4096, 33), (3001, 7), (12288, 4)]:
        img = rng.integers(0, 65536, (h, w), dtype=np.uint16)
        kb = np.stack([rng.uniform(-3, 3, w), rng.uniform(-70000, 70000, w)], 1)
        assert np.array_equal(oracle_mod.rrc(img, kb), oracle_mod.rrc_reference(img, kb))


def test_config1_rrc_4096x8192_restatement_equals_reference_loop(oracle_mod):
    """BASELINE config 1 (CPU plumbing): 4096-col x 8192-line single-band strip, RRC only, through the
    restatement (1 thread and threaded) and through the reference's own compiled loop."""
    from opticalimageprocessor_amd import synth
    W, L = 4096, 8192
    img = np.random.default_rng(41).integers(0, 65536, (L, W), dtype=np.uint16)
    kb = synth.lut(W)
    a = oracle_mod.rrc(img, kb)
    assert np.array_equal(a, oracle_mod.rrc(img, kb, threads=4))
    # k in [0.9, 1.1], b in [-8, 8]: spot-check the arithmetic itself on a column sample
    cols = np.arange(0, W, 97)
    want = (kb[cols, 0] * img[:, cols].astype(np.float64) + kb[cols, 1]).astype(np.int64) & 0xFFFF
    assert np.array_equal(a[:, cols], want.astype(np.uint16))
    if oracle_mod.ref_lib() is not None:
        assert np.array_equal(a, oracle_mod.rrc_reference(img, kb))


def test_rrc_wrap_examples(oracle_mod, ka):
    for v, want in ka["rrc_wrap_examples"].items():
        img = np.zeros((1, 1), np.uint16)
        assert oracle_mod.rrc(img, np.array([[0.0, float(v)]]))[0, 0] == want


def test_bicubic_coefficients(oracle_mod, ka):
    for t, want in ka["bicubic_coeffs_f32"].items():
        got = oracle_mod.interpolate_cubic(float(t))
        assert np.allclose(got, np.array(want, np.float32), rtol=0, atol=2e-9), (t, got)
    tab = oracle_mod.bicubic_tab()
    assert tab.shape == (1024, 16)
    # phase (fy=0, fx=0) is the identity kernel; every kernel sums to ~1
    assert np.array_equal(tab[0], np.outer([0, 1, 0, 0], [0, 1, 0, 0]).reshape(-1).astype(np.float32))
    assert np.abs(tab.sum(1) - 1).max() < 1e-6


def test_optimal_dft_size(ka):
    from oracle import phasecorr as pc
    for n, want in ka["optimal_dft_size"].items():
        assert pc.optimal_dft_size(int(n)) == want


def test_resize_x4_phases_and_constant(oracle_mod, ka):
    # a ramp resized x4: sample dx sits at x = (dx+0.5)/4 - 0.5 and is the 4-tap sum at the
    # published phase (the A=-0.75 kernel does not reproduce ramps exactly: +-0.041)
    src = np.tile(np.arange(64, dtype=np.float32), (8, 1))
    out = oracle_mod.resize_cubic(src, 256, 32)
    x = (np.arange(256) + 0.5) * 0.25 - 0.5
    assert np.abs(out[4, 8:-8] - x[8:-8]).max() < 0.0411
    for dx in (8, 9, 10, 11, 100, 201):
        sx = int(np.floor(x[dx]))
        c = oracle_mod.interpolate_cubic(float(x[dx] - sx))
        assert abs(out[4, dx] - float(np.dot(c.astype(np.float64), np.arange(sx - 1, sx + 3)))) < 1e-4
    frac = x - np.floor(x)
    assert np.allclose(sorted(set(np.round(frac, 6))), sorted(ka["resize_x4_phases"]))
    const = oracle_mod.resize_cubic(np.full((9, 9), 5.0, np.float32), 36, 36)
    assert np.abs(const - 5).max() < 1e-5


def test_remap_integer_shift_is_copy_with_zero_border(oracle_mod):
    rng = np.random.default_rng(3)
    src = rng.integers(0, 65536, (50, 40), dtype=np.uint16)
    mx, my = np.meshgrid(np.arange(40, dtype=np.float32) + 2, np.arange(50, dtype=np.float32) + 3)
    d = oracle_mod.remap_cubic(src, mx, my)
    assert np.array_equal(d[:47, :38], src[3:, 2:])
    assert (d[47:] == 0).all() and (d[:, 38:] == 0).all()


def test_remap_constant_stays_constant_inside(oracle_mod):
    src = np.full((60, 60), 1234, np.uint16)
    mx, my = np.meshgrid(np.arange(60, dtype=np.float32) + 0.37, np.arange(60, dtype=np.float32) - 0.81)
    d = oracle_mod.remap_cubic(src, mx, my)
    assert (d[3:-3, 3:-3] == 1234).all()


def test_prestitch_structure(oracle_mod):
    """dy = 0, dx integer: every section body is a shifted copy; seams lose no lines."""
    rng = np.random.default_rng(8)
    src = rng.integers(0, 65536, (700, 32), dtype=np.uint16)
    dst, off = oracle_mod.prestitch(src, 2.0, 0.0, 300, 327)
    assert dst.shape == src.shape and off == 700 - 1
    assert np.array_equal(dst[:690, :30], src[:690, 2:])
    with pytest.raises(ValueError):
        oracle_mod.prestitch(src[:300], 1.0, 1.0, 300, 327)          # imageop.h:242-244


def test_prestitch_seam_artefact_is_reproduced(oracle_mod):
    """App.B-3: with dy > 0 the last body line of every full section has its 4th tap outside
    the section buffer, so it differs from what an unsectioned remap would give."""
    rng = np.random.default_rng(9)
    src = rng.integers(1000, 3000, (700, 16), dtype=np.uint16)
    dst, _ = oracle_mod.prestitch(src, 0.0, 0.5, 300, 327)
    mx, my = np.meshgrid(np.arange(16, dtype=np.float32), np.arange(700, dtype=np.float32) + 0.5)
    whole = oracle_mod.remap_cubic(src, mx, my)
    step = 300 - 1
    same = (dst == whole).all(axis=1)
    # the last body line of section 0 loses its 4th tap, the first body line of section 1
    # (0 <= dy < 1: tap row -1) loses its 1st; every other line is unaffected
    assert not same[step - 1] and not same[step] and same[step - 2] and same[step + 1]


def test_stitch_raw_layout(oracle_mod):
    left = np.arange(5 * 10, dtype=np.uint16).reshape(5, 10)
    right = left + 1000
    out = oracle_mod.stitch_raw(left, right, 3)
    assert out.shape == (5, 14)
    assert np.array_equal(out[:, :7], left[:, :7]) and np.array_equal(out[:, 7:], right[:, 3:])


def test_split_mss(oracle_mod):
    bil = np.arange(3 * 16, dtype=np.uint16).reshape(3, 16)
    bands = oracle_mod.split_mss(bil)
    for b in range(4):
        assert np.array_equal(bands[b], bil[:, 4 * b:4 * b + 4])


def test_align_identity_coefficients(oracle_mod):
    rng = np.random.default_rng(4)
    bands = [rng.integers(0, 65536, (500, 24), dtype=np.uint16) for _ in range(4)]
    out, n = oracle_mod.align_mss(bands, np.zeros((4, 2)), np.zeros((4, 3)), 300, 0, 40, False, 100)
    assert n == 500 - 40 and out.shape == (460, 24, 4)
    for b in range(4):
        assert np.array_equal(out[:, :, b], bands[b][40:])
    out, n = oracle_mod.align_mss(bands, np.zeros((4, 2)), np.zeros((4, 3)), 300, 0, 40, True, 100)
    assert n == 500 and np.array_equal(out[:, :, 2], bands[2])


def test_phase_correlate_oracle_recovers_shifts():
    import _synth
    from oracle import phasecorr as pc
    sc = _synth.scene(460, 260, seed=4)
    a = sc[20:420, 20:220].astype(np.float32)
    for sx, sy in [(3, 0), (0, -4), (2, 5)]:
        b = sc[20 - sy:420 - sy, 20 - sx:220 - sx].astype(np.float32)
        (dx, dy), r = pc.phase_correlate(a, b)
        assert abs(dx - sx) < 0.6 and abs(dy - sy) < 0.6 and r > 0.5
    (dx, dy), r = pc.phase_correlate(a, a)
    assert abs(dx) < 1e-6 and abs(dy) < 1e-6 and abs(r - 1) < 1e-3


def test_rrc_param_file_loader_oracle(oracle_mod, tmp_path):
    p = tmp_path / "rrc.csv"
    p.write_text("1\n4\n0\n1.0 , 0.5\n0.9,-2\n 1.1 ,3.25\n1,0\n")
    kb = oracle_mod.load_rrc_param_file(str(p), 4)
    assert np.allclose(kb, [[1, .5], [.9, -2], [1.1, 3.25], [1, 0]])
    with pytest.raises(RuntimeError, match="expected 5 lines"):
        oracle_mod.load_rrc_param_file(str(p), 5)


def test_merge_subimages_restatement(oracle_mod):
    """aux_separator.h:341-393 for uncompressed frames, against the obvious numpy formulation: byte-swap every
    word, put the hparts sub-images of a row side by side, stack the rows of sub-images"""
    rng = np.random.default_rng(8)
    for shape in [(5, 8, 16, 24), (2, 3, 5, 7), (1, 1, 1, 1)]:
        tiles = rng.integers(0, 65536, shape).astype(np.uint16)
        got = oracle_mod.merge_subimages_be16(tiles)
        want = tiles.byteswap().transpose(0, 2, 1, 3).reshape(shape[0] * shape[2], shape[1] * shape[3])
        assert np.array_equal(got, want)
    # the reference frame geometry: (4 PAN + 1 MSS) x 8 sub-images of 256 x 1536 -> 1280 lines of 12288 pixels
    tiles = rng.integers(0, 65536, (5, 8, 256, 1536)).astype(np.uint16)
    got = oracle_mod.merge_subimages_be16(tiles)
    assert got.shape == (1280, 12288) and got[300, 1536 * 3 + 5] == int(tiles[1, 3, 44, 5]).to_bytes(2, "big")[1] * 256 + int(tiles[1, 3, 44, 5]).to_bytes(2, "big")[0]
