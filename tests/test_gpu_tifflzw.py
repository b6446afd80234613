"""The device LZW strip encoder (csrc/tifflzw.hip, oip_tiff_lzw_strips_u16) against an independent restatement of TIFF 6.0
sections 13 (LZW as libtiff writes it) and 14 (horizontal predictor): tests/_tiff.py::lzw_encode -- a dictionary of byte strings,
nothing in common with the open-addressing coders of the product -- on the predictor-2 byte stream of every strip, bit for bit.
The reference encodes these strips on the host through cv::imwrite (preproc.h:167-185) and GDAL (imageop.h:460-567)."""
import numpy as np
import pytest
import torch

import _tiff

pytestmark = pytest.mark.gpu


def _image(kind, rows, width, spp, seed):
    rng = np.random.default_rng(seed)
    n = rows * width * spp
    if kind == "noise12":                      # sensor-like: barely compresses, the table fills every ~5 KB
        a = np.clip(rng.normal(1800.0, 300.0, n), 64, 4095).astype(np.uint16)
    elif kind == "noise16":                    # nothing repeats: a code per byte, ClearCode every 3836 bytes
        a = rng.integers(0, 65536, n, dtype=np.uint16)
    elif kind == "ramp":                       # constant differences: long matches, slow code-width growth
        a = (np.arange(n, dtype=np.uint64) * 3 % 65536).astype(np.uint16)
    elif kind == "constant":
        a = np.full(n, 4242, dtype=np.uint16)
    else:                                      # smooth scene + a little noise
        x = np.arange(n, dtype=np.float64)
        a = (2000 + 800 * np.sin(x / 977.0) + rng.integers(0, 4, n)).astype(np.uint16)
    return a.reshape(rows, width * spp)


def _predict(block, spp):
    """TIFF predictor 2 on 16-bit samples, row by row, as the little-endian byte stream the coder sees"""
    d = block.astype(np.uint16).copy()
    d[:, spp:] = (block[:, spp:].astype(np.int32) - block[:, :-spp].astype(np.int32)).astype(np.uint16)
    return d.astype("<u2").tobytes()


@pytest.mark.parametrize("rows,width,spp,rps,kind", [
    (40, 720, 4, 11, "noise12"),               # 63 KB strips of 11 rows, the last one short
    (9, 7500, 4, 1, "scene"),                  # the product's own strips: one 60000-byte row each
    (6, 2048, 1, 3, "ramp"),                   # one sample per pixel
    (5, 1000, 4, 2, "constant"),
    (3, 16384, 4, 1, "noise16"),               # 128 KB strips: ~34 table generations each
    (150, 64, 4, 1, "noise12"),                # 150 small strips: three workgroups of lanes
    (4, 33, 1, 4, "scene"),                    # odd widths, an odd number of bytes per strip is likely
])
def test_device_lzw_strips_equal_the_independent_coder(ctx, rows, width, spp, rps, kind):
    oip = ctx
    img = _image(kind, rows, width, spp, seed=rows * 31 + width)
    d_img = torch.from_numpy(img.view(np.int16)).cuda()
    cap = oip.tiff_lzw_worst_bytes(rows, width, spp, rps)
    d_pay = torch.full((cap,), 0xEE, dtype=torch.uint8, device="cuda")
    off, ln, total = oip.tiff_lzw_strips(d_img, rows, width, spp, rps, d_pay)
    pay = d_pay.cpu().numpy()
    nstrips = (rows + rps - 1) // rps
    assert len(off) == nstrips and off[0] == 0 and total == off[-1] + ln[-1] and total <= cap
    pos = 0
    for k in range(nstrips):
        want = _tiff.lzw_encode(_predict(img[k * rps:(k + 1) * rps], spp))
        assert off[k] % 2 == 0 and off[k] == pos + (pos & 1), k
        got = pay[int(off[k]):int(off[k] + ln[k])].tobytes()
        assert got == want, (k, len(got), len(want))
        if ln[k] % 2 and k + 1 < nstrips:
            assert pay[int(off[k] + ln[k])] == 0          # the pad byte between an odd strip and the next
        pos = int(off[k] + ln[k])
    # and the streams decode (the independent decoder) to the predictor bytes
    k = nstrips - 1
    assert _tiff.lzw_decode(pay[int(off[k]):int(off[k] + ln[k])].tobytes()) == _predict(img[k * rps:(k + 1) * rps], spp)


def test_device_lzw_rejects_bad_arguments(ctx):
    oip = ctx
    d_img = torch.zeros(64 * 4, dtype=torch.int16, device="cuda")
    d_pay = torch.zeros(16, dtype=torch.uint8, device="cuda")
    with pytest.raises(ValueError):
        oip.tiff_lzw_strips(d_img, 1, 64, 3, 1, d_pay)               # spp 3
    with pytest.raises(ValueError):
        oip.tiff_lzw_strips(d_img, 1, 64, 4, 1, d_pay)               # payload too small


def _pack(strips):
    """strips one behind the other at even offsets, as a TIFF file holds them"""
    blob, off, ln = bytearray(b"\x00" * 6), [], []              # some bytes in front: the strips need not start the buffer
    for s in strips:
        if len(blob) & 1:
            blob.append(0)
        off.append(len(blob)); ln.append(len(s))
        blob += s
    return np.frombuffer(bytes(blob), dtype=np.uint8).copy(), off, ln


@pytest.mark.parametrize("rows,width,spp,rps,kind,pred", [
    (40, 720, 4, 11, "noise12", 2),
    (6, 7500, 4, 1, "scene", 2),
    (6, 2048, 1, 3, "ramp", 2),
    (5, 1000, 4, 2, "constant", 2),            # one long run: nothing but KwKwK-style growth
    (2, 16384, 4, 1, "noise16", 1),            # no predictor; many table generations
    (130, 64, 4, 1, "noise12", 2),
    (4, 33, 1, 4, "scene", 1),
])
def test_device_lzw_decoder_reads_the_independent_coders_strips(ctx, rows, width, spp, rps, kind, pred):
    """strips written by tests/_tiff.py::lzw_encode (an independent coder) decode on the device to the image, predictor undone"""
    img = _image(kind, rows, width, spp, seed=rows * 17 + width)
    strips = []
    for k in range((rows + rps - 1) // rps):
        block = img[k * rps:(k + 1) * rps]
        strips.append(_tiff.lzw_encode(_predict(block, spp) if pred == 2 else block.astype("<u2").tobytes()))
    blob, off, ln = _pack(strips)
    d_file = torch.from_numpy(blob).cuda()
    d_img = torch.full((rows, width * spp), -1, dtype=torch.int16, device="cuda")
    ctx.tiff_lzw_decode(d_file, off, ln, rows, width, spp, rps, pred, d_img)
    assert np.array_equal(d_img.cpu().numpy().view(np.uint16), img)


def test_device_lzw_round_trip_at_product_width(ctx):
    """the device encoder's strips through the device decoder: 300 rows of a 7500 x 4 product"""
    rows, width, spp = 300, 7500, 4
    img = _image("noise12", rows, width, spp, seed=5)
    d_img = torch.from_numpy(img.view(np.int16)).cuda()
    d_pay = torch.empty(ctx.tiff_lzw_worst_bytes(rows, width, spp, 1), dtype=torch.uint8, device="cuda")
    off, ln, total = ctx.tiff_lzw_strips(d_img, rows, width, spp, 1, d_pay)
    d_back = torch.zeros_like(d_img)
    ctx.tiff_lzw_decode(d_pay, off, ln, rows, width, spp, 1, 2, d_back)
    assert torch.equal(d_back, d_img) and total < img.nbytes * 1.2


def test_device_lzw_decoder_rejects_corrupt_strips(ctx):
    """the rules of the host decoder (csrc/oip_tiff.hpp::lzw_decode): a first code >= 256, a code beyond the table and a strip that
    does not decode to its rows are errors that name the strip; a stream cut before EndOfInformation is read as far as it goes"""
    rows, width, spp = 2, 64, 1
    img = _image("ramp", rows, width, spp, seed=1)
    good = [_tiff.lzw_encode(_predict(img[k:k + 1], spp)) for k in range(rows)]

    def codes(vals, w=9):
        acc, n, out = 0, 0, bytearray()
        for v in vals:
            acc = (acc << w) | v; n += w
            while n >= 8:
                out.append((acc >> (n - 8)) & 255); n -= 8
        if n:
            out.append((acc << (8 - n)) & 255)
        return bytes(out)

    d_img = torch.zeros((rows, width), dtype=torch.int16, device="cuda")
    for bad, what in ((codes([256, 300, 257]), "bad first code"), (codes([256, 65, 400, 257]), "code beyond the table"),
                      (good[1][:len(good[1]) // 2], "decodes to"), (codes([256, 65, 66, 257]), "decodes to")):
        blob, off, ln = _pack([good[0], bad])
        with pytest.raises(RuntimeError, match=what + ".*strip 1|strip 1.*" + what):
            ctx.tiff_lzw_decode(torch.from_numpy(blob).cuda(), off, ln, rows, width, spp, 1, 2, d_img)
    blob, off, ln = _pack(good)
    with pytest.raises(ValueError):
        ctx.tiff_lzw_decode(torch.from_numpy(blob).cuda(), off, [ln[0], len(blob)], rows, width, spp, 1, 2, d_img)     # strip outside the buffer
    # a stream without EndOfInformation that carries all its bytes is fine (libtiff tolerates it)
    full = _tiff.lzw_encode(_predict(img[1:2], spp))
    blob, off, ln = _pack([good[0], full[:-1]])
    try:
        ctx.tiff_lzw_decode(torch.from_numpy(blob).cuda(), off, ln, rows, width, spp, 1, 2, d_img)
        assert np.array_equal(d_img.cpu().numpy().view(np.uint16), img)
    except RuntimeError as e:                      # (cutting the last byte may also cut the last code: then the strip is short)
        assert "decodes to" in str(e)
