"""Minimal reader / writer for little-endian TIFF / BigTIFF with 16-bit chunky strips, uncompressed or LZW
with predictor 1 or 2 (what csrc/oip_tiff.hpp, cv::imwrite and GDAL's GTiff driver write).  Tests only: an
implementation of TIFF 6.0 section 13 (LZW) and section 14 (predictor) independent of the C++ codec."""
import struct

import numpy as np


def lzw_decode(data: bytes) -> bytes:
    """TIFF LZW: MSB-first 9..12-bit codes, Clear 256, EOI 257, early change"""
    table = [bytes([i]) for i in range(256)] + [b"", b""]
    out = bytearray()
    width, acc, nbits, old = 9, 0, 0, None
    for byte in data:
        acc = (acc << 8) | byte
        nbits += 8
        while nbits >= width:
            code = (acc >> (nbits - width)) & ((1 << width) - 1)
            nbits -= width
            if code == 257:
                return bytes(out)
            if code == 256:
                table = table[:258]
                width, old = 9, None
                continue
            if old is None:
                entry = table[code]
            elif code < len(table):
                entry = table[code]
                table.append(old + entry[:1])
            else:
                assert code == len(table), "corrupt LZW stream"
                entry = old + old[:1]
                table.append(entry)
            out += entry
            old = entry
            if len(table) >= (1 << width) - 1 and width < 12:
                width += 1
    return bytes(out)


def lzw_encode(data: bytes) -> bytes:
    codes = {bytes([i]): i for i in range(256)}
    out, acc, nbits = bytearray(), 0, 0
    width, nxt = 9, 258

    def put(code, w):
        nonlocal acc, nbits
        acc = (acc << w) | code
        nbits += w
        while nbits >= 8:
            out.append((acc >> (nbits - 8)) & 0xFF)
            nbits -= 8

    put(256, width)
    ent = b""
    for b in data:
        cur = ent + bytes([b])
        if cur in codes:
            ent = cur
            continue
        put(codes[ent], width)
        codes[cur] = nxt
        nxt += 1
        ent = bytes([b])
        if nxt == 4094:
            put(256, width)
            codes = {bytes([i]): i for i in range(256)}
            width, nxt = 9, 258
        elif nxt == (1 << width) and width < 12:
            width += 1
    if ent:
        put(codes[ent], width)
        nxt += 1
        if nxt == 4094:
            put(256, width)
            width = 9
        elif nxt == (1 << width) and width < 12:
            width += 1
    put(257, width)
    if nbits:
        out.append((acc << (8 - nbits)) & 0xFF)
    return bytes(out)


def read_tags(path):
    """directory only (no pixel decoding)"""
    return read_tiff_u16(path, tags_only=True)[1]


def read_tiff_u16(path, tags_only=False):
    with open(path, "rb") as f:
        buf = f.read()
    assert buf[:2] == b"II"
    ver = struct.unpack_from("<H", buf, 2)[0]
    big = ver == 43
    assert ver in (42, 43)
    if big:
        assert struct.unpack_from("<HH", buf, 4) == (8, 0)
        ifd = struct.unpack_from("<Q", buf, 8)[0]
        n = struct.unpack_from("<Q", buf, ifd)[0]
        ent, esz, osz, ofmt = ifd + 8, 20, 8, "<Q"
    else:
        ifd = struct.unpack_from("<I", buf, 4)[0]
        n = struct.unpack_from("<H", buf, ifd)[0]
        ent, esz, osz, ofmt = ifd + 2, 12, 4, "<I"
    tsize = {3: 2, 4: 4, 16: 8}
    tfmt = {3: "<H", 4: "<I", 16: "<Q"}
    tags = {}
    for i in range(n):
        o = ent + i * esz
        tid, typ = struct.unpack_from("<HH", buf, o)
        cnt = struct.unpack_from(ofmt, buf, o + 4)[0]
        voff = o + 4 + osz
        if cnt * tsize[typ] > osz:
            voff = struct.unpack_from(ofmt, buf, voff)[0]
        tags[tid] = [struct.unpack_from(tfmt[typ], buf, voff + k * tsize[typ])[0] for k in range(cnt)]
    if tags_only:
        return None, tags, big
    w, h, spp = tags[256][0], tags[257][0], tags.get(277, [1])[0]
    assert tags[258] == [16] * spp and tags[259][0] in (1, 5) and tags.get(284, [1]) == [1] and tags.get(339, [1] * spp) == [1] * spp
    strips = [buf[o:o + c] for o, c in zip(tags[273], tags[279])]
    if tags[259][0] == 5:
        strips = [lzw_decode(st) for st in strips]
    data = b"".join(strips)
    img = np.frombuffer(data, np.uint16).reshape(h, w, spp).copy()
    if tags.get(317, [1])[0] == 2:
        img = np.cumsum(img.astype(np.uint32), axis=1).astype(np.uint16)          # per channel along the row, mod 2^16
    if spp == 1:
        img = img.reshape(h, w)
    return img, tags, big


def write_tiff_u16(path, arr, lzw=False, predictor=1, rows_per_strip=None):
    """Minimal classic-TIFF writer (chunky; uncompressed single strip, or LZW strips with predictor 1 / 2) for
    test inputs; samples in file order."""
    import struct
    a = np.ascontiguousarray(arr, dtype="<u2")
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, s = a.shape
    if lzw:
        return _write_tiff_lzw(path, a, predictor, rows_per_strip or h)
    data = a.tobytes()
    extra = b""
    off_extra = 8 + len(data)

    def arr_tag(vals):
        nonlocal extra
        o = off_extra + len(extra)
        extra += struct.pack("<%dH" % len(vals), *vals)
        return o

    tags = [(256, 4, 1, w), (257, 4, 1, h)]
    tags.append((258, 3, s, arr_tag([16] * s) if s > 2 else 16 | (16 << 16) * (s == 2)))
    tags += [(259, 3, 1, 1), (262, 3, 1, 2 if s == 4 else 1), (273, 4, 1, 8), (277, 3, 1, s), (278, 4, 1, h),
             (279, 4, 1, len(data)), (284, 3, 1, 1)]
    if s == 4:
        tags.append((338, 3, 1, 2))
    tags.append((339, 3, s, arr_tag([1] * s) if s > 2 else 1 | (1 << 16) * (s == 2)))
    if len(extra) & 1:
        extra += b"\0"
    ifd = off_extra + len(extra)
    with open(path, "wb") as f:
        f.write(b"II" + struct.pack("<HI", 42, ifd))
        f.write(data)
        f.write(extra)
        f.write(struct.pack("<H", len(tags)))
        for t in tags:
            f.write(struct.pack("<HHII", *t))
        f.write(struct.pack("<I", 0))


def _write_tiff_lzw(path, a, predictor, rps):
    h, w, s = a.shape
    if predictor == 2:
        d = a.astype(np.int64)
        d[:, 1:, :] = d[:, 1:, :] - a[:, :-1, :].astype(np.int64)
        a = (d & 0xFFFF).astype("<u2")
    strips = [lzw_encode(a[r:r + rps].tobytes()) for r in range(0, h, rps)]
    body, offs = b"", []
    for st in strips:
        if len(body) & 1:
            body += b"\0"
        offs.append(8 + len(body))
        body += st
    if len(body) & 1:
        body += b"\0"
    extra = b""
    base = 8 + len(body)

    def arr_tag(vals, fmt="H"):
        nonlocal extra
        o = base + len(extra)
        extra += struct.pack("<%d%s" % (len(vals), fmt), *vals)
        if len(extra) & 1:
            extra += b"\0"
        return o

    n = len(strips)
    tags = [(256, 4, 1, w), (257, 4, 1, h),
            (258, 3, s, arr_tag([16] * s) if s > 2 else 16 | (16 << 16) * (s == 2)),
            (259, 3, 1, 5), (262, 3, 1, 2 if s == 4 else 1),
            (273, 4, n, arr_tag(offs, "I") if n > 1 else offs[0]), (277, 3, 1, s), (278, 4, 1, rps),
            (279, 4, n, arr_tag([len(st) for st in strips], "I") if n > 1 else len(strips[0])), (284, 3, 1, 1),
            (317, 3, 1, predictor)]
    if s == 4:
        tags.append((338, 3, 1, 2))
    tags.append((339, 3, s, arr_tag([1] * s) if s > 2 else 1 | (1 << 16) * (s == 2)))
    ifd = base + len(extra)
    with open(path, "wb") as f:
        f.write(b"II" + struct.pack("<HI", 42, ifd))
        f.write(body)
        f.write(extra)
        f.write(struct.pack("<H", len(tags)))
        for t in tags:
            f.write(struct.pack("<HHII", *t))
        f.write(struct.pack("<I", 0))
