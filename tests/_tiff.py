"""Minimal reader for uncompressed little-endian TIFF / BigTIFF with contiguous 16-bit strips
(what csrc/oip_tiff.hpp writes).  Tests only."""
import struct

import numpy as np


def read_tiff_u16(path):
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
    w, h, spp = tags[256][0], tags[257][0], tags[277][0]
    assert tags[258] == [16] * spp and tags[259] == [1] and tags[284] == [1] and tags[339] == [1] * spp
    data = b"".join(buf[o:o + c] for o, c in zip(tags[273], tags[279]))
    img = np.frombuffer(data, np.uint16).reshape(h, w, spp) if spp > 1 else np.frombuffer(data, np.uint16).reshape(h, w)
    return img, tags, big


def write_tiff_u16(path, arr):
    """Minimal classic-TIFF writer (one strip, uncompressed, chunky) for test inputs; samples in file order."""
    import struct
    a = np.ascontiguousarray(arr, dtype="<u2")
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, s = a.shape
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
