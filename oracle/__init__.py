"""oracle -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker.  The product (``opticalimageprocessor_amd``)
never imports it and has no CPU fallback.

Pieces
  * ``liboiporacle.so``  (oip_oracle.c)  -- RRC, RRC CSV loader, MSS split, cv::remap cubic,
    PreStitch/SectionaryRemap, DoInterBandAlignment, StitchBigRaw, cv::resize cubic.
  * ``phasecorr.py``     -- numpy restatement of cv::phaseCorrelate + the reference's section
    geometry, filtering and polynomial fit.
  * ``_ref/libref_rrc.so`` -- the reference's own InplaceRRC lines compiled in place
    (see Makefile); pins ``orc_inplace_rrc``.

Parity status: RRC pinned (bit-exact vs ``_ref`` and tests/golden); everything that lives in
OpenCV/NumCpp is PARITY UNPINNED (library absent, the reference has no tests or fixtures).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

_u16p = C.POINTER(C.c_uint16)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)


def build(quiet: bool = True) -> None:
    """Compile liboiporacle.so (and _ref/ when /root/reference is present)."""
    subprocess.run(["make", "-C", _HERE], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboiporacle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_inplace_rrc.argtypes = [_u16p, C.c_int, C.c_int, C.c_void_p]
        L.orc_inplace_rrc_mt.argtypes = [_u16p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_load_rrc_param_file.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_char_p, C.c_int]
        L.orc_load_rrc_param_file.restype = C.c_int
        L.orc_split_mss.argtypes = [_u16p, _u16p, _u16p, _u16p, _u16p, C.c_int, C.c_size_t]
        L.orc_bicubic_tab.argtypes = [_f32p]
        L.orc_interpolate_cubic.argtypes = [C.c_float, _f32p]
        L.orc_remap_cubic_u16.argtypes = [_u16p, C.c_int, C.c_int, C.c_size_t, _u16p, C.c_int,
                                          C.c_int, C.c_size_t, _f32p, _f32p, C.c_size_t]
        L.orc_prestitch.argtypes = [_u16p, _u16p, C.c_int, C.c_int, C.c_double, C.c_double,
                                    C.c_int, C.c_int]
        L.orc_prestitch.restype = C.c_long
        L.orc_align_mss.argtypes = [_u16p, _u16p, _u16p, _u16p, _u16p, C.c_int, C.c_long,
                                    _f64p, _f64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_align_mss.restype = C.c_long
        L.orc_stitch_raw.argtypes = [_u16p, _u16p, _u16p, C.c_int, C.c_long, C.c_int]
        L.orc_merge_subimages_be16.argtypes = [_u16p, _u16p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_resize_cubic_f32.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int]
        L.orc_window_u16_to_f32.argtypes = [_u16p, C.c_size_t, C.c_long, C.c_int, C.c_int,
                                            C.c_int, _f32p]
        _LIB = L
    return _LIB


def ref_lib():
    """The reference's own InplaceRRC (oracle/_ref), or None when it was never built."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libref_rrc.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.ref_inplace_rrc.argtypes = [_u16p, C.c_int, C.c_int, C.c_void_p]
        _REF = R
    return _REF


def _p16(a):
    return a.ctypes.data_as(_u16p)


def _c16(a):
    a = np.ascontiguousarray(a, dtype=np.uint16)
    return a


# --------------------------------------------------------------------------------------
def rrc(img: np.ndarray, kb: np.ndarray, threads: int = 1) -> np.ndarray:
    """imageop.h:129-138 on a copy.  kb is (w,2) float64 rows of (k, b)."""
    out = _c16(img).copy()
    h, w = out.shape
    kb = np.ascontiguousarray(kb, dtype=np.float64)
    assert kb.shape == (w, 2)
    if threads > 1:
        lib().orc_inplace_rrc_mt(_p16(out), w, h, kb.ctypes.data, threads)
    else:
        lib().orc_inplace_rrc(_p16(out), w, h, kb.ctypes.data)
    return out


def rrc_reference(img: np.ndarray, kb: np.ndarray) -> np.ndarray:
    """Same through oracle/_ref (the reference's own compiled loop)."""
    R = ref_lib()
    if R is None:
        raise RuntimeError("oracle/_ref/libref_rrc.so not built")
    out = _c16(img).copy()
    h, w = out.shape
    kb = np.ascontiguousarray(kb, dtype=np.float64)
    R.ref_inplace_rrc(_p16(out), w, h, kb.ctypes.data)
    return out


def load_rrc_param_file(path: str, expected: int) -> np.ndarray:
    out = np.zeros((expected, 2), dtype=np.float64)
    err = C.create_string_buffer(2048)
    rc = lib().orc_load_rrc_param_file(os.fsencode(path), expected, out.ctypes.data, err, 2048)
    if rc:
        exc = OSError if rc in (1, 2) else RuntimeError
        raise exc(err.value.decode())
    return out


def split_mss(bil: np.ndarray):
    bil = _c16(bil)
    lines, ppl = bil.shape
    bands = [np.empty((lines, ppl // 4), np.uint16) for _ in range(4)]
    lib().orc_split_mss(_p16(bil), *[_p16(b) for b in bands], ppl, lines)
    return bands


def bicubic_tab() -> np.ndarray:
    t = np.empty((32 * 32, 16), np.float32)
    lib().orc_bicubic_tab(t.ctypes.data_as(_f32p))
    return t


def interpolate_cubic(x: float) -> np.ndarray:
    c = np.empty(4, np.float32)
    lib().orc_interpolate_cubic(x, c.ctypes.data_as(_f32p))
    return c


def remap_cubic(src: np.ndarray, mapx: np.ndarray, mapy: np.ndarray) -> np.ndarray:
    src = _c16(src)
    mapx = np.ascontiguousarray(mapx, np.float32)
    mapy = np.ascontiguousarray(mapy, np.float32)
    dh, dw = mapx.shape
    dst = np.empty((dh, dw), np.uint16)
    lib().orc_remap_cubic_u16(_p16(src), src.shape[1], src.shape[0], src.shape[1], _p16(dst),
                              dw, dh, dw, mapx.ctypes.data_as(_f32p),
                              mapy.ctypes.data_as(_f32p), dw)
    return dst


def prestitch(src: np.ndarray, dx: float, dy: float, section_rows: int = 30000,
              row_guard: int = 32767):
    src = _c16(src)
    L, W = src.shape
    dst = np.zeros_like(src)
    r = lib().orc_prestitch(_p16(src), _p16(dst), W, L, dx, dy, section_rows, row_guard)
    if r < 0:
        raise ValueError("too few data rows, please use cv::remap()")
    return dst, int(r)


def align_mss(bands, cx, cy, lines_per_section=20000, line_offset=0, overlap=520,
              keep_leading=False, min_lines=1500):
    bands = [_c16(b) for b in bands]
    Lm, Wb = bands[0].shape
    cx = np.ascontiguousarray(cx, np.float64).reshape(4, 2)
    cy = np.ascontiguousarray(cy, np.float64).reshape(4, 3)
    out_rows = Lm - line_offset - (0 if keep_leading else overlap)
    dst = np.zeros((max(out_rows, 0), Wb, 4), np.uint16)
    n = lib().orc_align_mss(*[_p16(b) for b in bands], _p16(dst), Wb, Lm,
                            cx.ctypes.data_as(_f64p), cy.ctypes.data_as(_f64p),
                            lines_per_section, line_offset, overlap, int(keep_leading), min_lines)
    return dst, int(n)


def stitch_raw(left: np.ndarray, right: np.ndarray, fold: int) -> np.ndarray:
    left, right = _c16(left), _c16(right)
    L, W = left.shape
    out = np.empty((L, 2 * (W - fold)), np.uint16)
    lib().orc_stitch_raw(_p16(left), _p16(right), _p16(out), W, L, fold)
    return out


def merge_subimages_be16(tiles: np.ndarray) -> np.ndarray:
    """aux_separator.h:341-393 (uncompressed frames): tiles[vparts][hparts][sub_lines][sub_cols] of big-endian
    words (held in a uint16 array as read from the file) -> vparts*sub_lines lines of hparts*sub_cols pixels"""
    tiles = _c16(tiles)
    vparts, hparts, sub_lines, sub_cols = tiles.shape
    out = np.empty((vparts * sub_lines, hparts * sub_cols), np.uint16)
    lib().orc_merge_subimages_be16(_p16(tiles), _p16(out), vparts, hparts, sub_lines, sub_cols)
    return out


def resize_cubic(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    src = np.ascontiguousarray(src, np.float32)
    dst = np.empty((dh, dw), np.float32)
    lib().orc_resize_cubic_f32(src.ctypes.data_as(_f32p), src.shape[1], src.shape[0],
                               dst.ctypes.data_as(_f32p), dw, dh)
    return dst


def window_u16_to_f32(img: np.ndarray, row0: int, col0: int, rows: int, cols: int) -> np.ndarray:
    img = _c16(img)
    out = np.empty((rows, cols), np.float32)
    lib().orc_window_u16_to_f32(_p16(img), img.shape[1], row0, col0, rows, cols,
                                out.ctypes.data_as(_f32p))
    return out
