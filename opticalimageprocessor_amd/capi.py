"""ctypes binding of include/oip_c.h.  No arithmetic here -- every method forwards device
pointers to liboipgpu.so.  Anything that cannot reach the HIP library raises."""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
_LIB = None

STATUS_NAMES = {0: "OK", 1: "INVALID", 2: "RUNTIME", 3: "IO", 4: "DEVICE", 5: "NOMEM", 6: "UNSUPPORTED"}
# the exception classes the reference throws for each status (main.cpp:320-343 exit codes)
_STATUS_EXC = {1: ValueError, 2: RuntimeError, 3: OSError, 4: RuntimeError, 5: MemoryError, 6: NotImplementedError}


class OipError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("oip status %s: %s" % (STATUS_NAMES.get(status, status), msg))
        self.status = status


def library_path() -> str:
    # OIP_LIBRARY: another build of the same library (A/B experiments on one box: profiles/experiments/*.sh)
    return os.environ.get("OIP_LIBRARY") or os.path.join(_PKG, "lib", "liboipgpu.so")


def build(jobs: int = 8, quiet: bool = True) -> None:
    """Compile liboipgpu.so (+ the oip CLI) for gfx950 with hipcc."""
    subprocess.run(["make", "-C", os.path.join(_PKG, "csrc"), "-j%d" % jobs], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def declared_symbols():
    """Every function name include/oip_c.h declares."""
    text = open(os.path.join(_ROOT, "include", "oip_c.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(oip_[a-z0-9_]+)\s*\(", text)))


_vp, _i, _l, _d, _sz = C.c_void_p, C.c_int, C.c_long, C.c_double, C.c_size_t
_dp, _lp, _cp = C.POINTER(C.c_double), C.POINTER(C.c_long), C.c_char_p
_SIGS = {
    "oip_version": ([], _i),
    "oip_create": ([_i, C.POINTER(_vp)], _i),
    "oip_destroy": ([_vp], None),
    "oip_last_error": ([_vp], _cp),
    "oip_set_stream": ([_vp, _vp], _i),
    "oip_get_stream": ([_vp], _vp),
    "oip_sync": ([_vp], _i),
    "oip_malloc": ([_vp, C.POINTER(_vp), _sz], _i),
    "oip_free": ([_vp, _vp], _i),
    "oip_memset": ([_vp, _vp, _i, _sz], _i),
    "oip_memcpy_h2d": ([_vp, _vp, _vp, _sz], _i),
    "oip_memcpy_d2h": ([_vp, _vp, _vp, _sz], _i),
    "oip_host_alloc": ([_vp, C.POINTER(_vp), _sz], _i),
    "oip_host_free": ([_vp, _vp], _i),
    "oip_load_rrc_param_file": ([_cp, _i, _dp, _cp, _i], _i),
    "oip_rrc_u16": ([_vp, _vp, _vp, _i, _l, _vp], _i),
    "oip_rrc_u16_window": ([_vp, _vp, _l, _vp, _l, _i, _l, _vp], _i),
    "oip_rrc_u16_host": ([_vp, _vp, _i, _l, _dp], _i),
    "oip_read_file_to_device": ([_vp, _cp, _sz, _sz, _vp, C.POINTER(_sz), _lp], _i),
    "oip_write_device_to_file": ([_vp, _vp, _sz, _cp, _i], _i),
    "oip_write_device_to_file_at": ([_vp, _vp, _sz, _cp, _sz, _l], _i),
    "oip_file_sink_open": ([_vp, _cp, _sz, C.POINTER(_vp)], _i),
    "oip_file_sink_write": ([_vp, _vp, _sz, _vp, _sz, _l], _i),
    "oip_file_sink_close": ([_vp, _vp], _i),
    "oip_compute_mark": ([_vp, _lp], _i),
    "oip_compute_mark_sync": ([_vp, _l], _i),
    "oip_download_staged_after": ([_vp, _vp, _vp, _sz, _l], _i),
    "oip_permute_u16x4": ([_vp, _vp, _sz, C.POINTER(_i)], _i),
    "oip_tiff_lzw_worst_bytes": ([_l, _i, _i, _l], _sz),
    "oip_tiff_lzw_scratch_bytes": ([_l, _i, _i, _l], _sz),
    "oip_tiff_lzw_decode_u16": ([_vp, _vp, _sz, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _l, _l, _i, _i, _l, _i, _vp], _i),
    "oip_tiff_lzw_strips_u16": ([_vp, _vp, _l, _i, _i, _l, _vp, _sz, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(_sz), _vp, _sz], _i),
    "oip_upload_staged": ([_vp, _vp, _vp, _sz, _lp], _i),
    "oip_upload_staged_2d": ([_vp, _vp, _sz, _vp, _sz, _sz, _sz, _lp], _i),
    "oip_download_staged": ([_vp, _vp, _vp, _sz], _i),
    "oip_stage_wait": ([_vp, _l], _i),
    "oip_stage_sync": ([_vp], _i),
    "oip_stage_order_after_compute": ([_vp], _i),
    "oip_stage_threads": ([], _i),
    "oip_stage_stats": ([_vp, _dp, _i], _i),
    "oip_mss_split_rrc_u16": ([_vp, _vp, _vp, _sz, _i, _l, _vp], _i),
    "oip_phase_correlate_f32": ([_vp, _vp, _vp, _i, _i, _dp, _dp, _dp], _i),
    "oip_window_u16_to_f32": ([_vp, _vp, _sz, _l, _i, _i, _i, _vp], _i),
    "oip_resize_cubic_f32": ([_vp, _vp, _i, _i, _vp, _i, _i], _i),
    "oip_stt_correlate": ([_vp, _vp, _vp, _i, _l, _l, _l, _i, _i, _i, _i, _dp], _i),
    "oip_stt_correlate_windows": ([_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _dp], _i),
    "oip_interband_correlate_units": ([_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _dp], _i),
    "oip_interband_correlate": ([_vp, _vp, _l, _l, _l, _vp, _sz, _l, _l, _i, _i, _i, _i, _dp], _i),
    "oip_filter_and_fit": ([_dp, _i, _d, _i, _dp, _dp, _cp, _i], _i),
    "oip_filter_and_fit_mode": ([_dp, _i, _d, _i, _i, _dp, _dp, _cp, _i], _i),
    "oip_polyfit": ([_dp, _dp, _i, _i, _dp], _i),
    "oip_polyfit_reference": ([_dp, _dp, _i, _i, _dp], _i),
    "oip_stt_mean": ([_dp, _i, _d, _d, _dp, _dp, _dp, C.POINTER(_i)], _i),
    "oip_upsample_operator": ([_i, C.POINTER(C.c_float)], _i),
    "oip_remap_shift_bicubic_u16": ([_vp, _vp, _l, _l, _vp, _l, _l, _i, _l, _d, _d, _i, _i], _i),
    "oip_remap_shift_bicubic_u16_f16acc": ([_vp, _vp, _l, _l, _vp, _l, _l, _i, _l, _d, _d, _i, _i], _i),
    "oip_remap_shift_bicubic_u16_window": ([_vp, _vp, _l, _l, _vp, _l, _i, _l, _l, _l, _i, _l, _d, _d, _i, _i, _i], _i),
    "oip_remap_shift_rrc_bicubic_u16_window": ([_vp, _vp, _l, _l, _vp, _vp, _l, _i, _l, _l, _l, _i, _l, _d, _d, _i, _i, _i], _i),
    "oip_remap_shift_src_range": ([_l, _l, _l, _d, _i, _lp, _lp], _i),
    "oip_align_mss_bicubic_u16x4": ([_vp, _vp, _sz, _l, _l, _vp, _l, _l, _i, _l, _dp, _dp, _i, _i, _i, _i, _i, _lp], _i),
    "oip_align_mss_src_range": ([_l, _l, _l, _dp, _i, _i, _i, _i, _i, _i, _lp, _lp], _i),
    "oip_stitch_rows_u16": ([_vp, _vp, _vp, _vp, _i, _l, _i], _i),
    "oip_merge_subimages_be16": ([_vp, _vp, _vp, _i, _i, _i, _i], _i),
    "oip_profile_enable": ([_vp, _i], _i),
    "oip_profile_reset": ([_vp], _i),
    "oip_profile_filter": ([_vp, _cp], _i),
    "oip_profile_count": ([_vp], _i),
    "oip_profile_get": ([_vp, _i, _cp, _i, _dp, _lp], _i),
}


def load_library() -> C.CDLL:
    """dlopen liboipgpu.so (in-tree).  Raises if it has not been built."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise OSError("liboipgpu.so not built at %s -- run __graft_entry__.build() "
                          "(there is no CPU fallback)" % path)
        # torch wheels bundle their own ROCm runtime; whichever HIP runtime is loaded first
        # in a process must be torch's, or the second one finds no devices.  (The oip CLI
        # links the system runtime and never meets torch.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(path)
        for name, (args, res) in _SIGS.items():
            fn = getattr(lib, name)          # AttributeError if the symbol is missing
            fn.argtypes = args
            fn.restype = res
        _LIB = lib
    return _LIB


def _ptr(t):
    """device pointer of a torch tensor / raw int"""
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


def _dbl(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if n is not None:
        assert a.size == n, (a.size, n)
    return a


# ---- host-only entry points ---------------------------------------------------------------
def load_rrc_param_file(path: str, expected: int) -> np.ndarray:
    lib = load_library()
    out = np.zeros((expected, 2), np.float64)
    err = C.create_string_buffer(2048)
    rc = lib.oip_load_rrc_param_file(os.fsencode(path), expected, out.ctypes.data_as(_dp), err, 2048)
    if rc:
        raise _STATUS_EXC.get(rc, OipError)(err.value.decode())
    return out


FIT_MODES = {"reference": 0, "lstsq": 1}


def polyfit(x, y, deg: int, fit: str = "reference") -> np.ndarray:
    """fit="reference": NumCpp's operation order as the reference calls it (the product default);
    fit="lstsq": Householder QR on a scaled abscissa"""
    lib = load_library()
    x, y = _dbl(x), _dbl(y)
    out = np.zeros(deg + 1)
    fn = lib.oip_polyfit if FIT_MODES[fit] else lib.oip_polyfit_reference
    rc = fn(x.ctypes.data_as(_dp), y.ctypes.data_as(_dp), x.size, deg, out.ctypes.data_as(_dp))
    if rc:
        raise _STATUS_EXC.get(rc, OipError)("oip_polyfit failed (%s)" % STATUS_NAMES.get(rc))
    return out


def filter_and_fit(shifts, threshold=0.4, min_count=5, fit: str = "reference"):
    lib = load_library()
    s = _dbl(shifts)
    assert s.ndim == 3 and s.shape[0] == 4 and s.shape[2] == 4
    cx, cy = np.zeros((4, 2)), np.zeros((4, 3))
    err = C.create_string_buffer(1024)
    rc = lib.oip_filter_and_fit_mode(s.ctypes.data_as(_dp), s.shape[1], threshold, min_count, FIT_MODES[fit],
                                     cx.ctypes.data_as(_dp), cy.ctypes.data_as(_dp), err, 1024)
    if rc:
        raise _STATUS_EXC.get(rc, OipError)(err.value.decode())
    return cx, cy


def stt_mean(table, threshold=0.4, max_delta_y=0.0):
    """stitcher.h:181-198 on an (S, 3) table -> (dx, dy, response, valid)"""
    lib = load_library()
    t = _dbl(table)
    dx, dy, r, v = _d(), _d(), _d(), _i()
    rc = lib.oip_stt_mean(t.ctypes.data_as(_dp), t.shape[0], threshold, max_delta_y, C.byref(dx), C.byref(dy), C.byref(r), C.byref(v))
    if rc == 2:
        raise RuntimeError("No valid delta value found for stitching parameter calculating")
    if rc:
        raise ValueError("oip_stt_mean: bad argument")
    return dx.value, dy.value, r.value, v.value


def upsample_operator(n):
    """H, G_0..G_3 of the x4 cubic up-sampling n -> 4 n as an operator on spectra (include/oip_c.h): complex64 (5, 4 n)"""
    lib = load_library()
    out = np.zeros((5, 4 * n, 2), np.float32)
    rc = lib.oip_upsample_operator(n, out.ctypes.data_as(C.POINTER(C.c_float)))
    if rc:
        raise ValueError("oip_upsample_operator: unsupported length %d" % n)
    return out[..., 0] + 1j * out[..., 1]


def remap_shift_src_range(out_row0, out_rows, L, dy, section_rows=30000):
    lib = load_library()
    a, b = C.c_long(), C.c_long()
    rc = lib.oip_remap_shift_src_range(out_row0, out_rows, L, dy, section_rows, C.byref(a), C.byref(b))
    if rc:
        raise ValueError("oip_remap_shift_src_range: bad argument")
    return a.value, b.value


def align_mss_src_range(out_row0, out_rows, Lm, cy, Wb, lines_per_section=20000, line_offset=0,
                        overlap=520, keep_leading=False, min_lines=1500):
    lib = load_library()
    cy = _dbl(cy, 12)
    a, b = C.c_long(), C.c_long()
    rc = lib.oip_align_mss_src_range(out_row0, out_rows, Lm, cy.ctypes.data_as(_dp), Wb, lines_per_section,
                                     line_offset, overlap, int(keep_leading), min_lines, C.byref(a), C.byref(b))
    if rc:
        raise ValueError("oip_align_mss_src_range: bad argument")
    return a.value, b.value


# ---- device context -------------------------------------------------------------------------
class Context:
    """One oip_ctx (one GPU).  Methods take torch CUDA tensors (or raw device pointers)."""

    def __init__(self, device: int = 0, stream=None):
        self.lib = load_library()
        h = _vp()
        rc = self.lib.oip_create(device, C.byref(h))
        if rc:
            raise OipError(rc, "oip_create(%d) failed: no gfx950 device (no CPU fallback)" % device)
        self.h = h
        self.device = device
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if getattr(self, "h", None):
            self.lib.oip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc:
            msg = self.lib.oip_last_error(self.h).decode()
            exc = _STATUS_EXC.get(rc, OipError)
            raise exc(msg) if exc is not OipError else OipError(rc, msg)

    def set_stream(self, stream):
        """stream: torch.cuda.Stream, raw hipStream_t int, or None for the context's own."""
        s = getattr(stream, "cuda_stream", stream)
        self._ck(self.lib.oip_set_stream(self.h, s))

    def get_stream(self):
        """the hipStream_t the context's kernels run on, as an integer"""
        return self.lib.oip_get_stream(self.h) or 0

    def sync(self):
        self._ck(self.lib.oip_sync(self.h))

    # -- RRC
    def upload_kb(self, kb):
        import torch
        kb = _dbl(kb)
        return torch.from_numpy(kb.reshape(-1, 2)).to("cuda:%d" % self.device)

    def rrc_u16(self, src, dst, w, h, d_kb):
        self._ck(self.lib.oip_rrc_u16(self.h, _ptr(src), _ptr(dst), w, h, _ptr(d_kb)))

    def rrc_u16_window(self, src, src_pitch, dst, dst_pitch, w, h, d_kb):
        """RRC of columns [0, w) of h lines (pitches in pixels): e.g. straight into the left half of a stitched raster"""
        self._ck(self.lib.oip_rrc_u16_window(self.h, _ptr(src), src_pitch, _ptr(dst), dst_pitch, w, h, _ptr(d_kb)))

    def rrc_u16_host(self, buff: np.ndarray, kb):
        assert buff.dtype == np.uint16 and buff.flags.c_contiguous and buff.ndim == 2
        kb = _dbl(kb, buff.shape[1] * 2)
        self._ck(self.lib.oip_rrc_u16_host(self.h, buff.ctypes.data, buff.shape[1], buff.shape[0],
                                           kb.ctypes.data_as(_dp)))

    # -- raster I/O staging
    def read_file_to_device(self, path, d_dst, offset=0, nbytes=0, want_ticket=False):
        got, t = C.c_size_t(), C.c_long()
        self._ck(self.lib.oip_read_file_to_device(self.h, os.fsencode(path), offset, nbytes, _ptr(d_dst), C.byref(got),
                                                  C.byref(t) if want_ticket else None))
        return (got.value, t.value) if want_ticket else got.value

    def write_device_to_file(self, d_src, nbytes, path, append=False):
        self._ck(self.lib.oip_write_device_to_file(self.h, _ptr(d_src), nbytes, os.fsencode(path), int(append)))

    def write_device_to_file_at(self, d_src, nbytes, path, file_offset, mark=0, byte_offset=0):
        self._ck(self.lib.oip_write_device_to_file_at(self.h, _ptr(d_src) + byte_offset, nbytes, os.fsencode(path), file_offset, mark))

    def file_sink_open(self, path, nbytes):
        h = C.c_void_p()
        self._ck(self.lib.oip_file_sink_open(self.h, os.fsencode(path), nbytes, C.byref(h)))
        return h

    def file_sink_write(self, sink, file_offset, d_src, nbytes, mark=0, byte_offset=0):
        self._ck(self.lib.oip_file_sink_write(self.h, sink, file_offset, _ptr(d_src) + byte_offset, nbytes, mark))

    def file_sink_close(self, sink):
        self._ck(self.lib.oip_file_sink_close(self.h, sink))

    def compute_mark(self):
        m = C.c_long()
        self._ck(self.lib.oip_compute_mark(self.h, C.byref(m)))
        return m.value

    def compute_mark_sync(self, mark):
        self._ck(self.lib.oip_compute_mark_sync(self.h, mark))

    def permute_u16x4(self, img, npixels, order):
        o = (C.c_int * 4)(*[int(v) for v in order])
        self._ck(self.lib.oip_permute_u16x4(self.h, _ptr(img), npixels, o))

    def tiff_lzw_strips(self, img, rows, width, spp, rows_per_strip, payload, scratch=None):
        """LZW strips (predictor 2) of a device image into `payload` (device, uint8); returns (offsets, lengths, payload_bytes)"""
        n = (rows + rows_per_strip - 1) // rows_per_strip
        off = (C.c_uint64 * n)()
        ln = (C.c_uint64 * n)()
        total = C.c_size_t()
        self._ck(self.lib.oip_tiff_lzw_strips_u16(self.h, _ptr(img), rows, width, spp, rows_per_strip, _ptr(payload),
                                                  payload.numel() * payload.element_size(), off, ln, C.byref(total),
                                                  _ptr(scratch) if scratch is not None else None,
                                                  scratch.numel() * scratch.element_size() if scratch is not None else 0))
        return np.array(off[:], dtype=np.uint64), np.array(ln[:], dtype=np.uint64), total.value

    def tiff_lzw_decode(self, d_file, strip_off, strip_len, rows, width, spp, rows_per_strip, predictor, d_img):
        """LZW strips in `d_file` (device, uint8; offsets relative to it) decoded into d_img (device, rows x width x spp u16)"""
        n = len(strip_off)
        off = (C.c_uint64 * n)(*[int(v) for v in strip_off])
        ln = (C.c_uint64 * n)(*[int(v) for v in strip_len])
        self._ck(self.lib.oip_tiff_lzw_decode_u16(self.h, _ptr(d_file), d_file.numel() * d_file.element_size(), off, ln, n, rows, width, spp,
                                                  rows_per_strip, predictor, _ptr(d_img)))

    def tiff_lzw_scratch_bytes(self, rows, width, spp, rows_per_strip):
        return int(self.lib.oip_tiff_lzw_scratch_bytes(rows, width, spp, rows_per_strip))

    def tiff_lzw_worst_bytes(self, rows, width, spp, rows_per_strip):
        return int(self.lib.oip_tiff_lzw_worst_bytes(rows, width, spp, rows_per_strip))

    def upload_staged(self, d_dst, host: np.ndarray, want_ticket=False, byte_offset=0):
        assert host.flags.c_contiguous
        t = C.c_long()
        self._ck(self.lib.oip_upload_staged(self.h, _ptr(d_dst) + byte_offset, host.ctypes.data, host.nbytes,
                                            C.byref(t) if want_ticket else None))
        return t.value if want_ticket else None

    def upload_staged_2d(self, d_dst, dst_pitch_bytes, host: np.ndarray, want_ticket=False, byte_offset=0):
        """a 2-D host view (rows contiguous, any row stride) into device rows dst_pitch_bytes apart"""
        assert host.ndim == 2 and host.strides[1] == host.itemsize and host.strides[0] >= host.shape[1] * host.itemsize
        t = C.c_long()
        self._ck(self.lib.oip_upload_staged_2d(self.h, _ptr(d_dst) + byte_offset, dst_pitch_bytes, host.ctypes.data, host.strides[0],
                                               host.shape[1] * host.itemsize, host.shape[0], C.byref(t) if want_ticket else None))
        return t.value if want_ticket else None

    def download_staged(self, host: np.ndarray, d_src, byte_offset=0, mark=0):
        assert host.flags.c_contiguous and host.flags.writeable
        self._ck(self.lib.oip_download_staged_after(self.h, host.ctypes.data, _ptr(d_src) + byte_offset, host.nbytes, mark))

    def stage_wait(self, ticket):
        self._ck(self.lib.oip_stage_wait(self.h, ticket))

    def stage_stats(self, reset=False):
        """(seconds in pageable -> pinned copies, seconds waiting for a ring slot, bytes, calls) of the upload lane"""
        out = (C.c_double * 4)()
        self._ck(self.lib.oip_stage_stats(self.h, out, 1 if reset else 0))
        return tuple(out)

    def stage_order_after_compute(self):
        self._ck(self.lib.oip_stage_order_after_compute(self.h))

    def stage_sync(self):
        self._ck(self.lib.oip_stage_sync(self.h))

    def mss_split_rrc_u16(self, bil, planes, plane_stride, w, lines, d_kb4):
        self._ck(self.lib.oip_mss_split_rrc_u16(self.h, _ptr(bil), _ptr(planes), plane_stride, w, lines, _ptr(d_kb4)))

    # -- correlation
    def phase_correlate_f32(self, a, b, rows, cols):
        dx, dy, r = _d(), _d(), _d()
        self._ck(self.lib.oip_phase_correlate_f32(self.h, _ptr(a), _ptr(b), rows, cols, C.byref(dx), C.byref(dy), C.byref(r)))
        return (dx.value, dy.value), r.value

    def window_u16_to_f32(self, img, pitch, row0, col0, rows, cols, out):
        self._ck(self.lib.oip_window_u16_to_f32(self.h, _ptr(img), pitch, row0, col0, rows, cols, _ptr(out)))

    def resize_cubic_f32(self, src, sw, sh, dst, dw, dh):
        self._ck(self.lib.oip_resize_cubic_f32(self.h, _ptr(src), sw, sh, _ptr(dst), dw, dh))

    def stt_correlate(self, pan1, pan2, W, L, row0, nrows, sections=10, lines_per_section=16000,
                      overlap_cols=200, edge_cols=0):
        out = np.zeros((sections, 3))
        self._ck(self.lib.oip_stt_correlate(self.h, _ptr(pan1), _ptr(pan2), W, L, row0, nrows, sections,
                                            lines_per_section, overlap_cols, edge_cols, out.ctypes.data_as(_dp)))
        return out

    def interband_correlate(self, pan, Lp, prow0, pn, planes, plane_stride, mrow0, mn, W, slices=10,
                            sections=5, corr_lines=16000):
        out = np.zeros((4, sections * slices, 4))
        self._ck(self.lib.oip_interband_correlate(self.h, _ptr(pan), Lp, prow0, pn, _ptr(planes), plane_stride,
                                                  mrow0, mn, W, slices, sections, corr_lines,
                                                  out.ctypes.data_as(_dp)))
        return out

    def stt_correlate_windows(self, a_ptrs, a_pitch, b_ptrs, b_pitch, rows, cols):
        """n explicit CCD window pairs (device pointers + element pitches) -> (n, 3) dx, dy, response"""
        n = len(a_ptrs)
        out = np.zeros((n, 3))
        if n == 0:
            return out
        pa = (C.c_void_p * n)(*[_ptr(p) for p in a_ptrs]); pb = (C.c_void_p * n)(*[_ptr(p) for p in b_ptrs])
        qa = (C.c_size_t * n)(*a_pitch); qb = (C.c_size_t * n)(*b_pitch)
        self._ck(self.lib.oip_stt_correlate_windows(self.h, pa, qa, pb, qb, n, rows, cols, out.ctypes.data_as(_dp)))
        return out

    def interband_correlate_units(self, pan_ptrs, pan_pitch, band_ptrs, band_pitch, rows, cols):
        """n explicit (section, slice) units; band_ptrs is n x 4 device pointers -> (n, 4, 3) dx, dy, rs"""
        n = len(pan_ptrs)
        out = np.zeros((n, 4, 3))
        if n == 0:
            return out
        pp = (C.c_void_p * n)(*[_ptr(p) for p in pan_ptrs])
        bp = (C.c_void_p * (4 * n))(*[_ptr(p) for u in band_ptrs for p in u])
        qp = (C.c_size_t * n)(*pan_pitch); qb = (C.c_size_t * n)(*band_pitch)
        self._ck(self.lib.oip_interband_correlate_units(self.h, pp, qp, bp, qb, n, rows, cols, out.ctypes.data_as(_dp)))
        return out

    # -- resampling
    def remap_shift_bicubic_u16(self, src, dst, W, L, dx, dy, section_rows=30000, row_guard=32767,
                                src_row0=0, src_rows=None, out_row0=0, out_rows=None, f16acc=False):
        """f16acc=True: the fp16-accumulate variant (not the parity mode; tolerance in include/oip_c.h)"""
        src_rows = L if src_rows is None else src_rows
        out_rows = L if out_rows is None else out_rows
        fn = self.lib.oip_remap_shift_bicubic_u16_f16acc if f16acc else self.lib.oip_remap_shift_bicubic_u16
        self._ck(fn(self.h, _ptr(src), src_row0, src_rows, _ptr(dst), out_row0, out_rows, W, L, dx, dy,
                    section_rows, row_guard))

    def remap_shift_bicubic_u16_window(self, src, dst, dst_pitch, dst_col0, dst_col_off, W, L, dx, dy, section_rows=30000,
                                       row_guard=32767, src_row0=0, src_rows=None, out_row0=0, out_rows=None, f16acc=False):
        """the same resampling written into a window of another raster (columns >= dst_col0 only; see include/oip_c.h)"""
        src_rows = L if src_rows is None else src_rows
        out_rows = L if out_rows is None else out_rows
        self._ck(self.lib.oip_remap_shift_bicubic_u16_window(self.h, _ptr(src), src_row0, src_rows, _ptr(dst), dst_pitch, dst_col0,
                                                             dst_col_off, out_row0, out_rows, W, L, dx, dy, section_rows, row_guard,
                                                             1 if f16acc else 0))

    def remap_shift_rrc_bicubic_u16_window(self, src_raw, d_kb, dst, dst_pitch, dst_col0, dst_col_off, W, L, dx, dy, section_rows=30000,
                                           row_guard=32767, src_row0=0, src_rows=None, out_row0=0, out_rows=None, f16acc=False):
        """the windowed resampling with the RAW strip as source: RRC applied on load (see include/oip_c.h)"""
        src_rows = L if src_rows is None else src_rows
        out_rows = L if out_rows is None else out_rows
        self._ck(self.lib.oip_remap_shift_rrc_bicubic_u16_window(self.h, _ptr(src_raw), src_row0, src_rows, _ptr(d_kb), _ptr(dst), dst_pitch,
                                                                 dst_col0, dst_col_off, out_row0, out_rows, W, L, dx, dy, section_rows,
                                                                 row_guard, 1 if f16acc else 0))

    def align_mss_bicubic_u16x4(self, planes, plane_stride, dst, Wb, Lm, cx, cy, lines_per_section=20000,
                                line_offset=0, overlap=520, keep_leading=False, min_lines=1500,
                                src_row0=0, src_rows=None, out_row0=0, out_rows=None):
        cx, cy = _dbl(cx, 8), _dbl(cy, 12)
        src_rows = Lm if src_rows is None else src_rows
        total = Lm - line_offset - (0 if keep_leading else overlap)
        out_rows = total if out_rows is None else out_rows
        valid = C.c_long()
        self._ck(self.lib.oip_align_mss_bicubic_u16x4(self.h, _ptr(planes), plane_stride, src_row0, src_rows,
                                                      _ptr(dst), out_row0, out_rows, Wb, Lm,
                                                      cx.ctypes.data_as(_dp), cy.ctypes.data_as(_dp),
                                                      lines_per_section, line_offset, overlap, int(keep_leading),
                                                      min_lines, C.byref(valid)))
        return valid.value

    def stitch_rows_u16(self, left, right, out, W, L, fold):
        self._ck(self.lib.oip_stitch_rows_u16(self.h, _ptr(left), _ptr(right), _ptr(out), W, L, fold))

    # -- instrumentation
    def merge_subimages_be16(self, tiles, out, vparts, hparts, sub_lines, sub_cols):
        """aux_separator.h:341-393 for uncompressed frames: big-endian sub-images -> little-endian stripes"""
        self._ck(self.lib.oip_merge_subimages_be16(self.h, _ptr(tiles), _ptr(out), vparts, hparts, sub_lines, sub_cols))

    def profile_enable(self, on=True):
        self._ck(self.lib.oip_profile_enable(self.h, int(on)))

    def profile_filter(self, name=None):
        """time only the kernels profiled under `name` (None: all)"""
        self._ck(self.lib.oip_profile_filter(self.h, name.encode() if name else None))

    def profile_reset(self):
        self._ck(self.lib.oip_profile_reset(self.h))

    def profile(self):
        out = {}
        n = self.lib.oip_profile_count(self.h)
        for i in range(n):
            name = C.create_string_buffer(128)
            ms, cnt = _d(), _l()
            self._ck(self.lib.oip_profile_get(self.h, i, name, 128, C.byref(ms), C.byref(cnt)))
            out[name.value.decode()] = (ms.value, cnt.value)
        return out
