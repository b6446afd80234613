"""opticalimageprocessor_amd -- MI355X (gfx950) implementation of the
arloan/OpticalImageProcessor hot path (RRC, phase-correlation alignment, bicubic
resampling, strip stitching).

The product is ``lib/liboipgpu.so`` (hand-written HIP kernels behind the C ABI declared in
``include/oip_c.h``) plus the ``oip`` CLI built from ``csrc/oip_main.cpp``.  This Python
package is only a ctypes binding of that C ABI for tests and ``bench.py``; torch is used for
device allocations and ``torch.distributed``, never for arithmetic.  There is no CPU
fallback: loading fails loudly when the library is missing, and ``Context()`` fails when no
gfx950 device is present.
"""
from .capi import (  # noqa: F401
    Context,
    OipError,
    STATUS_NAMES,
    build,
    declared_symbols,
    filter_and_fit,
    library_path,
    load_library,
    load_rrc_param_file,
    polyfit,
    remap_shift_src_range,
    stt_mean,
    upsample_operator,
    align_mss_src_range,
)

__version__ = "1.1"
