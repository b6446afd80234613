#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/.

Run in the build container only (it needs oracle/_ref, i.e. /root/reference):
    python tests/golden/make_golden.py

rrc_reference.npz
    inputs (src u16, kb f64) and the output of the REFERENCE'S OWN IMO::InplaceRRC
    (imageop.h:129-138, compiled in place by oracle/Makefile into oracle/_ref/libref_rrc.so
    with the reference's CMake Release flags).  Columns 0..31 carry adversarial (k,b) pairs:
    wrap above 65535, negative results, |v| >= 2^31, NaN/inf, truncation boundaries.
known_answers.json
    analytic facts the OpenCV-restating parts must satisfy: bicubic
    coefficients (A=-0.75) at t = 0, 1/32, 1/2, 31/32 in f32; cv::getOptimalDFTSize values;
    the x4 resize phases.  These come from the published algorithm, not from a run of OpenCV
    (absent here): parity of those parts stays "unpinned".
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402


def main():
    oracle.build()
    if oracle.ref_lib() is None:
        sys.exit("oracle/_ref/libref_rrc.so missing: /root/reference not available")
    rng = np.random.default_rng(0x0A11CE)
    w, h = 1024, 48
    src = rng.integers(0, 65536, (h, w), dtype=np.uint16)
    src[0, :] = 0
    src[1, :] = 65535
    src[2, :] = np.arange(w) * 64 % 65536
    kb = np.stack([np.round(rng.uniform(0.9, 1.1, w), 6), np.round(rng.uniform(-8, 8, w), 4)], 1)
    special = [(1.5015, 0.0), (-1.0, 0.0), (1.0, -90.0), (1.0, 70000.0), (3e5, 0.0), (1e9, 0.0), (-1e9, 0.0),
               (float("nan"), 0.0), (1.0, 0.999999999), (0.0, 65535.99999), (0.0, 2147483647.5), (0.0, 2147483648.0),
               (0.0, -2147483648.5), (0.0, -2147483649.0), (1.0, float("inf")), (0.0, -0.9999), (1.0, -0.5),
               (0.5, 0.5), (2.0, -65535.0), (65536.0, 0.0), (32768.0, 1.0), (1.0000001, 0.0), (0.9999999, 0.0),
               (1.0, 1e-300), (1.0, -1e-300), (1e-310, 0.0), (0.0, float("-inf")), (1 / 3, 1 / 3), (1.1, -8.0),
               (0.9, 8.0), (32767.5, 0.5), (0.0, 98402.5)]
    for i, p in enumerate(special):
        kb[i] = p
    dst = oracle.rrc_reference(src, kb)
    assert np.array_equal(dst, oracle.rrc(src, kb)), "CPU restatement disagrees with the reference's loop"
    np.savez_compressed(os.path.join(HERE, "rrc_reference.npz"), src=src, kb=kb, dst=dst)

    ka = {
        "bicubic_A": -0.75,
        "bicubic_coeffs_f32": {
            "0": [0.0, 1.0, 0.0, 0.0],
            "0.03125": [-0.021995544, 0.99784088, 0.024864197, -0.00070953369],
            "0.5": [-0.09375, 0.59375, 0.59375, -0.09375],
            "0.96875": [-0.00070953369, 0.024864197, 0.99784088, -0.021995544],
        },
        "optimal_dft_size": {"200": 200, "307": 320, "1228": 1250, "3000": 3000, "16000": 16000, "4000": 4000,
                             "7": 8, "11": 12, "13": 15, "17": 18, "1": 1},
        "resize_x4_phases": [0.625, 0.875, 0.125, 0.375],
        "rrc_wrap_examples": {"98402.5": 32866, "-90": 65446},
    }
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(ka, f, indent=1)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
