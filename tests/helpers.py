"""Shared helpers for the parity tests (CPU oracle side and GPU side)."""
import math
import struct

import numpy as np


def f32(x):
    return float(np.float32(x))


def float_ulp_diff(a, b):
    """Distance in float32 ulps (gtest's ASSERT_FLOAT_EQ metric)."""
    def key(v):
        i = struct.unpack("<i", struct.pack("<f", v))[0]
        return i if i >= 0 else -(i & 0x7FFFFFFF)
    return abs(key(np.float32(a)) - key(np.float32(b)))


def check_case_values(case, out, norm_buf):
    """Assert the reference test's expectations on an output buffer / norm buffer."""
    off, st = case["pdf_offset"], case["pdf_stride"]
    for k, exp in case["expected_values"].items():
        got = float(out[off + int(k) * st])
        if exp == "nan":
            assert math.isnan(got), (case["name"], k, got)
        else:
            assert float_ulp_diff(got, exp) <= case["float_ulps"], (case["name"], k, got, exp)
    assert int(norm_buf[case["norm_offset"]]) == case["expected_norm"], case["name"]
    if case["expected_norm_buffer"] is not None:
        assert [int(x) for x in norm_buf] == case["expected_norm_buffer"], case["name"]


def eval_points_with_dataset(case, dataset=0.0):
    """Reference tests predate the dataset column: append it (see make_pdfz_known_answers.py)."""
    d = case["nobs"]
    pts = np.asarray(case["eval_points"], dtype=np.float32).reshape(-1, d)
    return np.concatenate([pts, np.full((pts.shape[0], 1), dataset, np.float32)], axis=1)


def philox4x32_10(ctr, key):
    """Philox4x32-10 (Salmon et al., SC'11) on Python ints: ctr = 4 words, key = 2 words."""
    M0, M1, W0, W1, MASK = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c3 ^ k1) & MASK, p0 & MASK
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return (c0, c1, c2, c3)
