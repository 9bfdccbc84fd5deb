"""The oracle's software binary16 (oracle/half.hpp) is pinned three ways: against the REFERENCE's vendored
half.hpp (oracle/_ref, only where /root/reference exists), against numpy's float16, and against committed
golden vectors generated from the reference library (tests/golden/half_ops.npz, tools/make_golden.py)."""
import os

import numpy as np
import pytest

from oracle import binding

GOLD = os.path.join(os.path.dirname(__file__), "golden", "half_ops.npz")


def _np_op(op, a, b):
    fa, fb = a.view(np.float16).astype(np.float32), b.view(np.float16).astype(np.float32)
    with np.errstate(all="ignore"):
        r = [fa + fb, fa - fb, fa * fb, fa / fb, np.sqrt(fa)][op]
    return r.astype(np.float16).view(np.uint16)


def _same(a, b):
    """bit equality, any NaN == any NaN"""
    fa, fb = a.view(np.float16), b.view(np.float16)
    return (a == b) | (np.isnan(fa) & np.isnan(fb))


def test_f16_to_f32_exhaustive():
    L = binding.lib()
    bits = np.arange(65536, dtype=np.uint16)
    got = np.array([L.orc_f16_to_f32(int(b)) for b in bits], dtype=np.float32)
    want = bits.view(np.float16).astype(np.float32)
    assert np.array_equal(got.view(np.uint32)[~np.isnan(want)], want.view(np.uint32)[~np.isnan(want)])
    assert np.isnan(got[np.isnan(want)]).all()


def test_f32_to_f16_rounding():
    L = binding.lib()
    rng = np.random.default_rng(1)
    xs = (rng.normal(size=40000) * rng.choice([1e-8, 1e-6, 1e-4, 1e-2, 1, 30, 1e3, 7e4], 40000)).astype(np.float32)
    edge = np.array([65504, 65519.996, 65520, 65536, 1e5, -1e5, 5.9604645e-8, 2.9802322e-8, 2.9802326e-8, 0.0, -0.0,
                     np.inf, -np.inf, 6.1035156e-5, 6.0975552e-5], dtype=np.float32)
    xs = np.concatenate([xs, edge])
    got = np.array([L.orc_f32_to_f16(float(x)) for x in xs], dtype=np.uint16)
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).view(np.uint16)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("op", range(5))
def test_half_ops_match_numpy(op):
    L = binding.lib()
    rng = np.random.default_rng(10 + op)
    a = rng.integers(0, 65536, 30000).astype(np.uint16)
    b = rng.integers(0, 65536, 30000).astype(np.uint16)
    got = np.array([L.orc_half_op(op, int(x), int(y)) for x, y in zip(a, b)], dtype=np.uint16)
    assert _same(got, _np_op(op, a, b)).all()


@pytest.mark.parametrize("op", range(5))
def test_half_ops_match_reference_library(op):
    R = binding.ref_half_lib()
    if R is None:
        pytest.skip("oracle/_ref not built (the reference is not present on this machine)")
    L = binding.lib()
    rng = np.random.default_rng(100 + op)
    a = rng.integers(0, 65536, 30000).astype(np.uint16)
    b = rng.integers(0, 65536, 30000).astype(np.uint16)
    got = np.array([L.orc_half_op(op, int(x), int(y)) for x, y in zip(a, b)], dtype=np.uint16)
    ref = np.array([R.ref_half_op(op, int(x), int(y)) for x, y in zip(a, b)], dtype=np.uint16)
    assert _same(got, ref).all()


def test_half_ops_golden_vectors():
    """Outputs of the reference's half.hpp, generated in the authoring container and committed."""
    g = np.load(GOLD)
    L = binding.lib()
    for op in range(5):
        a, b, want = g[f"a{op}"], g[f"b{op}"], g[f"r{op}"]
        got = np.array([L.orc_half_op(op, int(x), int(y)) for x, y in zip(a, b)], dtype=np.uint16)
        assert _same(got, want).all(), f"op {op}"
    f = g["f32_in"]
    got = np.array([L.orc_f32_to_f16(float(x)) for x in f], dtype=np.uint16)
    assert np.array_equal(got, g["f32_to_f16"])
