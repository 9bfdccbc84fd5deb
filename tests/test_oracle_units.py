"""Known-answer and property tests of the oracle's building blocks."""
import ctypes as C

import numpy as np

from oracle import binding
from lprt_amd import abi


def _brdf(prec, m, r, V, L_, N):
    lib = binding.lib()
    out = (C.c_float * 2)()
    v3 = lambda v: (C.c_float * 3)(*v)  # noqa: E731
    lib.orc_material_brdf(prec, m, r, v3(V), v3(L_), v3(N), out)
    return out[0], out[1]


def test_material_brdf_matches_reference_probe():
    """the reference's own bsdf.hpp, compiled on the host during the survey, gave these values
    for material_brdf(0.5, 0.5, V=(0,0,1), L=(.6,0,.8), N=(0,0,1))."""
    c, w = _brdf(abi.F32, 0.5, 0.5, (0, 0, 1), (.6, 0, .8), (0, 0, 1))
    assert abs(c - 0.279012978) < 1e-8 and abs(w - 0.0050490615) < 1e-9
    c, w = _brdf(abi.F16, 0.5, 0.5, (0, 0, 1), (.6, 0, .8), (0, 0, 1))
    assert c == np.float32(0.27758789) and w == np.float32(0.0049896240)


def test_brdf_below_horizon_is_zero():
    assert _brdf(abi.F32, 0.3, 0.4, (0, 0, 1), (0.6, 0, -0.8), (0, 0, 1)) == (0.0, 0.0)


def _xorwow(seed, sub, kind, n):
    lib = binding.lib()
    if kind == 0:
        out = np.zeros(n, dtype=np.uint32)
    else:
        out = np.zeros(n, dtype=np.float32)
    lib.orc_xorwow(int(seed), int(sub), kind, n, out.ctypes.data_as(C.c_void_p))
    return out


def _state(seed, sub):
    out = np.zeros(6, dtype=np.uint32)
    binding.lib().orc_xorwow_state(int(seed), int(sub), out.ctypes.data_as(C.c_void_p))
    return out


def test_xorwow_seed_scramble_and_first_draw():
    """curand_init(seed=0, 0, 0): state is the xorwow constants plus the documented scramble; the first output is
    one plain Marsaglia step + Weyl increment of it."""
    s0, s1 = 0 ^ 0xaad26b49, 0 ^ 0xf7dcefdd
    t0, t1 = (1099087573 * s0) & 0xffffffff, (2591861531 * s1) & 0xffffffff
    want = np.array([(123456789 + t0) & 0xffffffff, 362436069 ^ t0, (521288629 + t1) & 0xffffffff, 88675123 ^ t1,
                     (5783321 + t0) & 0xffffffff, (6615241 + t1 + t0) & 0xffffffff], dtype=np.uint32)
    st = _state(0, 0)
    assert np.array_equal(st, want)
    v = [int(x) for x in st[:5]]
    t = v[0] ^ (v[0] >> 2)
    v4 = (v[4] ^ ((v[4] << 4) & 0xffffffff)) ^ (t ^ ((t << 1) & 0xffffffff))
    first = (v4 + int(st[5]) + 362437) & 0xffffffff
    assert int(_xorwow(0, 0, 0, 1)[0]) == first


def test_xorwow_subsequence_is_a_linear_group_action():
    """state(seed, a+b) == skip_b(state(seed, a)); d never changes (2^67 * 362437 = 0 mod 2^32)."""
    lib = binding.lib()
    m = np.zeros((32, 160, 5), dtype=np.uint32)
    for k in range(32):
        lib.orc_xorwow_matrix(k, m[k].ctypes.data_as(C.c_void_p))

    def apply(mat, v):
        r = np.zeros(5, dtype=np.uint32)
        for i in range(160):
            if (int(v[i // 32]) >> (i % 32)) & 1:
                r ^= mat[i]
        return r

    for seed, a, b in [(1, 5, 9), (123456789012345, 1000, 77), (7, 2073599, 1)]:
        sa, sab = _state(seed, a), _state(seed, a + b)
        v = sa[:5].copy()
        for k in range(32):
            if (b >> k) & 1:
                v = apply(m[k], v)
        assert np.array_equal(v, sab[:5]) and sa[5] == sab[5]
    # matrices are powers of one another: M^(2^(k+1)) = (M^(2^k))^2 on random vectors
    rng = np.random.default_rng(3)
    for k in (0, 5, 17):
        v = rng.integers(0, 2**32, 5, dtype=np.uint64).astype(np.uint32)
        assert np.array_equal(apply(m[k + 1], v), apply(m[k], apply(m[k], v)))


def test_xorwow_uniform_and_normal_ranges():
    u = _xorwow(42, 3, 1, 20000)
    assert u.min() > 0.0 and u.max() <= 1.0 and abs(u.mean() - 0.5) < 0.01
    n = _xorwow(42, 3, 2, 20000)
    assert abs(n.mean()) < 0.03 and abs(n.std() - 1.0) < 0.03
    # distinct subsequences are distinct streams
    assert not np.array_equal(_xorwow(42, 3, 0, 8), _xorwow(42, 4, 0, 8))
