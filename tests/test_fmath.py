"""include/lprt_fmath.h: the single-precision transcendentals that shape rays (Box-Muller's logf / sinf / cosf of
curand_normal, rt/rtrt/cuda.hpp:1927-1928, 1967-1969; tan(fovy / 2) of the primary rays, cuda.hpp:803-805).  One header
is compiled into the HIP kernels and into the CPU oracle so that both cast bit-identical rays.  CPU: accuracy against
double-precision libm on the arguments curand_normal produces (CUDA documents 1 / 2 / 2 / 4 ulp for the logf / sinf /
cosf / tanf the reference calls).  GPU: the device evaluates every function to the same bits as the host."""
import numpy as np
import pytest

from oracle import binding


def boxmuller_arguments(n, seed):
    x = np.random.default_rng(seed).integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    x[:4] = [0, 1, 2**32 - 1, 2**31]
    u = x.astype(np.float32) * np.float32(2.3283064e-10) + np.float32(2.3283064e-10 / 2.0)   # curand_uniform
    v = x.astype(np.float32) * np.float32(1.46291807e-09) + np.float32(1.46291807e-09 / 2.0)  # 2 pi u
    return u, v


def ulps(got, want):
    w32 = want.astype(np.float32)
    return np.abs(got.astype(np.float64) - want) / np.spacing(np.abs(w32)).astype(np.float64)


def test_accuracy_against_double_precision():
    u, v = boxmuller_arguments(400000, 1)
    assert ulps(binding.fmath("log", u), np.log(u.astype(np.float64))).max() <= 0.9
    for op, f in (("sin", np.sin), ("cos", np.cos)):
        got, want = binding.fmath(op, v), f(v.astype(np.float64))
        assert (np.abs(got - want) / np.spacing(np.float32(1.0))).max() <= 0.8  # absolute: what a direction component sees
        big = np.abs(want) > 1e-3
        assert ulps(got[big], want[big]).max() <= 1.7
    x = np.linspace(1e-3, 1.5, 300001, dtype=np.float32)  # fovy / 2
    assert ulps(binding.fmath("tan", x), np.tan(x.astype(np.float64))).max() <= 3.5
    assert binding.fmath("log", np.float32([1.0]))[0] == 0.0
    s = binding.fmath("sin", v[:1000]).astype(np.float64)
    c = binding.fmath("cos", v[:1000]).astype(np.float64)
    assert np.abs(s * s + c * c - 1).max() < 3e-7


@pytest.mark.gpu
def test_device_evaluates_the_same_bits(cornell):
    import lprt_amd
    from lprt_amd import abi

    u, v = boxmuller_arguments(1 << 20, 2)
    with lprt_amd.Renderer(16, 16, abi.F32) as r:
        for op, x in (("log", u), ("sin", v), ("cos", v), ("tan", np.linspace(1e-3, 1.5, 1 << 18, dtype=np.float32)),
                      ("sin", np.linspace(-60, 60, 1 << 18, dtype=np.float32)), ("log", np.float32([1e-40, 1e-38, 3e-39, 1e30]))):
            a, b = r.fmath(op, x), binding.fmath(op, x)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (op, int((a.view(np.uint32) != b.view(np.uint32)).sum()))
