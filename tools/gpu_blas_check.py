"""BLAS orderings of the all-global and the global+local builders must be identical; build times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lprt_amd
from lprt_amd import abi, scenes
for name in ("sponza_like", "small_test", "cornell", "instanced_field"):
    sc = scenes.SCENES[name]()
    orders, times = {}, {}
    for mode in ("0", "1"):
        r = lprt_amd.Renderer(64, 64, abi.F32)
        r.set_option("blas_builder", 0)
        r.set_option("blas_local", int(mode))
        r.set_rdresource(sc)            # builds once (warm-up: allocations)
        r.enable_stage_timing(True)
        r.reset_stage_times()
        for _ in range(5):
            r.build_blas()
        r.synchronize()
        times[mode] = r.stage_times()["build_blas"][0] / 5
        orders[mode] = [r.blas_order(v) for v in range(len(sc.vaos))]
        r.close()
    same = all(np.array_equal(a, b) for a, b in zip(orders["0"], orders["1"]))
    print(f"{name:16s} tris={sc.n_tris:7d} identical order: {same}  build ms: global {times['0']:.3f}  global+local {times['1']:.3f}")
    assert same
