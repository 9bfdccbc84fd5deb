"""Generates tests/golden/*.npz.

  half_ops.npz        outputs of the REFERENCE's vendored third_party/half/half.hpp (through oracle/_ref, which only
                      exists where /root/reference is mounted) on fixed random operands: the one piece of the
                      reference that can run here.
  cornell_c1_f32.npz  BASELINE.json configs[0]: Cornell box, 256x256, 1 spp, fp32, seeds (1,2,3), frame 0 — rendered
                      by the CPU oracle (the reference has no CPU path, tests or fixtures of its own: these vectors
                      pin the oracle against regressions and pin the GPU path on machines without the oracle).
  cornell_f16_64.npz  same scene, 64x64, fp16, two frames (temporal path).
  rays_small.npz      hit records of a fixed ray batch on the random test scene (fp32 + fp16).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import lprt_amd  # noqa: E402,F401
from lprt_amd import abi, scenes  # noqa: E402
from oracle import binding  # noqa: E402
from oracle.binding import Oracle  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def golden_rays(sc, n, seed):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, dtype=abi.RAY_DTYPE)
    lo = sc.aabbs[:, :3].min(axis=0) - 0.25
    hi = sc.aabbs[:, 3:].max(axis=0) + 0.25
    rays["source"] = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    rays["direction"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays["min_distance"] = 1e-4
    rays["max_distance"] = 1e5
    rays["skip_vao"] = 0xFFFFFFFF
    return rays


def main():
    R = binding.ref_half_lib()
    if R is None:
        raise SystemExit("oracle/_ref/libref_half.so missing: run `make -C oracle ref` where /root/reference exists")
    rng = np.random.default_rng(20250304)
    d = {}
    for op in range(5):
        a = rng.integers(0, 65536, 4000).astype(np.uint16)
        b = rng.integers(0, 65536, 4000).astype(np.uint16)
        d[f"a{op}"], d[f"b{op}"] = a, b
        d[f"r{op}"] = np.array([R.ref_half_op(op, int(x), int(y)) for x, y in zip(a, b)], dtype=np.uint16)
    f = (rng.normal(size=4000) * rng.choice([1e-8, 1e-5, 1e-2, 1, 100, 7e4], 4000)).astype(np.float32)
    d["f32_in"] = f
    d["f32_to_f16"] = np.array([R.ref_f32_to_f16(float(x)) for x in f], dtype=np.uint16)
    np.savez_compressed(os.path.join(OUT, "half_ops.npz"), **d)

    sc = scenes.cornell_box()
    o = Oracle(256, 256, abi.F32)
    o.load_scene(sc)
    o.render_frame(abi.default_settings(spp=1))
    np.savez_compressed(os.path.join(OUT, "cornell_c1_f32.npz"),
                        output=o.read("OUTPUT")[..., :3].astype(np.float32),
                        raw_gi_colored=o.read("RAW_GI_COLORED").astype(np.float16),
                        g_objid=o.read("G_OBJID")[..., 0].astype(np.uint8), g_offset=o.read("G_OFFSET")[..., 0].astype(np.uint8),
                        g_alpha=(o.read("G_COLOR")[..., 3] > 0))
    o.close()

    o = Oracle(64, 64, abi.F16)
    o.load_scene(sc)
    frames = []
    for _ in range(2):
        o.set_scene(sc)
        o.render_frame(abi.default_settings(spp=2))
        frames.append(o.read("OUTPUT")[..., :3].astype(np.float32))
    np.savez_compressed(os.path.join(OUT, "cornell_f16_64.npz"), frame0=frames[0], frame1=frames[1],
                        g_objid=o.read("G_OBJID")[..., 0].astype(np.uint8))
    o.close()

    st = scenes.small_test_scene()
    rays = golden_rays(st, 20000, 5)
    d = {"rays": rays}
    for prec, tag in ((abi.F32, "f32"), (abi.F16, "f16")):
        o = Oracle(32, 32, prec)
        o.load_scene(st)
        r = rays.copy()
        if prec == abi.F16:
            r["min_distance"] = 0.1
        d[f"closest_{tag}"] = o.trace(r, False)
        d[f"any_{tag}"] = o.trace(r, True)["hit"].astype(np.uint8)
        o.close()
    np.savez_compressed(os.path.join(OUT, "rays_small.npz"), **d)
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
