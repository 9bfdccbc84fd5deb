"""A/B: per-pixel traversal kernels vs ray streams on one workload; stage times per frame.
usage: gpu_stream_probe.py [scene] [W] [H] [spp] [frames]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lprt_amd
from lprt_amd import abi, scenes
a = sys.argv[1:]
scene = a[0] if a else "sponza_like"
W, H, spp, frames = (int(a[1]), int(a[2]), int(a[3]), int(a[4])) if len(a) > 4 else (1920, 1080, 8, 4)
sc = scenes.SCENES[scene]()
variants = [dict(ray_stream=0), dict(ray_stream=1), dict(ray_stream=1, stream_wide_closest=1), dict(ray_stream=1, stream_refill_min=16),
            dict(ray_stream=1, stream_refill_min=48), dict(ray_stream=1, stream_node_min=8), dict(ray_stream=1, stream_node_min=32),
            dict(ray_stream=1, stream_node_min=44)]
if os.environ.get("PROBE_VARIANTS"):
    variants = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in v.split(",") if kv) for v in os.environ["PROBE_VARIANTS"].split(";")]
ref = {}
for prec in (abi.F32, abi.F16):
    for v in variants:
        with lprt_amd.Renderer(W, H, prec) as r:
            for k, val in v.items():
                r.set_option(k, val)
            r.load_scene(sc)
            fs = abi.default_settings(spp=spp)
            for _ in range(2):
                r.set_scene(sc); r.render(fs)
            r.synchronize()
            r.enable_stage_timing(True); r.reset_stage_times()
            t0 = time.perf_counter()
            for _ in range(frames):
                r.set_scene(sc); r.render(fs)
            r.synchronize()
            dt = (time.perf_counter() - t0) / frames * 1e3
            st = {k: round(x[0] / frames, 2) for k, x in r.stage_times().items() if x[1] and x[0] / frames > 0.3}
            out = r.read("OUTPUT")
            key = prec
            same = None
            if key in ref:
                same = bool(np.array_equal(out, ref[key], equal_nan=True))
                ndiff = int((out != ref[key]).any(axis=-1).sum())
            else:
                ref[key] = out; ndiff = 0
            print(f"prec={prec} {v}: {dt:.2f} ms/frame  identical_to_first={same} (px differing {ndiff})  {json.dumps(st)}", flush=True)
