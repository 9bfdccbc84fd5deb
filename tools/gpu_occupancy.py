"""Lane occupancy of the traversal loops (instrumented kernels): per stage and phase, lane_steps / (64 * wave_steps), and steps
per ray.  usage: gpu_occupancy.py [scene] [W] [H] [spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lprt_amd
from lprt_amd import abi, scenes
a = sys.argv[1:]
scene = a[0] if a else "sponza_like"
W, H, spp = (int(a[1]), int(a[2]), int(a[3])) if len(a) > 3 else (1920, 1080, 2)
sc = scenes.SCENES[scene]()
PH = ("entry", "node", "leaf")
for prec in (abi.F32, abi.F16):
    for opts in (dict(ray_stream=0), dict(ray_stream=1)):
        with lprt_amd.Renderer(W, H, prec) as r:
            for k, v in opts.items():
                r.set_option(k, v)
            for kv in os.environ.get("LPRT_OPTS", "").split(","):
                if "=" in kv:
                    r.set_option(kv.split("=")[0], int(kv.split("=")[1]))
            r.load_scene(sc)
            r.enable_counters(True)
            r.reset_counters()
            r.render(abi.default_settings(spp=spp))
            for st in ("primary", "trace_di1", "trace_gi", "trace_di2"):
                c = r.stage_counters(st)
                if not c["rays"]:
                    continue
                occ = " ".join(f"{PH[k]} {c['lane_steps'][k] / max(1, 64 * c['wave_steps'][k]):.2f} ({c['wave_steps'][k] * 64 / c['rays']:.1f} wave-steps/ray-lane, {c['lane_steps'][k] / c['rays']:.1f} steps/ray)" for k in range(3))
                print(f"prec={prec} {opts} {st:9s} rays {c['rays']:9d} nodes/ray {c['blas_node_visits'] / c['rays']:.1f} tris/ray {c['tri_tests'] / c['rays']:.2f} objects/ray {c['object_visits'] / c['rays']:.2f} | {occ}", flush=True)
