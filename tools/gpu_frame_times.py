"""Per-frame wall time and stage times of consecutive frames: python tools/gpu_frame_times.py scene spp frames"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lprt_amd
from lprt_amd import abi, scenes
name, spp, frames = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sc = scenes.SCENES[name]()
for prec in (abi.F32, abi.F16):
    r = lprt_amd.Renderer(1920, 1080, prec)
    r.load_scene(sc)
    r.enable_stage_timing(True)
    for k in range(frames):
        r.reset_stage_times()
        r.set_scene(sc)
        t0 = time.perf_counter()
        r.render(abi.default_settings(spp=spp))
        r.synchronize()
        dt = time.perf_counter() - t0
        st = {k_: round(v[0], 2) for k_, v in r.stage_times().items() if v[1] and v[0] > 0.5}
        print("prec", prec, "frame", k, f"{dt * 1e3:.1f} ms", st, flush=True)
    r.close()
