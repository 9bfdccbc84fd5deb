"""Profiling driver: N frames of one workload, nothing else (for rocprofv3 --kernel-trace / --pmc passes).
usage: gpu_atrium_frames.py [scene] [W] [H] [spp] [f32|f16] [frames] [rebuild]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lprt_amd
from lprt_amd import abi, scenes
a = sys.argv[1:]
scene = a[0] if len(a) > 0 else "sponza_like"
W, H, spp = int(a[1]) if len(a) > 1 else 1920, int(a[2]) if len(a) > 2 else 1080, int(a[3]) if len(a) > 3 else 8
prec = abi.F16 if (a[4] if len(a) > 4 else "f16") == "f16" else abi.F32
frames, rebuild = int(a[5]) if len(a) > 5 else 3, int(a[6]) if len(a) > 6 else 0
sc = scenes.SCENES[scene]()
with lprt_amd.Renderer(W, H, prec) as r:
    for opt in os.environ.get("LPRT_OPTS", "").split(","):
        if "=" in opt:
            k, v = opt.split("=")
            r.set_option(k, int(v))
    r.load_scene(sc)
    fs = abi.default_settings(spp=spp)
    fs.rebuild_blas = rebuild
    for _ in range(frames):
        r.set_scene(sc)
        r.render(fs)
    r.synchronize()
print("done", scene, W, H, spp, prec, frames)
