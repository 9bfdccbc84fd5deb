"""The host-side floor of a frame: set_scene + render of a frame so small that the GPU time is negligible, timed with and
without stage timing; the number of kernel launches comes from rocprofv3 --kernel-trace of the same script.
usage: gpu_launch_floor.py [W] [H] [spp] [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lprt_amd
from lprt_amd import abi, scenes
a = sys.argv[1:]
W, H, spp, frames = (int(a[0]), int(a[1]), int(a[2]), int(a[3])) if len(a) > 3 else (64, 36, 4, 400)
sc = scenes.cornell_box()
for prec in (abi.F32, abi.F16):
    with lprt_amd.Renderer(W, H, prec) as r:
        r.load_scene(sc)
        prepared = r.prepare_scene(sc)
        fs = abi.default_settings(spp=spp)
        for _ in range(20):
            r.set_scene(prepared=prepared); r.render(fs)
        r.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            r.set_scene(prepared=prepared); r.render(fs)
        t_enq = (time.perf_counter() - t0) / frames * 1e3
        r.synchronize()
        t_all = (time.perf_counter() - t0) / frames * 1e3
        print(f"prec={prec} {W}x{H} {spp} spp: enqueue {t_enq:.3f} ms/frame, with the final synchronise {t_all:.3f} ms/frame")
