"""Host time of one frame's API calls (no synchronisation inside the loop) at a tiny frame size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lprt_amd
from lprt_amd import abi, scenes
sc = scenes.cornell_box()
r = lprt_amd.Renderer(64, 64, abi.F32)
r.load_scene(sc)
fs = abi.default_settings(spp=4)
prep = r.prepare_scene(sc)
for _ in range(20):
    r.set_scene(prepared=prep); r.render(fs)
r.synchronize()
N = 300
t_set = t_render = 0.0
t0 = time.perf_counter()
for _ in range(N):
    a = time.perf_counter(); r.set_scene(prepared=prep); b = time.perf_counter(); r.render(fs); c = time.perf_counter()
    t_set += b - a; t_render += c - b
t1 = time.perf_counter()
r.synchronize()
t2 = time.perf_counter()
print(f"host: set_scene {1e6*t_set/N:.1f} us, render_frame {1e6*t_render/N:.1f} us per frame; loop {1e3*(t1-t0)/N:.3f} ms/frame; drain {1e3*(t2-t1):.2f} ms")
