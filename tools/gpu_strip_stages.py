import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lprt_amd
from lprt_amd import abi, scenes, dist as lp_dist
sc = scenes.SCENES["cornell"]()
W, H, spp, world = 1920, 1080, 4, 8
rank = 4
own_b, own_e = lp_dist.strip_rows(H, world, rank)
win_b, win_e = lp_dist.window_rows(H, world, rank)
for label, (b, e), lean in (("full", (0, H), False), ("lean strip", (win_b, win_e), True)):
    r = lprt_amd.Renderer(W, H, abi.F32, device=0, row_begin=b, row_end=e)
    r.load_scene(sc)
    if lean:
        r.set_sample_rows(own_b, own_e - own_b)
    fs = abi.default_settings(spp=spp, gi_on=True, svgf=True)
    prepared = r.prepare_scene(sc)
    for _ in range(3):
        r.set_scene(prepared=prepared); r.render(fs)
    r.enable_stage_timing(True); r.reset_stage_times()
    n = 10
    for _ in range(n):
        r.set_scene(prepared=prepared); r.render(fs)
    r.synchronize()
    st = r.stage_times()
    print(label, {k: (round(1e3 * v[0] / n, 1), v[1] // n) for k, v in st.items() if v[1]}, "sum us", round(1e3 * sum(v[0] for v in st.values()) / n, 1))
    r.close()
