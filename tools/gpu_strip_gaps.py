"""Exploration: one rank's strip (halo untraced) of the 1080p Cornell frame, N frames and nothing else -- for a
rocprofv3 --kernel-trace run whose summed kernel time per frame is compared with the wall time per frame (launch gaps).
usage: gpu_strip_gaps.py [world] [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lprt_amd
from lprt_amd import abi, scenes, dist as lp_dist

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
W, H, spp = 1920, 1080, 4
sc = scenes.SCENES["cornell"]()
rank = world // 2
own_b, own_e = lp_dist.strip_rows(H, world, rank)
win_b, win_e = lp_dist.window_rows(H, world, rank) if world > 1 else (0, H)
with lprt_amd.Renderer(W, H, abi.F32, device=0, row_begin=win_b, row_end=win_e) as r:
    r.load_scene(sc)
    out = torch.zeros((own_e - own_b, W, 4), dtype=torch.float32, device="cuda:0")  # as bench.py: the rank's rows of the gathered frame
    r.set_output_target(out.data_ptr(), own_b, own_e - own_b)
    if world > 1:
        r.set_sample_rows(own_b, own_e - own_b)
    fs = abi.default_settings(spp=spp, gi_on=True, svgf=True)
    prepared = r.prepare_scene(sc)
    for _ in range(5):
        r.set_scene(prepared=prepared); r.render(fs)
    r.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        r.set_scene(prepared=prepared); r.render(fs)
    r.synchronize()
    dt = time.perf_counter() - t0
print(f"world={world} frames={frames} wall {1e3 * dt / frames:.4f} ms/frame")
