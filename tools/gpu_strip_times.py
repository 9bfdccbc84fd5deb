#!/usr/bin/env python3
"""Frame time of one rank's row window (strip + halo) against the strip alone: what the redundant halo tracing costs.
usage: gpu_strip_times.py [scene] [W H spp] [world]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lprt_amd
from lprt_amd import abi, scenes, dist as lp_dist

name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
W, H, spp = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080, 4)
world = int(sys.argv[5]) if len(sys.argv) > 5 else 8
sc = scenes.SCENES[name]()
for prec, pn in ((abi.F32, "f32"), (abi.F16, "f16")):
    rank = world // 2
    own_b, own_e = lp_dist.strip_rows(H, world, rank)
    win_b, win_e = lp_dist.window_rows(H, world, rank)
    for label, (b, e) in (("full", (0, H)), ("strip+halo", (win_b, win_e)), ("halo untraced", (win_b, win_e)), ("strip", (own_b, own_e))):
        r = lprt_amd.Renderer(W, H, prec, device=0, row_begin=b, row_end=e)
        r.load_scene(sc)
        if label == "halo untraced":  # what a rank computes by default: spp loop on the strip, everything else on the window
            r.set_sample_rows(own_b, own_e - own_b)
        fs = abi.default_settings(spp=spp, gi_on=True, svgf=True)
        prepared = r.prepare_scene(sc)
        for _ in range(3):
            r.set_scene(prepared=prepared); r.render(fs)
        r.synchronize()
        n = 20 if name == "cornell" else 5
        t0 = time.perf_counter()
        for _ in range(n):
            r.set_scene(prepared=prepared); r.render(fs)
        r.synchronize()
        print(f"{name} {W}x{H} {spp}spp {pn} world={world} {label:13s} rows={e-b:5d}  {1e3*(time.perf_counter()-t0)/n:8.3f} ms", flush=True)
        r.close()
