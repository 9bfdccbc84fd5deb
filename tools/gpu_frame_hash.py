"""Hashes of every readable buffer after a short moving-Cornell sequence (odd frame size, both precisions): run it under two
libraries (LPRT_LIBRARY=...) and diff the outputs -- an A/B that must not change a bit prints the same lines.
HASH_DUMP=<prefix> also saves the denoised planes as <prefix>_<prec>_<name>.npy.
usage: gpu_frame_hash.py [W] [H] [spp] [frames] [scene] [rebuild_blas]   (scene other than cornell: static, camera and all)"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lprt_amd
from lprt_amd import abi, scenes
a = sys.argv[1:]
W, H, spp, frames = (int(a[0]), int(a[1]), int(a[2]), int(a[3])) if len(a) > 3 else (333, 187, 2, 4)
scene_name = a[4] if len(a) > 4 else "cornell"
rebuild = int(a[5]) if len(a) > 5 else 0
base = scenes.cornell_box() if scene_name == "cornell" else scenes.SCENES[scene_name]()
for prec in (abi.F32, abi.F16):
    with lprt_amd.Renderer(W, H, prec) as r:
        r.load_scene(base)
        fs = abi.default_settings(spp=spp)
        fs.rebuild_blas = rebuild
        for k in range(frames):
            r.set_scene(scenes.moving_cornell(k, base) if scene_name == "cornell" else base); r.render(fs)
        r.synchronize()
        for name in sorted(abi.BUF, key=abi.BUF.get):
            try:
                x = r.read(name)
            except Exception as e:  # a buffer this configuration does not keep
                continue
            if os.environ.get("HASH_DUMP") and name in ("OUTPUT", "MUL_GI_COLORED", "MUL_GI_WHITE"):
                np.save(f"{os.environ['HASH_DUMP']}_{prec}_{name}.npy", x)
            print(f"prec={prec} {name:24s} {hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest()[:16]} nonfinite={int((~np.isfinite(x)).sum())}")
