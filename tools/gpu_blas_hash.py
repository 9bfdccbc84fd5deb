"""Hash of every BLAS of a scene (node records: root, child boxes, leaf words; leaf order) -- run under two libraries
(LPRT_LIBRARY=...) to check that a builder change leaves the trees bit for bit alone.  usage: gpu_blas_hash.py [scene] [f32|f16]"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lprt_amd
from lprt_amd import abi, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "sponza_like"
prec = abi.F16 if (sys.argv[2] if len(sys.argv) > 2 else "f16") == "f16" else abi.F32
sc = scenes.SCENES[name]()
h = hashlib.sha256()
with lprt_amd.Renderer(64, 64, prec) as r:
    r.load_scene(sc)
    r.build_blas()
    n = len(sc.c_vaos()) if hasattr(sc, "c_vaos") else 0
    for v in range(n):
        info = r.blas_info(v)
        for arr in r.blas_nodes(v):
            h.update(np.ascontiguousarray(arr).tobytes())
        h.update(np.ascontiguousarray(r.blas_order(v)).tobytes())
    print(name, "prec", prec, "vaos", n, h.hexdigest()[:24])
