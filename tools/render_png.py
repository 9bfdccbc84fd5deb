"""Render a generated scene or a glTF file to PNG on the GPU: python tools/render_png.py cornell out.png [W H frames spp]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lprt_amd
from lprt_amd import abi, scenes, gltf, image_io

what, out = sys.argv[1], sys.argv[2]
W, H, frames, spp = (int(x) for x in (sys.argv[3:7] + ["640", "360", "8", "4"][len(sys.argv) - 3:]))
prec = abi.F16 if os.environ.get("LPRT_PREC") == "f16" else abi.F32
sc = scenes.SCENES[what]() if what in scenes.SCENES else gltf.load_gltf(what)
r = lprt_amd.Renderer(W, H, prec)
r.load_scene(sc)
for _ in range(frames):
    r.set_scene(sc)
    r.render(abi.default_settings(spp=spp))
image_io.save_png(out, r.read("OUTPUT"))
print("wrote", out)
