import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lprt_amd
from lprt_amd import abi, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
prec = abi.F16 if (len(sys.argv) > 2 and sys.argv[2] == "f16") else abi.F32
sc = scenes.SCENES[name]()
r = lprt_amd.Renderer(1920, 1080, prec); r.load_scene(sc)
r.enable_counters(True); r.reset_counters()
r.render(abi.default_settings(spp=1)); r.synchronize()
for st in ("primary", "trace_di1", "trace_gi", "trace_di2"):
    c = r.stage_counters(st)
    n = max(1, c["rays"])
    print(st, "rays", c["rays"], {k: round(v / n, 3) for k, v in c.items() if k != "rays"})
