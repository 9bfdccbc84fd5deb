"""Time the GI stage for individual seed triples: python tools/gpu_seed_probe.py scene n [prec]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lprt_amd
from lprt_amd import abi, scenes
name, n = sys.argv[1], int(sys.argv[2])
prec = abi.F16 if len(sys.argv) > 3 and sys.argv[3] == "f16" else abi.F32
sc = scenes.SCENES[name]()
r = lprt_amd.Renderer(1920, 1080, prec)
r.load_scene(sc)
r.enable_stage_timing(True)
r.enable_counters(True)
for k in range(n):
    a = 3 * k + 1
    r.reset_history(); r.reset_stage_times(); r.reset_counters()
    r.set_scene(sc)
    r.render(abi.default_settings(spp=1, seeds=[a, a + 1, a + 2]))
    r.synchronize()
    st = r.stage_times()
    c = r.stage_counters("trace_gi")
    print(f"k={k} seedB={a+1} gi={st['trace_gi'][0]:.2f} ms di2={st['trace_di2'][0]:.2f} rays={c['rays']} nodes/ray={c['blas_node_visits']/max(1,c['rays']):.1f} tris/ray={c['tri_tests']/max(1,c['rays']):.2f}", flush=True)
