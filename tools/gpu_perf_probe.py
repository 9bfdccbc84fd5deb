"""Exploration harness (not the headline bench): A/B timing of renderer variants in ONE process, Sponza-like configs."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lprt_amd
from lprt_amd import abi, scenes


def timed(r, sc, fs, frames=10, warm=2):
    for _ in range(warm):
        r.set_scene(sc); r.render(fs)
    r.synchronize()
    r.enable_stage_timing(True); r.reset_stage_times()
    t0 = time.perf_counter()
    for _ in range(frames):
        r.set_scene(sc); r.render(fs)
    r.synchronize()
    dt = (time.perf_counter() - t0) / frames
    st = {k: round(v[0] / frames, 3) for k, v in r.stage_times().items() if v[1]}
    r.enable_stage_timing(False)
    return dt * 1e3, st


def main():
    which = sys.argv[1:] or ["modes", "sponza"]
    if "modes" in which:
        sc = scenes.cornell_box()
        for prec in (abi.F32, abi.F16):
            r = lprt_amd.Renderer(1920, 1080, prec); r.load_scene(sc)
            fs = abi.default_settings(spp=4)
            for rep in range(2):
                for mode in ("0", "1"):
                    r.set_option("tlas_mode", int(mode))
                    ms, st = timed(r, sc, fs, frames=8)
                    print(f"cornell prec={prec} tlas_mode={mode} rep={rep}: {ms:.3f} ms  primary={st['primary']} di1={st['trace_di1']} gi={st['trace_gi']} di2={st['trace_di2']}", flush=True)
            r.close()
    if "sponza" in which:
        sc = scenes.sponza_like()
        for prec in (abi.F32, abi.F16):
            r = lprt_amd.Renderer(1920, 1080, prec); r.load_scene(sc)
            r.enable_counters(True); r.reset_counters(); r.set_scene(sc); r.render(abi.default_settings(spp=8)); r.synchronize()
            rays = r.counters()["rays"]; r.enable_counters(False)
            for rebuild in (0, 1):
                fs = abi.default_settings(spp=8); fs.rebuild_blas = rebuild
                for mode in ("0", "1"):
                    r.set_option("tlas_mode", int(mode))
                    ms, st = timed(r, sc, fs, frames=5)
                    print(f"sponza_like 1080p 8spp prec={prec} rebuild={rebuild} tlas_mode={mode}: {ms:.3f} ms/frame {rays/ms/1e3:.0f} Mray/s  {json.dumps(st)}", flush=True)
            r.close()
    if "sponza_parity" in which:
        from oracle.binding import Oracle
        sc = scenes.sponza_like()
        for prec, tol in ((abi.F32, 1e-3), (abi.F16, 5e-2)):
            W, H = 320, 180
            o = Oracle(W, H, prec, threads=min(len(os.sched_getaffinity(0)), 32)); o.load_scene(sc)
            r = lprt_amd.Renderer(W, H, prec); r.load_scene(sc)
            for f in range(2):
                fs = abi.default_settings(spp=2)
                o.set_scene(sc); o.render_frame(fs); r.set_scene(sc); r.render(fs)
                for name in ("G_OBJID", "INTENSITY0", "INTENSITY1", "RAW_GI_COLORED", "MUL_GI_COLORED", "COLOR_INPROGRESS", "OUTPUT"):
                    a, b = r.read(name).astype(np.float64), o.read(name).astype(np.float64)
                    a, b = np.nan_to_num(a, nan=0, posinf=0, neginf=0), np.nan_to_num(b, nan=0, posinf=0, neginf=0)
                    print(f"sponza parity prec={prec} frame={f} {name:18s} rel_l2={np.linalg.norm(a-b)/max(np.linalg.norm(b),1e-30):.3e} mismatch_px={int((np.abs(a-b)>1e-3*(1+np.abs(b))).any(axis=-1).sum())}", flush=True)


if __name__ == "__main__":
    main()
