"""Round-3 parity probe (GPU box): everything the fp16 parity statements of DESIGN.md section 2 are made of, printed as JSON lines.

  A  hit records of random rays on the atrium, half precision: HIP (default slab, reference slab) vs the oracle under
     three tie orders, pairwise mismatch counts, and the disputed rays adjudicated by exhaustive search;
  B  atrium frames (256x144, 2 spp, BLAS rebuilt per frame): ray-by-ray classes + relative L2 of every radiance plane
     over ALL pixels, fp32 and fp16 (both slab modes), next to the oracle-vs-oracle topology baseline;
  C  moving Cornell, fp16, both slab modes: relative L2 of every radiance plane over all pixels per frame.

    python tools/gpu_parity_probe.py [A] [B] [C]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lprt_amd  # noqa: E402
from lprt_amd import abi, scenes  # noqa: E402
from oracle.binding import Oracle  # noqa: E402
import replay  # noqa: E402
from test_gpu_trace import random_rays  # noqa: E402

PLANES = ["INTENSITY0", "INTENSITY1", "RAW_COLOR", "RAW_GI_COLORED", "RAW_GI_WHITE", "MUL_GI_COLORED", "MUL_GI_WHITE", "COLOR_INPROGRESS", "OUTPUT"]


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    fin = np.isfinite(a) & np.isfinite(b)
    return float(np.linalg.norm(np.where(fin, a - b, 0)) / max(np.linalg.norm(np.where(fin, b, 0)), 1e-30))


def emit(**kw):
    print(json.dumps(kw), flush=True)


def part_a(sc):
    prec = abi.F16
    rays = random_rays(sc, 40000, 13, 0.1)
    ans = {}
    orcs = {}
    for tie in (0, 1, 2):
        o = Oracle(16, 16, prec)
        o.set_tie_order(tie)
        o.load_scene(sc)
        ans[f"oracle{tie}"] = o.trace(rays, False)
        orcs[tie] = o
    for mode in (0, 1):
        with lprt_amd.Renderer(16, 16, prec) as r:
            r.set_option("reference_slab", mode)
            r.load_scene(sc)
            t0 = time.time()
            ans[f"hip_slab{mode}"] = r.trace(rays, False)
            emit(part="A", what="trace_seconds", mode=mode, s=time.time() - t0)
    names = list(ans)
    for i, a in enumerate(names):
        for b in names[i + 1:]:
            d = replay.closest_differs(ans[a], ans[b])
            emit(part="A", pair=[a, b], differ=int(d.sum()), hitflag=int((ans[a]["hit"] != ans[b]["hit"]).sum()), n=len(rays))
    for g in ("hip_slab0", "hip_slab1", "oracle1", "oracle2"):
        d = replay.closest_differs(ans[g], ans["oracle0"])
        j = replay.adjudicate(orcs[0], rays[d], ans[g][d], ans["oracle0"][d])
        gn, on = j["gpu_nearest"], j["oracle_nearest"]
        emit(part="A", adjudicate=g, against="oracle0", disputed=int(d.sum()), both_nearest=int((gn & on).sum()), only_this=int((gn & ~on).sum()),
             only_oracle0=int((~gn & on).sum()), neither=int((~gn & ~on).sum()))
    # undisputed rays: a sample against the exhaustive search
    same = ~replay.closest_differs(ans["hip_slab0"], ans["oracle0"])
    idx = np.flatnonzero(same)[:600]
    j = replay.adjudicate(orcs[0], rays[idx], ans["hip_slab0"][idx], ans["oracle0"][idx])
    emit(part="A", undisputed_sample=len(idx), nearest=int(j["gpu_nearest"].sum()))
    for o in orcs.values():
        o.close()


def frame_rows(tag, r, o, c, npx):
    return dict(tag=tag, rel_l2={n: rel_l2(r.read(n), o.read(n)) for n in PLANES},
                px={"primary": int(c["primary"].sum()), "shadow0": int(c["shadow0"].sum()), "any": int(c["any"].sum()),
                    "gi": [int(m.sum()) for m in c["gi"]], "gi_tie": [int(m.sum()) for m in c["gi_tie"]],
                    "gi_other_ray": [int(m.sum()) for m in c["gi_other_ray"]], "gi_cull": [int(m.sum()) for m in c["gi_cull"]],
                    "shadow1": [int(m.sum()) for m in c["shadow1"]]}, npx=npx)


def part_b(sc):
    W, H, spp = 256, 144, 2
    for prec, modes in ((abi.F32, (0,)), (abi.F16, (0, 1))):
        for mode in modes:
            o, probe_o = Oracle(W, H, prec), Oracle(W, H, prec)
            r, probe_g = lprt_amd.Renderer(W, H, prec), lprt_amd.Renderer(W, H, prec)
            for x in (r, probe_g):
                x.set_option("reference_slab", mode)
            for x in (o, probe_o, r, probe_g):
                x.load_scene(sc)
            for f in range(2):
                fs = abi.default_settings(spp=spp)
                fs.rebuild_blas = 1
                o.set_scene(sc)
                o.render_frame(fs)
                r.set_scene(sc)
                r.render(fs)
                c = replay.classify(r, o, sc, prec, probe_g, probe_o, f, spp, rebuild_blas=1)
                row = frame_rows(f"hip prec={prec} slab={mode} frame={f}", r, o, c, W * H)
                emit(part="B", **row)
            for x in (o, probe_o, r, probe_g):
                x.close()
    # oracle-vs-oracle topology baseline, fp16
    prec = abi.F16
    pair = []
    for tie in (1, 2):
        o, probe = Oracle(W, H, prec), Oracle(W, H, prec)
        for x in (o, probe):
            x.set_tie_order(tie)
            x.load_scene(sc)
        pair.append((o, probe))
    (oa, pa), (ob, pb) = pair
    for f in range(2):
        fs = abi.default_settings(spp=spp)
        for o in (oa, ob):
            o.set_scene(sc)
            o.render_frame(fs)
        c = replay.classify(oa, ob, sc, prec, pa, pb, f, spp)
        emit(part="B", **frame_rows(f"oracle tie1 vs tie2 frame={f}", oa, ob, c, W * H))


def part_c():
    W, H = 160, 96
    cornell = scenes.cornell_box()
    for mode in (0, 1):
        o = Oracle(W, H, abi.F16)
        o.load_scene(cornell)
        with lprt_amd.Renderer(W, H, abi.F16) as r:
            r.set_option("reference_slab", mode)
            r.load_scene(cornell)
            for k in range(4):
                sc = scenes.moving_cornell(k, cornell)
                fs = abi.default_settings(spp=1)
                o.set_scene(sc)
                o.render_frame(fs)
                r.set_scene(sc)
                r.render(fs)
                emit(part="C", slab=mode, frame=k, rel_l2={n: rel_l2(r.read(n), o.read(n)) for n in PLANES},
                     primary=int(replay.primary_mismatch(r, o).sum()))
        o.close()


if __name__ == "__main__":
    parts = [a for a in sys.argv[1:] if a in ("A", "B", "C")] or ["A", "B", "C"]
    sponza = scenes.SCENES["sponza_like"]()
    if "A" in parts:
        part_a(sponza)
    if "C" in parts:
        part_c()
    if "B" in parts:
        part_b(sponza)
