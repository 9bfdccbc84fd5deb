"""How much of a half-precision frame does the BLAS topology alone decide?  (test infrastructure; CPU only)

The reference builds its BLAS with std::nth_element (rt/trace/object_bvh.hpp:122-125); which of two triangles with equal
split keys goes left is implementation-defined (MSVC's there, libstdc++'s in the oracle).  Both choices are the
reference.  This tool renders the same frames with the oracle under two such orders (oracle/render.hpp: tie_order 1 / 2)
and prints, per plane, the relative L2 between them and the ray-by-ray mismatch classes of tests/replay.py -- the
baseline the HIP-vs-oracle numbers of the fp16 parity tests are held against.

    python tools/oracle_topology_baseline.py [--scene sponza_like] [--size 256x144] [--spp 2] [--frames 2] [--prec f16]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from lprt_amd import abi, scenes  # noqa: E402
from oracle.binding import Oracle  # noqa: E402
import replay  # noqa: E402

PLANES = ["INTENSITY0", "INTENSITY1", "RAW_COLOR", "RAW_GI_COLORED", "RAW_GI_WHITE", "MUL_GI_COLORED", "MUL_GI_WHITE", "COLOR_INPROGRESS", "OUTPUT"]


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    fin = np.isfinite(a) & np.isfinite(b)
    return float(np.linalg.norm((a - b)[fin]) / max(np.linalg.norm(b[fin]), 1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="sponza_like")
    ap.add_argument("--size", default="256x144")
    ap.add_argument("--spp", type=int, default=2)
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--prec", default="f16")
    ap.add_argument("--orders", default="1,2")
    ap.add_argument("--out", default=None, help="write the rows as a JSON fixture (tests/golden/topology_baseline_*.json)")
    a = ap.parse_args()
    W, H = map(int, a.size.split("x"))
    prec = abi.F16 if a.prec == "f16" else abi.F32
    sc = scenes.SCENES[a.scene]()
    ta, tb = map(int, a.orders.split(","))
    pair = []
    for tie in (ta, tb):
        o, probe = Oracle(W, H, prec), Oracle(W, H, prec)
        for x in (o, probe):
            x.set_tie_order(tie)
            x.load_scene(sc)
        pair.append((o, probe))
    (oa, pa), (ob, pb) = pair
    out = []
    for f in range(a.frames):
        fs = abi.default_settings(spp=a.spp)
        for o in (oa, ob):
            o.set_scene(sc)
            o.render_frame(fs)
        c = replay.classify(oa, ob, sc, prec, pa, pb, f, a.spp)
        npx = W * H
        row = {"frame": f, "rel_l2": {n: rel_l2(oa.read(n), ob.read(n)) for n in PLANES},
               "px": {"primary": int(c["primary"].sum()), "shadow0": int(c["shadow0"].sum()), "any": int(c["any"].sum()),
                      "gi": [int(m.sum()) for m in c["gi"]], "gi_tie": [int(m.sum()) for m in c["gi_tie"]],
                      "gi_other_ray": [int(m.sum()) for m in c["gi_other_ray"]], "gi_cull": [int(m.sum()) for m in c["gi_cull"]],
                      "shadow1": [int(m.sum()) for m in c["shadow1"]]}, "npx": npx}
        out.append(row)
        print(json.dumps(row))
    if a.out:
        with open(a.out, "w") as f:
            json.dump({"scene": a.scene, "size": [W, H], "spp": a.spp, "precision": a.prec, "tie_orders": [ta, tb], "frames": out,
                       "made_by": "tools/oracle_topology_baseline.py"}, f, indent=1)
    return out


if __name__ == "__main__":
    main()
