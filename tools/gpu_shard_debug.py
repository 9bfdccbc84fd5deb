import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lprt_amd
from lprt_amd import abi, scenes, dist
sc = scenes.cornell_box()
W, H, world = 160, 300, 3
halo = int(sys.argv[1]) if len(sys.argv) > 1 else dist.HALO_ROWS
full = lprt_amd.Renderer(W, H, abi.F32); full.load_scene(sc)
wins = [dist.window_rows(H, world, r, halo) for r in range(world)]
strips = [dist.strip_rows(H, world, r) for r in range(world)]
rs = []
for wb, we in wins:
    r = lprt_amd.Renderer(W, H, abi.F32, row_begin=wb, row_end=we); r.load_scene(sc); rs.append(r)
row_bytes = rs[0].history_row_bytes()
plan = dist.history_transfers(H, world, halo)
print("wins", wins, "strips", strips, "plan", plan)
for k in range(3):
    fs = abi.default_settings(spp=1)
    full.set_scene(sc); full.render(fs)
    for r in rs:
        r.set_scene(sc); r.render(fs)
    for name in ("G_OBJID", "RAW_COLOR", "RAW_GI_COLORED", "MUL_GI_COLORED", "MUL_GI_WHITE", "COLOR_INPROGRESS", "OUTPUT"):
        want = full.read(name)
        for rank, (r, (wb, we), (b, e)) in enumerate(zip(rs, wins, strips)):
            got = r.read(name)
            bad_rows = np.where((got != want[wb:we]).reshape(we - wb, -1).any(axis=1))[0] + wb
            own_bad = [x for x in bad_rows if b <= x < e]
            print(f"frame {k} {name:16s} rank {rank}: window-bad rows {bad_rows.min() if len(bad_rows) else '-'}..{bad_rows.max() if len(bad_rows) else '-'} ({len(bad_rows)}), owned-bad {own_bad[:5]}..{own_bad[-3:]} ({len(own_bad)})")
    for src, dst, b, e in plan:
        buf = torch.empty((e - b) * row_bytes, dtype=torch.uint8, device="cuda")
        rs[src].export_history_rows(b, e - b, buf.data_ptr()); rs[src].synchronize()
        rs[dst].import_history_rows(b, e - b, buf.data_ptr()); rs[dst].synchronize()
