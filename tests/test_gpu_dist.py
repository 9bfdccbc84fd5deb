"""The N > 1 path end to end on one GPU: two ranks (gloo transport, both rendering on cuda:0) run the code bench.py runs
per frame — render the strip + halo (or, with trace_halo=False, trace the strip alone and receive the halo rows' sample
means between the two render phases), exchange the halo rows' temporal history point-to-point, all-gather the image —
and every gathered frame must equal the single-context frame bit for bit."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, W, H, frames, trace_halo, moving, q):
    import torch
    import torch.distributed as tdist
    import lprt_amd
    from lprt_amd import abi, scenes, dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    sc = scenes.cornell_box()
    halo = dist.HALO_ROWS + (12 if moving else 0)  # 68 rows + the vertical motion per frame
    wb, we = dist.window_rows(H, world, rank, halo)
    r = lprt_amd.Renderer(W, H, abi.F32, device=0, row_begin=wb, row_end=we)
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    frame = dist.ShardedFrame(W, H, rank, world, torch.device("cuda", 0), backend="gloo", halo=halo, trace_halo=trace_halo)
    frame.attach(r)
    r.load_scene(sc)
    out = []
    for k in range(frames):
        r.set_scene(scenes.moving_cornell(k, sc, climb=0.06) if moving else sc)
        moved = frame.render(r, abi.default_settings(spp=1))
        img = frame.all_gather()
        torch.cuda.synchronize()
        out.append(img.cpu().numpy().copy())
    q.put((rank, out, moved))
    tdist.barrier()
    r.close()  # while torch's stream and the HIP runtime are alive
    tdist.destroy_process_group()


# 3 ranks: the middle one has two neighbours; moving: objects move, the camera dollies and climbs, history taps cross strips
@pytest.mark.parametrize("world,trace_halo,moving", [(2, True, False), (2, False, False), (3, False, False), (3, False, True), (2, True, True)])
def test_ranks_render_exchange_gather_equal_one_context(world, trace_halo, moving):
    import torch.multiprocessing as mp
    import lprt_amd
    from lprt_amd import abi, scenes
    W, H, frames = 128, 288, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, W, H, frames, trace_halo, moving, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sc = scenes.cornell_box()
    full = lprt_amd.Renderer(W, H, abi.F32)
    full.load_scene(sc)
    want = []
    for k in range(frames):
        full.set_scene(scenes.moving_cornell(k, sc, climb=0.06) if moving else sc)
        full.render(abi.default_settings(spp=1))
        want.append(full.read("OUTPUT"))
    for rank, out, moved in got:
        assert moved > 0
        for k in range(frames):
            assert np.array_equal(out[k], want[k], equal_nan=True), f"rank {rank} frame {k}"
