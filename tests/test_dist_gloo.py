"""world_size-2 rehearsal of the N>1 path on CPU (gloo): strip ownership, in-place all-gather of the final image,
max-over-ranks timing reduction — the same torch.distributed calls bench.py makes with backend 'nccl'."""
import os
import socket

import numpy as np
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

from lprt_amd import dist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, H, W, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    frame = dist.ShardedFrame(W, H, rank, world, torch.device("cpu"), backend="gloo")
    # each rank "renders" its strip: value encodes (row, col, channel) so misplaced rows are detected
    b, e = frame.begin, frame.end
    rows = torch.arange(b, e, dtype=torch.float32).view(-1, 1, 1)
    cols = torch.arange(W, dtype=torch.float32).view(1, -1, 1)
    ch = torch.arange(4, dtype=torch.float32).view(1, 1, -1)
    strip = rows * 1000 + cols * 4 + ch
    frame.send[:e - b] = strip
    img = frame.all_gather()
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
    q.put((rank, img.numpy().copy(), float(t.item())))
    tdist.barrier()
    tdist.destroy_process_group()


def _run(H, W, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, W, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = (np.arange(H, dtype=np.float32).reshape(-1, 1, 1) * 1000 + np.arange(W, dtype=np.float32).reshape(1, -1, 1) * 4 +
            np.arange(4, dtype=np.float32).reshape(1, 1, -1))
    for rank, img, tmax in got:
        assert np.array_equal(img, want), f"rank {rank} assembled a wrong frame"
        assert abs(tmax - 0.1 * world) < 1e-12


def test_all_gather_equal_strips():
    _run(H=16, W=5)


def test_all_gather_ragged_strips():
    _run(H=15, W=3)
