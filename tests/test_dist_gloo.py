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


def _deferred_worker(rank, world, port, H, W, q):
    """The deferred all-gather (ShardedFrame(defer_gather=True)): frame n's gather runs under frame n + 1.  Six frames with
    distinct contents: all_gather() returns frame n - 1, flush() the last one -- every frame, in order, complete on every
    rank, bit-equal to what the immediate gather assembles."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)

    class Target:  # stands in for lprt_set_output_target: where the renderer would write this rank's rows
        def set_output_target(self, ptr, first_row, n_rows):
            self.ptr = ptr

    def strip_of(frame, k):
        b, e = frame.begin, frame.end
        rows = torch.arange(b, e, dtype=torch.float32).view(-1, 1, 1)
        cols = torch.arange(W, dtype=torch.float32).view(1, -1, 1)
        ch = torch.arange(4, dtype=torch.float32).view(1, 1, -1)
        return rows * 1000 + cols * 4 + ch + 100000 * k

    got = {}
    for defer in (False, True):
        frame = dist.ShardedFrame(W, H, rank, world, torch.device("cpu"), backend="gloo", defer_gather=defer)
        tgt = Target()
        frame.attach(tgt)
        imgs = []
        for k in range(6):
            assert tgt.ptr == frame.send.data_ptr()  # the renderer's target is the buffer the NEXT gather will read
            frame.send[:frame.end - frame.begin] = strip_of(frame, k)
            out = frame.all_gather()
            if not defer:
                imgs.append(out.numpy().copy())
            elif k == 0:
                assert out is None
            else:
                imgs.append(out.numpy().copy())
        if defer:
            imgs.append(frame.flush().numpy().copy())
            assert frame.flush() is None
        got[defer] = imgs
        assert frame.gathers == 6 and frame.bytes_gathered == 6 * frame.send.numel() * 4
    ok = len(got[True]) == 6 and all(np.array_equal(a, b) for a, b in zip(got[False], got[True]))
    want0 = (np.arange(H, dtype=np.float32).reshape(-1, 1, 1) * 1000 + np.arange(W, dtype=np.float32).reshape(1, -1, 1) * 4 +
             np.arange(4, dtype=np.float32).reshape(1, 1, -1))
    ok = ok and all(np.array_equal(img, want0 + 100000 * k) for k, img in enumerate(got[True]))
    q.put((rank, ok))
    tdist.barrier()
    tdist.destroy_process_group()


def test_deferred_all_gather_returns_the_same_frames_in_order():
    for world in (2, 4):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_deferred_worker, args=(r, world, port, 16, 5, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=120) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert all(ok for _, ok in res), res


def test_all_gather_equal_strips():
    _run(H=16, W=5)


def test_all_gather_ragged_strips():
    _run(H=15, W=3)


# ---- temporal-history halo exchange --------------------------------------------------------
class _FakeRenderer:
    """Stands in for lprt_export/import_history_rows on CPU: per group (filter history 1, TAA history 2, sample means 4)
    one 'plane' of ROW_BYTES bytes per image row; rows a rank owns carry a value derived from (group, row, frame), halo
    rows carry garbage until imported."""
    ROW_BYTES = 24
    GROUPS = (1, 2, 4)

    def __init__(self, frame):
        self.f = frame
        self.planes = {g: {} for g in self.GROUPS}
        self.resolved_ok = True

    @staticmethod
    def value(g, row, k):
        return (row * 7 + k * 3 + g) % 251

    def _fill(self, g, k):
        for r in range(self.f.win_begin, self.f.win_end):
            owned = self.f.begin <= r < self.f.end
            self.planes[g][r] = np.full(self.ROW_BYTES, self.value(g, r, k) if owned else 255, dtype=np.uint8)

    def _holds(self, g, k, reach):
        rows = range(max(self.f.win_begin, self.f.begin - reach), min(self.f.win_end, self.f.end + reach))
        return all(bool((self.planes[g][r] == self.value(g, r, k)).all()) for r in rows)

    def render(self, k):  # a whole frame: leaves history on the owned rows
        self._fill(1, k)
        self._fill(2, k)

    def history_ok(self, k):  # filter history on the whole window, TAA history on the rows next to the strip
        return self._holds(1, k, self.f.halo) and self._holds(2, k, min(self.f.halo, dist.TAA_ROWS + max(0, self.f.halo - dist.HALO_ROWS)))

    # the two-phase frame of trace_halo=False: the resolve phase must find this frame's sample means on every row of the
    # window and (from frame 1 on) the history the last frame left
    def set_output_target(self, *a):
        pass

    def set_sample_rows(self, first, n):
        self.sample_rows = (first, n)

    def render_phase(self, k, phase):
        if phase == 1:
            self._fill(4, k)
        else:
            self.resolved_ok = self.resolved_ok and self._holds(4, k, self.f.halo) and (k == 0 or self.history_ok(k - 1))
            self.render(k)

    def history_row_bytes(self, groups=3):
        return self.ROW_BYTES * sum(1 for g in self.GROUPS if groups & g)

    def synchronize(self):
        pass

    def export_history_rows(self, first, n, ptr, groups=3):
        import ctypes
        gs = [g for g in self.GROUPS if groups & g]
        buf = (ctypes.c_uint8 * (n * self.ROW_BYTES * len(gs))).from_address(ptr)
        np.frombuffer(buf, dtype=np.uint8)[:] = np.concatenate([self.planes[g][first + i] for g in gs for i in range(n)])

    def import_history_rows(self, first, n, ptr, groups=3):
        import ctypes
        gs = [g for g in self.GROUPS if groups & g]
        buf = (ctypes.c_uint8 * (n * self.ROW_BYTES * len(gs))).from_address(ptr)
        a = np.frombuffer(buf, dtype=np.uint8).reshape(len(gs), n, self.ROW_BYTES)
        for j, g in enumerate(gs):
            for i in range(n):
                self.planes[g][first + i] = a[j, i].copy()


def _hist_worker(rank, world, port, H, halo, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    frame = dist.ShardedFrame(4, H, rank, world, torch.device("cpu"), backend="gloo", halo=halo)
    r = _FakeRenderer(frame)
    ok = True
    for k in range(3):
        r.render(k)
        frame.exchange_history(r)
        ok = ok and r.history_ok(k)  # every halo row now holds its OWNER's value
    q.put((rank, ok))
    tdist.barrier()
    tdist.destroy_process_group()


def _lean_worker(rank, world, port, H, halo, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    frame = dist.ShardedFrame(4, H, rank, world, torch.device("cpu"), backend="gloo", halo=halo, trace_halo=False)
    r = _FakeRenderer(frame)
    frame.attach(r)
    ok = r.sample_rows == (frame.begin, frame.end - frame.begin)
    for k in range(4):
        # sample phase (last frame's history in flight) -> sample means -> resolve phase -> history packed for the next frame
        moved = frame.render(r, k)
        frame.all_gather()
        ok = ok and r.resolved_ok and moved > 0
    q.put((rank, ok))
    tdist.barrier()
    tdist.destroy_process_group()


def _run_hist(H, world, halo, worker=_hist_worker):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, H, halo, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in got), got


def test_history_exchange_two_ranks():
    _run_hist(H=40, world=2, halo=6)


def test_history_exchange_halo_spans_several_strips():
    _run_hist(H=30, world=4, halo=11)  # strips of 7-8 rows: a halo reaches two neighbours on each side


def test_halo_samples_arrive_between_the_two_render_phases():
    _run_hist(H=40, world=2, halo=6, worker=_lean_worker)
    _run_hist(H=30, world=4, halo=11, worker=_lean_worker)


def test_history_transfer_plan_covers_every_halo_row_once():
    for H, world, halo in ((1080, 8, dist.HALO_ROWS), (1080, 2, dist.HALO_ROWS), (30, 4, 11), (15, 4, 3)):
        plan = dist.history_transfers(H, world, halo)
        for dst in range(world):
            wb, we = dist.window_rows(H, world, dst, halo)
            ob, oe = dist.strip_rows(H, world, dst)
            got = sorted(r for s, d, b, e in plan if d == dst for r in range(b, e))
            assert got == [r for r in range(wb, we) if not ob <= r < oe]
        for s, d, b, e in plan:
            sb, se = dist.strip_rows(H, world, s)
            assert sb <= b < e <= se
