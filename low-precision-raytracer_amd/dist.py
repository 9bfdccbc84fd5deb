"""Frame sharding across the GPUs of one node.

Every pixel is independent up to the denoiser, and the RNG is keyed by the GLOBAL pixel index, so a frame is cut
into row strips: rank r renders rows [r*H/N, (r+1)*H/N) plus a halo of HALO_ROWS on each side (the reach of the
SVGF stencils), denoises locally, and writes only its own rows into its slot of the gathered image; one
all-gather (RCCL over xGMI through torch.distributed, backend "nccl") completes the frame on every rank.  Scene
data (geometry, BVH, materials) is replicated.

Across frames the temporal history (SVGF colour/moment history, TAA history, frame counters) of a rank's HALO rows
is not right -- those rows lack part of their own filter support -- so after each frame every rank sends the owned
rows its neighbours hold as halo and receives its halo rows from their owners (`exchange_history`, point-to-point
over xGMI: 2 x 68 rows x 49 B/pixel + 2 x 2 rows x 16 B/pixel of TAA history, about 13 MB per rank at 1080p fp32).  With that the sharded sequence equals the
single-GPU sequence as long as reprojection moves a pixel by no more than `halo - 68` rows per frame (a static
camera needs none; pass a larger `halo` for fast vertical motion).

`trace_halo=False` removes the redundant rays: a rank runs the per-sample stages (RNG, shade, shadow/GI rays) on the rows
it OWNS only and, between the two render phases, receives the sample means of its halo rows from their owners together
(32 B/pixel, fp32); the history of the previous frame travels under the sample phase.  At 4K / 8 GPUs a rank traces 270
rows instead of 406.
"""
import numpy as np

# a-trous reach 2*(1+2+4+8+16) = 62, + 1 (3x3 variance / gradient) + 4 (9x9 clamp) = 67 rows of current-frame data,
# + 1 for the bilinear history taps of the temporal accumulation at the outermost of those rows (frames >= 1)
HALO_ROWS = 68


def strip_rows(height, world_size, rank):
    """Rows [begin, end) owned by `rank`: contiguous, near-equal strips (row-major => all-gather lands in place)."""
    base, rem = divmod(height, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def window_rows(height, world_size, rank, halo=HALO_ROWS):
    """Rows [begin, end) a rank RENDERS: its strip plus the halo, clipped to the image."""
    b, e = strip_rows(height, world_size, rank)
    return max(0, b - halo), min(height, e + halo)


# The TAA history is only read within +-1 row of a pixel (jittered bilinear taps), so a rank needs it on 2 halo rows (plus
# whatever vertical motion the halo was widened for), not on all 68
TAA_ROWS = 2


def history_transfers(height, world_size, halo=HALO_ROWS):
    """[(src_rank, dst_rank, row_begin, row_end)]: rows src OWNS that dst holds within `halo` rows of its strip, in a
    fixed global order."""
    out = []
    for dst in range(world_size):
        wb, we = window_rows(height, world_size, dst, halo)
        for src in range(world_size):
            if src == dst:
                continue
            ob, oe = strip_rows(height, world_size, src)
            b, e = max(wb, ob), min(we, oe)
            if b < e:
                out.append((src, dst, b, e))
    return out


def equal_strips(height, world_size):
    """all_gather_into_tensor needs equal counts per rank."""
    return height % world_size == 0


class ShardedFrame:
    """Owns one rank's strip buffer and the gathered RGBA32F image, and performs the per-frame all-gather.

    The renderer writes this rank's rows into `send` (lprt_set_output_target); `all_gather()` completes `image` on
    every rank.  Equal strips (H % N == 0: 1080p and 4K on 1/2/4/8 GPUs) gather straight into the image; ragged
    strips go through padded per-rank slots.  backend "nccl" (= RCCL) gathers device tensors over xGMI; "gloo" (CPU
    rehearsal) stages through host memory.
    """

    def __init__(self, width, height, rank, world_size, device, backend="nccl", halo=HALO_ROWS, trace_halo=True, defer_gather=False):
        import torch
        self.trace_halo = trace_halo or world_size == 1
        self.torch = torch
        self.W, self.H, self.rank, self.world = width, height, rank, world_size
        self.device, self.backend, self.halo = device, backend, halo
        self.begin, self.end = strip_rows(height, world_size, rank)
        self.win_begin, self.win_end = window_rows(height, world_size, rank, halo)
        self._hist = {}
        self._hist_packed = False
        self.image = torch.zeros((height, width, 4), dtype=torch.float32, device=device)
        self.equal = equal_strips(height, world_size)
        self.max_rows = -(-height // world_size)
        rows = (self.end - self.begin) if self.equal else self.max_rows
        self.send = torch.zeros((rows, width, 4), dtype=torch.float32, device=device)
        if not self.equal:
            self.recv = torch.zeros((world_size, self.max_rows, width, 4), dtype=torch.float32, device=device)
        # Deferred all-gather (opt-in; equal strips only): the gather of frame n is started asynchronously and runs under
        # frame n + 1 -- the renderer writes frame n + 1's rows into the OTHER of two strip buffers, the gathered image
        # alternates likewise -- and is waited for when frame n + 1's gather is posted.  all_gather() then returns the image
        # of the PREVIOUS frame (None after the first); flush() waits for and returns the newest.  Frames come out in
        # order, one call later (tests/test_dist_gloo.py).
        self.defer = bool(defer_gather) and world_size > 1 and self.equal
        self.bytes_gathered = 0  # bytes this rank contributed to all-gathers so far
        self.gathers = 0
        if self.defer:
            self._sends = [self.send, torch.zeros_like(self.send)]
            self._images = [self.image, torch.zeros_like(self.image)]
            self._slot, self._pending, self._renderer = 0, None, None

    def target(self):
        """(device pointer, first_row, n_rows) for lprt_set_output_target."""
        return self.send.data_ptr(), self.begin, self.end - self.begin

    def all_gather(self):
        import torch.distributed as dist
        torch = self.torch
        if self.world == 1:
            return self.send  # one GPU: the strip is the frame
        self.bytes_gathered += self.send.numel() * self.send.element_size()
        self.gathers += 1
        if self.defer:
            return self._all_gather_deferred()
        staged = self.backend != "nccl" and self.send.is_cuda
        if self.equal:
            if staged:
                out = torch.empty(self.image.shape, dtype=torch.float32)
                dist.all_gather_into_tensor(out.view(-1), self.send.cpu().view(-1))
                self.image.copy_(out)
            else:
                dist.all_gather_into_tensor(self.image.view(-1), self.send.view(-1))
        else:
            if staged:
                out = torch.empty(self.recv.shape, dtype=torch.float32)
                dist.all_gather_into_tensor(out.view(-1), self.send.cpu().view(-1))
                self.recv.copy_(out)
            else:
                dist.all_gather_into_tensor(self.recv.view(-1), self.send.view(-1))
            for r in range(self.world):
                b, e = strip_rows(self.H, self.world, r)
                self.image[b:e].copy_(self.recv[r, :e - b])
        return self.image

    def _all_gather_deferred(self):
        import torch.distributed as dist
        prev = self.flush()  # at most one gather in flight: frame n - 1's has had all of frame n to finish
        send, image = self._sends[self._slot], self._images[self._slot]
        if self.backend != "nccl" and send.is_cuda:  # rehearsal through host memory
            host_in, host_out = send.cpu(), self.torch.empty(image.shape, dtype=self.torch.float32)
            work = dist.all_gather_into_tensor(host_out.view(-1), host_in.view(-1), async_op=True)
            self._pending = (work, image, host_out)
        else:
            work = dist.all_gather_into_tensor(image.view(-1), send.view(-1), async_op=True)
            self._pending = (work, image, None)
        # the next frame's rows go to the other strip buffer while this one is being read by the transport
        self._slot ^= 1
        self.send, self.image = self._sends[self._slot], self._images[self._slot]
        if self._renderer is not None:
            ptr, first_row, n_rows = self.target()
            self._renderer.set_output_target(ptr, first_row, n_rows)
        return prev

    def flush(self):
        """Deferred mode: wait for the gather in flight and return its image (None if there is none)."""
        if not getattr(self, "defer", False) or self._pending is None:
            return None
        work, image, host_out = self._pending
        self._pending = None
        work.wait()
        if host_out is not None:
            image.copy_(host_out)
        return image

    def attach(self, renderer):
        """Output target + the rows the per-sample stages run on (the strip alone unless the halo is traced locally)."""
        ptr, first_row, n_rows = self.target()
        if getattr(self, "defer", False):
            self._renderer = renderer
        renderer.set_output_target(ptr, first_row, n_rows)
        if not self.trace_halo:
            renderer.set_sample_rows(self.begin, self.end - self.begin)

    def render(self, renderer, settings, stream_ordered=False):
        """One sharded frame up to (not including) the all-gather; returns the bytes this rank exchanged.

        trace_halo=False: the history frame n-1 left was packed when that frame ended; its transfer is posted here, after
        the caller's all-gather of frame n-1, and runs under the sample phase of frame n.  Only the sample means of frame
        n (32 of the 97 B per pixel and row, fp32) travel between the two phases with the stream waiting for them."""
        from . import abi
        if self.world == 1:
            renderer.render(settings)
            return 0
        if self.trace_halo:
            renderer.render(settings)
            return self.exchange_history(renderer, stream_ordered)
        ordered = stream_ordered and self.backend == "nccl"
        hist, samp = self._plan(renderer, "history"), self._plan(renderer, "samples")
        in_flight = self._post(renderer, hist, ordered) if self._hist_packed else None
        renderer.render_phase(settings, abi.PHASE_SAMPLE)
        self._export(renderer, samp)
        moved = self._finish(renderer, samp, self._post(renderer, samp, ordered), ordered)
        if in_flight is not None:
            moved += self._finish(renderer, hist, in_flight, ordered)
        renderer.render_phase(settings, abi.PHASE_RESOLVE)
        self._export(renderer, hist)  # for the next frame
        self._hist_packed = True
        return moved

    def reset(self):
        """Call with renderer.reset_history(): the history packed after the last frame is dropped with it."""
        self._hist_packed = False

    def exchange_history(self, renderer, stream_ordered=False, samples=False):
        """Owned rows -> the neighbours that hold them as halo, halo rows <- their owners.  The SVGF history (and with
        samples=True this frame's sample means) travel for the whole halo, the TAA history for TAA_ROWS (+ motion margin)
        rows.  stream_ordered=True: the renderer enqueues on torch's current stream and the transport is NCCL, which
        orders itself after that stream — no host synchronisation is needed."""
        if self.world == 1:
            return 0
        plan = self._plan(renderer, "history+samples" if samples else "history")
        ordered = stream_ordered and self.backend == "nccl"
        self._export(renderer, plan)
        return self._finish(renderer, plan, self._post(renderer, plan, ordered), ordered)

    # ---- the exchange in steps: plan (buffers, once) -> export (pack kernels) -> post (isend/irecv) -> finish (wait, unpack)
    def _plan(self, renderer, what):
        from . import abi
        torch = self.torch
        if what not in self._hist:
            staged = self.backend != "nccl"
            plan = self._hist[what] = []
            taa_halo = min(self.halo, TAA_ROWS + max(0, self.halo - HALO_ROWS))
            group_halos = {"history": ((abi.HIST_FILTER, self.halo), (abi.HIST_TAA, taa_halo)),
                           "history+samples": ((abi.HIST_FILTER | abi.HIST_SAMPLES, self.halo), (abi.HIST_TAA, taa_halo)),
                           "samples": ((abi.HIST_SAMPLES, self.halo),)}[what]
            for groups, halo in group_halos:
                row_bytes = renderer.history_row_bytes(groups)
                for src, dst, b, e in history_transfers(self.H, self.world, halo):
                    if self.rank not in (src, dst) or not (self.win_begin <= b and e <= self.win_end):
                        continue
                    dev = torch.empty((e - b) * row_bytes, dtype=torch.uint8, device=self.device)
                    host = torch.empty((e - b) * row_bytes, dtype=torch.uint8) if staged and dev.is_cuda else None
                    plan.append((groups, src, dst, b, e, dev, host))
        return self._hist[what]

    def _export(self, renderer, plan):
        for groups, src, dst, b, e, dev, host in plan:
            if src == self.rank:
                renderer.export_history_rows(b, e - b, dev.data_ptr(), groups)

    def _post(self, renderer, plan, ordered):
        import torch.distributed as dist
        if not ordered:
            renderer.synchronize()  # the exports run on the renderer's stream, the transport on torch's
        ops = []
        for groups, src, dst, b, e, dev, host in plan:
            if src == self.rank:
                if host is not None:
                    host.copy_(dev)
                ops.append(dist.P2POp(dist.isend, host if host is not None else dev, dst))
            else:
                ops.append(dist.P2POp(dist.irecv, host if host is not None else dev, src))
        return dist.batch_isend_irecv(ops) if ops else []

    def _finish(self, renderer, plan, reqs, ordered):
        torch = self.torch
        for req in reqs:
            req.wait()
        if not ordered and torch.cuda.is_available() and str(self.device).startswith("cuda"):
            torch.cuda.synchronize()
        moved = 0
        for groups, src, dst, b, e, dev, host in plan:
            moved += dev.numel()
            if dst == self.rank:
                if host is not None:
                    dev.copy_(host)
                    if dev.is_cuda:
                        torch.cuda.synchronize()
                renderer.import_history_rows(b, e - b, dev.data_ptr(), groups)
        return moved


def assemble_strips(strips, height, world_size):
    """CPU helper (tests): stitch per-rank strips (numpy [rows, W, C]) into the full frame."""
    out = None
    for r, s in enumerate(strips):
        b, e = strip_rows(height, world_size, r)
        assert s.shape[0] == e - b
        if out is None:
            out = np.zeros((height,) + s.shape[1:], dtype=s.dtype)
        out[b:e] = s
    return out
