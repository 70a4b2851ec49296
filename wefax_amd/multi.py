"""Multi-GPU plumbing of the hot path: one process per GPU, captures sharded across
ranks, finished images gathered to rank 0 with ONE collective per step.

The decode itself never communicates (the exact path is global per capture, so the unit
that shards is the capture).  ``torch.distributed`` is used only as the transport:
backend ``nccl`` (= RCCL over xGMI) on GPUs, ``gloo`` in the CPU tests.  Nothing here
touches the arithmetic.
"""
from __future__ import annotations


def capture_shard(n_items: int, world: int, rank: int) -> range:
    """Contiguous block partition of ``n_items`` captures over ``world`` ranks; the first
    ``n_items % world`` ranks take one extra item."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank {rank} for world size {world}")
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


class ImageExchange:
    """Fixed-capacity send buffer per rank + receive buffers on the root.

    ``gather(nbytes, width)`` moves every rank's image (its first ``nbytes`` bytes) to the
    root in one ``dist.gather`` of the payload buffers; the (nbytes, width) pairs travel in
    an 16-byte header inside the same buffer, so there is exactly one collective per step.
    """

    HEADER = 16

    def __init__(self, dist, torch, capacity: int, device, root: int = 0):
        self.dist, self.torch = dist, torch
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.root = root
        self.capacity = int(capacity)
        total = self.HEADER + self.capacity
        self.send = torch.zeros(total, dtype=torch.uint8, device=device)
        self.recv = ([torch.zeros(total, dtype=torch.uint8, device=device) for _ in range(self.world)]
                     if self.rank == root else None)

    @property
    def payload_ptr(self) -> int:
        """Device (or host) address where the local image bytes go."""
        return self.send.data_ptr() + self.HEADER

    def payload_view(self):
        return self.send[self.HEADER:]

    def gather(self, nbytes: int, width: int):
        """Returns, on the root, a list of (uint8 tensor of nbytes_r, width_r) per rank."""
        torch = self.torch
        if nbytes > self.capacity:
            raise ValueError(f"image of {nbytes} bytes exceeds the exchange capacity {self.capacity}")
        hdr = torch.tensor([nbytes, width], dtype=torch.int64).view(torch.uint8).to(self.send.device)
        self.send[:self.HEADER] = hdr
        self.dist.gather(self.send, self.recv, dst=self.root)
        if self.rank != self.root:
            return None
        out = []
        for buf in self.recv:
            meta = buf[:self.HEADER].cpu().view(torch.int64)
            nb, w = int(meta[0]), int(meta[1])
            out.append((buf[self.HEADER:self.HEADER + nb], w))
        return out


class PipelinedExchange:
    """The image gather of step k overlapped with the decode of step k + 1.

    Nothing waits on the host: the decode's stream (``lib_stream``, the native context's
    hipStream_t wrapped as a torch ExternalStream) writes {header, image} into send slot
    k % slots with ``wfx_decode_export_async``; an event orders the collective, issued on a
    separate communication stream, behind that copy; another event keeps the decode from
    reusing a slot before the gather that read it has finished.  One RCCL gather per step.
    """

    HEADER = ImageExchange.HEADER

    def __init__(self, dist, torch, capacity: int, device, lib_stream: int, slots: int = 2, root: int = 0):
        self.dist, self.torch = dist, torch
        self.world, self.rank, self.root = dist.get_world_size(), dist.get_rank(), root
        self.capacity = int(capacity)
        total = self.HEADER + self.capacity
        self.send = [torch.zeros(total, dtype=torch.uint8, device=device) for _ in range(slots)]
        self.recv = [[torch.zeros(total, dtype=torch.uint8, device=device) for _ in range(self.world)]
                     if self.rank == root else None for _ in range(slots)]
        self.lib = torch.cuda.ExternalStream(lib_stream, device=device)
        self.comm = torch.cuda.Stream(device=device)
        self.free = [None] * slots          # event: the gather that used the slot is complete
        self.k = 0

    def prepare(self, ctx) -> None:
        """Call BEFORE the decode is enqueued: the decode then writes header and image straight into the step's send
        slot (``wfx_decode_bind_image``), and ``submit`` has nothing to copy."""
        s = self.k % len(self.send)
        if self.free[s] is not None:
            self.lib.wait_event(self.free[s])
            self.free[s] = None
        ctx.decode_bind_image(self.send[s].data_ptr(), self.HEADER + self.capacity)

    def submit(self, ctx, buffer_id: int) -> int:
        """Call right after the decode was enqueued on ``ctx``; returns the step's slot."""
        torch = self.torch
        s = self.k % len(self.send)
        self.k += 1
        if self.free[s] is not None:
            self.lib.wait_event(self.free[s])
        ctx.decode_export_async(buffer_id, self.send[s].data_ptr(), self.HEADER + self.capacity)   # no-op after prepare()
        ready = self.lib.record_event()
        self.comm.wait_event(ready)
        with torch.cuda.stream(self.comm):
            self.dist.gather(self.send[s], self.recv[s], dst=self.root)
            self.free[s] = self.comm.record_event()
        return s

    def result(self, slot: int):
        """Root: list of (uint8 tensor, width) per rank for the step that used ``slot`` (waits for it)."""
        self.comm.synchronize()
        if self.rank != self.root:
            return None
        out = []
        for buf in self.recv[slot]:
            meta = buf[:self.HEADER].cpu().view(self.torch.int64)
            nb, w = int(meta[0]), int(meta[1])
            out.append((buf[self.HEADER:self.HEADER + nb], w))
        return out
