"""Multi-GPU row tiling of one frame (no reference counterpart: the reference is single-GPU,
src/main.cu:86).

Pixels are independent and every generator is keyed on the GLOBAL pixel id
(src/pathtrace.cu:206,265), so rank g renders the contiguous row block
row_range(H, G, g) into its own HBM and the frame is assembled by ONE gather to rank 0 at
frame end.  Because the buffer is [row][col][14], a row block is one contiguous span: the
root posts G-1 receives straight into the final frame at the tile offsets and renders its
own tile in place (zero-copy, ragged tiles allowed); every other rank posts one send.  The
G-1 transfers are issued as one group (torch batch_isend_irecv = ncclGroupStart/End with
backend "nccl", which is RCCL on ROCm), i.e. exactly what ncclGather does internally, and
they arrive over G-1 distinct xGMI links.  With "gloo" the same code runs on CPU tensors.
"""
import torch
import torch.distributed as dist

CHANNELS = 14


def row_range(height, world_size, rank):
    """Contiguous, balanced row blocks: the first (height % world) ranks get one extra row."""
    base, extra = divmod(height, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


class FrameGather:
    """Buffers + the frame-end gather of row tiles to rank `dst`."""

    def __init__(self, width, height, device, dst=0, group=None):
        self.width, self.height, self.dst, self.group = width, height, dst, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rows = row_range(height, self.world, self.rank)
        self.is_root = self.rank == dst
        # gloo cannot send/recv device tensors: stage through the host (functional testing of the
        # multi-rank path on a box without RCCL peers; the production backend is "nccl" = RCCL)
        self.stage_host = (dist.is_initialized() and dist.get_backend(group) == "gloo"
                           and torch.device(device).type == "cuda")
        rf = width * CHANNELS
        if self.is_root:
            self.frame = torch.empty(height * rf, dtype=torch.float32, device=device)
            self.tile = self.frame[self.rows[0] * rf : self.rows[1] * rf]  # rendered in place
        else:
            self.frame = None
            self.tile = torch.empty((self.rows[1] - self.rows[0]) * rf, dtype=torch.float32, device=device)

    def gather(self):
        """Collective over the group.  Returns the list of outstanding requests; after
        `wait_all()` (and a stream sync for GPU tensors) rank dst's `frame` is complete."""
        if self.world == 1:
            return []
        rf = self.width * CHANNELS
        if self.stage_host:
            if self.is_root:
                for r in range(self.world):
                    b, e = row_range(self.height, self.world, r)
                    if r != self.rank and e > b:
                        buf = torch.empty((e - b) * rf, dtype=torch.float32)
                        dist.recv(buf, src=r, group=self.group)
                        self.frame[b * rf : e * rf].copy_(buf)
            elif self.tile.numel():
                dist.send(self.tile.cpu(), dst=self.dst, group=self.group)
            return []
        ops = []
        if self.is_root:
            for r in range(self.world):
                if r == self.rank:
                    continue
                b, e = row_range(self.height, self.world, r)
                if e > b:
                    ops.append(dist.P2POp(dist.irecv, self.frame[b * rf : e * rf], r, self.group))
        elif self.tile.numel():
            ops.append(dist.P2POp(dist.isend, self.tile, self.dst, self.group))
        return dist.batch_isend_irecv(ops) if ops else []

    @staticmethod
    def wait_all(reqs):
        for r in reqs:
            r.wait()
