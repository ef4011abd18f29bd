"""Scanline-stripe sharding of one frame over the GPUs of a node.

The reference maps a pure function over pixels (generatePixels, src/Image.hs:34-36),
so rows are independent.  Cost per row is uneven (the dragon sits in the middle
rows), therefore ranks take small INTERLEAVED stripes: stripe s of `stripe_rows`
rows belongs to rank s % world.  The scene is replicated; the only exchange is
one gather of the finished rows to rank 0 (RCCL over xGMI when the backend is
"nccl").  The same selection rule is implemented by the kernel launcher
(rayhs_render_params.stripe_*, include/rayhs.h).
"""
import torch
import torch.distributed as dist


def stripe_rows_of(rank, world, height, stripe_rows, row0=0, row1=None):
    """Image rows rendered by `rank`, in increasing order (what the kernel writes compactly)."""
    row1 = height if row1 is None else row1
    rows = []
    total = row1 - row0
    n_stripes = (total + stripe_rows - 1) // stripe_rows
    for s in range(rank, n_stripes, world):
        a = row0 + s * stripe_rows
        rows.extend(range(a, min(a + stripe_rows, row1)))
    return rows


def max_rows(world, height, stripe_rows):
    return max(len(stripe_rows_of(r, world, height, stripe_rows)) for r in range(world))


class FrameGather:
    """Pre-planned gather of per-rank stripes into the full frame on rank 0.

    Buffers and index tensors are created once; `gather()` is then one
    collective plus one indexed copy, safe to call every frame.
    """

    def __init__(self, rank, world, height, width, stripe_rows, device, dtype=torch.float32, channels=3):
        self.rank, self.world = rank, world
        self.height, self.width, self.channels = height, width, channels
        self.rows = stripe_rows_of(rank, world, height, stripe_rows)
        self.pad_rows = max_rows(world, height, stripe_rows)
        # local buffer padded to the largest share so that every rank sends the same size
        self.local = torch.zeros((self.pad_rows, width, channels), dtype=dtype, device=device)
        if rank == 0:
            # one allocation; the gather lands in its per-rank slices
            self.stacked = torch.empty((world, self.pad_rows, width, channels), dtype=dtype, device=device) if world > 1 else None
            self.parts = [self.stacked[r] for r in range(world)] if world > 1 else None
            # with one rank there is nothing to gather or to put in order: the frame IS the local buffer (no copy per frame)
            self.frame = self.local[:height] if world == 1 else torch.empty((height, width, channels), dtype=dtype, device=device)
            src_rank, src_row, dst_row = [], [], []
            for r in range(world):
                rr = stripe_rows_of(r, world, height, stripe_rows)
                src_rank += [r] * len(rr)
                src_row += list(range(len(rr)))
                dst_row += rr
            order = sorted(range(len(dst_row)), key=lambda i: dst_row[i])
            assert [dst_row[i] for i in order] == list(range(height))
            # flat row index into the stacked [world, pad_rows] gather result, in image order
            self.take = torch.tensor([src_rank[i] * self.pad_rows + src_row[i] for i in order], dtype=torch.long, device=device)
        else:
            self.parts, self.frame, self.take = None, None, None

    def gather(self):
        """Returns the assembled frame on rank 0 (None elsewhere)."""
        self.gather_start()
        return self.gather_finish()

    def gather_start(self):
        """Enqueue the collective without blocking the calling stream; pair with gather_finish().
        Between the two the caller may enqueue work that does not touch `local` (e.g. render the next
        frame into another FrameGather), which then overlaps with the transfer."""
        self._work = None
        if self.world > 1:
            self._work = dist.gather(self.local, self.parts if self.rank == 0 else None, dst=0, async_op=True)

    def gather_finish(self):
        if self.world == 1:
            return self.frame   # a view of `local`
        self._work.wait()  # makes the current stream wait for the collective; does not block the host
        self._work = None
        if self.rank != 0:
            return None
        flat = self.stacked.view(self.world * self.pad_rows, self.width, self.channels)
        torch.index_select(flat, 0, self.take, out=self.frame)
        return self.frame
