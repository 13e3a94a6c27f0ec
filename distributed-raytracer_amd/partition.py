"""Single-node scanline partitioner and image gather: one process per GPU.

This takes the place of the reference's ``src/distributed`` tree for the render
path.  There, the orchestrator deals one primary ray at a time to ray servers
(``orchestrator_server.rs:59-68``) and pixel results come back one message at a
time (``ray_server.rs:94-102`` -> ``orchestrator_server.rs:253-256`` ->
``client.rs:104-121``).  Here every (pixel, sample) is independent, the whole
scene is replicated on each GPU, rank r renders the interleaved scanlines
r, r+G, r+2G, ... and the packed 8-bit slabs are gathered once per render with a
single all-gather (RCCL over xGMI when the process group's backend is "nccl").

Interleaving balances cost: sky rows take about one segment per sample, ground
rows many.  The Philox streams are keyed on the global pixel index, so the
gathered image is the same for every G.
"""
import numpy as np


def row_partition(image_height, world_size, rank):
    """(row_begin, row_step, n_rows) of ``rank``: rows rank, rank+world_size, ..."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank %d / world size %d" % (rank, world_size))
    n_rows = (image_height - rank + world_size - 1) // world_size if rank < image_height else 0
    return rank, world_size, n_rows


def slab_rows(image_height, world_size):
    """Rows of the padded per-rank slab: ceil(H / G).  Ranks with fewer rows leave the tail unused."""
    return (image_height + world_size - 1) // world_size


def deinterleave(gathered, image_height, world_size):
    """[G, slab_rows, W] slabs -> [H, W] image: image row j = slab j % G, slab row j // G.

    Works on numpy arrays and torch tensors alike."""
    G = world_size
    S = gathered.shape[1]
    W = gathered.shape[2]
    # [G, S, W] -> [S, G, W] -> [S*G, W]; row s*G + r is image row s*G + r.
    if hasattr(gathered, "permute"):
        full = gathered.permute(1, 0, 2).reshape(S * G, W)
    else:
        full = np.transpose(gathered, (1, 0, 2)).reshape(S * G, W)
    return full[:image_height]


def gather_image(slab, image_height, group=None, always_collective=False):
    """All-gather the per-rank colour slabs [slab_rows, W] and return the full [H, W] image on every rank.

    A single rank returns its slab without a collective unless ``always_collective`` is set (tests use it to run
    the RCCL call on one GPU)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return slab[:image_height]
    if dist.get_world_size(group) == 1 and not always_collective:
        return slab[:image_height]
    G = dist.get_world_size(group)
    S, W = slab.shape
    if slab.is_cuda and dist.get_backend(group) == "gloo":
        # gloo gathers host tensors only: stage through host memory (rehearsals of the multi-rank path on one GPU)
        host = torch.empty((G * S, W), dtype=slab.dtype)
        dist.all_gather_into_tensor(host, slab.contiguous().cpu(), group=group)
        return deinterleave(host.to(slab.device).view(G, S, W), image_height, G)
    gathered = torch.empty((G * S, W), dtype=slab.dtype, device=slab.device)  # rank r's slab = rows r*S .. r*S+S-1
    dist.all_gather_into_tensor(gathered, slab.contiguous(), group=group)
    return deinterleave(gathered.view(G, S, W), image_height, G)


def gather_slabs(slab, image_height, group=None):
    """All-gather per-rank slabs [slab_rows, W, ...] of any trailing shape (the running sums [.., 3], the counts) and
    return the de-interleaved [H, W, ...] array on every rank."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return slab[:image_height]
    G = dist.get_world_size(group)
    S, W = slab.shape[0], slab.shape[1]
    tail = tuple(slab.shape[2:])
    gathered = torch.empty((G * S, W) + tail, dtype=slab.dtype, device=slab.device)
    dist.all_gather_into_tensor(gathered, slab.contiguous(), group=group)
    full = gathered.view((G, S, W) + tail).transpose(0, 1).reshape((S * G, W) + tail)  # row s*G + r = image row s*G + r
    return full[:image_height]


class DistributedRenderer:
    """Renders this rank's scanlines into persistent slab buffers and gathers the image.

    ``render_rows`` is the per-rank render: ``render_rows(cam, row_begin, row_step, color, raw, count) -> stats``
    writing compact slabs (row j at slab row (j - row_begin) // row_step).  The GPU path passes
    ``gpu_render_rows(scene)``; tests substitute a CPU checker to exercise the partition and the gather.
    """

    def __init__(self, cam, render_rows, alloc, rank=0, world_size=1, group=None):
        self.cam = cam
        self.rank, self.world_size, self.group = rank, world_size, group
        cam.initialize()
        self.H, self.W = cam.image_height, cam.image_width
        self.row_begin, self.row_step, self.n_rows = row_partition(self.H, world_size, rank)
        self.S = slab_rows(self.H, world_size)
        self.color, self.raw, self.count = alloc(self.S, self.W)
        self.render_rows = render_rows
        self.samples_done = 0
        self._gather_events = []

    def step(self, gather=True, time_gather=False, stats_later=False):
        """One pass of the camera's samples_per_pixel over this rank's rows, then the gather.  With ``time_gather`` the
        gather is bracketed by two events on the current stream (read later through gather_ms, without a wait here).
        With ``stats_later`` the render is only enqueued and what comes back in the statistics' place has a
        ``stats()`` that waits for it (a backend without that simply returns the statistics)."""
        if stats_later:
            stats = self.render_rows(self.cam, self.row_begin, self.row_step, self.color, self.raw, self.count,
                                     self.samples_done, want_stats="later")
        else:
            stats = self.render_rows(self.cam, self.row_begin, self.row_step, self.color, self.raw, self.count,
                                     self.samples_done)
        self.samples_done += self.cam.samples_per_pixel
        timed = gather and time_gather and getattr(self.color, "is_cuda", False)
        if timed:
            import torch
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        image = gather_image(self.color, self.H, self.group) if gather else None
        if timed:
            e1.record()
            self._gather_events.append((e0, e1))
        return image, stats

    def gather_ms(self):
        """Total device time of the gathers timed so far (all-gather + de-interleave), in ms; waits for them."""
        total = 0.0
        for e0, e1 in self._gather_events:
            e1.synchronize()
            total += e0.elapsed_time(e1)
        self._gather_events = []
        return total

    def gather_sums(self):
        """The other two buffers of the render call (camera.rs:212-213) for the whole image on every rank: the running
        colour sums [H, W, 3] and sample counts [H, W] (colors.rs:38-39), gathered from the ranks' slabs.  Two more
        all-gathers (6 x the colour's bytes + 1 x): asked for when needed, not every pass."""
        if self.raw is None or self.count is None:
            raise ValueError("this renderer was made without raw / count buffers")
        return gather_slabs(self.raw, self.H, self.group), gather_slabs(self.count, self.H, self.group)


def gpu_render_rows(scene, **opt_overrides):
    """Per-rank render callback for DistributedRenderer backed by the HIP library."""
    from . import _ffi
    from .device import render_device

    def fn(cam, row_begin, row_step, color, raw, count, sample_offset, want_stats=True):
        flags = cam.flags | _ffi.FLAG_COMPACT_ROWS
        return render_device(cam, scene, color, raw, count, row_begin=row_begin, row_step=row_step, flags=flags,
                             sample_offset=sample_offset, want_stats=want_stats, **opt_overrides)

    return fn


def gpu_alloc(device, with_raw=True):
    def fn(rows, width):
        import torch
        dev = torch.device("cuda", device)
        color = torch.zeros((rows, width), dtype=torch.int32, device=dev)
        raw = torch.zeros((rows, width, 3), dtype=torch.float64, device=dev) if with_raw else None
        count = torch.zeros((rows, width), dtype=torch.int32, device=dev) if with_raw else None
        return color, raw, count

    return fn
