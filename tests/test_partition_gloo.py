"""The N > 1 path on CPU: two processes, `gloo` backend.  Each rank renders its interleaved scanlines into a
compact slab (here with the CPU oracle standing in as the per-rank renderer -- this test is about the
partition, the slab layout and the gather, which are the same code the GPU ranks run) and the all-gathered,
de-interleaved image must equal the single-process render."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import _oracle as O
    import dray_amd as d
    from dray_amd import partition as P

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spheres = d.scene_three_spheres()
    cam = d.camera_three_spheres(width=40, spp=3, depth=8)   # H = 22: not a multiple of 3 -> ragged slabs at G = 3 too

    def cpu_rows(cam_, row_begin, row_step, color, raw, count, sample_offset):
        oc = O.camera_three_spheres(width=cam_.image_width, spp=cam_.samples_per_pixel, depth=cam_.max_depth)
        H, W = O.image_height(oc), oc.image_width
        fc = np.zeros((H, W), dtype=np.uint32)
        fr = np.zeros((H, W, 3))
        fn = np.zeros((H, W), dtype=np.int32)
        rows = list(range(row_begin, H, row_step))
        fr[rows] = raw.numpy()[:len(rows)]
        fn[rows] = count.numpy()[:len(rows)]
        _, _, _, st = O.render(oc, spheres, seed=cam_.seed, sample_offset=sample_offset, row_begin=row_begin,
                               row_step=row_step, color=fc, raw=fr, count=fn)
        color[:len(rows)] = torch.from_numpy(fc[rows].view(np.int32))
        raw[:len(rows)] = torch.from_numpy(fr[rows])
        count[:len(rows)] = torch.from_numpy(fn[rows])
        return st

    def alloc(rows, width):
        return (torch.zeros((rows, width), dtype=torch.int32), torch.zeros((rows, width, 3), dtype=torch.float64),
                torch.zeros((rows, width), dtype=torch.int32))

    cam.seed = 77
    r = P.DistributedRenderer(cam, cpu_rows, alloc, rank=rank, world_size=world)
    img1, st1 = r.step()
    img2, st2 = r.step()          # second pass continues the running sums with the next samples' streams
    segs = torch.tensor([float(st1.segments + st2.segments)], dtype=torch.float64)
    dist.all_reduce(segs)
    np.save(os.path.join(out_dir, "img_rank%d.npy" % rank), img2.numpy().view(np.uint32))
    raw_full, count_full = r.gather_sums()
    np.save(os.path.join(out_dir, "raw_rank%d.npy" % rank), raw_full.numpy())
    np.save(os.path.join(out_dir, "count_rank%d.npy" % rank), count_full.numpy())
    if rank == 0:
        np.save(os.path.join(out_dir, "segs.npy"), segs.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_two_rank_gather_equals_single_render(tmp_path, world):
    import torch.multiprocessing as mp
    import _oracle as O
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    cam = O.camera_three_spheres(width=40, spp=6, depth=8)
    col, raw, cnt, st = O.render(cam, O.scene_three_spheres(), seed=77)
    for rank in range(world):
        img = np.load(os.path.join(str(tmp_path), "img_rank%d.npy" % rank))
        assert img.shape == col.shape
        assert (img == col).all()           # every rank holds the whole image, equal to the one-process render
        # ... and, when asked, the running sums and counts of the whole image (colors.rs:38-39)
        assert (np.load(os.path.join(str(tmp_path), "raw_rank%d.npy" % rank)) == raw).all()
        assert (np.load(os.path.join(str(tmp_path), "count_rank%d.npy" % rank)) == cnt).all()
    assert int(np.load(os.path.join(str(tmp_path), "segs.npy"))[0]) == st.segments
