"""The multi-GPU path on CPU: world_size 2, 3 (ragged tiles) and 8 (the driver's largest launch; ragged) over gloo.  Each rank renders
its row block with the oracle standing in for the HIP kernel, the product's FrameGather
assembles the frame on rank 0, and the result must equal a single-process full frame bit for
bit (generators are keyed on the global pixel id, so tiling cannot change any pixel)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, size, spp, rng, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge

    ge.load_package()
    from cuda_pathtrace_amd import tiling

    oracle = ge.load_oracle()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fg = tiling.FrameGather(size, size, torch.device("cpu"))
        b, e = fg.rows
        assert (b, e) == tiling.row_range(size, world, rank)
        tile = oracle.render(size, size, spp, rng_mode=rng, row_begin=b, row_end=e, threads=2)
        fg.tile.copy_(torch.from_numpy(tile.reshape(-1)))
        fg.wait_all(fg.gather())
        dist.barrier()
        if rank == 0:
            q.put(fg.frame.numpy().reshape(size, size, 14).copy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,size", [(2, 32), (3, 31)])
@pytest.mark.parametrize("rng", [0, 1])
def test_row_tiled_frame_equals_single_process_frame(oracle, world, size, rng):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, size, 2, rng, q)) for r in range(world)]
    for p in procs:
        p.start()
    frame = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = oracle.render(size, size, 2, rng_mode=rng)
    assert np.array_equal(frame.view(np.uint32), full.view(np.uint32))


def test_eight_ranks_ragged_frame_equals_single_process_frame(oracle):
    """World size 8 -- the largest launch the driver makes (`bench.py --gpus 8`), never run on hardware -- with a frame height that does
    not divide (36 rows -> 5,5,5,5,4,4,4,4): seven receives posted by the root straight into the frame, one send per peer."""
    world, size = 8, 36
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, size, 2, 0, q)) for r in range(world)]
    for p in procs:
        p.start()
    frame = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    full = oracle.render(size, size, 2, rng_mode=0)
    assert np.array_equal(frame.view(np.uint32), full.view(np.uint32))


def test_row_range_partitions_exactly():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge

    ge.load_package()
    from cuda_pathtrace_amd import tiling

    for h in (1, 7, 64, 1024, 4096):
        for g in (1, 2, 3, 4, 8):
            spans = [tiling.row_range(h, g, r) for r in range(g)]
            assert spans[0][0] == 0 and spans[-1][1] == h
            assert all(spans[i][1] == spans[i + 1][0] for i in range(g - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
