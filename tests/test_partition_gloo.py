"""N > 1 path on CPU: two gloo ranks each render their interleaved row blocks (with the oracle
standing in for the GPU kernel -- test only), gather the packed rows to rank 0 exactly as
bench.py does on RCCL, and rank 0 checks the reassembled frame against the single-rank render."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mort_amd import partition

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_row_maps_cover_image_exactly_once():
    for H, N, rpb in ((675, 8, 8), (112, 2, 8), (90, 4, 16), (7, 2, 8), (64, 3, 8)):
        seen = np.zeros(H, dtype=int)
        for r in range(N):
            lr = partition.local_rows(H, r, N, rpb)
            g = partition.global_rows(r, N, rpb, lr).numpy()
            assert (g < H).all() and lr <= partition.max_local_rows(H, N, rpb)
            seen[g] += 1
        assert (seen == 1).all(), (H, N, rpb)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world_size, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        from mort_amd import host
        from tests import oracle_lib as O
        rpb = 8
        world, cam = host.build_scene(1, width=96, spp=4)
        W, H = cam.image_width, cam.image_height
        fg = partition.FrameGather(H, W, 4, torch.uint8, rank, world_size, rpb, torch.device("cpu"))
        # this rank's rows: render them (full-frame buffers, only owned rows are written)
        lr = partition.local_rows(H, rank, world_size, rpb)
        rows = partition.global_rows(rank, world_size, rpb, lr).numpy()
        states = O.seed_states(69420, W, H)
        rgba = np.zeros((H, W, 4), dtype=np.uint8)
        for b0 in range(0, lr, rpb):
            g0 = int(rows[b0])
            g1 = min(g0 + rpb, H)
            part = O.render(world, cam, states=states, rows=(g0, g1), nthreads=1, want_accum=False, want_segments=False)
            rgba[g0:g1] = part["rgba"][g0:g1]
        fg.tile[:lr] = torch.from_numpy(rgba[rows])
        frame = fg.gather(dist)
        if rank == 0:
            full = O.render(world, cam, nthreads=2, want_accum=False, want_segments=False)["rgba"]
            q.put(bool((frame.numpy() == full).all()))
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_reassembles_the_frame():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
