"""world_size-2 gloo test (CPU) of the multi-GPU plumbing: contiguous batch sharding by rank and the
single all-gather of packed detections [B_local, Q, C+4] (dinov2_od_amd/dist.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dinov2_od_amd.dist import shard_bounds


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, global_batch, ragged, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from dinov2_od_amd import dist as ddist
    ddist.init_from_env("gloo")
    lo, hi = shard_bounds(global_batch, rank, world)
    Q, C = 5, 7
    full = torch.arange(global_batch * Q * (C + 4), dtype=torch.float32).view(global_batch, Q, C + 4)
    local = full[lo:hi].clone()                  # stands for this rank's forward_packed() output
    out = ddist.gather_detections(local) if ragged else ddist.gather_detections_equal(local)
    ok = torch.equal(out, full)
    q.put((rank, ok, tuple(out.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("global_batch,ragged", [(8, False), (8, True), (7, True)])
def test_gather_detections_world2(global_batch, ragged):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, global_batch, ragged, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, shape in res:
        assert ok, (rank, shape)
        assert shape[0] == global_batch


def test_shard_bounds_cover_batch():
    for gb in (1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(gb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
