"""N>1 path on CPU: world_size-2 gloo run of the sharding + final reduction bench.py uses."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from radiosaber_amd import sharding


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, cells, S, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = sharding.cell_ids_for_rank(rank, world, cells)
    seeds = sharding.seeds_for_cells(ids)
    # stand-in for the per-rank device result: a deterministic function of the global cell ids
    per_cell = (ids[:, None].astype(np.int64) * 1000003 + np.arange(S)[None, :]) % 99991
    t = torch.from_numpy(per_cell.sum(0).astype(np.int64))
    sharding.all_reduce_slice_bytes(t, dist)
    wall = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(wall, op=dist.ReduceOp.MAX)
    if rank == 0:
        out.put((t.numpy().tolist(), float(wall.item()), seeds[:3].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_reduce():
    world, cells, S = 2, 6, 20
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cells, S, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, wall, seeds0 = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ids = np.arange(world * cells, dtype=np.int64)
    exp = ((ids[:, None] * 1000003 + np.arange(S)[None, :]) % 99991).sum(0)
    assert got == exp.tolist()
    assert abs(wall - 0.2) < 1e-12  # MAX over ranks
    assert seeds0 == sharding.seeds_for_cells(np.arange(3)).tolist()


def test_shards_partition_the_cells():
    world, cells = 8, 512
    allc = np.concatenate([sharding.cell_ids_for_rank(r, world, cells) for r in range(world)])
    assert (allc == np.arange(world * cells)).all()
    s = sharding.seeds_for_cells(allc)
    assert len(np.unique(s)) == len(s) and s.max() < 2**31 - 1
    assert [sharding.first_cell_for_rank(r, world, cells) for r in (0, 1, 7)] == [0, 512, 3584]
    # a cell's seed does not depend on the number of ranks
    assert (sharding.seeds_for_cells(sharding.cell_ids_for_rank(3, 4, 1024))[:512] ==
            sharding.seeds_for_cells(sharding.cell_ids_for_rank(6, 8, 512))).all()


def test_throughput_reducer():
    mb = sharding.slice_throughput_mbps([1250000, 0], 10.0)
    assert mb.tolist() == [1.0, 0.0]
