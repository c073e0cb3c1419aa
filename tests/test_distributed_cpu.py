"""N>1 host path on CPU: two gloo ranks shard the env index space and all-gather the per-step
record exactly as bench.py does with RCCL."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from hcr_genesis_lr_cl_amd.distributed import StepGather, global_mean, shard


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    off, n = shard(64, world, rank)
    g = StepGather(n, 5, world, "cpu")
    ids = torch.arange(off, off + n, dtype=torch.float32)
    obs = ids[:, None] * torch.ones(1, 5) + torch.arange(5) * 0.1
    out = g(obs, ids * 2, (ids % 3 == 0).float())
    o, r, d = g.split()
    # command-curriculum mean over the resetting envs of all ranks: rank 0 has 3 of them (sum 6), rank 1 none
    gm = global_mean(torch.tensor(6.0) if rank == 0 else torch.tensor(0.0), 3 if rank == 0 else 0)
    assert gm == (2.0, 3.0), gm
    # overlapped mode: three steps in flight over two buffer pairs, results read after finish()
    g2 = StepGather(n, 5, world, "cpu", overlap=True)
    outs = [g2(obs + k, ids * (2 + k), (ids % 3 == 0).float()) for k in range(3)]
    g2.finish()
    full = torch.arange(64, dtype=torch.float32)
    assert torch.allclose(outs[2][:, 5], full * 4) and torch.allclose(outs[1][:, 5], full * 3)
    assert torch.allclose(outs[2][:, 0], full + 2)
    q.put((rank, o.numpy().copy(), r.numpy().copy(), d.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=120) for _ in range(world)]
    [p.join(timeout=60) for p in ps]
    ids = np.arange(64, dtype=np.float32)
    for rank, o, r, d in res:
        np.testing.assert_allclose(o, ids[:, None] + np.arange(5) * 0.1, rtol=1e-6)
        np.testing.assert_allclose(r, ids * 2)
        np.testing.assert_array_equal(d, ids % 3 == 0)


def test_shard_is_contiguous_and_even():
    assert [shard(32768, 8, r) for r in (0, 3, 7)] == [(0, 4096), (12288, 4096), (28672, 4096)]
    import pytest
    with pytest.raises(ValueError):
        shard(10, 4, 0)


def test_global_mean_without_process_group_is_local():
    assert global_mean(torch.tensor(9.0), 3) == (3.0, 3.0)
    assert global_mean(torch.tensor(0.0), 0) == (0.0, 0.0)
