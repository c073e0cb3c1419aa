"""N>1 host path on CPU: two gloo ranks shard the env index space and all-gather the per-step
record exactly as bench.py does with RCCL."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from hcr_genesis_lr_cl_amd.distributed import GATHER_MODES, PeerGather, StepGather, global_mean, make_gather, pick_fastest, shard


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
    (o,), r, d = g.split()
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
    # several steps per collective: 7 steps in batches of 3 (the last one partly filled and flushed by finish()), overlapped
    g3 = StepGather(n, 5, world, "cpu", overlap=True, batch=3)
    g3.prime()       # communicator warm-up on both buffer pairs: must not disturb the step count or the records that follow
    assert g3.t == 0
    outs3 = [g3(obs + k, ids * (2 + k), (ids % 3 == 0).float()) for k in range(7)]
    g3.finish()
    assert g3.last_fill == 1          # 7 steps in batches of 3: the flushed batch holds one step and travels at that size
    for k, slot in ((3, 0), (5, 2), (6, 0)):
        for rk in range(world):
            o_k, n_k = shard(64, world, rk)
            rec = g3.step_view(outs3[k], rk, slot, g3.last_fill if k == 6 else None)
            ids_k = torch.arange(o_k, o_k + n_k, dtype=torch.float32)
            assert torch.allclose(rec[:, 5], ids_k * (2 + k)) and torch.allclose(rec[:, 0], ids_k + k), (k, rk)
    (p3,), r3, d3 = g3.split(g3.step_view(outs3[5], 1, 2))
    assert p3.shape == (n, 5) and d3.dtype == torch.bool
    # the three transports behind one interface (make_gather): the same 7 steps, the same records after finish() -- "copy-engine" = peer
    # writes into mapped buffers (here: files in /dev/shm), no collective on the data path
    for mode in GATHER_MODES:
        gm_ = make_gather(mode, n, 5, world, "cpu", batch=3)
        assert (mode == "copy-engine") == isinstance(gm_, PeerGather) and gm_.mode == mode
        gm_.prime()
        assert gm_.t == 0
        outs_m = [gm_(obs + k, ids * (2 + k), (ids % 3 == 0).float()) for k in range(7)]
        gm_.finish()
        b = gm_.batch
        for k in ((6,) if mode == "rccl-sync" else (3, 5, 6)):     # awaited mode: one buffer, a record is read before the next step overwrites it
            slot, fill = k % b, (gm_.last_fill if k // b == 6 // b else None)
            for rk in range(world):
                o_k, n_k = shard(64, world, rk)
                rec = gm_.step_view(outs_m[k], rk, slot, fill)
                ids_k = torch.arange(o_k, o_k + n_k, dtype=torch.float32)
                assert torch.allclose(rec[:, 5], ids_k * (2 + k)) and torch.allclose(rec[:, 0], ids_k + k), (mode, k, rk)
        assert gm_.wait_s >= 0.0 and gm_.finish_s > 0.0
        if hasattr(gm_, "close"):
            gm_.close()
    # a rank that is a whole batch behind: the other one reaches finish() and flushes its partly filled batch first -- the flush must not go
    # out before the slow rank's full batches have landed in the same buffer pair (compacted layout, distributed.PeerGather.finish)
    import time
    gs_ = make_gather("copy-engine", n, 5, world, "cpu", batch=3)
    gs_.prime()
    if rank == 0:
        time.sleep(0.6)
    outs_s = [gs_(obs + k, ids * (2 + k), (ids % 3 == 0).float()) for k in range(7)]
    gs_.finish()
    for rk in range(world):
        o_k, n_k = shard(64, world, rk)
        rec = gs_.step_view(outs_s[6], rk, 0, gs_.last_fill)
        ids_k = torch.arange(o_k, o_k + n_k, dtype=torch.float32)
        assert torch.allclose(rec[:, 5], ids_k * 8) and torch.allclose(rec[:, 0], ids_k + 6), ("late rank", rk)
    gs_.close()
    # the ranks agree on the transport: each rank's slowest-rank times decide (rank 1 alone would have picked rccl-sync)
    mode, agreed = pick_fastest({"rccl": 30.0 + 10 * rank, "rccl-sync": 50.0 - 20 * rank, "copy-engine": 35.0})
    assert mode == "copy-engine" and agreed == {"rccl": 40.0, "rccl-sync": 50.0, "copy-engine": 35.0}, (mode, agreed)
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


def test_gather_packs_strided_windows_bool_done_and_several_outputs():
    """What bench.py hands it for the history tasks: the actor stack is a strided window of a wider row, `done` is bool,
    and the critic stack and labels ride in the same record."""
    g = StepGather(6, [4, 3, 2], 1, "cpu")
    rows = torch.arange(6 * 10, dtype=torch.float32).reshape(6, 10)
    obs, crit, lab = rows[:, 3:7], rows[:, 1:4] * 2, rows[:, 8:] + 0.5
    assert not obs.is_contiguous()
    done = torch.tensor([True, False, False, True, False, False])
    g(obs=[obs, crit, lab], rew=torch.arange(6.0), done=done)
    (o, c, l), r, d = g.split()
    assert torch.equal(o, obs) and torch.equal(c, crit) and torch.equal(l, lab)
    assert torch.equal(r, torch.arange(6.0)) and torch.equal(d, done) and d.dtype == torch.bool


def test_bench_launcher_spawns_the_ranks_it_was_asked_for():
    """`python bench.py --gpus 2` from a plain shell (no WORLD_SIZE): the file itself starts two ranks through
    torch.distributed.run and rank 0 prints ONE JSON line saying n_gpus = 2 (dry run: gloo, no GPU, stand-in step)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--launcher-dry-run"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["gather_ok"] is True
    # the transport was chosen by measurement: all three candidates were timed, the line says which one ran
    g = out["gather"]
    assert set(g["candidates_us"]) == set(GATHER_MODES) and g["mode"] == min(g["candidates_us"], key=g["candidates_us"].get)
    assert g["step_us_alone"] > 0 and g["steps_per_exchange"] in (1, 4) and g["selected"].startswith("fastest")
    for forced in ("copy-engine", "rccl-sync"):
        rf = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--launcher-dry-run",
                             "--gather-mode", forced], capture_output=True, text=True, timeout=300, env=env)
        assert rf.returncode == 0, rf.stderr[-2000:]
        of = json.loads([l for l in rf.stdout.splitlines() if l.startswith("{")][0])
        assert of["gather_ok"] is True and of["gather"]["mode"] == forced and of["gather"]["selected"] == "forced"
    # a launcher that started a different number of ranks than --gpus says is refused, not silently accepted
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launcher-dry-run"], capture_output=True, text=True,
                        timeout=120, env=env2)
    assert r2.returncode != 0 and "--gpus 2" in (r2.stderr + r2.stdout)
