"""Env sharding across the GPUs of one node: one process per GPU, contiguous
blocks of envs per rank, no collective inside the physics/MDP path.  The only exchange is the
per-step all-gather of the returned record (obs | reward | done) named by BASELINE.json's
north_star -- RCCL (backend "nccl") over xGMI on the GPUs, gloo in the CPU tests.
"""
from __future__ import annotations

import torch


def shard(global_num_envs: int, world: int, rank: int):
    """(offset, count) of the contiguous env block owned by `rank`."""
    if global_num_envs % world:
        raise ValueError("global env count must divide evenly over ranks (weak scaling: fixed envs per GPU)")
    n = global_num_envs // world
    return rank * n, n


def global_mean(local_sum, local_count):
    """Mean over the envs of ALL ranks from each rank's (sum, count) -- the command-curriculum decision of
    legged_robot.py:336-348 is a mean over the envs that reset at the gate step, wherever they live (one
    small all-reduce every max_episode_length steps).  Every rank must call it at the gate step, also with count 0.
    Returns (mean, total_count) as Python floats; without an initialised process group it is the local mean."""
    import torch.distributed as dist
    dev = local_sum.device if torch.is_tensor(local_sum) else torch.device("cpu")
    t = torch.zeros(2, dtype=torch.float64, device=dev)
    t[0] = local_sum
    t[1] = float(local_count)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    s, c = float(t[0]), float(t[1])
    return (s / c if c > 0 else 0.0), c


class StepGather:
    """Packs one step's outputs into a (n_local, obs+2) record and all-gathers it."""

    def __init__(self, n_local, num_obs, world, device, dtype=torch.float32):
        self.world, self.num_obs = world, num_obs
        self.rec = torch.empty(n_local, num_obs + 2, device=device, dtype=dtype)
        self.out = torch.empty(world * n_local, num_obs + 2, device=device, dtype=dtype) if world > 1 else self.rec

    def __call__(self, obs, rew, done):
        r = self.rec
        r[:, :self.num_obs] = obs
        r[:, self.num_obs] = rew
        r[:, self.num_obs + 1] = done
        if self.world > 1:
            import torch.distributed as dist
            dist.all_gather_into_tensor(self.out, r)
        return self.out

    def split(self):
        o = self.out
        return o[:, :self.num_obs], o[:, self.num_obs], o[:, self.num_obs + 1] > 0.5
