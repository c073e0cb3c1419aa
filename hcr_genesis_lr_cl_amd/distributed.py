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
    """Packs one step's outputs -- every observation tensor step() returned (actor obs; privileged / critic obs and
    estimator labels where the task has them), reward and done -- into one (n_local, sum(widths) + 2) float record and
    all-gathers it.

    `overlap=True` (bench.py at N > 1): the collective of step t is issued asynchronously and only awaited when its
    double-buffered record is about to be reused at step t+2 (or in `finish()`), so the RCCL latency hides behind the next
    step's kernel.  That is the data flow SURVEY 8e recommends -- every rank steps its own envs with its own copy of the
    policy, the gathered records feed the learner once the rollout is over -- and it keeps one all-gather per step on the
    wire.  `__call__` then returns the buffer the gather is landing in; read it after `finish()` (or after the call two
    steps later).

    The inputs may be strided views (sliding history windows) and `done` a bool tensor: the packing is one `torch.cat`
    into the float record."""

    def __init__(self, n_local, widths, world, device, dtype=torch.float32, overlap=False, batch=1):
        """`batch` = control steps per collective: the records of `batch` consecutive steps travel in ONE all-gather (issued after
        the last of them).  A per-step gather of the go2 record is 0.77 MB per rank -- at 8 ranks 5.4 MB in per rank every 28 us,
        i.e. ~190 GB/s of ingress in messages small enough to be latency-bound; 4 steps per collective make it one ~24 MB
        message per 110 us, which RCCL moves near its large-message bandwidth.  The consumer (a learner reading the rollout after
        its last step) sees the same records either way."""
        self.widths = [int(widths)] if isinstance(widths, int) else [int(w) for w in widths]
        self.num_obs = sum(self.widths)
        self.world, self.overlap = world, bool(overlap) and world > 1
        self.batch, self.n_local = max(int(batch), 1), int(n_local)
        nb = 2 if self.overlap else 1
        W = self.num_obs + 2
        self.recs = [torch.empty(self.batch * n_local, W, device=device, dtype=dtype) for _ in range(nb)]
        self.outs = [torch.empty(world * self.batch * n_local, W, device=device, dtype=dtype) if world > 1 else r for r in self.recs]
        self.work = [None] * nb
        self.t = 0
        self.last_fill = self.batch
        self.rec, self.out = self.recs[0], self.outs[0]

    def _issue(self, i, fill=None):
        """All-gather of buffer pair i; `fill` < batch sends only the first `fill` steps' records (a partly filled batch)."""
        if self.world > 1:
            import torch.distributed as dist
            rec, out = self.recs[i], self.outs[i]
            if fill is not None and fill < self.batch:
                rec, out = rec[:fill * self.n_local], out[:self.world * fill * self.n_local]
            if self.overlap:
                self.work[i] = dist.all_gather_into_tensor(out, rec, async_op=True)
            else:
                dist.all_gather_into_tensor(out, rec)

    def __call__(self, obs, rew, done):
        """Pack this step's outputs; every `batch`-th call issues the collective.  Returns the buffer the records of the current
        batch land in (complete after that collective; with overlap, after `finish()` or two batches later)."""
        i = (self.t // self.batch) % len(self.recs)
        slot = self.t % self.batch
        self.t += 1
        if slot == 0 and self.work[i] is not None:   # the collective that last used this buffer pair must have finished
            self.work[i].wait()
            self.work[i] = None
        self.out = self.outs[i]
        r = self.rec = self.recs[i][slot * self.n_local:(slot + 1) * self.n_local]
        parts = [obs] if torch.is_tensor(obs) else list(obs)
        # one fused packing kernel on the compute stream (slice copies cost a launch each per step)
        torch.cat((*parts, rew.unsqueeze(1), done.unsqueeze(1).to(r.dtype)), dim=1, out=r)
        if slot == self.batch - 1:
            self._issue(i)
        return self.out

    def prime(self):
        """One all-gather on each buffer pair, awaited: RCCL sets up the connections of an algorithm at its first use, and the first
        collective of a run would otherwise land `batch` steps into it (inside a short timed region).  Leaves `t` untouched."""
        if self.world > 1:
            for i in range(len(self.recs)):
                self.recs[i].zero_()
                self._issue(i)
                if self.work[i] is not None:
                    self.work[i].wait()
                    self.work[i] = None
            if self.recs[0].is_cuda:
                torch.cuda.synchronize()

    def finish(self):
        """Send a partly filled batch (only the steps it holds: `last_fill` of them, laid out as a batch of that size -- pass it to
        `step_view`), then wait for every outstanding gather (end of rollout / end of the timed region)."""
        self.last_fill = self.batch
        if self.t % self.batch != 0:
            self.last_fill = self.t % self.batch
            self._issue((self.t // self.batch) % len(self.recs), self.last_fill)
            self.t += self.batch - self.t % self.batch
        for i, w in enumerate(self.work):
            if w is not None:
                w.wait()
                self.work[i] = None

    def step_view(self, out, rank, slot, fill=None):
        """The (n_local, W) record of `rank` at step `slot` of a gathered batch (`fill`: steps in it, for the partly filled batch
        `finish()` flushed)."""
        b = self.batch if fill is None else int(fill)
        w = self.world if self.world > 1 else 1
        o = out[:w * b * self.n_local].view(w, b, self.n_local, self.num_obs + 2)
        return o[rank, slot]

    def split(self, out=None):
        """-> ([obs tensors in the order given], rew, done) views of a gathered record."""
        o = self.out if out is None else out
        parts, c = [], 0
        for w in self.widths:
            parts.append(o[:, c:c + w])
            c += w
        return parts, o[:, self.num_obs], o[:, self.num_obs + 1] > 0.5
