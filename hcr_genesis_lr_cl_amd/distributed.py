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

    MODE = "rccl"

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
        self.wait_s = 0.0          # host time spent waiting for a collective inside __call__ (attribution: bench.py `gather.wait_us_per_step`)
        self.finish_s = 0.0        # ... and inside finish(): what of the exchange was NOT hidden behind the steps
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
                import time
                t0 = time.perf_counter()
                dist.all_gather_into_tensor(out, rec)
                if rec.is_cuda:
                    torch.cuda.current_stream(rec.device).synchronize()      # "sync": the step that follows starts after the records arrived
                self.wait_s += time.perf_counter() - t0

    def __call__(self, obs, rew, done):
        """Pack this step's outputs; every `batch`-th call issues the collective.  Returns the buffer the records of the current
        batch land in (complete after that collective; with overlap, after `finish()` or two batches later)."""
        i = (self.t // self.batch) % len(self.recs)
        slot = self.t % self.batch
        self.t += 1
        if slot == 0 and self.work[i] is not None:   # the collective that last used this buffer pair must have finished
            import time
            t0 = time.perf_counter()
            self.work[i].wait()
            self.wait_s += time.perf_counter() - t0
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
        import time
        if self.recs[0].is_cuda:
            torch.cuda.current_stream(self.recs[0].device).synchronize()    # the steps themselves: not part of the exchange's exposed time
        t0 = time.perf_counter()
        self.last_fill = self.batch
        if self.t % self.batch != 0:
            self.last_fill = self.t % self.batch
            self._issue((self.t // self.batch) % len(self.recs), self.last_fill)
            self.t += self.batch - self.t % self.batch
        for i, w in enumerate(self.work):
            if w is not None:
                w.wait()
                self.work[i] = None
        if self.recs[0].is_cuda:
            torch.cuda.current_stream(self.recs[0].device).synchronize()
        self.finish_s += time.perf_counter() - t0

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


class PeerGather(StepGather):
    """The same record exchange WITHOUT a collective kernel: every rank owns a receive buffer of (world, batch * n_local, W) floats, the
    other ranks map it once (GPU: IPC memory handles exchanged through the process group; CPU tests: a file in /dev/shm) and from then on
    WRITE their batch straight into their slot of every peer's buffer -- device-to-device copies enqueued on a side stream, which the
    runtime hands to the SDMA copy engines, so no compute unit is taken from the step kernel.  (RCCL's all-gather runs in a kernel of
    its own -- `rcclGenericKernel`, 256-thread workgroups at 261-280 registers per wave on gfx950 -- which cannot share a SIMD with the
    256-register step kernel: DESIGN.md section 5.)  north_star's RCCL all-gather stays the control plane here: the process group carries
    the handles and the final barrier.

    Contract = StepGather(overlap=True): `__call__` packs the step's record; every `batch`-th call enqueues the peer writes and returns
    at once; the gathered records are complete after `finish()` (every rank drains its copy stream, then all ranks meet at a barrier).
    Buffers are reused every other batch, as there."""

    MODE = "copy-engine"

    def __init__(self, n_local, widths, world, device, dtype=torch.float32, batch=1, rank=None):
        super().__init__(n_local, widths, world, device, dtype=dtype, overlap=True, batch=batch)
        import torch.distributed as dist
        self.rank = int(rank if rank is not None else (dist.get_rank() if world > 1 else 0))
        self.device = torch.device(device)
        self._shm = []
        self.peers = [None, None]          # [buffer pair][rank] -> that rank's receive buffer (ours: self.outs[i])
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self.sent = [None, None]           # event after the last copy out of recs[i]
        if world > 1:
            self._open_peers()

    # ---- peer mapping -------------------------------------------------------------------------------------------------
    def _open_peers(self):
        import torch.distributed as dist
        W = self.num_obs + 2
        shape = (self.world * self.batch * self.n_local, W)
        if self.device.type == "cuda":
            from torch.multiprocessing.reductions import reduce_tensor
            mine = [reduce_tensor(o) for o in self.outs]                 # (rebuild function, arguments with the IPC memory handle)
        else:
            import os
            import uuid
            import numpy as np
            mine = []
            for i in range(2):
                path = f"/dev/shm/lg_gather_{uuid.uuid4().hex}_{self.rank}_{i}"
                mm = np.memmap(path, dtype=np.float32, mode="w+", shape=shape)
                self._shm.append(path)
                self.outs[i] = torch.from_numpy(mm)                      # our receive buffer lives in the shared file
                mine.append(path)
            self.out = self.outs[0]
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine)
        for i in range(2):
            row = []
            for r in range(self.world):
                if r == self.rank:
                    row.append(self.outs[i])
                elif self.device.type == "cuda":
                    fn, args = everyone[r][i]
                    row.append(fn(*args))                                # opens the peer's allocation in this process
                else:
                    import numpy as np
                    row.append(torch.from_numpy(np.memmap(everyone[r][i], dtype=np.float32, mode="r+", shape=shape)))
            self.peers[i] = row
        dist.barrier()

    # ---- the exchange ---------------------------------------------------------------------------------------------------
    def _issue(self, i, fill=None):
        if self.world <= 1:
            return
        b = self.batch if fill is None or fill >= self.batch else int(fill)
        rows = b * self.n_local
        src = self.recs[i][:rows]
        lo = self.rank * rows                                             # layout of a gathered batch of b steps: (world, b, n_local, W)
        if self.copy_stream is not None:
            self.copy_stream.wait_stream(torch.cuda.current_stream(self.device))      # the records were packed on the compute stream
            with torch.cuda.stream(self.copy_stream):
                for r in range(self.world):
                    self.peers[i][r][lo:lo + rows].copy_(src, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.copy_stream)
            self.sent[i] = ev
        else:
            for r in range(self.world):
                self.peers[i][r][lo:lo + rows].copy_(src)

    def __call__(self, obs, rew, done):
        i = (self.t // self.batch) % 2
        if self.t % self.batch == 0 and self.sent[i] is not None:
            # the copies out of this record buffer (two batches ago) must have been issued to the engines before it is overwritten:
            # the compute stream waits for them on the device, the host does not block
            torch.cuda.current_stream(self.device).wait_event(self.sent[i])
            self.sent[i] = None
        return super().__call__(obs, rew, done)

    def prime(self):
        if self.world > 1:
            import torch.distributed as dist
            for i in range(2):
                self.recs[i].zero_()
                self._issue(i)
            if self.copy_stream is not None:
                self.copy_stream.synchronize()
            self.sent = [None, None]
            dist.barrier()

    def finish(self):
        import time
        import torch.distributed as dist
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()        # the steps themselves: not part of the exchange's exposed time
        t0 = time.perf_counter()
        self.last_fill = self.batch
        if self.t % self.batch != 0:
            # A partly filled batch travels compacted -- (world, fill, n_local, W), as the collective transports send it -- so rank r's rows
            # lie where the FULL batches of lower ranks put theirs.  Nothing but this call orders the ranks' writes: a rank that is a whole
            # batch ahead would have its flushed rows overwritten by a slower rank's earlier full batch into the same buffer pair.  So every
            # rank's full batches land first (drain, barrier), then the flush goes out.
            if self.copy_stream is not None:
                self.copy_stream.synchronize()
            if self.world > 1:
                dist.barrier()
            self.last_fill = self.t % self.batch
            self._issue((self.t // self.batch) % 2, self.last_fill)
            self.t += self.batch - self.t % self.batch
        if self.copy_stream is not None:
            self.copy_stream.synchronize()
        if self.world > 1:
            dist.barrier()                                               # every rank's writes into every buffer have landed
        self.finish_s += time.perf_counter() - t0

    def close(self):
        import os
        self.peers = [None, None]
        for pth in self._shm:
            try:
                os.remove(pth)
            except OSError:
                pass
        self._shm = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


GATHER_MODES = ("rccl", "rccl-sync", "copy-engine")


def make_gather(mode, n_local, widths, world, device, batch=4):
    """The record exchange of bench.py / a learner at N > 1 in one of the three transports:
      "rccl"         one RCCL all-gather per `batch` steps, overlapped with the following steps (double-buffered)
      "rccl-sync"    one RCCL all-gather per step, awaited before the next step (north_star's literal reading)
      "copy-engine"  peer writes by the copy engines, `batch` steps per write (PeerGather)"""
    if mode == "rccl":
        g = StepGather(n_local, widths, world, device, overlap=True, batch=batch)
    elif mode == "rccl-sync":
        g = StepGather(n_local, widths, world, device, overlap=False, batch=1)
    elif mode == "copy-engine":
        g = PeerGather(n_local, widths, world, device, batch=batch)
    else:
        raise ValueError(f"gather mode {mode!r}: one of {GATHER_MODES}")
    g.mode = mode
    return g


def pick_fastest(times_us):
    """{mode: microseconds per step measured on this rank} -> the mode every rank agrees on: the one whose SLOWEST rank is fastest
    (all-reduce MAX over ranks, then argmin; ties go to the earlier entry of GATHER_MODES).  Returns (mode, {mode: max over ranks})."""
    import torch.distributed as dist
    modes = [m for m in GATHER_MODES if m in times_us]
    t = torch.tensor([float(times_us[m]) for m in modes], dtype=torch.float64)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t = t.cpu()
    best = int(torch.argmin(t))
    return modes[best], {m: float(v) for m, v in zip(modes, t)}
