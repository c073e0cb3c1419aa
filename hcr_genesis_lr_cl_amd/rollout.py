"""RolloutStorage with the per-step record and the return computation as HIP kernels.

Mirrors the reference's rsl_rl/storage/rollout_storage.py: same constructor, same tensors under the same names and shapes
((T, N, width) float32; dones (T, N, 1) uint8), `add_transitions`, `clear`, `compute_returns`, `get_statistics`,
`mini_batch_generator` -- so rsl_rl's PPO with a feed-forward ActorCritic takes it in place of its own (recurrent policies are refused:
hidden states are not stored).  What changes is how the rows are filled:

  * `record(t-less API: add_step)`: reward (with the time-out bootstrap of rsl_rl/algorithms/ppo.py:106-113), done flag and any
    observation rows in ONE launch (`lg_rollout_record`) instead of nine `copy_` launches (rollout_storage.py:92-100);
  * `compute_returns`: GAE over the whole rollout plus the advantage normalisation in two launches (`lg_rollout_gae`) instead
    of a 24-iteration Python loop of elementwise ops (rollout_storage.py:124-138);
  * with `attach_env(env)` on a task with unstacked observations the storage's observation rows ARE the env's observation
    copies (LgTaskCfg.obs_sets = T + 1): step() writes each observation straight into its row and nothing is copied.

There is no CPU path: the kernels live in csrc/liblgsim.so (include/lgrollout.h)."""
from __future__ import annotations

import ctypes as C

import torch

from . import abi


class RolloutStorage:
    class Transition:                      # rollout_storage.py:37-52
        def __init__(self):
            self.observations = None
            self.critic_observations = None
            self.actions = None
            self.rewards = None
            self.dones = None
            self.values = None
            self.actions_log_prob = None
            self.action_mean = None
            self.action_sigma = None
            self.hidden_states = None

        def clear(self):
            self.__init__()

    def __init__(self, num_envs, num_transitions_per_env, obs_shape, privileged_obs_shape, actions_shape, device="cuda:0", env=None):
        self.lib = abi.load_lib()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("hcr_genesis_lr_cl_amd.rollout.RolloutStorage needs a HIP device (no CPU fallback)")
        self.obs_shape, self.privileged_obs_shape, self.actions_shape = obs_shape, privileged_obs_shape, actions_shape
        T, N, dev = int(num_transitions_per_env), int(num_envs), self.device
        self.num_transitions_per_env, self.num_envs = T, N
        z = lambda *s, **k: torch.zeros(*s, device=dev, **k)
        self._env = None
        self.observations = None
        if env is not None:
            self.attach_env(env)
        if self.observations is None:
            self.observations = z(T, N, *obs_shape)
        self.privileged_observations = z(T, N, *privileged_obs_shape) if privileged_obs_shape[0] is not None else None
        self.rewards = z(T, N, 1)
        self.actions = z(T, N, *actions_shape)
        self.dones = z(T, N, 1, dtype=torch.uint8)
        self.actions_log_prob = z(T, N, 1)
        self.values = z(T, N, 1)
        self.returns = z(T, N, 1)
        self.advantages = z(T, N, 1)
        self.mu = z(T, N, *actions_shape)
        self.sigma = z(T, N, *actions_shape)
        self._scratch = z(2, dtype=torch.float64)
        self.saved_hidden_states_a = self.saved_hidden_states_c = None
        self.step = 0

    # ---- zero-copy observation rows -----------------------------------------------------------------------------------
    def attach_env(self, env):
        """Lay the observation rows over the env's observation copies.  Needs an env built with cfg.hip.obs_sets = T + 1 and
        unstacked observations (go2); otherwise the rows stay separate and are filled by the record kernel."""
        eng = env._engine
        raw = eng.buf.raw("obs_buf")
        T = self.num_transitions_per_env
        if int(eng.task.obs_sets) != T + 1 or int(eng.task.obs_stack) != 1 or raw.shape[1:] != (self.num_envs, *self.obs_shape):
            return False
        self._env = env
        self.observations = raw[:T]
        self._obs_all = raw
        self._sync_env_cycle()
        return True

    def _sync_env_cycle(self):
        """Start of a rollout: the env's current observation becomes row 0 and the env's next T observations land in rows 1 .. T
        (row T is the observation after the last step, i.e. row 0 of the next rollout)."""
        eng = self._env._engine
        cur = eng.obs_set()
        if cur != 0:
            self._obs_all[0].copy_(self._obs_all[cur])
        abi.check(self.lib.lg_obs_set_select(eng.handle, 0), self.lib)
        eng._refresh_obs_slot()

    @property
    def zero_copy(self):
        return self._env is not None

    # ---- filling rows -------------------------------------------------------------------------------------------------
    def add_step(self, rew, reset, time_outs, gamma, observations=None, critic_observations=None, extra_copies=()):
        """One launch for everything the env contributes to row `self.step`: rewards (+ gamma * value * time_out), dones and
        the observation rows (skipped when they are zero-copy).  `values[self.step]` must already hold the critic's output for
        this step (the bootstrap reads it).  Policy-side rows (actions, values, log-prob, mean, std) are written by the caller
        straight into `self.actions[self.step]` ... as outputs of its own ops."""
        t = self.step
        if t >= self.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        copies = []
        if observations is not None and not self.zero_copy:
            copies.append((observations, self.observations[t]))
        if critic_observations is not None and self.privileged_observations is not None:
            copies.append((critic_observations, self.privileged_observations[t]))
        copies += list(extra_copies)
        arr = (abi.LgRowCopy * max(len(copies), 1))()
        for i, (src, dst) in enumerate(copies):
            if src.dim() != 2 or src.stride(1) != 1 or not dst.is_contiguous() or src.shape != dst.shape or src.dtype != torch.float32:
                raise ValueError("row copies take (N, width) float32 views with unit inner stride")
            arr[i].src, arr[i].dst, arr[i].width, arr[i].src_stride = src.data_ptr(), dst.data_ptr(), src.shape[1], src.stride(0)
        rst = reset.view(torch.uint8) if reset.dtype == torch.bool else reset
        to = None if time_outs is None else (time_outs.view(torch.uint8) if time_outs.dtype == torch.bool else time_outs)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        abi.check(self.lib.lg_rollout_record(self.num_envs, rew.data_ptr(), rst.data_ptr(), 0 if to is None else to.data_ptr(),
                                             self.values[t].data_ptr(), float(gamma), self.rewards[t].data_ptr(), self.dones[t].data_ptr(),
                                             arr, len(copies), stream), self.lib)
        self.step += 1

    def add_transitions(self, transition):
        """rollout_storage.py:89-102, the reference's entry point (rsl_rl's PPO.process_env_step has already bootstrapped
        `transition.rewards`): policy-side rows by copy_, env-side rows by the record kernel."""
        t = self.step
        if t >= self.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        hs = getattr(transition, "hidden_states", None)
        if hs is not None and tuple(hs) != (None, None):
            raise NotImplementedError("hidden states (recurrent ActorCritic, rollout_storage.py:101-122) are not stored here: "
                                      "use rsl_rl's own RolloutStorage for recurrent policies")
        self.actions[t].copy_(transition.actions)
        self.values[t].copy_(transition.values)
        self.actions_log_prob[t].copy_(transition.actions_log_prob.view(-1, 1))
        self.mu[t].copy_(transition.action_mean)
        self.sigma[t].copy_(transition.action_sigma)
        crit = transition.critic_observations if self.privileged_observations is not None else None
        self.add_step(transition.rewards.reshape(-1).contiguous(), transition.dones.reshape(-1), None, 0.0,
                      observations=transition.observations, critic_observations=crit)

    def clear(self):
        self.step = 0
        if self.zero_copy:
            self._sync_env_cycle()

    def compute_returns(self, last_values, gamma, lam):
        """rollout_storage.py:124-138 in two launches."""
        lv = last_values.reshape(-1).contiguous().float()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        abi.check(self.lib.lg_rollout_gae(self.num_transitions_per_env, self.num_envs, self.values.data_ptr(), self.rewards.data_ptr(),
                                          self.dones.data_ptr(), lv.data_ptr(), float(gamma), float(lam), self.returns.data_ptr(),
                                          self.advantages.data_ptr(), self._scratch.data_ptr(), stream), self.lib)

    def get_statistics(self):
        """(mean trajectory length, mean reward) of the stored rollout -- what rollout_storage.py:140-146 reports: a trajectory ends at
        a done flag or at the last stored step.  Leaves `dones` untouched."""
        ends = self.dones.squeeze(-1).t().bool().clone()        # (N, T), env-major like the reference's flattening
        ends[:, -1] = True
        idx = torch.nonzero(ends.reshape(-1)).squeeze(1)
        lengths = torch.diff(idx, prepend=idx.new_full((1,), -1))
        return lengths.float().mean(), self.rewards.mean()

    def mini_batch_generator(self, num_mini_batches, num_epochs=8):
        """Mini-batches in the tuple order rsl_rl's PPO.update unpacks (rollout_storage.py:148-186): obs, critic obs, actions, target
        values, advantages, returns, old log-prob, old mean, old std, hidden states (None, None), masks None.  One random permutation of
        the T x N samples per call, cut into `num_mini_batches` equal index blocks and replayed every epoch."""
        flat = {k: getattr(self, k).flatten(0, 1) for k in
                ("observations", "actions", "values", "advantages", "returns", "actions_log_prob", "mu", "sigma")}
        flat["critic"] = self.privileged_observations.flatten(0, 1) if self.privileged_observations is not None else flat["observations"]
        order = ("observations", "critic", "actions", "values", "advantages", "returns", "actions_log_prob", "mu", "sigma")
        per = (self.num_envs * self.num_transitions_per_env) // num_mini_batches
        blocks = torch.randperm(num_mini_batches * per, device=self.device).view(num_mini_batches, per)
        for _ in range(num_epochs):
            for b in blocks:
                yield (*(flat[k][b] for k in order), (None, None), None)

    def reccurent_mini_batch_generator(self, num_mini_batches, num_epochs=8):
        raise NotImplementedError("recurrent policies are outside the hot path (SURVEY 8): use rsl_rl's own RolloutStorage for them")
