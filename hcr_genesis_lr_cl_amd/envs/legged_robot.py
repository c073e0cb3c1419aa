"""LeggedRobot -- the rsl_rl ``VecEnv`` surface of the reference, with the whole control step
(actions -> physics -> termination / rewards / resets / observations) fused into one HIP launch.

API parity targets (reference file:line):
  * ``step(actions) -> (obs, privileged_obs, rew, reset, extras)``  legged_robot.py:37-53
  * ``reset()``                                                     base_task.py:60-64
  * ``get_observations() / get_privileged_observations()``          base_task.py:50-54
  * attributes rsl_rl reads: num_envs, num_obs, num_privileged_obs, num_actions,
    max_episode_length, episode_length_buf, obs_buf, rew_buf, reset_buf, extras, device
    (rsl_rl/env/vec_env.py:36-60, on_policy_runner.py:100-124)
  * ``self.simulator`` is a HipSimulator exposing the reference's Simulator properties.

Differences by design: data-dependent ``nonzero()`` host syncs of the reference
(legged_robot.py:69,305) are replaced by masked in-kernel logic; random draws come from a
counter-based Philox stream keyed on (seed, global env id, step, slot).
"""
from __future__ import annotations

from collections.abc import MutableMapping

import numpy as np
import torch

from .. import abi
from .. import config as cfgmod
from ..simulator import HipSimulator


class _EpisodeExtras(MutableMapping):
    """Lazy ``extras["episode"]`` (legged_robot.py:128-138): per-term mean of the episode sums over the envs
    reset at one step, divided by episode_length_s.  The kernel snapshots every resetting env's sums and the
    step index; the means are formed on access, so the hot loop issues no extra launches and no atomics.
    When a step had no reset the reference leaves the previous dict in place: the same happens here by
    falling back to the most recent step (<= this one) that had resets.  (An env that resets twice before
    the dict is read contributes only its latest episode -- logging only.)
    The reference's dict is a plain one and its runner writes into it (on_policy_runner.py:182-185 rebinds
    ``ep_info[key]`` while logging): writes land in a local overlay, deletes hide a key."""

    def __init__(self, env, step):
        self._env, self._step = env, step
        self._mask = None
        self._over, self._gone = {}, set()

    def __setitem__(self, key, value):
        self._over[key] = value
        self._gone.discard(key)

    def __delitem__(self, key):
        if key not in self:
            raise KeyError(key)
        self._over.pop(key, None)
        self._gone.add(key)

    def _ids(self):
        if self._mask is None:
            ds = self._env._engine.buf["episode_done_step"]
            ok = (ds <= self._step) & (ds >= 0)
            if bool(ok.any()):
                self._mask = ds == ds[ok].max()
            else:
                self._mask = torch.zeros_like(ds, dtype=torch.bool)
        return self._mask

    def __getitem__(self, key):
        if key in self._gone:
            raise KeyError(key)
        if key in self._over:
            return self._over[key]
        env = self._env
        if key == "max_command_x":
            return env.command_ranges["lin_vel_x"][1]
        if key == "terrain_level":
            return torch.mean(env.simulator.terrain_levels.float())
        if key in env._lazy_episode_extras:      # task-specific entries formed on access (go2_cat: "cstr_probs")
            return env._lazy_episode_extras[key]()
        name = key[4:]
        if not key.startswith("rew_") or name not in env.episode_sums:
            raise KeyError(key)
        m = self._ids()
        if not bool(m.any()):
            return torch.zeros((), device=env.device)
        if name.startswith("cstr_"):      # constraint violation counters (go2_cat): logged like reward sums (constraint_manager.py:91-97)
            done = env._engine.buf["cstr_done_sums"][abi.CSTR_NAMES.index(name[5:])]
        else:
            done = env._engine.buf["episode_done_sums"][abi.reward_id(name, env.simulator._model.joints_per_leg)]
        return torch.mean(done[m]) / env.max_episode_length_s

    def _lazy_keys(self):
        env = self._env
        for k in env.episode_sums:
            yield "rew_" + k
        if env.cfg.terrain.curriculum:
            yield "terrain_level"
        if env.cfg.commands.curriculum:
            yield "max_command_x"
        yield from env._lazy_episode_extras

    def __iter__(self):
        seen = set()
        for k in self._lazy_keys():
            seen.add(k)
            if k not in self._gone:
                yield k
        for k in self._over:
            if k not in seen:
                yield k

    def __len__(self):
        return sum(1 for _ in self)


class LeggedRobot:
    _lazy_episode_extras = {}      # key -> callable, evaluated when extras["episode"][key] is read (see _EpisodeExtras)

    def __init__(self, cfg, sim_params=None, sim_device="cuda:0", headless=True, inject_rand=False,
                 env_id_offset=0, global_num_envs=None):
        self.cfg = cfg
        self.init_done = False
        self._parse_cfg(cfg)
        self.device = sim_device
        self.headless = headless
        self.num_envs = cfg.env.num_envs
        self.num_obs = cfg.env.num_observations
        self.num_privileged_obs = cfg.env.num_privileged_obs
        self.num_privileged_obs = cfg.env.num_privileged_obs
        self.num_actions = cfg.env.num_actions
        self.simulator = HipSimulator(cfg, sim_params or cfgmod.class_to_dict(cfg.sim), sim_device, headless,
                                      inject_rand=inject_rand, env_id_offset=env_id_offset,
                                      global_num_envs=global_num_envs)
        self._engine = self.simulator._engine
        b = self._engine.buf
        # base_task.py:29-37
        self.rew_buf = b["rew_buf"]
        self.reset_buf = b["reset_buf"].view(torch.bool)
        self.time_out_buf = b["time_out_buf"].view(torch.bool)
        self._episode_length_buf = b["episode_length_buf"]
        self.reset_buf.fill_(True)
        self.extras = {}
        self._init_buffers()
        self._prepare_reward_function()
        self._warm_rare_paths()
        self.init_done = True

    @property
    def episode_length_buf(self):
        """The kernel's per-env step counter.  rsl_rl REBINDS this attribute (`env.episode_length_buf = torch.randint_like(...)`,
        on_policy_runner.py:105-106 init_at_random_ep_len): the setter copies into the device buffer the kernel reads."""
        return self._episode_length_buf

    @episode_length_buf.setter
    def episode_length_buf(self, value):
        self._episode_length_buf.copy_(torch.as_tensor(value, device=self._episode_length_buf.device).to(self._episode_length_buf.dtype))

    # The observation outputs alternate between two copies and history-stacked ones are windows that move one frame per step
    # (engine._Buffers): look them up per access.  The tensors step() returns stay intact while the NEXT step() runs
    # (rsl_rl keeps them across env.step(), ppo.py:103-104 / rollout_storage.py:92) and are overwritten by the one after.
    @property
    def obs_buf(self):
        return self._engine.buf["obs_buf"]

    @property
    def privileged_obs_buf(self):
        return self._engine.buf.get("priv_obs_buf")

    # ------------------------------------------------------------------------------------------
    def step(self, actions):
        """legged_robot.py:37-53 in one launch (two on the rare command-curriculum steps)."""
        self.common_step_counter += 1
        c = self.common_step_counter
        self._engine_step(actions, c)
        self.extras["episode"] = _EpisodeExtras(self, c)
        return self.obs_buf, self.privileged_obs_buf, self.rew_buf, self.reset_buf, self.extras

    def _engine_step(self, actions, c):
        if self.cfg.commands.curriculum and (c % self.max_episode_length == 0):
            # the curriculum decision needs the episode sums of the envs that reset at this very
            # step, before they are zeroed (legged_robot.py:110-111, 336-348): split the launch
            self._engine.step(abi.PHASE_PRE | abi.PHASE_SIM | abi.PHASE_POST, actions, c)
            self._command_curriculum_gate()
            self._engine.step(abi.PHASE_RESET, None, c)
        else:
            self._engine.step(abi.PHASE_ALL, actions, c)

    def _command_curriculum_gate(self):
        env_ids = self.reset_buf.nonzero(as_tuple=False).flatten()
        self._update_command_curriculum(env_ids)      # collective across ranks: called even with no local reset
        self._on_curriculum_gate(env_ids)             # likewise (task curricula reduce over all ranks too)

    def reset(self):
        """base_task.py:60-64: reset every env, then one zero-action step."""
        self.reset_idx(torch.arange(self.num_envs, device=self.device))
        obs, priv, _, _, _ = self.step(torch.zeros(self.num_envs, self.num_actions, device=self.device))
        return obs, priv

    def reset_idx(self, env_ids):
        """legged_robot.py:94-148 for an explicit id list: flag the envs and run the RESET phase."""
        if len(env_ids) == 0:
            return
        self.reset_buf.fill_(False)
        self.reset_buf[env_ids] = True
        self._engine.step(abi.PHASE_RESET, None, self.common_step_counter)

    def get_observations(self):
        return self.obs_buf

    # ---- checkpoint / resume ---------------------------------------------------------------------------
    _HOST_STATE = ("common_step_counter", "command_ranges")

    def state_dict(self):
        """Everything needed to continue this env bit for bit: the engine's buffers (physics state, MDP state, episode sums,
        observation histories, DR parameters, terrain levels / origins), the step counter that keys the Philox draws, and the
        host-side curriculum state (command ranges; task classes add theirs through `_HOST_STATE`)."""
        import copy
        return {"engine": self._engine.state_dict(), "host": {k: copy.deepcopy(getattr(self, k)) for k in self._HOST_STATE},
                "task": type(self).__name__, "seed": int(self._engine.task.seed), "env_id_offset": int(self._engine.task.env_id_offset)}

    def load_state_dict(self, sd):
        import copy
        if sd["task"] != type(self).__name__:
            raise ValueError(f"checkpoint of task {sd['task']} loaded into {type(self).__name__}")
        if int(sd["seed"]) != int(self._engine.task.seed) or int(sd["env_id_offset"]) != int(self._engine.task.env_id_offset):
            raise ValueError("checkpoint seed / env shard differ from this env's: the random streams would not continue")
        self._engine.load_state_dict(sd["engine"])
        for k, v in sd["host"].items():
            setattr(self, k, copy.deepcopy(v))
        self._upload_command_ranges()          # also rebinds nothing: the device copy is part of the buffers, this keeps host and device equal

    def save_checkpoint(self, path):
        torch.save(self.state_dict(), path)

    def load_checkpoint(self, path):
        self.load_state_dict(torch.load(path, map_location=self.device, weights_only=True))

    def get_privileged_observations(self):
        return self.privileged_obs_buf

    # ------------------------------------------------------------------------------------------
    def _update_command_curriculum(self, env_ids):
        """legged_robot.py:336-348; the mean runs over the resetting envs of every rank (distributed.global_mean)."""
        from ..distributed import global_mean
        k = abi.REWARD_ID["tracking_lin_vel"]
        local = self._engine.buf["episode_sums"][k][env_ids].sum() if len(env_ids) > 0 else torch.zeros((), device=self.device)
        mean, count = global_mean(local, len(env_ids))
        if count > 0 and mean / self.max_episode_length > self.cfg.commands.curriculum_threshold * self.reward_scales["tracking_lin_vel"]:
            r = self.command_ranges["lin_vel_x"]
            r[0] = float(np.clip(r[0] - 0.5, -self.cfg.commands.max_curriculum, 0.))
            r[1] = float(np.clip(r[1] + 0.5, 0., self.cfg.commands.max_curriculum))
            self._upload_command_ranges()

    def _warm_rare_paths(self):
        """Touch, at construction, every torch op the command-curriculum gate uses (nonzero, gather, mean, compare,
        host->device upload).  On ROCm the first use of each op loads its code object (~120 ms in total, measured);
        left lazy that lands in the middle of a rollout, on the first gate step (1 step in 1000)."""
        if not self.cfg.commands.curriculum:
            return
        ids = self.reset_buf.nonzero(as_tuple=False).flatten()
        if len(ids) > 0:
            k = abi.REWARD_ID["tracking_lin_vel"]
            t = torch.zeros(2, dtype=torch.float64, device=self.device)     # same ops as distributed.global_mean, no collective
            t[0] = self._engine.buf["episode_sums"][k][ids].sum()
            t[1] = float(len(ids))
            float(t[0]) / max(float(t[1]), 1.0)
            float(torch.mean(self._engine.buf["episode_sums"][k][ids]))      # task curricula use means (go2_wtw.py:220-247)
        self._upload_command_ranges()

    def _on_curriculum_gate(self, env_ids):
        """Task hook on command-curriculum gate steps (go2_wtw.py:119-122)."""

    def _extra_ranges(self):
        return []

    def _upload_command_ranges(self):
        cr = self.command_ranges
        vals = list(cr["lin_vel_x"]) + list(cr["lin_vel_y"]) + list(cr["ang_vel_yaw"]) + list(cr["heading"])
        vals = vals + list(self._extra_ranges())
        vals = vals + [0.0] * (abi.CMD_RANGE_FLOATS - len(vals))
        self._engine.buf["command_ranges"].copy_(torch.tensor(vals, dtype=torch.float32))

    def _parse_cfg(self, cfg):
        """legged_robot.py:436-455."""
        self.dt = cfgmod.control_dt(cfg)
        self.debug = cfg.env.debug
        self.obs_scales = cfg.normalization.obs_scales
        self.reward_scales = cfgmod.class_to_dict(cfg.rewards.scales)
        self.command_ranges = cfgmod.class_to_dict(cfg.commands.ranges)
        if cfg.terrain.mesh_type not in ("heightfield", "trimesh"):
            cfg.terrain.curriculum = False
        self.max_episode_length_s = cfg.env.episode_length_s
        self.max_episode_length = np.ceil(self.max_episode_length_s / self.dt)
        cfg.domain_rand.push_interval = np.ceil(cfg.domain_rand.push_interval_s / self.dt)

    def _init_buffers(self):
        """legged_robot.py:380-409: aliases onto the engine's device buffers."""
        b = self._engine.buf
        self.common_step_counter = 0
        self.commands = b["commands"]
        self.actions, self.last_actions, self.llast_actions = b["actions"], b["last_actions"], b["llast_actions"]
        self.feet_air_time = b["feet_air_time"]
        self.last_contacts = b["last_contacts"].view(torch.bool)
        self.fail_buf = b["fail_buf"]
        self.commands_scale = torch.tensor([self.obs_scales.lin_vel, self.obs_scales.lin_vel, self.obs_scales.ang_vel],
                                           device=self.device)
        self.noise_scale_vec = torch.tensor(np.ctypeslib.as_array(self._engine.task.noise_vec)[:self._engine.task.obs_frame].copy(),
                                            device=self.device)
        self.add_noise = self.cfg.noise.add_noise
        self._upload_command_ranges()
        self.extras["time_outs"] = self.time_out_buf

    def _prepare_reward_function(self):
        """legged_robot.py:411-434: drop zero scales, multiply by dt, per-term episode sums."""
        for key in list(self.reward_scales.keys()):
            if self.reward_scales[key] == 0:
                self.reward_scales.pop(key)
            else:
                self.reward_scales[key] *= self.dt
        self.reward_names = [n for n in self.reward_scales if n != "termination"]
        es = self._engine.buf["episode_sums"]
        jpl = self.simulator._model.joints_per_leg      # four-joint legs: three slots carry the sole-foot terms (include/lgsim.h)
        self.episode_sums = {name: es[abi.reward_id(name, jpl)] for name in self.reward_scales}
