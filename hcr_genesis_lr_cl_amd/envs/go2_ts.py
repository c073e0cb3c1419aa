"""The other Go2-rough task heads (reference legged_gym/envs/__init__.py:82-86): Go2TS, Go2CTS, Go2Dreamwaq, Go2CaT.
Same robot, heightfield terrain, rewards, resets and domain randomisation as Go2EE; they differ in how one step's outputs
are packaged, and Go2CaT adds constraints-as-terminations.  All per-step work runs in the HIP kernel: the critic frame and
the auxiliary output are "observation programs" (include/lgsim.h LgObsSeg) assembled by config.py / builders.py.

Return arities follow the reference:
  * Go2TS / Go2CTS / Go2CaT  (legged_robot_ts.py:59-76): (obs, privileged_obs, obs_history, critic_obs, rew, reset, extras)
  * Go2Dreamwaq (legged_robot_dreamwaq.py:63-79): (obs, privileged_obs, obs_history, explicit_labels, next_state, rew, reset, extras)

Buffers: obs_history is the 20-frame actor stack (engine "obs_buf"); obs is its newest frame (the reference clips obs but
not the history, legged_robot_ts.py:71 -- with clip_observations = 100 and actions clipped to +-100 / +-10 that clip never
binds, so one buffer serves both); the critic stack is engine "priv_obs_buf"; the single-frame auxiliary output (privileged
encoder input / explicit labels + next state) is engine "labels_buf".

Documented deviation: the reference's size fields were written for 12 contact-state links while the asset config selects
17 (config.py); its critic deque therefore starts with 172-wide zero frames that are replaced by 177-wide real ones during
the first five steps.  Here every frame is 177 wide from the start (what the reference emits from step 5 on)."""
import torch

from .. import abi
from .legged_robot import LeggedRobot


class LeggedRobotTS(LeggedRobot):
    def _parse_cfg(self, cfg):
        super()._parse_cfg(cfg)
        self.num_history_obs = cfg.env.num_history_obs        # legged_robot_ts.py:84-88
        self.num_latent_dims = cfg.env.num_latent_dims
        self.num_critic_obs = getattr(cfg.env, "num_critic_obs", None)

    # the observation outputs alternate between two copies (engine._Buffers): look them up per access
    @property
    def obs_history(self):
        return self._engine.buf["obs_buf"]

    @property
    def obs_buf(self):
        h = self._engine.buf["obs_buf"]
        return h[:, h.shape[1] - int(self.cfg.env.num_observations):]

    @property
    def critic_obs_buf(self):
        return self._engine.buf["priv_obs_buf"]

    @property
    def privileged_obs_buf(self):
        return self._engine.buf["labels_buf"]

    def step(self, actions):
        _, _, rew, done, extras = super().step(actions)
        return self.obs_buf, self.privileged_obs_buf, self.obs_history, self.critic_obs_buf, rew, done, extras

    def reset(self):
        self.reset_idx(torch.arange(self.num_envs, device=self.device))
        return self.step(torch.zeros(self.num_envs, self.num_actions, device=self.device))[:4]

    def get_observations(self):
        return self.obs_buf, self.privileged_obs_buf, self.obs_history, self.critic_obs_buf


class Go2TS(LeggedRobotTS):
    pass


class Go2CTS(LeggedRobotTS):
    def _parse_cfg(self, cfg):
        super()._parse_cfg(cfg)
        self.num_teacher = cfg.env.num_teacher                # legged_robot_cts.py:5-6

    def step(self, actions):
        out = super().step(actions)
        if self.cfg.terrain.curriculum:                       # go2_cts.py:89-95
            lv = self.simulator.terrain_levels.float()
            ep = out[-1]["episode"]
            ep["teacher_terrain_level"] = torch.mean(lv[:self.num_teacher])
            ep["student_terrain_level"] = torch.mean(lv[self.num_teacher:])
        return out


class Go2Dreamwaq(LeggedRobot):
    def _parse_cfg(self, cfg):
        super()._parse_cfg(cfg)
        self.num_history_obs = cfg.env.num_history_obs        # legged_robot_dreamwaq.py:91-96
        self.num_latent_dims = cfg.env.num_latent_dims
        self.num_explicit_dims = cfg.env.num_explicit_dims
        self.num_decoder_output = cfg.env.num_decoder_output

    @property
    def obs_history(self):
        return self._engine.buf["obs_buf"]

    @property
    def obs_buf(self):
        h = self._engine.buf["obs_buf"]
        return h[:, h.shape[1] - int(self.cfg.env.num_observations):]

    @property
    def privileged_obs_buf(self):
        return self._engine.buf["priv_obs_buf"]

    @property
    def explicit_labels_buf(self):
        return self._engine.buf["labels_buf"][:, :self.num_explicit_dims]

    @property
    def next_state_buf(self):
        return self._engine.buf["labels_buf"][:, self.num_explicit_dims:self.num_explicit_dims + self.num_decoder_output]

    def step(self, actions):
        _, _, rew, done, extras = super().step(actions)
        return self.obs_buf, self.privileged_obs_buf, self.obs_history, self.explicit_labels_buf, self.next_state_buf, rew, done, extras

    def reset(self):
        self.reset_idx(torch.arange(self.num_envs, device=self.device))
        return self.step(torch.zeros(self.num_envs, self.num_actions, device=self.device))[:5]

    def get_observations(self):
        return self.obs_buf, self.privileged_obs_buf, self.obs_history, self.explicit_labels_buf, self.next_state_buf


class Go2CaT(LeggedRobotTS):
    """go2_cat.py: the TS packaging plus constraints as terminations.  The nine violation flags, the termination probability,
    the (1 - p) scaling of the reward and the per-episode violation counters are computed in the kernel's POST phase
    (LgTaskCfg.cat_enable).  One input is job-wide: the reference's style constraint multiplies a (N,) flag with a (N,1) flag
    (go2_cat.py:179-180), which makes an env violate it when its own command is ~0 and ANY env moves a joint faster than
    4 rad/s.  That flag has to exist before the rewards are computed, so the step is two launches here -- physics, which raises
    the flag in `command_ranges`, then the MDP phases -- with a one-float max-reduction over the ranks of a sharded job in between;
    no host sync."""

    def _prepare_reward_function(self):
        super()._prepare_reward_function()
        b = self._engine.buf
        self.cstr_prob = b["cstr_prob"]
        for k, n in enumerate(abi.CSTR_NAMES):                     # constraint_manager.py:91-97 logs them into episode_sums
            self.episode_sums["cstr_" + n] = b["cstr_sums"][k]
        # go2_cat.py:98-101 puts mean(cstr_prob) into extras["episode"] every step; here it is formed when that entry is read (the
        # value of the latest step: read it before stepping again, as the runner does) -- a reduction launch per step otherwise
        self._lazy_episode_extras = {"cstr_probs": lambda: torch.mean(self.cstr_prob)}

    def _engine_step(self, actions, c):
        e = self._engine
        e.step(abi.PHASE_PRE | abi.PHASE_SIM, actions, c)         # raises command_ranges[CR_ANY_FAST + (c & 1)] in-kernel
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            k = abi.CR_ANY_FAST + (int(c) & 1)
            dist.all_reduce(e.buf["command_ranges"][k:k + 1], op=dist.ReduceOp.MAX)
        if self.cfg.commands.curriculum and (c % self.max_episode_length == 0):
            e.step(abi.PHASE_POST, None, c)
            self._command_curriculum_gate()
            e.step(abi.PHASE_RESET, None, c)
        else:
            e.step(abi.PHASE_POST | abi.PHASE_RESET, None, c)

