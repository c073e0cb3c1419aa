"""GO2WTW task (reference legged_gym/envs/go2/go2_wtw/go2_wtw.py): periodic-gait reward framework,
per-env behaviour parameters with their own curriculum, 5-frame actor / critic histories.

Everything per-step runs in the fused kernel (gait clock :29-36, behaviour resampling :180-218 and
:258-263, rewards :377-519, observation stacks :53-111, second action-history shift :45-46);
this class only keeps the host-side curriculum state (:220-247, 348-376) and exposes the task's
tensors under the reference's attribute names.

Deviations from the reference, by design:
  * go2_wtw.py:121 calls `self.update_command_curriculum` (no underscore), which does not exist and
    would raise at the first gate step with a reset; here the working
    `_update_command_curriculum` runs.
  * two index-flatten bugs make env 0 special in the reference (its gait clock restarts whenever any
    env's does, :33-34; its swing/stance indicator is overwritten, :455-462).  They couple env 0 to
    the whole batch and are not reproduced; the CPU oracle reproduces them to stay pinned to the
    reference's golden vectors, and the GPU parity test compares every env but env 0 on the
    affected quantities.
"""
import torch

from .. import abi
from .legged_robot import LeggedRobot


class GO2WTW(LeggedRobot):
    _HOST_STATE = LeggedRobot._HOST_STATE + ("gait_period_range", "foot_clearance_target_range", "base_height_target_range",
                                             "pitch_target_range", "num_gaits")

    def _parse_cfg(self, cfg):
        super()._parse_cfg(cfg)
        bp = cfg.rewards.behavior_params_range                      # go2_wtw.py:348-376
        mid = lambda r: [(r[0] + r[1]) / 2] * 2
        self.gait_period_min, self.gait_period_max = bp.gait_period_range
        self.gait_period_range = mid(bp.gait_period_range)
        self.foot_clearance_target_min, self.foot_clearance_target_max = bp.foot_clearance_target_range
        self.foot_clearance_target_range = [self.foot_clearance_target_min] * 2
        self.base_height_target_min, self.base_height_target_max = bp.base_height_target_range
        self.base_height_target_range = mid(bp.base_height_target_range)
        self.pitch_target_min, self.pitch_target_max = bp.pitch_target_range
        self.pitch_target_range = mid(bp.pitch_target_range)
        self.num_gaits = 1
        self.num_gait_max = len(cfg.rewards.periodic_reward_framework.theta_fl_list)

    def _extra_ranges(self):
        return (self.gait_period_range + self.base_height_target_range + self.foot_clearance_target_range
                + self.pitch_target_range + [float(self.num_gaits)])

    def _init_buffers(self):
        super()._init_buffers()
        ts = self._engine.buf["task_state"]                          # (N, 22), layout in include/lgsim.h
        self.gait_time, self.phi, self.gait_period = ts[:, 0:1], ts[:, 1:2], ts[:, 2:3]
        self.base_height_target, self.foot_clearance_target, self.pitch_target = ts[:, 3:4], ts[:, 4:5], ts[:, 5:6]
        self.theta, self.clock_input, self.exp_C_frc = ts[:, 6:10], ts[:, 10:18], ts[:, 18:22]
        prf = self.cfg.rewards.periodic_reward_framework              # go2_wtw.py:319-346
        self.theta[:] = torch.tensor([prf.theta_fl_list[0], prf.theta_fr_list[0], prf.theta_rl_list[0], prf.theta_rr_list[0]],
                                     device=self.device)
        self.gait_period[:] = self.gait_period_range[0]
        self.base_height_target[:] = self.base_height_target_range[0]
        self.foot_clearance_target[:] = self.foot_clearance_target_range[0]
        self.pitch_target[:] = self.pitch_target_range[0]

    def _on_curriculum_gate(self, env_ids):
        self._update_behavior_param_curriculum(env_ids)

    def _update_behavior_param_curriculum(self, env_ids):
        """go2_wtw.py:220-247.  The means run over the resetting envs of EVERY rank (distributed.global_mean, called on all
        ranks at the gate step, also with no local reset), so the behaviour ranges and the gait set advance identically on
        all shards, as in a single-process run."""
        from ..distributed import global_mean
        es, L = self._engine.buf["episode_sums"], self.max_episode_length
        n = len(env_ids)
        stats = {}
        for name in ("quad_periodic_gait", "tracking_base_height", "tracking_foot_clearance", "tracking_orientation"):
            local = es[abi.REWARD_ID[name]][env_ids].sum() if n > 0 else torch.zeros((), device=self.device)
            stats[name], count = global_mean(local, n)
        if count == 0:
            return
        m = lambda name: stats[name] / L
        if m("quad_periodic_gait") > 0.8 * self.reward_scales["quad_periodic_gait"]:
            self.gait_period_range[0] = max(self.gait_period_range[0] - 0.05, self.gait_period_min)
            self.gait_period_range[1] = min(self.gait_period_range[1] + 0.05, self.gait_period_max)
            self.num_gaits = min(self.num_gaits + 1, self.num_gait_max)
        if m("tracking_base_height") > 0.9 * self.reward_scales["tracking_base_height"]:
            self.base_height_target_range[0] = max(self.base_height_target_range[0] - 0.02, self.base_height_target_min)
            self.base_height_target_range[1] = min(self.base_height_target_range[1] + 0.02, self.base_height_target_max)
        if m("tracking_foot_clearance") > 0.8 * self.reward_scales["tracking_foot_clearance"]:
            self.foot_clearance_target_range[0] = max(self.foot_clearance_target_range[0] - 0.01, self.foot_clearance_target_min)
            self.foot_clearance_target_range[1] = min(self.foot_clearance_target_range[1] + 0.01, self.foot_clearance_target_max)
        if m("tracking_orientation") > 0.9 * self.reward_scales["tracking_orientation"]:
            self.pitch_target_range[0] = max(self.pitch_target_range[0] - 0.05, self.pitch_target_min)
            self.pitch_target_range[1] = min(self.pitch_target_range[1] + 0.05, self.pitch_target_max)
        self._upload_command_ranges()
