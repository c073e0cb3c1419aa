"""Go2EE task (reference legged_gym/envs/go2/go2_ee/go2_ee.py on legged_robot_ee.py; experiment name
"go2_rough"): explicit-estimator variant on heightfield terrain with the terrain curriculum.
`step` returns the reference's 6-tuple (legged_robot_ee.py:56-73):
    (estimator_features (N, 45*20), estimator_labels (N, 24), critic_obs (N, 174*5), rew, reset, extras)
all produced by the fused kernel (terrain sampling genesis_simulator.py:552-610, terrain curriculum
legged_robot.py:254-272, stacks go2_ee.py:10-75)."""
import torch

from .legged_robot import LeggedRobot


class LeggedRobotEE(LeggedRobot):
    def _parse_cfg(self, cfg):
        super()._parse_cfg(cfg)
        self.num_estimator_features = cfg.env.num_estimator_features
        self.num_estimator_labels = cfg.env.num_estimator_labels

    # the observation outputs alternate between two copies (engine._Buffers): look them up per access
    @property
    def estimator_labels_buf(self):
        return self._engine.buf["labels_buf"]

    @property
    def estimator_features_buf(self):
        return self._engine.buf["obs_buf"]

    def step(self, actions):
        _, priv, rew, done, extras = super().step(actions)
        return self.estimator_features_buf, self.estimator_labels_buf, priv, rew, done, extras

    def reset(self):
        self.reset_idx(torch.arange(self.num_envs, device=self.device))
        f, l, p, _, _, _ = self.step(torch.zeros(self.num_envs, self.num_actions, device=self.device))
        return f, l, p

    def get_observations(self):
        return self.estimator_features_buf, self.estimator_labels_buf, self.privileged_obs_buf


class Go2EE(LeggedRobotEE):
    pass
