"""TRON1PF_EE task (reference legged_gym/envs/tron1_pf/tron1_pf_ee/tron1_pf_ee.py, experiment
"tron1_pf_rough"): 6-DOF point-foot biped on heightfield terrain.  Per-step logic lives in the fused
kernel: biped periodic gait (:28-35, 347-433), sit-pose resets chosen by one coin per reset batch
(:204-210, 277-310), 10-frame estimator / critic stacks (:53-141), all domain randomisation incl.
per-env joint armature / friction / damping (genesis_simulator.py:704-733).

Deviations from the reference: `self.update_command_curriculum` (:201) and `self.simulator.dof_names`
(:171) do not exist there; here the curriculum works and `dof_names` is a property of
HipSimulator.  Env 0's index-flatten anomalies are not reproduced (see envs/go2_wtw.py)."""
import torch

from .go2_ee import LeggedRobotEE


class TRON1PF_EE(LeggedRobotEE):
    def _init_buffers(self):
        super()._init_buffers()
        ts = self._engine.buf["task_state"]                 # layout LG_TASK_STATE_BIPED (include/lgsim.h)
        self.gait_time, self.phi, self.gait_period = ts[:, 0:1], ts[:, 1:2], ts[:, 2:3]
        self.theta, self.clock_input, self.exp_C_frc = ts[:, 4:6], ts[:, 6:10], ts[:, 10:12]
        prf = self.cfg.rewards.periodic_reward_framework      # tron1_pf_ee.py:167-184
        self.theta[:, 0], self.theta[:, 1] = prf.theta_left, prf.theta_right
        self.gait_period[:] = prf.gait_period
        ini = self.cfg.init_state
        self.sit_pos = torch.tensor(ini.sit_pos, device=self.device)
        self.sit_joint_angles = torch.tensor([ini.sit_joint_angles[n] for n in self.simulator.dof_names], device=self.device)
