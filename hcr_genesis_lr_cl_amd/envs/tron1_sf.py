"""TRON1SF task (reference legged_gym/envs/tron1_sf/tron1_sf.py, experiment "tron1_sf"): the 8-DOF sole-foot biped on the plane --
four joints per leg (abad, hip, knee, ankle), the ankle body with its box sole is the foot.  Plain 5-tuple VecEnv surface;
obs = 10 x 33 actor stack, privileged obs = 10 x 72 critic stack (tron1_sf.py:15-81), both produced by the kernel (critic frame =
observation program, config.py TRON1SFCfg).  Per-step logic in the kernel: the 6-DOF index pairs of `_reset_dofs` (:213-233), sit-pose
resets chosen by one coin per reset batch (:160-166, 235-254), air time gated by |commands[:, :3]| (:256-267), `no_fly` above 1 N
(:275-278), `hip_pos_zero_command`, `keep_ankle_pitch_zero_in_air`, `foot_flat` (:281-308; the foot orientation follows from the base
orientation and the leg's joint angles).  Physics: leg-per-lane kernel with four-joint chains; the sole is the exact foot contact at
its centre plus four corner points."""
import torch

from .legged_robot import LeggedRobot


class TRON1SF(LeggedRobot):
    def _init_buffers(self):
        super()._init_buffers()
        ini = self.cfg.init_state                              # tron1_sf.py:115-122
        self.sit_pos = torch.tensor(ini.sit_pos, device=self.device)
        self.sit_joint_angles = torch.tensor([ini.sit_joint_angles[n] for n in self.simulator.dof_names], device=self.device)
