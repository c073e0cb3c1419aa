"""TRON1PF task (reference legged_gym/envs/tron1_pf/tron1_pf.py, experiment "tron1_pf"): the 6-DOF point-foot biped on the plane.
Plain 5-tuple VecEnv surface; obs = 5 x 27 actor stack, privileged obs = 5 x 45 critic stack (tron1_pf.py:15-66), both produced by
the fused kernel (critic frame = observation program, config.py TRON1PFCfg).  Rewards add `no_fly` (tron1_pf.py:151-154) to the
biped set."""
from .legged_robot import LeggedRobot


class TRON1PF(LeggedRobot):
    pass
