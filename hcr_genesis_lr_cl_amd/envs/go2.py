"""GO2 task (reference legged_gym/envs/go2/go2.py): 45-wide observation (go2.py:40-56), per-joint
reset ranges (go2.py:17-37), zero reset twist (go2.py:131-133).  All of it is data in GO2Cfg
(`reset` section + obs layout id) consumed by the fused kernel; the class exists so that
`GO2(cfg, sim_params, device, headless)` constructs exactly like the reference task."""
from .legged_robot import LeggedRobot


class GO2(LeggedRobot):
    pass
