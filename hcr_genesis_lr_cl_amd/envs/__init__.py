"""Task registry mirroring legged_gym/envs/__init__.py:80-91 for the tasks in scope."""
from .legged_robot import LeggedRobot
from .go2 import GO2
from .go2_wtw import GO2WTW
from .go2_ee import Go2EE
from .tron1_pf_ee import TRON1PF_EE
from .tron1_pf import TRON1PF
from .tron1_sf import TRON1SF
from .go2_ts import Go2TS, Go2CTS, Go2Dreamwaq, Go2CaT
from ..config import GO2Cfg, GO2WTWCfg, GO2EECfg, TRON1PFEECfg, GO2TSCfg, GO2CTSCfg, GO2DreamwaqCfg, GO2CaTCfg, TRON1PFCfg, TRON1SFCfg

# registry names of the reference (legged_gym/envs/__init__.py:80-91); go2_ee is the "go2_rough" experiment
TASKS = {"go2": (GO2, GO2Cfg), "go2_wtw": (GO2WTW, GO2WTWCfg), "go2_ee": (Go2EE, GO2EECfg),
         "tron1_pf_ee": (TRON1PF_EE, TRON1PFEECfg), "go2_ts": (Go2TS, GO2TSCfg), "go2_cts": (Go2CTS, GO2CTSCfg),
         "go2_dreamwaq": (Go2Dreamwaq, GO2DreamwaqCfg), "go2_cat": (Go2CaT, GO2CaTCfg), "tron1_pf": (TRON1PF, TRON1PFCfg), "tron1_sf": (TRON1SF, TRON1SFCfg)}


def set_seed(seed):
    """helpers.py:36-46 (the heightfield generator draws from np.random, like the reference's)."""
    import os
    import random
    import numpy as np
    import torch
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)


def make_env(name, num_envs=None, device="cuda:0", **kw):
    """task_registry.make_env equivalent (task_registry.py:35-72) for this backend, including its set_seed(cfg.seed)."""
    cls, cfg_cls = TASKS[name]
    cfg = cfg_cls()
    if num_envs is not None:
        cfg.env.num_envs = int(num_envs)
    set_seed(int(cfg.seed))
    return cls(cfg, None, device, True, **kw), cfg
