"""Golden heightfields from the reference's own Terrain class (build container only).
Stores the int16 grids (compressed) + env origins for the two rough-terrain configs in scope."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

rh.load_reference()
import legged_gym.envs  # noqa: E402,F401  (resolves the reference's import cycle first)
from legged_gym.utils.terrain import Terrain  # noqa: E402
from legged_gym.envs.go2.go2_ee.go2_ee_config import Go2EECfg  # noqa: E402
from legged_gym.envs.tron1_pf.tron1_pf_ee.tron1_pf_ee_config import TRON1PF_EECfg  # noqa: E402

for name, cfg_cls, curriculum in (("go2_ee", Go2EECfg, True), ("tron1_pf_ee", TRON1PF_EECfg, True), ("go2_ee_random", Go2EECfg, False)):
    cfg = cfg_cls()
    cfg.terrain.curriculum = curriculum
    np.random.seed(3)
    t = Terrain(cfg.terrain)
    np.savez_compressed(os.path.join(HERE, f"terrain_{name}.npz"), height_field_raw=t.height_field_raw,
                        env_origins=t.env_origins, seed=3, curriculum=curriculum)
    print(name, t.height_field_raw.shape, t.height_field_raw.min(), t.height_field_raw.max(),
          os.path.getsize(os.path.join(HERE, f"terrain_{name}.npz")))
