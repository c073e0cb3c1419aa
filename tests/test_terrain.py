"""The heightfield generator reproduces the reference's Terrain class sample for sample
(golden grids from tests/golden/gen_terrain_fixtures.py, same np.random seed)."""
import os

import numpy as np
import pytest

from hcr_genesis_lr_cl_amd.config import GO2EECfg
from hcr_genesis_lr_cl_amd.terrain import Terrain

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name,curriculum", [("go2_ee", True), ("go2_ee_random", False)])
def test_go2_rough_heightfield_matches_reference(name, curriculum):
    g = np.load(os.path.join(G, f"terrain_{name}.npz"))
    cfg = GO2EECfg()
    cfg.terrain.curriculum = curriculum
    np.random.seed(int(g["seed"]))
    t = Terrain(cfg.terrain)
    assert t.height_field_raw.dtype == np.int16 and t.height_field_raw.shape == (1200, 1200)
    np.testing.assert_array_equal(t.height_field_raw, g["height_field_raw"])
    np.testing.assert_allclose(t.env_origins, g["env_origins"], atol=0)
    # all five tile families are present on the curriculum map (columns = types)
    assert t.height_field_raw.min() < -100 and t.height_field_raw.max() > 100


def test_bad_terrain_types_raise_like_the_reference():
    cfg = GO2EECfg()
    cfg.terrain.mesh_type = "trimesh"
    with pytest.raises(NotImplementedError):      # genesis_simulator.py:271
        Terrain(cfg.terrain)
