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


@pytest.mark.gpu
@pytest.mark.parametrize("name,curriculum,cfg_name", [("go2_ee", True, "GO2EECfg"), ("go2_ee_random", False, "GO2EECfg"), ("tron1_pf_ee", True, "TRON1PFEECfg")])
def test_device_generated_heightfield_matches_reference(name, curriculum, cfg_name):
    """the same grid from the init-time kernel (include/lgsim.h lg_terrain_generate) -- slopes in the reference's float64
    order, stairs and obstacles in integers, the up-sampled random-uniform tiles through FITPACK's own degree-1 recurrences on the
    host-fitted knots / coefficients; numpy draws taken on the host in the reference's call order.  Bit for bit, origins included."""
    import torch
    from hcr_genesis_lr_cl_amd import config as cfgmod
    g = np.load(os.path.join(G, f"terrain_{name}.npz"))
    cfg = getattr(cfgmod, cfg_name)()
    cfg.terrain.curriculum = curriculum
    np.random.seed(int(g["seed"]))
    t = Terrain(cfg.terrain, device="cuda:0")
    assert t.heightsamples_dev is not None and t.heightsamples_dev.dtype == torch.int16 and t.heightsamples_dev.is_cuda
    np.testing.assert_array_equal(t.heightsamples_dev.cpu().numpy(), g["height_field_raw"])
    np.testing.assert_allclose(t.env_origins, g["env_origins"], atol=0)
    # and the host generator draws the same numbers in the same order: the two paths leave numpy's generator in the same state
    after_dev = np.random.random()
    np.random.seed(int(g["seed"]))
    Terrain(cfg.terrain)
    assert np.random.random() == after_dev
