import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def go2():
    from hcr_genesis_lr_cl_amd.model_compiler import load_model
    from hcr_genesis_lr_cl_amd.config import GO2Cfg
    from hcr_genesis_lr_cl_amd import builders
    model, cfg = load_model("go2"), GO2Cfg()
    return dict(model=model, cfg=cfg, desc=builders.make_model_desc(model, cfg),
                opts=builders.make_sim_options(model, cfg))
