import sys, os, copy
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from hcr_genesis_lr_cl_amd import abi, builders
from hcr_genesis_lr_cl_amd.config import GO2Cfg
from hcr_genesis_lr_cl_amd.engine import Engine
from hcr_genesis_lr_cl_amd.model_compiler import load_model
from tests.util import random_sim_state, load_state_into_engine, engine_arrays
from tests.test_gpu_physics import SIM_OUT
model, cfg = load_model("go2"), GO2Cfg()
desc = builders.make_model_desc(model, cfg)
task = builders.make_task_cfg(model, cfg)
dec = int(sys.argv[1]) if len(sys.argv) > 1 else 1
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
res = {}
for lay in (1, 2):
    opts = builders.make_sim_options(model, cfg)
    opts.sim_layout = lay; opts.decimation = dec
    eng = Engine(model, desc, opts, task, 512, "cuda:0")
    st, actions = random_sim_state(model, cfg, 512, seed, 0.3, 0.0)
    load_state_into_engine(eng, st)
    eng.step(abi.PHASE_SIM, torch.from_numpy(actions).cuda(), 0)
    res[lay] = engine_arrays(eng, SIM_OUT)
for k in SIM_OUT:
    a, b = res[1][k], res[2][k]
    d = np.abs(a - b).reshape(512, -1)
    worst = d.max(1).argmax()
    print(f"{k:22s} max {d.max():.3e} env {worst} nbad(>1e-4) {(d.max(1) > 1e-4).sum()}")
f1, f2 = res[1]["link_contact_forces"].reshape(512, 17, 3), res[2]["link_contact_forces"].reshape(512, 17, 3)
d = np.abs(f1 - f2).max(2)
print("per-link max force diff:", np.round(d.max(0), 3))
e = d.max(1).argmax()
print("env", e); print(np.round(f1[e], 2)); print(np.round(f2[e], 2))
