import sys, os, copy
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from hcr_genesis_lr_cl_amd import abi, builders
from hcr_genesis_lr_cl_amd.config import GO2Cfg
from hcr_genesis_lr_cl_amd.engine import Engine
from hcr_genesis_lr_cl_amd.model_compiler import load_model
from tests.util import random_sim_state, load_state_into_engine, engine_arrays
from tests.test_gpu_physics import SIM_OUT
from oracle import oracle as orc
model, cfg = load_model("go2"), GO2Cfg()
desc = builders.make_model_desc(model, cfg)
task = builders.make_task_cfg(model, cfg)
dec = int(sys.argv[1]) if len(sys.argv) > 1 else 1
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
air = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
res = {}
for lay in (1, 2):
    opts = builders.make_sim_options(model, cfg)
    opts.sim_layout = lay; opts.decimation = dec
    eng = Engine(model, desc, opts, task, 512, "cuda:0")
    st, actions = random_sim_state(model, cfg, 512, seed, air, 0.0)
    st0 = st.copy()
    load_state_into_engine(eng, st)
    eng.step(abi.PHASE_SIM, torch.from_numpy(actions).cuda(), 0)
    res[lay] = engine_arrays(eng, SIM_OUT)
orc.sim_step(desc, opts, st, actions, "f64", threads=8)
for k in ("base_quat", "base_ang_vel_w", "base_lin_vel_w", "dof_pos", "dof_vel", "link_contact_forces"):
    ref = st.arr[k].reshape(512, -1)
    out = []
    for lay in (1, 2):
        d = np.abs(res[lay][k] - ref).max(1)
        out.append(f"lay{lay}: max {d.max():.2e} p99 {np.percentile(d, 99):.2e} p50 {np.percentile(d, 50):.2e} argmax {d.argmax()}")
    print(f"{k:20s} " + " | ".join(out))
e = int(np.abs(res[2]["base_quat"] - st.arr["base_quat"]).max(1).argmax())
print("env", e, "ww0", st0.arr["base_ang_vel_w"][e], "quat0", st0.arr["base_quat"][e], "z", st0.arr["base_pos"][e])
print("quat ref", st.arr["base_quat"][e], "lay1", res[1]["base_quat"][e], "lay2", res[2]["base_quat"][e])
print("ww ref", st.arr["base_ang_vel_w"][e], "lay1", res[1]["base_ang_vel_w"][e], "lay2", res[2]["base_ang_vel_w"][e])
