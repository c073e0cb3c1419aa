import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, copy, sys
from oracle import oracle as orc
from hcr_genesis_lr_cl_amd.model_compiler import load_model
from hcr_genesis_lr_cl_amd.config import TRON1SFCfg
from hcr_genesis_lr_cl_amd import builders, config as cfgmod
model, cfg = load_model("tron1_sf"), TRON1SFCfg()
desc = builders.make_model_desc(model, cfg); opts = builders.make_sim_options(model, cfg)
arm = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
for k in range(8): desc.armature[k] = arm
st = orc.HostState(model, 1, cfgmod.default_dof_pos(cfg), 0.84)
st.arr["added_base_mass"][:] = 0
act = np.zeros((1, 8), np.float32)
L = model.n_links
for k in range(300):
    orc.sim_step(desc, opts, st, act, "f64")
    if k % 25 == 0 or k > 296:
        f = st.arr["link_contact_forces"][0].reshape(L, 3)
        print(k, "z", round(float(st.arr["base_pos"][0,2]),4), "feet z", np.round(st.arr["feet_pos"][0].reshape(2,3)[:,2],4), "Fz", np.round(f[[0,3,4,7,8],2],1), "q", np.round(st.arr["dof_pos"][0][:4],3), "qd", np.round(st.arr["dof_vel"][0][:4],2), "pg", np.round(st.arr["projected_gravity"][0],2))
