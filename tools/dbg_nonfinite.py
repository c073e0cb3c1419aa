"""Developer check of the non-finite guard: inject NaN / Inf into single envs and print what the step reports."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hcr_genesis_lr_cl_amd.envs import make_env
for task in sys.argv[1:] or ["go2"]:
    for what in ("nan_qd", "inf_z", "nan_quat", "inf_vw"):
        env, cfg = make_env(task, 64)
        env.reset()
        g = torch.Generator(device="cuda"); g.manual_seed(3)
        for _ in range(3):
            env.step(torch.randn(64, env.num_actions, generator=g, device="cuda"))
        eng = env._engine
        if what == "nan_qd": eng.buf["dof_vel"][5, 1] = float("nan")
        if what == "inf_z": eng.buf["base_pos"][5, 2] = float("inf")
        if what == "nan_quat": eng.buf["base_quat"][5, 0] = float("nan")
        if what == "inf_vw": eng.buf["base_lin_vel_w"][5, 0] = float("inf")
        out = env.step(torch.randn(64, env.num_actions, generator=g, device="cuda"))
        print(task, what, "count", eng.nonfinite_count(), "done", bool(out[-2][5]), "timeout", bool(out[-1]["time_outs"][5]),
              "fail_buf", int(eng.buf["fail_buf"][5]), "pos", eng.buf["base_pos"][5].tolist(), "n_done", int(out[-2].sum()))
