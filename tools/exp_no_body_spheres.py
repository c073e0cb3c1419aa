"""Experiment: upper bound of what the body (non-foot) collision spheres cost in the go2 step -- same bench with their radii set to
-1e30 (never inside the margin).  usage: python tools/exp_no_body_spheres.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hcr_genesis_lr_cl_amd import builders
import hcr_genesis_lr_cl_amd.builders as B
from hcr_genesis_lr_cl_amd.envs import make_env

orig = B.make_model_desc
for strip in (False, True):
    def patched(m, cfg, _strip=strip):
        d = orig(m, cfg)
        if _strip:
            feet = set(int(d.foot_sphere[i]) for i in range(d.n_legs))
            for s in range(d.n_spheres):
                if s not in feet:
                    d.sph_r[s] = -1e30
        return d
    B.make_model_desc = patched
    import hcr_genesis_lr_cl_amd.simulator as S
    S.builders.make_model_desc = patched
    n = 4096
    env, cfg = make_env("go2", n)
    env.reset()
    env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=1000)
    act = torch.randn(n, 12, device="cuda")
    for _ in range(100):
        env.step(act)
    torch.cuda.synchronize()
    ms = env._engine.time_steps(act, env.common_step_counter + 1, 400)
    print("body spheres", "stripped" if strip else "present", f"{ms*1e3:.2f} us/step")
    del env
