import sys, os
sys.path.insert(0, os.getcwd())
import torch
from hcr_genesis_lr_cl_amd.envs import make_env
for task in ("go2", "go2_wtw", "go2_ee", "tron1_pf_ee"):
    env, cfg = make_env(task, 4096)
    env.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    env.episode_length_buf[:] = torch.randint(0, 1000, (4096,), generator=g, device="cuda", dtype=torch.int32)
    tot = 0
    for i in range(1200):
        out = env.step(torch.randn(4096, env.num_actions, generator=g, device="cuda"))
        if i >= 200: tot += int(out[-2].sum())
    print(task, "resets per step:", tot / 1000.0, "of 4096")
