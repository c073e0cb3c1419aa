"""Steady-state fused-step workload for rocprofv3 (kernel trace / PMC): warm-up, then a fixed number of measured launches.
usage: python tools/prof_steady.py [n_envs] [warm] [measured] [task]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hcr_genesis_lr_cl_amd.envs import make_env

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 400
meas = int(sys.argv[3]) if len(sys.argv) > 3 else 100
task = sys.argv[4] if len(sys.argv) > 4 else "go2"
env, cfg = make_env(task, n)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
env.episode_length_buf = torch.randint(0, 1000, (n,), generator=g, device="cuda", dtype=torch.int32)
bank = [torch.randn(n, env.num_actions, generator=g, device="cuda") for _ in range(16)]
for i in range(warm + meas):
    env.step(bank[i % 16])
torch.cuda.synchronize()
print("done", float(env.rew_buf.mean()))
