"""Diagnostic: shader-cycle stamps inside quad_sim_kernel (build with -DLG_DBG_STAMPS, LG_LIB=..., physics-only launches)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hcr_genesis_lr_cl_amd import abi
from hcr_genesis_lr_cl_amd.envs import make_env
env, cfg = make_env("go2", 4096)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
env.episode_length_buf[:] = torch.randint(0, 1000, (4096,), generator=g, device="cuda", dtype=torch.int32)
bank = [torch.randn(4096, 12, generator=g, device="cuda") for _ in range(8)]
for i in range(600):
    env.step(bank[i % 8])
names = {12: "prologue+consts", 13: "kinematics", 14: "spheres", 15: "PD + pass 2", 16: "base + inverse", 17: "pass 3",
         18: "foot frame + W columns", 19: "sweeps", 20: "integrate", 21: "sub-steps 2-4", 22: "read-back + stores"}
acc = torch.zeros(32); n = 0
for i in range(100):
    env._engine.step(abi.PHASE_SIM, bank[i % 8], 0)
    torch.cuda.synchronize()
    acc += env._engine.buf["episode_done_sums"].flatten()[:32].cpu(); n += 1
acc /= n
prev = 0.0
for k in range(12, 23):
    print(f"{names[k]:26s} +{acc[k]-prev:8.0f} cycles (cum {acc[k]:8.0f})")
    prev = acc[k]
