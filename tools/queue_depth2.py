import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hcr_genesis_lr_cl_amd import abi
from hcr_genesis_lr_cl_amd.envs import make_env
n = 4096
env, cfg = make_env("go2", n)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
env.episode_length_buf[:] = torch.randint(0, 1000, (n,), generator=g, device="cuda", dtype=torch.int32)
bank = [torch.randn(n, 12, generator=g, device="cuda") for _ in range(8)]
for i in range(600):
    env.step(bank[i % 8])
torch.cuda.synchronize()
ts = [time.perf_counter()]
for i in range(3000):
    env.step(bank[i % 8])
    if (i + 1) % 100 == 0:
        ts.append(time.perf_counter())
torch.cuda.synchronize()
tend = time.perf_counter()
print("host us/step per 100-step block:", " ".join(f"{(b - a) * 1e4:.0f}" for a, b in zip(ts[:-1], ts[1:])))
print("total", (tend - ts[0]) / 3000 * 1e6, "us/step; counter", env.common_step_counter)
