"""Diagnostic: host-side time of every env.step() call in a bench-like short run (1 + 5 steps, then regions of 20)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hcr_genesis_lr_cl_amd.envs import make_env
env, cfg = make_env("go2", 4096)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
env.episode_length_buf[:] = torch.randint(0, 1000, (4096,), generator=g, device="cuda", dtype=torch.int32)
bank = [torch.randn(4096, 12, generator=g, device="cuda") for _ in range(16)]
for i in range(6):
    env.step(bank[i])
STRIDE = int(os.environ.get("STRIDE", "0"))
env._engine.profile(STRIDE)
for region in range(4):
    torch.cuda.synchronize()
    ts = [time.perf_counter()]
    for i in range(20):
        env.step(bank[i % 16])
        ts.append(time.perf_counter())
    torch.cuda.synchronize()
    te = time.perf_counter()
    print("region", region, "wall %.1f us/step | host per call:" % ((te - ts[0]) / 20 * 1e6),
          " ".join("%.0f" % ((ts[i + 1] - ts[i]) * 1e6) for i in range(20)), "| final sync %.0f us" % ((te - ts[-1]) * 1e6))
