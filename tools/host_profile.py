"""Developer tool: where the host spends its time per env.step() (cProfile over the bench loop)."""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hcr_genesis_lr_cl_amd.envs import make_env
n = 4096
env, cfg = make_env("go2", n)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
bank = [torch.randn(n, 12, generator=g, device="cuda") for _ in range(16)]
for i in range(200):
    env.step(bank[i % 16])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(3000):
    env.step(bank[i % 16])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e6*(t1-t0)/3000:.1f} us/step, drained after {1e6*(t2-t0)/3000:.1f} us/step")
pr = cProfile.Profile()
pr.enable()
for i in range(3000):
    env.step(bank[i % 16])
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
