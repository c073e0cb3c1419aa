"""Diagnostic: distribution over workgroups of (physics end, step end) - start, in shader-clock cycles (LG_DBG_STAMPS build)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hcr_genesis_lr_cl_amd.envs import make_env
env, cfg = make_env("go2", 4096)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
env.episode_length_buf[:] = torch.randint(0, 1000, (4096,), generator=g, device="cuda", dtype=torch.int32)
bank = [torch.randn(4096, 12, generator=g, device="cuda") for _ in range(8)]
for i in range(700):
    env.step(bank[i % 8])
for rep in range(3):
    env.step(bank[rep])
    torch.cuda.synchronize()
    d = env._engine.buf["episode_done_sums"].flatten()
    t0, t1, t2 = d[4096:4096 + 1024], d[8192:8192 + 1024], d[12288:12288 + 1024]
    w = lambda a, b: ((b - a) % (1 << 24))
    first = t0.min()
    for name, v in (("start - first start", w(first.expand(1024), t0)), ("physics", w(t0, t1)), ("mdp tail", w(t1, t2)), ("end - first start", w(first.expand(1024), t2))):
        q = torch.quantile(v, torch.tensor([0.0, 0.5, 0.9, 0.99, 1.0], device=v.device))
        print(f"{name:20s} min {q[0]:8.0f}  p50 {q[1]:8.0f}  p90 {q[2]:8.0f}  p99 {q[3]:8.0f}  max {q[4]:8.0f}")
    print("resets this step:", int(env.reset_buf.sum()))
