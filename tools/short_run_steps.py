"""Diagnostic: device time of each of the first steps after a synchronize (why a 20-step timed region reads slower than a
2000-step one)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hcr_genesis_lr_cl_amd.envs import make_env
env, cfg = make_env("go2", 4096)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
env.episode_length_buf[:] = torch.randint(0, 1000, (4096,), generator=g, device="cuda", dtype=torch.int32)
bank = [torch.randn(4096, 12, generator=g, device="cuda") for _ in range(16)]
for i in range(4000):
    env.step(bank[i % 16])
torch.cuda.synchronize()
for trial in range(4):
    K = 24
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(K):
        env.step(bank[i % 16])
        ev[i + 1].record()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tt = time.perf_counter() - t0
    print("trial", trial, "host issue %.1f us/step, wall %.1f us/step;" % (th / K * 1e6, tt / K * 1e6),
          " ".join("%.1f" % (ev[i].elapsed_time(ev[i + 1]) * 1e3) for i in range(K)))
# without per-step events
for trial in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        env.step(bank[i % 16])
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tt = time.perf_counter() - t0
    print("plain 20 steps: host issue %.1f us/step, wall %.1f us/step" % (th / 20 * 1e6, tt / 20 * 1e6))
