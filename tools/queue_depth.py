"""Developer experiment: per-step time of the fused step as a function of how many launches the host enqueues back to back."""
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
eng = env._engine
c = 1000
def run(reps, ev_every=0, use_env=False):
    global c
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    evs = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for i in range(reps):
        if use_env: env.step(bank[i % 8])
        else:
            eng.step(abi.PHASE_ALL, bank[i % 8], c); c += 1
        if ev_every and i % ev_every == 0:
            ev = torch.cuda.Event(); ev.record(); evs.append(ev)
    t1 = time.perf_counter()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, (t1 - t0) / reps * 1e6
for reps in (200, 500, 1000, 2000, 4000):
    d, h = run(reps)
    print(f"reps {reps:5d}: device {d:6.1f} us/step, host enqueue {h:6.1f} us/step", flush=True)
for ev in (1, 8, 64):
    d, h = run(2000, ev)
    print(f"reps  2000 + event every {ev:2d}: device {d:6.1f} us/step, host {h:6.1f}", flush=True)
d, h = run(2000, 0, True)
print(f"env.step x2000: device {d:6.1f} us/step, host {h:6.1f}")
print("counter now", env.common_step_counter)
for seg in range(12):
    d, h = run(250, 0, True)
    print(f"  env.step seg {seg:2d} (counter -> {env.common_step_counter}): device {d:6.1f} us/step, host {h:6.1f}", flush=True)
