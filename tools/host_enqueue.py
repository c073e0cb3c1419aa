"""Developer tool: host time to ENQUEUE a short burst of env.step() calls on an idle queue (no back-pressure), and where it goes."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hcr_genesis_lr_cl_amd.envs import make_env
n = 4096
env, cfg = make_env(sys.argv[1] if len(sys.argv) > 1 else "go2", n)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
bank = [torch.randn(n, env.num_actions, generator=g, device="cuda") for _ in range(16)]
for i in range(200):
    env.step(bank[i % 16])
res = []
for rep in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        env.step(bank[i % 16])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    res.append(((t1 - t0) / 20 * 1e6, (t2 - t0) / 20 * 1e6))
res.sort()
print("enqueue us/step (median, min): %.1f %.1f   wall us/step (median): %.1f" % (res[len(res) // 2][0], res[0][0], sorted(r[1] for r in res)[len(res) // 2]))
# raw lg_step cost
eng = env._engine
from hcr_genesis_lr_cl_amd import abi
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    eng.step(abi.PHASE_ALL, bank[i % 16], 1000 + i)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("engine.step enqueue us/step: %.1f" % ((t1 - t0) / 20 * 1e6))
st = torch.cuda.current_stream().cuda_stream
a = bank[0].data_ptr()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    eng.lib.lg_step(eng.handle, abi.PHASE_ALL, a, 2000 + i, st)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("bare lg_step enqueue us/step: %.1f" % ((t1 - t0) / 20 * 1e6))
