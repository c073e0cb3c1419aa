"""Developer micro-benchmark: fused step time of any task at 4096 envs (not the contract bench).  usage: python tools/quick_task_bench.py task [n]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hcr_genesis_lr_cl_amd.envs import make_env
for task in sys.argv[1:]:
    n = 4096
    env, cfg = make_env(task, n)
    env.reset()
    env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
    bank = [torch.randn(n, env.num_actions, device="cuda") for _ in range(8)]
    for i in range(100):
        env.step(bank[i % 8])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(600):
        env.step(bank[i % 8])
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 600 * 1e3
    print(f"{task:14s} {us:7.1f} us/step {n / us:7.1f} M env-steps/s", flush=True)
    del env
