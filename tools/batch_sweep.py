"""Developer micro-benchmark: step time of a task over batch sizes and physics layouts (LG_SIM_LAYOUT 1 = leg-per-lane, 2 = component-per-lane).
usage: python tools/batch_sweep.py task n1 n2 ..."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hcr_genesis_lr_cl_amd.envs import make_env
task = sys.argv[1]
for n in [int(x) for x in sys.argv[2:]]:
    row = []
    for lay in ("1", "2"):
        os.environ["LG_SIM_LAYOUT"] = lay
        env, cfg = make_env(task, n)
        env.reset()
        env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
        bank = [torch.randn(n, env.num_actions, device="cuda") for _ in range(8)]
        for i in range(60):
            env.step(bank[i % 8])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(300):
            env.step(bank[i % 8])
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 300 * 1e3
        row.append(f"layout {lay}: {us:7.1f} us {n / us:6.1f} M/s")
        del env
    print(f"{task} n={n:7d}  " + "   ".join(row), flush=True)
