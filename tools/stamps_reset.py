"""Diagnostic: MDP stamps of workgroup 0 when every env resets each step (forced time-outs)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hcr_genesis_lr_cl_amd.envs import make_env
task = sys.argv[1] if len(sys.argv) > 1 else "go2"
env, cfg = make_env(task, 4096)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
bank = [torch.randn(4096, env.num_actions, generator=g, device="cuda") for _ in range(8)]
names = ["start", "lds staged", "prologue loads", "lane consts", "sub-steps", "sim epilogue", "callback", "rewards", "post end", "reset blk", "obs", "end"]
acc = torch.zeros(12); n = 0
for i in range(300):
    if len(sys.argv) > 2: env.episode_length_buf[:] = 1001          # everything times out
    env.step(bank[i % 8])
    if i >= 100:
        torch.cuda.synchronize()
        acc += env._engine.buf["episode_done_sums"].flatten()[:12].cpu(); n += 1
acc /= n
prev = 0.0
for k in range(5, 12):
    print(f"{names[k]:16s} +{acc[k]-prev:9.0f} cycles   (cum {acc[k]:9.0f})")
    prev = acc[k]
