"""Diagnostic: per-section shader-cycle stamps of workgroup 0 (build with -DLG_DBG_STAMPS, run with LG_LIB=...)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hcr_genesis_lr_cl_amd.envs import make_env
env, cfg = make_env(sys.argv[1] if len(sys.argv) > 1 else "go2", 4096)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
env.episode_length_buf[:] = torch.randint(0, 1000, (4096,), generator=g, device="cuda", dtype=torch.int32)
bank = [torch.randn(4096, env.num_actions, generator=g, device="cuda") for _ in range(8)]
names = ["start", "lds staged", "prologue loads", "lane consts", "sub-steps", "sim epilogue", "callback", "rewards", "post end", "reset blk", "obs", "end"]
acc = torch.zeros(32)
n = 0
for i in range(700):
    env.step(bank[i % 8])
    if i >= 500:
        torch.cuda.synchronize()
        acc += env._engine.buf["episode_done_sums"].flatten()[:32].cpu()
        n += 1
acc /= n
prev = 0.0
for k in range(1, 12):
    print(f"{names[k]:16s} +{acc[k]-prev:9.0f} cycles   (cum {acc[k]:9.0f})")
    prev = acc[k]
