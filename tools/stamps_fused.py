"""Diagnostic: cycle stamps of workgroup 0 through the fused step kernel (quad physics stamps 12..22, MDP tail 5..11)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hcr_genesis_lr_cl_amd.envs import make_env
NE = int(os.environ.get("NE", "4096"))
TASK = os.environ.get("TASK", "go2")
env, cfg = make_env(TASK, NE)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
env.episode_length_buf[:] = torch.randint(0, 1000, (NE,), generator=g, device="cuda", dtype=torch.int32)
bank = [torch.randn(NE, env.num_actions, generator=g, device="cuda") for _ in range(8)]
acc = torch.zeros(32); n = 0
for i in range(800):
    env.step(bank[i % 8])
    if i >= 600:
        torch.cuda.synchronize()
        acc += env._engine.buf["episode_done_sums"].flatten()[:32].cpu(); n += 1
acc /= n
order = [(23, "quad: load burst + LDS staging"), (24, "quad: lane constants"), (12, "quad: rest of prologue"), (21, "4 sub-steps (to read-back)"), (22, "read-back + stores"),
         (5, "mdp: start"), (6, "mdp: callback"), (7, "mdp: termination + rewards"), (9, "mdp: reset blk"), (25, "mdp: obs pointers + blanking"), (26, "mdp: obs noise draws"), (27, "mdp: obs actor frame"), (10, "mdp: obs task blocks"), (11, "mdp: state stores")]
prev = 0.0
for k, name in order:
    print(f"{name:30s} +{acc[k]-prev:8.0f} cycles (cum {acc[k]:8.0f})")
    prev = acc[k]
