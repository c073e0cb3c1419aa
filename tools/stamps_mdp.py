"""Diagnostic: cycle stamps of workgroup 0 / lane 0 through the MDP launch (env_step_kernel<.., POST|RESET>) of a two-launch task.
usage: LG_LIB=dbg/lg_STAMPS.so python tools/stamps_mdp.py <task> [n_envs]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hcr_genesis_lr_cl_amd.envs import make_env
task = sys.argv[1] if len(sys.argv) > 1 else "tron1_pf_ee"
NE = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
env, cfg = make_env(task, NE)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
env.episode_length_buf[:] = torch.randint(0, 1000, (NE,), generator=g, device="cuda", dtype=torch.int32)
bank = [torch.randn(NE, env.num_actions, generator=g, device="cuda") for _ in range(8)]
acc = torch.zeros(32); n = 0
for i in range(600):
    env.step(bank[i % 8])
    if i >= 400:
        torch.cuda.synchronize()
        acc += env._engine.buf["episode_done_sums"].flatten()[:32].cpu(); n += 1
acc /= n
order = [(2, "staging + barrier"), (5, "read-back loads"), (6, "callback"), (7, "termination + rewards"), (8, "after rewards"), (9, "reset block"), (25, "obs: pointers + blanking"), (26, "obs: noise draws"), (27, "obs: actor frame puts"), (10, "obs: task blocks / programs"), (11, "state stores")]
prev = 0.0
print(task, NE)
for k, name in order:
    print(f"{name:26s} +{acc[k]-prev:8.0f} cycles (cum {acc[k]:8.0f})")
    prev = acc[k]
