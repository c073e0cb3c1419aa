"""Diagnostic: how often does the rough-terrain contact model throw a robot (vertical speed / height above the tile far outside
anything legs can produce)?  usage: python tools/launch_events.py [task] [n_envs] [steps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hcr_genesis_lr_cl_amd.envs import make_env
task = sys.argv[1] if len(sys.argv) > 1 else "go2_ee"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
env, cfg = make_env(task, n)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(11)
env.episode_length_buf = torch.randint(0, 1000, (n,), generator=g, device="cuda", dtype=torch.int32)
s = env.simulator
fast = 0; high = 0; worst_v = 0.0; worst_h = 0.0
first = None
for t in range(steps):
    env.step(torch.randn(n, env.num_actions, generator=g, device="cuda"))
    vz = s._base_lin_vel_w[:, 2]
    rel = s.base_pos[:, 2] - s.env_origins[:, 2]
    f = (vz.abs() > 6.0)
    fast += int(f.sum()); high += int((rel > 4.0).sum())
    worst_v = max(worst_v, float(vz.abs().max())); worst_h = max(worst_h, float(rel.max()))
    if first is None and bool(f.any()):
        e = int(f.nonzero()[0])
        first = (t, e, float(vz[e]), s.base_pos[e].tolist(), s.env_origins[e].tolist(), float(s.link_contact_forces[e].norm(dim=-1).max()), int(s.terrain_levels[e]), int(s.terrain_types[e]))
print(f"{task}: {n} envs x {steps} steps: |vz| > 6 m/s in {fast} env-steps ({100*fast/(n*steps):.4f} %), height above tile > 4 m in {high}; worst |vz| {worst_v:.1f} m/s, worst height {worst_h:.1f} m")
print("first event:", first)
