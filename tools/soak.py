"""Soak run: N(0,1) actions (every 7th batch x4) for many control steps on every registered task; checks every 500 steps that the
whole state and every returned tensor is finite and inside the engine's clamps, counts resets.
usage: python tools/soak.py [steps] [n_envs] [tasks...]   (GPU)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hcr_genesis_lr_cl_amd.envs import make_env, TASKS

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
tasks = sys.argv[3:] or list(TASKS)
bad = 0
for task in tasks:
    env, cfg = make_env(task, n)
    env.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(17)
    bank = [torch.randn(n, env.num_actions, generator=g, device="cuda") * (4.0 if i % 7 == 6 else 1.0) for i in range(32)]
    resets = torch.zeros((), device="cuda", dtype=torch.int64)
    t0 = time.perf_counter()
    worst = {}
    for t in range(steps):
        out = env.step(bank[t % 32])
        resets += out[-2].sum()
        if t % 500 == 499 or t == steps - 1:
            b = env._engine.buf
            for k in ("dof_pos", "dof_vel", "base_pos", "base_quat", "base_lin_vel_w", "base_ang_vel_w", "torques", "link_contact_forces",
                      "feet_pos", "feet_vel", "episode_sums", "commands", "feet_air_time"):
                if k in b and not torch.isfinite(b[k]).all():
                    print(f"  {task}: non-finite {k} at step {t}"); bad += 1
            for o in out[:-1]:
                if torch.is_tensor(o) and o.dtype == torch.float32 and not torch.isfinite(o).all():
                    print(f"  {task}: non-finite output at step {t}"); bad += 1
            qn = (b["base_quat"].norm(dim=1) - 1).abs().max().item()
            worst["quat"] = max(worst.get("quat", 0.0), qn)
            worst["z"] = max(worst.get("z", 0.0), b["base_pos"][:, 2].abs().max().item())
            worst["v"] = max(worst.get("v", 0.0), b["base_lin_vel_w"].abs().max().item())
            worst["qd"] = max(worst.get("qd", 0.0), b["dof_vel"].abs().max().item())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{task:14s} {steps} steps x {n} envs in {dt:6.2f} s ({n * steps / dt / 1e6:6.1f} M env-steps/s incl. checks), resets {int(resets)}, "
          f"|quat|-1 <= {worst['quat']:.1e}, |z| <= {worst['z']:.2f} m, |v| <= {worst['v']:.1f} m/s, |qd| <= {worst['qd']:.1f} rad/s", flush=True)
    del env
print("non-finite findings:", bad)
sys.exit(1 if bad else 0)
