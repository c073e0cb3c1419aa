"""Developer check: one control step of tron1_pf_ee through the fused launch and through SIM + PRE|POST|RESET; where do they differ?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from hcr_genesis_lr_cl_amd import abi
from hcr_genesis_lr_cl_amd.envs import make_env
N = 256
task = sys.argv[1] if len(sys.argv) > 1 else "tron1_pf_ee"
e1, e2 = make_env(task, N, "cuda:0")[0], make_env(task, N, "cuda:0")[0]
e1.reset(); e2.reset()
g = torch.Generator(device="cuda"); g.manual_seed(5)
e1.episode_length_buf[:] = torch.randint(0, 1000, (N,), generator=g, device="cuda", dtype=torch.int32)
e1.common_step_counter = e2.common_step_counter = 480
for t in range(30):
    for k in e1._engine.buf.keys():
        e2._engine.buf.raw(k).copy_(e1._engine.buf.raw(k))
    e2.common_step_counter = e1.common_step_counter
    act = torch.randn(N, e1.num_actions, generator=g, device="cuda") * (1.0 if t % 5 else 4.0)
    e1.step(act)
    e2.common_step_counter += 1
    ca = float(e2.cfg.normalization.clip_actions)
    e2._engine.step(abi.PHASE_SIM, torch.clip(act, -ca, ca), e2.common_step_counter)
    e2._engine.step(abi.PHASE_PRE | abi.PHASE_POST | abi.PHASE_RESET, act, e2.common_step_counter)
    torch.cuda.synchronize()
    for k in e1._engine.buf.keys():
        a, b = e1._engine.buf[k].cpu().numpy().astype(np.float64), e2._engine.buf[k].cpu().numpy().astype(np.float64)
        d = np.abs(a - b).reshape(a.shape[0], -1) if a.ndim > 1 else np.abs(a - b).reshape(-1, 1)
        tol = 2e-3 if "vel" in k or k in ("torques", "link_contact_forces") else 2e-4
        if d.max() > tol:
            cols = np.unique(np.nonzero(d > tol)[1])
            print(f"step {t} {k}: max {d.max():.4g} at cols {cols[:24].tolist()} ({(d > tol).any(axis=1).sum()} rows), resets {int(e1.reset_buf.sum())}")
# which uniform did the sin entries get?  noise = entry - noise-free copy in the critic frame
o1, o2 = e1._engine.buf["obs_buf"].cpu().numpy(), e2._engine.buf["obs_buf"].cpu().numpy()
p1 = e1._engine.buf["priv_obs_buf"].cpu().numpy()
F0, P0 = 9 * 31, 9 * 134
for r in range(3):
    print("env", r, "fused noise sin0 sin1 cos0 cos1:", (o1[r, F0 + 27:F0 + 31] - p1[r, P0 + 27:P0 + 31]).round(5).tolist(),
          "| split:", (o2[r, F0 + 27:F0 + 31] - p1[r, P0 + 27:P0 + 31]).round(5).tolist(),
          "| act noise fused", (o1[r, F0 + 21:F0 + 27] - p1[r, P0 + 21:P0 + 27]).round(5).tolist())
