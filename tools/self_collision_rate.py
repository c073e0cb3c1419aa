"""How often would self-collision matter?  The reference enables it (genesis_simulator.py:250 `enable_self_collision=True`); this
engine does not model it.  This diagnostic runs a random-policy rollout and counts, from the engine's own
state, how often collision spheres of NON-ADJACENT bodies overlap: spheres of different legs, and thigh / calf / foot spheres
against base spheres (hip-base and consecutive bodies of a leg are adjacent links, which Genesis excludes as well).

usage: python tools/self_collision_rate.py [task] [n_envs] [steps]   (GPU)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from hcr_genesis_lr_cl_amd.envs import make_env

task = sys.argv[1] if len(sys.argv) > 1 else "go2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
env, cfg = make_env(task, n)
env.reset()
m = env.simulator._model
A = m.arrays
dev = "cuda"
nb = int(A["n_bodies"]); ns = int(A["n_spheres"]); legs = int(A["n_legs"])
jpos = torch.tensor(A["jpos"][:nb], device=dev, dtype=torch.float32)
axis = torch.tensor(A["axis"][:nb], device=dev, dtype=torch.float32)
sph_body = torch.tensor(A["sph_body"][:ns], device=dev).long()
sph_pos = torch.tensor(A["sph_pos"][:ns], device=dev, dtype=torch.float32)
sph_r = torch.tensor(A["sph_r"][:ns], device=dev, dtype=torch.float32)


def quat_to_mat(q):
    x, y, z, w = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(*q.shape[:-1], 3, 3)


def axis_angle(ax, ang):
    c, s = torch.cos(ang)[..., None, None], torch.sin(ang)[..., None, None]
    K = torch.zeros(*ang.shape, 3, 3, device=dev)
    K[..., 0, 1], K[..., 0, 2], K[..., 1, 0], K[..., 1, 2], K[..., 2, 0], K[..., 2, 1] = -ax[2], ax[1], ax[2], -ax[0], -ax[1], ax[0]
    return torch.eye(3, device=dev) * c + s * K + (1 - c) * torch.outer(ax, ax)


def sphere_centres(sim):
    N = sim.base_pos.shape[0]
    R = [quat_to_mat(sim.base_quat)]
    P = [sim.base_pos.clone()]
    for b in range(1, nb):
        par = 0 if (b - 1) % 3 == 0 else b - 1
        Pb = P[par] + (R[par] @ jpos[b]).reshape(N, 3)
        Rb = R[par] @ axis_angle(axis[b], sim.dof_pos[:, b - 1])
        R.append(Rb); P.append(Pb)
    R, P = torch.stack(R, 1), torch.stack(P, 1)                      # (N, nb, 3, 3), (N, nb, 3)
    return P[:, sph_body] + (R[:, sph_body] @ sph_pos[None, :, :, None])[..., 0]


leg_of = torch.where(sph_body == 0, torch.full_like(sph_body, -1), (sph_body - 1) // 3)
seg_of = torch.where(sph_body == 0, torch.full_like(sph_body, -1), (sph_body - 1) % 3)
pair = (leg_of[:, None] != leg_of[None, :])                          # different legs, or leg vs base
pair &= ~(((leg_of[:, None] == -1) & (seg_of[None, :] == 0)) | ((leg_of[None, :] == -1) & (seg_of[:, None] == 0)))   # hip-base adjacent
pair &= torch.triu(torch.ones(ns, ns, dtype=torch.bool, device=dev), 1)
rsum = sph_r[:, None] + sph_r[None, :]
g = torch.Generator(device=dev); g.manual_seed(1)
env.episode_length_buf = torch.randint(0, 1000, (n,), generator=g, device=dev, dtype=torch.int32)
hit_steps = 0; deep_steps = 0; total = 0; worst = 0.0
for t in range(steps):
    env.step(torch.randn(n, env.num_actions, generator=g, device=dev))
    if t % 4 == 0:
        c = sphere_centres(env.simulator)
        d = torch.cdist(c, c)
        pen = (rsum[None] - d).masked_fill(~pair[None], -1.0)
        mx = pen.amax(dim=(1, 2))
        hit_steps += int((mx > 0).sum()); deep_steps += int((mx > 0.01).sum()); total += n
        worst = max(worst, float(mx.max()))
print(f"{task}: random N(0,1) policy, {n} envs x {steps} steps (every 4th sampled): non-adjacent sphere overlap in "
      f"{100 * hit_steps / total:.2f} % of env-steps, deeper than 1 cm in {100 * deep_steps / total:.2f} %, worst {worst * 100:.1f} cm")
