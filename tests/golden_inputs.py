"""Synthetic physics read-backs / draws for MDP parity tests at arbitrary env counts."""
import numpy as np


def random_mdp_inputs(rng, model, cfg, N, n_slots):
    A, L, F = model.n_dof, model.n_links, model.n_legs
    q0 = np.array([cfg.init_state.default_joint_angles[n] for n in cfg.asset.dof_names], np.float32)
    s = {}
    s["base_pos"] = (rng.normal(size=(N, 3)) * [2, 2, 0.03] + [0, 0, 0.32]).astype(np.float32)
    rpy = rng.normal(size=(N, 3)) * np.where(rng.random((N, 1)) < 0.1, 1.3, 0.15) * [1, 1, 10]
    cr, sr, cp, sp, cy, sy = [f(rpy[:, i] / 2) for i in range(3) for f in (np.cos, np.sin)]
    s["base_quat"] = np.stack([cy * sr * cp - sy * cr * sp, cy * cr * sp + sy * sr * cp, sy * cr * cp - cy * sr * sp,
                               cy * cr * cp + sy * sr * sp], 1).astype(np.float32)
    s["base_lin_vel_w"] = (rng.normal(size=(N, 3)) * [0.6, 0.4, 0.15]).astype(np.float32)
    s["base_ang_vel_w"] = (rng.normal(size=(N, 3)) * [0.5, 0.5, 0.8]).astype(np.float32)
    s["dof_pos"] = (q0 + rng.normal(size=(N, A)) * 0.35).astype(np.float32)
    s["dof_vel"] = (rng.normal(size=(N, A)) * 3).astype(np.float32)
    s["last_dof_vel"] = (rng.normal(size=(N, A)) * 3).astype(np.float32)
    s["torques"] = (rng.normal(size=(N, A)) * 8).astype(np.float32)
    f = np.zeros((N, L, 3), np.float32)
    feet = [int(i) for i in model.arrays["foot_link"][:F]]
    for l in range(L):
        if l in feet:
            on = rng.random(N) < 0.6
            f[:, l, 2] = on * rng.uniform(0.05, 80, N)
            f[:, l, :2] = on[:, None] * rng.normal(size=(N, 2)) * 10
        else:
            on = rng.random(N) < (0.06 if l == 0 else 0.1)
            f[:, l] = on[:, None] * rng.normal(size=(N, 3)) * (12 if l == 0 else 3)
    s["link_contact_forces"] = f
    s["feet_pos"] = (rng.normal(size=(N, F, 3)) * [0.2, 0.15, 0.03] + [0, 0, 0.05]).astype(np.float32)
    s["feet_vel"] = (rng.normal(size=(N, F, 3)) * [0.8, 0.4, 0.5]).astype(np.float32)
    s["last_feet_vel"] = (rng.normal(size=(N, F, 3)) * [0.8, 0.4, 0.5]).astype(np.float32)
    actions = (rng.normal(size=(N, A)) * np.where(rng.random((N, 1)) < 0.05, 80, 1)).astype(np.float32)
    R = rng.random((N, n_slots), dtype=np.float32)
    return s, actions, R
