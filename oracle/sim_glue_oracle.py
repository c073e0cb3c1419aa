"""numpy restatement of the reference's simulator glue around the physics engine
(legged_gym/simulator/genesis_simulator.py).  TEST INFRASTRUCTURE ONLY (see oracle/lg_oracle.c header):
imported by tests/, never by the product.

Pinned by tests/golden/sim_glue_<task>.npz, which tests/golden/gen_sim_glue_fixtures.py produced by running the
reference's own GenesisSimulator methods unbound on a recording stand-in (tests/test_sim_glue_oracle.py).
All arithmetic is float32, like torch's.
"""
from __future__ import annotations

import numpy as np

from . import mdp_oracle as mo

f32 = np.float32


def compute_torques(actions, action_scale, kp_scale, p_gains, default_dof_pos, dof_pos, kd_scale, d_gains, dof_vel):
    """genesis_simulator.py:630-642: tau = kp_s Kp (a_scaled + q0 - q) - kd_s Kd qd, unclipped."""
    a = actions.astype(f32) * f32(action_scale)
    return (kp_scale * p_gains * (a + default_dof_pos - dof_pos) - kd_scale * d_gains * dof_vel).astype(f32)


def out_of_bound(base_pos, x_range, y_range):
    """genesis_simulator.py:615-619: on the bound counts as outside."""
    x, y = base_pos[:, 0], base_pos[:, 1]
    return (x >= x_range[1]) | (x <= x_range[0]) | (y >= y_range[1]) | (y <= y_range[0])


def teleport_out_of_bound(base_pos, x_range, y_range, base_init_pos, env_origins):
    """genesis_simulator.py:612-628: base_init_pos + env_origins for the envs outside; set_pos(zero_velocity=False)
    leaves the twist alone."""
    m = out_of_bound(base_pos, x_range, y_range)
    out = base_pos.astype(f32).copy()
    out[m] = (base_init_pos.astype(f32) + env_origins[m].astype(f32)).astype(f32)
    return out, m


def read_back(base_quat_wxyz, lin_vel_w, ang_vel_w):
    """genesis_simulator.py:40-46: wxyz -> xyzw, euler, body-frame twist, projected unit gravity."""
    q = np.concatenate([base_quat_wxyz[:, 1:4], base_quat_wxyz[:, 0:1]], 1).astype(f32)
    g = np.tile(np.array([0, 0, -1], f32), (len(q), 1))
    return dict(base_quat=q, base_euler=mo.get_euler_xyz(q), base_lin_vel=mo.quat_rotate_inverse(q, lin_vel_w),
                base_ang_vel=mo.quat_rotate_inverse(q, ang_vel_w), projected_gravity=mo.quat_rotate_inverse(q, g))


def contact_states(link_contact_forces, link_ids):
    """genesis_simulator.py:53-55."""
    return (np.linalg.norm(link_contact_forces[:, link_ids, :], axis=-1) > 1.0).astype(f32)


def affine(u, lo, hi):
    """u * (hi - lo) + lo with the python-float range folded to f32 the way torch does for a f32 tensor."""
    return (u.astype(f32) * f32(hi - lo) + f32(lo)).astype(f32)


def randomize(cfg_dr, n_actions, draws):
    """genesis_simulator.py:62-82, 665-739: draw order and formulas of reset_idx's domain randomisation.
    `draws`: the uniforms in call order.  Returns name -> values for the reset envs."""
    it = iter(draws)
    out = {}
    if cfg_dr.randomize_friction:
        out["friction_values"] = affine(next(it), *cfg_dr.friction_range)
    if cfg_dr.randomize_base_mass:
        out["added_base_mass"] = affine(next(it), *cfg_dr.added_mass_range)
    if cfg_dr.randomize_com_displacement:
        out["base_com_bias"] = np.concatenate([affine(next(it), *r) for r in (cfg_dr.com_pos_x_range, cfg_dr.com_pos_y_range, cfg_dr.com_pos_z_range)], 1)
    for flag, rng, key in (("randomize_joint_armature", "joint_armature_range", "joint_armature"),
                           ("randomize_joint_friction", "joint_friction_range", "joint_friction"),
                           ("randomize_joint_damping", "joint_damping_range", "joint_damping")):
        if getattr(cfg_dr, flag):
            out[key] = affine(next(it), *getattr(cfg_dr, rng)).reshape(-1, 1)
    if cfg_dr.randomize_pd_gain:
        # torch_rand_float (math_utils.py:79-81): (upper - lower) * rand + lower
        out["kp_scale"] = (f32(cfg_dr.kp_range[1] - cfg_dr.kp_range[0]) * next(it) + f32(cfg_dr.kp_range[0])).astype(f32)
        out["kd_scale"] = (f32(cfg_dr.kd_range[1] - cfg_dr.kd_range[0]) * next(it) + f32(cfg_dr.kd_range[0])).astype(f32)
    return out


def terrain_curriculum(levels, types, origins, ids, up, down, max_level, randint):
    """genesis_simulator.py:140-148."""
    lv = levels.copy()
    lv[ids] = lv[ids] + 1 * up - 1 * down
    lv[ids] = np.where(lv[ids] >= max_level, randint, np.clip(lv[ids], 0, None))
    return lv, origins[lv[ids], types[ids]]


def quat_from_euler_xyz(roll, pitch, yaw):
    """math_utils.py:111-125."""
    cy, sy, cr, sr, cp, sp = np.cos(yaw * f32(0.5)), np.sin(yaw * f32(0.5)), np.cos(roll * f32(0.5)), np.sin(roll * f32(0.5)), \
        np.cos(pitch * f32(0.5)), np.sin(pitch * f32(0.5))
    qw = cy * cr * cp + sy * sr * sp
    qx = cy * sr * cp - sy * cr * sp
    qy = cy * cr * sp + sy * sr * cp
    qz = sy * cr * cp - cy * sr * sp
    return np.stack([qx, qy, qz, qw], -1).astype(f32)
