"""numpy restatement of the MDP part of one control step (everything LeggedRobot.step does
around the physics).  TEST INFRASTRUCTURE ONLY (see oracle/lg_oracle.c header).

Pinned by golden vectors generated from the reference's own env classes
(tests/golden/gen_mdp_fixtures.py -> tests/golden/*_mdp.npz; tests/test_mdp_oracle.py).
Each block cites the reference lines it follows.  All arithmetic is float32 like torch's.
"""
from __future__ import annotations

import numpy as np

from hcr_genesis_lr_cl_amd import abi
from hcr_genesis_lr_cl_amd import config as cfgmod

f32 = np.float32


def quat_rotate_inverse(q, v):
    """math_utils.py:64-76."""
    qw = q[:, 3:4]
    qv = q[:, :3]
    a = v * (f32(2.0) * qw ** 2 - f32(1.0))
    b = np.cross(qv, v) * (f32(2.0) * qw)
    c = qv * (f32(2.0) * np.sum(qv * v, axis=1, keepdims=True))
    return (a - b + c).astype(f32)


def quat_apply(q, v):
    """math_utils.py:34-40."""
    xyz = q[:, :3]
    t = np.cross(xyz, v) * f32(2)
    return (v + q[:, 3:4] * t + np.cross(xyz, t)).astype(f32)


def wrap_to_pi(a):
    """math_utils.py:50-53.  Under TorchScript the in-place ``angles %= 2*pi`` lowers to
    aten::fmod_ (C-style remainder, sign of the dividend), NOT torch.remainder: negative angles
    are left in (-2*pi, 0] and only values above +pi are shifted.  Verified on the scripted
    function's graph; reproduced here because the heading command depends on it."""
    two_pi = f32(2 * np.pi)
    a = np.fmod(a.astype(f32), two_pi)
    return (a - two_pi * (a > f32(np.pi))).astype(f32)


def get_euler_xyz(q):
    """math_utils.py:90-109."""
    qx, qy, qz, qw = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    roll = np.arctan2(f32(2) * (qw * qx + qy * qz), qw * qw - qx * qx - qy * qy + qz * qz)
    sinp = f32(2) * (qw * qy - qz * qx)
    pitch = np.where(np.abs(sinp) >= 1, np.copysign(f32(np.pi / 2), sinp), np.arcsin(np.clip(sinp, -1, 1)))
    yaw = np.arctan2(f32(2) * (qw * qz + qx * qy), qw * qw + qx * qx - qy * qy - qz * qz)
    return np.stack([roll, pitch, yaw], 1).astype(f32)


def quat_apply_yaw(q, v):
    """math_utils.py:43-47: zero x,y of the quaternion, renormalise, rotate."""
    qy = q.copy().astype(f32)
    qy[:, :2] = 0
    n = np.clip(np.linalg.norm(qy, axis=1, keepdims=True), 1e-9, None)
    qy = (qy / n).astype(f32)
    P = v.shape[1]
    qq = np.repeat(qy, P, axis=0)
    return quat_apply(qq, v.reshape(-1, 3)).reshape(v.shape)


def sample_heights(base_pos, base_quat, height_points, hf, border, hscale, vscale):
    """genesis_simulator.py:552-577: grid around the base, truncation to cells, min of 3 neighbours."""
    N, P = base_pos.shape[0], height_points.shape[0]
    hp = np.zeros((N, P, 3), f32)
    hp[:, :, :2] = height_points
    pts = quat_apply_yaw(base_quat, hp) + base_pos[:, None, :]
    pts = pts + f32(border)
    idx = (pts / f32(hscale)).astype(np.int64)        # .long(): truncation towards zero
    px = np.clip(idx[:, :, 0].reshape(-1), 0, hf.shape[0] - 2)
    py = np.clip(idx[:, :, 1].reshape(-1), 0, hf.shape[1] - 2)
    h = np.minimum(np.minimum(hf[px, py], hf[px + 1, py]), hf[px, py + 1])
    return (h.reshape(N, P) * f32(vscale)).astype(f32)


def feet_terrain_info(feet_pos, hf, border, hscale, vscale):
    """genesis_simulator.py:579-610: 9 heights per foot; normal from RAW int16 differences."""
    N, F = feet_pos.shape[:2]
    idx = ((feet_pos + f32(border)) / f32(hscale)).astype(np.int64)
    px = np.clip(idx[:, :, 0].reshape(-1), 0, hf.shape[0] - 2)
    py = np.clip(idx[:, :, 1].reshape(-1), 0, hf.shape[1] - 2)
    xm, ym = np.maximum(px - 1, 0), np.maximum(py - 1, 0)
    hs = [hf[xm, py], hf[px + 1, py], hf[px, ym], hf[px, py + 1], hf[px, py], hf[xm, ym], hf[px + 1, py + 1],
          hf[xm, py + 1], hf[px + 1, ym]]
    dx = ((hs[1].astype(np.int16) - hs[0].astype(np.int16)) / (hscale * 2)).reshape(N, F).astype(f32)
    dy = ((hs[3].astype(np.int16) - hs[2].astype(np.int16)) / (hscale * 2)).reshape(N, F).astype(f32)
    nrm = np.stack([dx, dy, -np.ones_like(dx)], -1)
    nrm = (nrm / np.linalg.norm(nrm, axis=-1, keepdims=True)).astype(f32)
    har = np.stack([h.reshape(N, F) * f32(vscale) for h in hs], -1).astype(f32)
    return har, nrm.reshape(N, F * 3)


def foot_rotations(model, base_quat, dof_pos):
    """World-from-foot-body rotation matrices (N, F, 3, 3) of the chains' last bodies: base rotation times, joint by joint, the
    fixed joint frame and the rotation about the joint axis (the link orientation Genesis reports in rigid_body_states[:, :, 3:7])."""
    a = model.arrays
    N, F, J = base_quat.shape[0], model.n_legs, model.joints_per_leg
    q = base_quat.astype(np.float64)
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    Rb = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                   np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                   np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], 1)
    out = np.zeros((N, F, 3, 3))
    for l in range(F):
        R = Rb
        for j in range(J):
            b = 1 + J * l + j
            ax = a["axis"][b] / np.linalg.norm(a["axis"][b])
            K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
            th = dof_pos[:, J * l + j].astype(np.float64)[:, None, None]
            Rj = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
            R = R @ a["jrot"][b].reshape(3, 3) @ Rj
        out[:, l] = R
    return out


class MdpOracle:
    def __init__(self, model, cfg, task, n_envs, env_origins=None):
        self.model, self.cfg, self.task, self.N = model, cfg, task, n_envs
        N, A, F = n_envs, model.n_dof, model.n_legs
        self.A, self.F, self.L = A, F, model.n_links
        self.dt = f32(task.control_dt)
        self.q0 = cfgmod.default_dof_pos(cfg)
        self.soft = cfgmod.soft_dof_limits(model, cfg)
        self.feet = [int(i) for i in sorted(model.arrays["foot_link"][:F])]
        self.term = model.find_link_indices(cfg.asset.terminate_after_contacts_on)
        self.pen = model.find_link_indices(cfg.asset.penalize_contacts_on)
        self.scales = np.ctypeslib.as_array(task.reward_scales).copy()
        self.noise_vec = np.ctypeslib.as_array(task.noise_vec)[:task.obs_frame].copy()
        z = lambda *s: np.zeros(s, f32)
        self.actions, self.last_actions, self.llast_actions = z(N, A), z(N, A), z(N, A)
        self.commands = z(N, 4)
        self.feet_air_time = z(N, F)
        self.last_contacts = np.zeros((N, F), bool)
        self.episode_length_buf = np.zeros(N, np.int32)
        self.fail_buf = np.zeros(N, np.int64)
        self.reset_buf = np.zeros(N, bool)
        self.time_out_buf = np.zeros(N, bool)
        self.rew_buf = z(N)
        self.obs_buf = z(N, task.num_obs)
        self.episode_sums = z(abi.R_COUNT, N)
        self.env_origins = z(N, 3) if env_origins is None else env_origins.astype(f32)
        cr = cfg.commands.ranges
        self.command_ranges = np.array(list(cr.lin_vel_x) + list(cr.lin_vel_y) + list(cr.ang_vel_yaw) + list(cr.heading), f32)
        # DR / engine-side per-env values written at reset
        self.friction_values, self.added_base_mass = np.ones((N, 1), f32), z(N, 1)
        self.base_com_bias, self.rand_push_vels = z(N, 3), z(N, 3)
        self.kp_scale, self.kd_scale = np.ones((N, A), f32), np.ones((N, A), f32)
        self.priv_obs_buf = z(N, max(task.num_priv_obs, 1))
        self.labels_buf = z(N, max(task.num_labels, 1))
        self.state_links = model.find_link_indices(cfg.asset.contact_state_link_names) if cfg.asset.obtain_link_contact_states else []
        self.cstr_prob, self.cstr_sums = z(N), z(abi.NUM_CSTR, N)
        self.terrain_levels = np.zeros(N, np.int64)
        self.terrain_types = np.zeros(N, np.int64)
        self.terrain_origins = None
        if task.obs_layout in (abi.OBS_GO2_EE, abi.OBS_PROGRAM):
            self.obs_hist = z(N, task.obs_stack, task.obs_frame)
            self.priv_hist = z(N, task.priv_stack, task.priv_frame)
        if task.gait_mode == 1:
            self._init_wtw()
        self.joint_armature, self.joint_friction, self.joint_damping = z(N, 1), z(N, 1), z(N, 1)
        if task.gait_mode == 2:                                       # tron1_pf_ee.py:150-184
            self.theta = np.tile(np.ctypeslib.as_array(task.theta_table).reshape(4, 4)[0, :2], (N, 1)).astype(f32)
            self.gait_time, self.phi = z(N, 1), z(N, 1)
            self.gait_period = np.full((N, 1), task.gait_period_fixed, f32)
            self.clock_input = z(N, 4)
            self.exp_C_frc = z(N, 2)
        if task.obs_layout == abi.OBS_TRON1_EE:
            self.obs_hist = z(N, task.obs_stack, task.obs_frame)
            self.priv_hist = z(N, task.priv_stack, task.priv_frame)

    def _init_wtw(self):
        """go2_wtw.py:295-376 (_init_buffers + _parse_cfg)."""
        N, T = self.N, self.task
        bp = self.cfg.rewards.behavior_params_range
        mid = lambda r: [(r[0] + r[1]) / 2] * 2
        self.gait_period_range = mid(bp.gait_period_range)
        self.foot_clearance_target_range = [bp.foot_clearance_target_range[0]] * 2
        self.base_height_target_range = mid(bp.base_height_target_range)
        self.pitch_target_range = mid(bp.pitch_target_range)
        self.num_gaits = 1
        self.theta_table = np.ctypeslib.as_array(T.theta_table).reshape(4, 4).copy()
        z = lambda *s: np.zeros(s, f32)
        self.theta = np.tile(self.theta_table[0], (N, 1)).astype(f32)
        self.gait_time, self.phi = z(N, 1), z(N, 1)
        self.gait_period = np.full((N, 1), self.gait_period_range[0], f32)
        self.clock_input = z(N, 8)
        self.base_height_target = np.full((N, 1), self.base_height_target_range[0], f32)
        self.foot_clearance_target = np.full((N, 1), self.foot_clearance_target_range[0], f32)
        self.pitch_target = np.full((N, 1), self.pitch_target_range[0], f32)
        self.exp_C_frc = z(N, 4)
        self.obs_hist = z(N, T.obs_stack, T.obs_frame)
        self.priv_hist = z(N, T.priv_stack, T.priv_frame)

    def behavior_ranges(self):
        return np.array(self.gait_period_range + self.base_height_target_range + self.foot_clearance_target_range
                        + self.pitch_target_range + [self.num_gaits], f32)

    def _gait_indicator(self, feet_f, feet_vel, n_feet):
        """"step" indicator of go2_wtw.py:377-470 / tron1_pf_ee.py:347-424, including the reference's index-flatten bug
        that overwrites env 0 (see _resample_behavior docstring of the wtw notes)."""
        T, N = self.task, self.N
        acc = np.zeros(N, f32)
        b_swing = f32(T.b_swing) * f32(2 * np.pi)
        for i in range(n_feet):
            q_frc = np.linalg.norm(feet_f[:, i], axis=-1)
            q_spd = np.linalg.norm(feet_vel[:, i], axis=-1)
            ph = np.remainder(self.phi[:, 0] + self.theta[:, i], f32(1.0)).astype(f32) * f32(2 * np.pi)
            swing = (ph >= 0) & (ph < b_swing)
            stance = (ph >= b_swing) & (ph < f32(2 * np.pi))
            c_frc = np.where(swing, -1.0, 0.0).astype(f32)
            c_spd = np.where(stance, -1.0, 0.0).astype(f32)
            if swing.any():
                c_frc[0], c_spd[0] = -1.0, 0.0
            if stance.any():
                c_frc[0], c_spd[0] = 0.0, -1.0
            self.exp_C_frc[:, i] = c_frc
            acc = acc + (c_spd * q_spd + c_frc * q_frc).astype(f32)
        return acc

    def _resample_behavior(self, ids, R, slot):
        """go2_wtw.py:180-218.  One gait index per call for the whole batch (quirk 6); the pronk/bound
        clearance clamp is idempotent per env because the lower clearance bound never moves."""
        if len(ids) == 0:
            return
        u = lambda k: R[ids, slot + k][:, None]
        rg = self.gait_period_range, self.base_height_target_range, self.foot_clearance_target_range, self.pitch_target_range
        for k, (arr, r) in enumerate(zip((self.gait_period, self.base_height_target, self.foot_clearance_target, self.pitch_target), rg)):
            arr[ids] = f32(r[1] - r[0]) * u(k) + f32(r[0])
        sel = min(int(np.floor(R[ids[0], slot + 4] * self.num_gaits)), self.num_gaits - 1)
        self.theta[ids] = self.theta_table[sel]
        th = self.theta
        pronk = (th[:, 0] == 0) & (th[:, 1] == 0) & (th[:, 2] == 0) & (th[:, 3] == 0)
        bound = (th[:, 0] == 0) & (th[:, 1] == 0) & (th[:, 2] == 0.5) & (th[:, 3] == 0.5)
        self.foot_clearance_target[pronk | bound] = f32(self.foot_clearance_target_range[0])

    # ---------------------------------------------------------------------------------------
    def _resample(self, ids, R, slot):
        """legged_robot.py:317-334."""
        cr, c = self.command_ranges, self.commands
        c[ids, 0] = (cr[1] - cr[0]) * R[ids, slot] + cr[0]
        c[ids, 1] = (cr[3] - cr[2]) * R[ids, slot + 1] + cr[2]
        if self.task.heading_command:
            c[ids, 3] = (cr[7] - cr[6]) * R[ids, slot + 2] + cr[6]
        else:
            c[ids, 2] = (cr[5] - cr[4]) * R[ids, slot + 2] + cr[4]
        keep = np.sqrt(np.sum(c[ids, :3] ** 2, axis=1)) > f32(0.2)
        c[ids, :3] *= keep[:, None]

    def step(self, sim, actions, R, counter):
        """sim: dict of physics read-backs AFTER the physics of this step (modified in place for
        resets / pushes, like the engine state).  R: (N, n_slots) uniforms.  counter: value of
        common_step_counter after this step's increment."""
        T, S, N, A, F = self.task, self.task.slots, self.N, self.A, self.F
        sc = self.scales
        # ---- _pre_sim_step: legged_robot.py:230-239
        a = np.clip(actions.astype(f32), -f32(T.clip_actions), f32(T.clip_actions))
        self.llast_actions[:] = self.last_actions
        self.last_actions[:] = self.actions
        self.actions[:] = a
        # ---- read-back: genesis_simulator.py:40-46
        q = sim["base_quat"]
        blv = quat_rotate_inverse(q, sim["base_lin_vel_w"])
        bav = quat_rotate_inverse(q, sim["base_ang_vel_w"])
        g = np.tile(np.array([0, 0, -1], f32), (N, 1))
        pg = quat_rotate_inverse(q, g)
        # ---- post_physics_step: legged_robot.py:55-76
        self.episode_length_buf += 1
        ids = np.nonzero(self.episode_length_buf % T.resample_steps == 0)[0]
        self._resample(ids, R, S.cb_cmd)
        if T.heading_command:                                              # :307-312
            fwd = quat_apply(q, np.tile(np.array([1, 0, 0], f32), (N, 1)))
            heading = np.arctan2(fwd[:, 1], fwd[:, 0]).astype(f32)
            self.commands[:, 2] = np.clip(f32(0.5) * wrap_to_pi(self.commands[:, 3] - heading), f32(T.yaw_clip[0]), f32(T.yaw_clip[1]))
        if T.push_interval > 0 and counter % T.push_interval == 0:       # :314-315, genesis_simulator.py:150-158
            m = f32(T.max_push_vel_xy)
            push = (m + m) * R[:, S.push:S.push + 2] - m
            self.rand_push_vels[:, :2] = push
            sim["base_lin_vel_w"][:, :2] += push
        if T.gait_mode == 1 and T.behavior_resample_steps > 0:           # go2_wtw.py:258-263
            ids = np.nonzero(self.episode_length_buf % T.behavior_resample_steps == 0)[0]
            self._resample_behavior(ids, R, S.task_cb)
        # ---- check_termination: :78-92
        F_l = sim["link_contact_forces"].reshape(N, self.L, 3)
        fail = np.zeros(N, bool)
        if self.term:
            fail = np.any(np.linalg.norm(F_l[:, self.term], axis=-1) > 10.0, axis=1)
        fail |= pg[:, 2] > f32(T.max_projected_gravity)
        self.fail_buf += fail
        self.time_out_buf = self.episode_length_buf > T.max_episode_length
        self.reset_buf = (self.fail_buf > T.fail_threshold) | self.time_out_buf
        # ---- constraints as terminations: go2_cat.py:143-205 + utils/constraint_manager.py:25-97.  Every constraint is a 0/1 flag, so
        #      the manager's ratio to the running max clamps to 1: p = max_p of the worst violated constraint
        cat_keep = f32(1.0)
        if T.cat_enable:
            tq_lim = self.model.arrays["effort"].astype(f32)
            vlim = np.ctypeslib.as_array(T.dof_vel_limits)[:A]
            lo_, hi_ = np.ctypeslib.as_array(T.soft_dof_lo)[:A], np.ctypeslib.as_array(T.soft_dof_hi)[:A]
            mh = sim["measured_heights"] if "measured_heights" in sim else np.zeros((N, 1), f32)
            still = (np.linalg.norm(self.commands[:, :3], axis=1) < 0.1)
            flags = [
                np.any(np.abs(sim["torques"]) > tq_lim, axis=1),
                np.any(np.abs(sim["dof_vel"]) > vlim, axis=1),
                np.any(np.abs(self.actions - self.last_actions) / self.dt > f32(T.cat_action_rate), axis=1),
                np.mean(sim["base_pos"][:, 2:3] - mh, axis=1) < f32(T.cat_min_base_height),
                np.any(np.linalg.norm(F_l[:, self.pen], axis=-1) > 10.0, axis=1),
                np.any(np.linalg.norm(F_l[:, self.feet], axis=-1) > 4 * np.abs(F_l[:, self.feet, 2]), axis=1),
                np.any(sim["dof_pos"] < lo_, axis=1) & np.any(sim["dof_pos"] > hi_, axis=1),
                pg[:, 2] > f32(T.cat_max_projected_gravity),
                # (N,) * (N,1) in the reference (go2_cat.py:179-180): row j of the (N,N) product is still_j * any-env-fast
                still & bool(np.any(np.abs(sim["dof_vel"]) > 4.0)),
            ]
            max_p = [T.cat_soft_p] * 4 + [1.0] * 4 + [T.cat_soft_p]
            self.cstr_prob = np.max(np.stack([f.astype(f32) * f32(p_) for f, p_ in zip(flags, max_p)]), axis=0).astype(f32)
            self.cstr_sums += np.stack(flags).astype(f32)
            cat_keep = (f32(1.0) - self.cstr_prob).astype(f32)
        # ---- compute_reward: :150-168 (alphabetical)
        total = np.zeros(N, f32)
        cmd = self.commands
        dof_pos, dof_vel = sim["dof_pos"], sim["dof_vel"]
        feet_f = F_l[:, self.feet]
        feet_pos, feet_vel = sim["feet_pos"].reshape(N, F, 3), sim["feet_vel"].reshape(N, F, 3)
        cmd_xy = np.sqrt(np.sum(cmd[:, :2] ** 2, axis=1))
        cmd_xyz = np.sqrt(np.sum(cmd[:, :3] ** 2, axis=1))

        def add(name, r):
            k = abi.REWARD_ID[name]
            if sc[k] == 0:
                return
            rew = (r.astype(f32) * sc[k]).astype(f32)
            total[:] = total + rew
            self.episode_sums[k] += rew
        R_ = abi.REWARD_ID
        on = lambda n: sc[R_[n]] != 0
        JPL4 = self.model.joints_per_leg == 4     # sole-foot biped: three quadruped-only slots carry its own terms (include/lgsim.h)
        if on("action_rate"):
            add("action_rate", np.sum((self.last_actions - self.actions) ** 2, axis=1))
        if on("action_smoothness"):
            add("action_smoothness", np.sum((self.actions - f32(2) * self.last_actions + self.llast_actions) ** 2, axis=1))
        if on("ang_vel_xy"):
            add("ang_vel_xy", np.sum(bav[:, :2] ** 2, axis=1))
        if on("base_height"):
            add("base_height", (sim["base_pos"][:, 2] - f32(T.base_height_target)) ** 2)
        if on("biped_periodic_gait"):                                      # tron1_pf_ee.py:347-433
            add("biped_periodic_gait", np.exp(self._gait_indicator(feet_f, feet_vel, 2)))
        if on("collision"):
            add("collision", np.sum(1.0 * (np.linalg.norm(F_l[:, self.pen], axis=-1) > 0.1), axis=1))
        if on("dof_acc"):
            add("dof_acc", np.sum(((sim["last_dof_vel"] - dof_vel) / self.dt) ** 2, axis=1))
        if on("dof_close_to_default"):
            add("dof_close_to_default", np.sum((dof_pos - self.q0) ** 2, axis=1))
        if on("dof_pos_limits"):
            out = -np.clip(dof_pos - self.soft[:, 0], None, 0) + np.clip(dof_pos - self.soft[:, 1], 0, None)
            add("dof_pos_limits", np.sum(out, axis=1))
        if on("dof_pos_stand_still"):
            add("dof_pos_stand_still", np.sum((dof_pos - self.q0) ** 2, axis=1) * (cmd_xyz < 0.1))
        if on("dof_power"):
            add("dof_power", np.sum(np.abs(sim["torques"] * dof_vel), axis=1))
        if on("dof_vel"):
            add("dof_vel", np.sum(dof_vel ** 2, axis=1))
        if on("dof_vel_stand_still"):
            add("dof_vel_stand_still", np.sum(np.abs(dof_vel), axis=1) * (cmd_xyz < 0.1))
        if on("feet_air_time"):                                            # :545-555
            contact = feet_f[:, :, 2] > 1.0
            filt = contact | self.last_contacts
            self.last_contacts = contact
            first = (self.feet_air_time > 0) * filt
            self.feet_air_time += self.dt
            r = np.sum((self.feet_air_time - f32(T.feet_air_time_threshold)) * first, axis=1)
            r = r * ((cmd_xyz if T.air_time_cmd_dims == 3 else cmd_xy) > 0.1)      # legged_robot.py:553 | tron1_sf.py:264
            self.feet_air_time *= ~filt
            add("feet_air_time", r)
        if on("feet_contact_stand_still"):
            full = np.sum(1.0 * (feet_f[:, :, 2] > 0.1), axis=1) == F
            add("feet_contact_stand_still", 1.0 * full * (cmd_xyz < 0.1))
        if on("feet_distance"):                                            # tron1_pf_ee.py:458-463
            d = np.linalg.norm(feet_pos[:, 0, :2] - feet_pos[:, 1, :2], axis=-1)
            add("feet_distance", np.maximum(0, f32(T.foot_distance_threshold) - d))
        if on("foot_acc"):
            acc = (feet_vel - sim["last_feet_vel"].reshape(N, F, 3)) / self.dt
            add("foot_acc", np.sum(acc ** 2, axis=(1, 2)))
        if on("foot_clearance"):                                           # :575-588
            vxy = np.linalg.norm(feet_vel[:, :, :2], axis=-1)
            zf = feet_pos[:, :, 2]
            if T.foot_clearance_ref == 1:                                     # go2_ee.py:136-150, go2_ts.py:146-160
                zf = zf - np.mean(sim["height_around_feet"].reshape(N, F, 9), axis=-1)
            elif T.foot_clearance_ref == 2:                                   # tron1_pf_ee.py:442-456, go2_cts.py:156-170
                zf = zf - np.max(sim["height_around_feet"].reshape(N, F, 9), axis=-1)
            err = np.sum(vxy * (zf - f32(T.foot_clearance_target) - f32(T.foot_height_offset)) ** 2, axis=-1)
            add("foot_clearance", np.exp(-err / f32(T.foot_clearance_sigma)))
        if JPL4 and on("quad_periodic_gait"):                              # tron1_sf.py:297-308 _reward_foot_flat (slot LG_R_FOOT_FLAT)
            Rf = foot_rotations(self.model, sim["base_quat"], dof_pos)         # rigid_body_states[:, feet, 3:7] as rotation matrices
            tilt = np.abs(Rf[:, :, 2, 0]) + np.abs(Rf[:, :, 2, 1])             # x, y of the world z axis seen from the foot
            add("quad_periodic_gait", np.sum(np.exp(-tilt.astype(f32) / f32(0.1)), axis=1))
        if on("foot_landing_vel"):
            zv = feet_vel[:, :, 2]
            land = ((feet_pos[:, :, 2] - f32(T.foot_height_offset)) < f32(T.about_landing_threshold)) & ~(feet_f[:, :, 2] > 0.1) & (zv < 0)
            add("foot_landing_vel", np.sum(np.where(land, zv, 0) ** 2, axis=1))
        if on("hip_pos"):
            if JPL4:                                                           # tron1_sf.py:281-285 _reward_hip_pos_zero_command
                add("hip_pos", np.sum((dof_pos[:, [1, 5]] - self.q0[[1, 5]]) ** 2, axis=1) * (cmd_xyz < 0.1))
            else:
                add("hip_pos", np.sum((dof_pos[:, 0::3] - self.q0[0::3]) ** 2, axis=1))
        if JPL4 and on("tracking_foot_clearance"):                         # tron1_sf.py:287-295 _reward_keep_ankle_pitch_zero_in_air
            c = feet_f[:, :, 2] > 1.0
            ank = np.abs(dof_pos[:, 3]) * ~c[:, 0] + np.abs(dof_pos[:, 7]) * ~c[:, 1]
            add("tracking_foot_clearance", np.exp(-np.abs(ank) / f32(0.2)))
        if on("keep_balance"):
            add("keep_balance", np.ones(N, f32))
        if on("lin_vel_z"):
            add("lin_vel_z", blv[:, 2] ** 2)
        if on("no_fly"):                                                   # tron1_pf.py:151-154
            add("no_fly", 1.0 * (np.sum(1.0 * (feet_f[:, :, 2] > f32(T.no_fly_contact_threshold)), axis=1) == 1))   # tron1_sf.py:275-278: 1.0
        if on("orientation"):
            add("orientation", np.sum(pg[:, :2] ** 2, axis=1))
        if not JPL4 and on("quad_periodic_gait"):                          # go2_wtw.py:377-484 ("step" indicator)
            acc = np.zeros(N, f32)
            b_swing = f32(T.b_swing) * f32(2 * np.pi)
            for i in range(4):
                q_frc = np.linalg.norm(feet_f[:, i], axis=-1)
                q_spd = np.linalg.norm(feet_vel[:, i], axis=-1)
                ph = np.remainder(self.phi[:, 0] + self.theta[:, i], f32(1.0)).astype(f32) * f32(2 * np.pi)
                swing = (ph >= 0) & (ph < b_swing)
                stance = (ph >= b_swing) & (ph < f32(2 * np.pi))
                c_frc = np.where(swing, -1.0, 0.0).astype(f32)
                c_spd = np.where(stance, -1.0, 0.0).astype(f32)
                # reference bug reproduced (go2_wtw.py:455-462): `(mask of shape (N,1)).nonzero().flatten()`
                # interleaves row and COLUMN indices, so index 0 is in both index lists whenever they are
                # non-empty: env 0 is overwritten as "swing", then as "stance" if any env is in stance.
                if swing.any():
                    c_frc[0], c_spd[0] = -1.0, 0.0
                if stance.any():
                    c_frc[0], c_spd[0] = 0.0, -1.0
                self.exp_C_frc[:, i] = c_frc
                acc = acc + (c_spd * q_spd + c_frc * q_frc).astype(f32)
            add("quad_periodic_gait", np.exp(acc))
        if on("torques"):
            add("torques", np.sum(sim["torques"] ** 2, axis=1))
        if on("tracking_ang_vel"):
            add("tracking_ang_vel", np.exp(-((cmd[:, 2] - bav[:, 2]) ** 2) / f32(T.tracking_sigma)))
        if on("tracking_base_height"):                                     # go2_wtw.py:495-500 (plane: heights are 0)
            if T.gait_mode == 1:
                d = sim["base_pos"][:, 2] - self.base_height_target[:, 0]
            else:                                                              # tron1_pf_ee.py:435-440
                d = np.mean(sim["base_pos"][:, 2:3] - sim["measured_heights"], axis=1) - f32(T.base_height_target)
            add("tracking_base_height", np.exp(-(d ** 2) / f32(T.base_height_sigma)))
        if not JPL4 and on("tracking_foot_clearance"):                     # go2_wtw.py:507-519
            vxy = np.linalg.norm(feet_vel[:, :, :2], axis=-1)
            err = np.sum(vxy * (feet_pos[:, :, 2] - self.foot_clearance_target - f32(T.foot_height_offset)) ** 2, axis=-1)
            add("tracking_foot_clearance", np.exp(-err / f32(T.foot_clearance_sigma)))
        if on("tracking_lin_vel"):
            add("tracking_lin_vel", np.exp(-np.sum((cmd[:, :2] - blv[:, :2]) ** 2, axis=1) / f32(T.tracking_sigma)))
        if on("tracking_orientation"):                                     # go2_wtw.py:502-505
            eul = get_euler_xyz(q)
            add("tracking_orientation", np.exp(-(eul[:, 0] ** 2 + (eul[:, 1] - self.pitch_target[:, 0]) ** 2) / f32(T.euler_sigma)))
        if T.only_positive_rewards:
            total = np.clip(total * cat_keep, 0, None)                        # go2_cat.py:219-223
        if on("termination"):
            k = R_["termination"]
            rew = (1.0 * (self.reset_buf & ~self.time_out_buf) * sc[k]).astype(f32)
            total = total + rew
            self.episode_sums[k] += rew
        self.rew_buf = total.astype(f32)
        if T.gait_mode in (1, 2):                                          # go2_wtw.py:29-36, tron1_pf_ee.py:28-35
            self.gait_time += self.dt
            over = self.gait_time >= (self.gait_period - self.dt / f32(2))
            self.gait_time[over] = 0
            if over.any():      # same (N,1)-mask .nonzero().flatten() bug as in the gait indicator (go2_wtw.py:33-34):
                self.gait_time[0] = 0   # env 0's clock restarts whenever ANY env's does
            self.phi = (self.gait_time / self.gait_period).astype(f32)
        # ---- reset_idx: :94-148 (command curriculum is applied by the caller beforehand on gate steps)
        ids = np.nonzero(self.reset_buf)[0]
        self.done_sums = None
        if len(ids):
            if self.cfg.commands.curriculum and counter % int(T.max_episode_length) == 0:   # :110-111
                self.update_command_curriculum(ids)
            if T.terrain_curriculum and counter > 0:                          # legged_robot.py:254-272, genesis_simulator.py:140-148
                dist = np.linalg.norm(sim["base_pos"][ids, :2] - self.env_origins[ids, :2], axis=1)
                up = dist > T.terrain_env_length / 2
                down = (dist < np.linalg.norm(self.commands[ids, :2], axis=1) * f32(T.episode_length_s) * f32(0.5)) & ~up
                lv = self.terrain_levels[ids] + up.astype(np.int64) - down.astype(np.int64)
                rnd = np.minimum(np.floor(R[ids, S.terrain_level] * T.max_terrain_level).astype(np.int64), T.max_terrain_level - 1)
                lv = np.where(lv >= T.max_terrain_level, rnd, np.clip(lv, 0, None))
                self.terrain_levels[ids] = lv
                self.env_origins[ids] = self.terrain_origins[lv, self.terrain_types[ids]]
            if T.gait_mode == 1:
                self._resample_behavior(ids, R, S.task_reset)
            self._resample(ids, R, S.reset_cmd)
            sit = T.sit_percent > 0 and R[ids[0], S.task_reset] < T.sit_percent      # one draw per call (quirk 11)
            lo = np.ctypeslib.as_array(T.reset_dof_lo)[:A]
            span = np.ctypeslib.as_array(T.reset_dof_span)[:A]
            if sit:                                                            # tron1_pf_ee.py:277-283
                sim["dof_pos"][ids] = np.ctypeslib.as_array(T.sit_dof_pos)[:A]
            else:
                sim["dof_pos"][ids] = self.q0 + (span * R[ids, S.reset_dof:S.reset_dof + A] + lo)   # go2.py:30-35
            sim["dof_vel"][ids] = 0
            pos = (np.ctypeslib.as_array(T.sit_pos) if sit else np.array(self.cfg.init_state.pos, f32)) + self.env_origins[ids]
            if T.custom_origins:
                pos[:, :2] += f32(T.reset_root_xy_span) * R[ids, S.reset_root_xy:S.reset_root_xy + 2] + f32(T.reset_root_xy_lo)
            sim["base_pos"][ids] = pos
            sim["base_quat"][ids] = np.ctypeslib.as_array(T.sit_quat) if sit else np.ctypeslib.as_array(T.base_init_quat)
            lv = f32(T.reset_lin_vel_span) * R[ids, S.reset_lin_vel:S.reset_lin_vel + 3] + f32(T.reset_lin_vel_lo)
            av = f32(T.reset_ang_vel_span) * R[ids, S.reset_ang_vel:S.reset_ang_vel + 3] + f32(T.reset_ang_vel_lo)
            if sit:
                lv, av = np.zeros_like(lv), np.zeros_like(av)
            sim["base_lin_vel_w"][ids], sim["base_ang_vel_w"][ids] = lv, av
            blv[ids], bav[ids] = lv, av                                    # genesis_simulator.py:128-129
            pg = quat_rotate_inverse(sim["base_quat"], g)                  # :125
            if T.dr_friction_on:
                self.friction_values[ids, 0] = f32(T.dr_friction_span) * R[ids, S.dr_friction] + f32(T.dr_friction_lo)
            if T.dr_mass_on:
                self.added_base_mass[ids, 0] = f32(T.dr_mass_span) * R[ids, S.dr_mass] + f32(T.dr_mass_lo)
            if T.dr_com_on:
                for k in range(3):
                    self.base_com_bias[ids, k] = f32(T.dr_com_span[k]) * R[ids, S.dr_com + k] + f32(T.dr_com_lo[k])
            if T.dr_joint_on:                                              # genesis_simulator.py:704-733
                self.joint_armature[ids, 0] = f32(T.dr_joint_span[0]) * R[ids, S.dr_joint] + f32(T.dr_joint_lo[0])
                self.joint_friction[ids, 0] = f32(T.dr_joint_span[1]) * R[ids, S.dr_joint + 1] + f32(T.dr_joint_lo[1])
                self.joint_damping[ids, 0] = f32(T.dr_joint_span[2]) * R[ids, S.dr_joint + 2] + f32(T.dr_joint_lo[2])
            if T.dr_pd_on:
                self.kp_scale[ids] = f32(T.dr_kp_span) * R[ids, S.dr_kp:S.dr_kp + A] + f32(T.dr_kp_lo)
                self.kd_scale[ids] = f32(T.dr_kd_span) * R[ids, S.dr_kd:S.dr_kd + A] + f32(T.dr_kd_lo)
            sim["last_dof_vel"][ids] = 0
            sim["last_feet_vel"].reshape(N, F, 3)[ids] = 0
            self.llast_actions[ids] = 0; self.last_actions[ids] = 0; self.actions[ids] = 0
            self.feet_air_time[ids] = 0
            self.episode_length_buf[ids] = 0
            self.fail_buf[ids] = 0
            if T.gait_mode == 1:                                           # go2_wtw.py:139-142, 174-178
                self.gait_time[ids] = 0; self.phi[ids] = 0; self.clock_input[ids] = 0
            if T.gait_mode == 2:                                           # tron1_pf_ee.py:220-226
                tt = np.ctypeslib.as_array(T.theta_table).reshape(4, 4)
                self.theta[ids, 0] = f32(tt[0, 0]) + R[ids, S.task_reset + 1]
                self.theta[ids, 1] = self.theta[ids, 0] + f32(tt[0, 1] - tt[0, 0])
                self.gait_time[ids, 0] = R[ids, S.task_reset + 2] * self.gait_period[ids, 0]
                self.phi[ids] = self.gait_time[ids] / self.gait_period[ids]
                self.clock_input[ids] = 0
            if T.obs_stack > 1:                                            # go2_wtw.py:174-178, legged_robot_ee.py:115-121
                self.obs_hist[ids] = 0; self.priv_hist[ids] = 0
            self.done_sums = (self.episode_sums[:, ids].sum(1), len(ids))
            if T.cat_enable:
                self.cstr_sums[:, ids] = 0
            self.episode_sums[:, ids] = np.where((sc[:abi.R_COUNT] != 0)[:, None], 0, self.episode_sums[:, ids])
        # ---- compute_observations: go2.py:40-64, clip legged_robot.py:48-49
        if T.obs_layout == abi.OBS_GO2:
            cs = np.array([T.obs_scale_lin_vel, T.obs_scale_lin_vel, T.obs_scale_ang_vel], f32)
            obs = np.concatenate([self.commands[:, :3] * cs, pg, bav * f32(T.obs_scale_ang_vel),
                                  (sim["dof_pos"] - self.q0) * f32(T.obs_scale_dof_pos),
                                  sim["dof_vel"] * f32(T.obs_scale_dof_vel), self.actions], axis=1).astype(f32)
            if T.add_noise:
                obs = obs + (f32(2) * R[:, S.noise:S.noise + obs.shape[1]] - f32(1)) * self.noise_vec
            self.obs_buf = np.clip(obs, -f32(T.clip_obs), f32(T.clip_obs)).astype(f32)
        elif T.obs_layout == abi.OBS_GO2_WTW:                                # go2_wtw.py:53-111, 251-256
            for i in range(4):
                ang = f32(2 * np.pi) * (self.phi[:, 0] + self.theta[:, i])
                self.clock_input[:, i] = np.sin(ang)
                self.clock_input[:, i + 4] = np.cos(ang)
            cs = np.array([T.obs_scale_lin_vel, T.obs_scale_lin_vel, T.obs_scale_ang_vel], f32)
            frame = np.concatenate([self.commands[:, :3] * cs, pg, bav * f32(T.obs_scale_ang_vel),
                                    (sim["dof_pos"] - self.q0) * f32(T.obs_scale_dof_pos),
                                    sim["dof_vel"] * f32(T.obs_scale_dof_vel), self.actions, self.clock_input,
                                    self.gait_period, self.base_height_target, self.foot_clearance_target,
                                    self.pitch_target, self.theta], axis=1).astype(f32)
            priv = np.concatenate([frame, blv * f32(T.obs_scale_lin_vel), self.rand_push_vels[:, :2], self.added_base_mass,
                                   self.friction_values, self.base_com_bias, self.kp_scale, self.kd_scale,
                                   self.exp_C_frc], axis=1).astype(f32)
            now = frame
            if T.add_noise:
                now = frame + (f32(2) * R[:, S.noise:S.noise + frame.shape[1]] - f32(1)) * self.noise_vec
            self.obs_hist = np.concatenate([self.obs_hist[:, 1:], now[:, None]], axis=1)
            self.priv_hist = np.concatenate([self.priv_hist[:, 1:], priv[:, None]], axis=1)
            co = f32(T.clip_obs)
            self.obs_buf = np.clip(self.obs_hist.reshape(N, -1), -co, co).astype(f32)
            self.priv_obs_buf = np.clip(self.priv_hist.reshape(N, -1), -co, co).astype(f32)
        elif T.obs_layout == abi.OBS_GO2_EE:                                 # go2_ee.py:10-75
            cs = np.array([T.obs_scale_lin_vel, T.obs_scale_lin_vel, T.obs_scale_ang_vel], f32)
            frame = np.concatenate([self.commands[:, :3] * cs, pg, bav * f32(T.obs_scale_ang_vel),
                                    (sim["dof_pos"] - self.q0) * f32(T.obs_scale_dof_pos),
                                    sim["dof_vel"] * f32(T.obs_scale_dof_vel), self.actions], axis=1).astype(f32)
            states = (1.0 * (np.linalg.norm(F_l[:, self.state_links], axis=-1) > 1.0)).astype(f32)
            dr = np.concatenate([self.friction_values - f32(T.friction_offset), self.added_base_mass, self.base_com_bias,
                                 self.rand_push_vels[:, :2], self.kp_scale - f32(T.kp_offset), self.kd_scale - f32(T.kd_offset)], axis=1)
            heights = np.clip(sim["base_pos"][:, 2:3] - f32(T.heights_offset) - sim["measured_heights"], -1, 1) * f32(T.obs_scale_height)
            crit = np.concatenate([frame, dr, states, heights], axis=1).astype(f32)
            now = frame
            if T.add_noise:
                now = frame + (f32(2) * R[:, S.noise:S.noise + frame.shape[1]] - f32(1)) * self.noise_vec
            self.obs_hist = np.concatenate([self.obs_hist[:, 1:], now[:, None]], axis=1)
            self.priv_hist = np.concatenate([self.priv_hist[:, 1:], crit[:, None]], axis=1)
            co = f32(T.clip_obs)
            self.obs_buf = np.clip(self.obs_hist.reshape(N, -1), -co, co).astype(f32)
            self.priv_obs_buf = np.clip(self.priv_hist.reshape(N, -1), -co, co).astype(f32)
            fh = np.clip(sim["feet_pos"].reshape(N, F, 3)[:, :, 2] - np.mean(sim["height_around_feet"].reshape(N, F, 9), axis=-1)
                         - f32(T.foot_height_offset), -1, 1)
            self.labels_buf = np.concatenate([blv * f32(T.obs_scale_lin_vel), states, fh], axis=1).astype(f32)
        elif T.obs_layout == abi.OBS_TRON1_EE:                               # tron1_pf_ee.py:53-141, 251-256
            for i in range(2):
                ang = f32(2 * np.pi) * (self.phi[:, 0] + self.theta[:, i])
                self.clock_input[:, i] = np.sin(ang)
                self.clock_input[:, i + 2] = np.cos(ang)
            cs = np.array([T.obs_scale_lin_vel, T.obs_scale_lin_vel, T.obs_scale_ang_vel], f32)
            frame = np.concatenate([self.commands[:, :3] * cs, pg, bav * f32(T.obs_scale_ang_vel),
                                    (sim["dof_pos"] - self.q0) * f32(T.obs_scale_dof_pos),
                                    sim["dof_vel"] * f32(T.obs_scale_dof_vel), self.actions, self.clock_input], axis=1).astype(f32)
            states = (1.0 * (np.linalg.norm(F_l[:, self.state_links], axis=-1) > 1.0)).astype(f32)
            dr = np.concatenate([self.friction_values - f32(T.friction_offset), self.added_base_mass, self.base_com_bias,
                                 self.rand_push_vels[:, :2], self.kp_scale - f32(T.kp_offset), self.kd_scale - f32(T.kd_offset),
                                 self.joint_armature, self.joint_friction, self.joint_damping], axis=1)
            heights = np.clip(sim["base_pos"][:, 2:3] - f32(T.heights_offset) - sim["measured_heights"], -1, 1) * f32(T.obs_scale_height)
            har = sim["height_around_feet"].reshape(N, F, 9)
            fpz = sim["feet_pos"].reshape(N, F, 3)[:, :, 2]
            rel = np.clip((fpz[:, :, None] - har).reshape(N, -1), -1.0, 1.0)
            crit = np.concatenate([frame, dr, self.exp_C_frc, states, heights, sim["normals"], rel], axis=1).astype(f32)
            now = frame
            if T.add_noise:
                now = frame + (f32(2) * R[:, S.noise:S.noise + frame.shape[1]] - f32(1)) * self.noise_vec
            self.obs_hist = np.concatenate([self.obs_hist[:, 1:], now[:, None]], axis=1)
            self.priv_hist = np.concatenate([self.priv_hist[:, 1:], crit[:, None]], axis=1)
            co = f32(T.clip_obs)
            self.obs_buf = np.clip(self.obs_hist.reshape(N, -1), -co, co).astype(f32)
            self.priv_obs_buf = np.clip(self.priv_hist.reshape(N, -1), -co, co).astype(f32)
            fh = np.clip(fpz - np.max(har, axis=-1) - f32(T.foot_height_offset), -1, 1)
            self.labels_buf = np.concatenate([blv * f32(T.obs_scale_lin_vel), states, fh, sim["normals"]], axis=1).astype(f32)
        elif T.obs_layout == abi.OBS_PROGRAM:   # go2_ts.py:5-86, go2_cts.py:11-87, go2_dreamwaq.py:7-84, go2_cat.py:19-95
            cs = np.array([T.obs_scale_lin_vel, T.obs_scale_lin_vel, T.obs_scale_ang_vel], f32)
            head = [self.commands[:, :3] * cs, pg, bav * f32(T.obs_scale_ang_vel), (sim["dof_pos"] - self.q0) * f32(T.obs_scale_dof_pos),
                    sim["dof_vel"] * f32(T.obs_scale_dof_vel)]
            frame = np.concatenate(head + [self.actions], axis=1).astype(f32)
            har = sim["height_around_feet"].reshape(N, F, 9) if "height_around_feet" in sim else np.zeros((N, F, 9), f32)
            fpz = sim["feet_pos"].reshape(N, F, 3)[:, :, 2]
            dr_base = lambda sc_: np.concatenate([self.friction_values - f32(T.friction_offset), self.added_base_mass, self.base_com_bias,
                                                  self.rand_push_vels[:, :2]], axis=1)
            blocks = {
                abi.SEG_LAST_ACTIONS: lambda sc_: self.last_actions,           # tron1_pf.py:36
                abi.SEG_DR_BASE: dr_base,                                      # tron1_pf.py:37-41
                abi.SEG_FEET_AIR_TIME: lambda sc_: self.feet_air_time,         # tron1_pf.py:42
                abi.SEG_FRAME: lambda sc_: frame,
                abi.SEG_KP: lambda sc_: self.kp_scale - f32(T.kp_offset),           # tron1_sf.py:40-41
                abi.SEG_KD: lambda sc_: self.kd_scale - f32(T.kd_offset),           # tron1_sf.py:42-43
                abi.SEG_DR: lambda sc_: np.concatenate([self.friction_values - f32(T.friction_offset), self.added_base_mass, self.base_com_bias,
                                                        self.rand_push_vels[:, :2], self.kp_scale - f32(T.kp_offset),
                                                        self.kd_scale - f32(T.kd_offset)], axis=1),
                abi.SEG_DR_JOINT: lambda sc_: np.concatenate([self.joint_armature, self.joint_friction, self.joint_damping], axis=1),
                abi.SEG_BASE_LIN_VEL: lambda sc_: blv * f32(T.obs_scale_lin_vel) * f32(sc_),
                abi.SEG_CONTACT_STATES: lambda sc_: (1.0 * (np.linalg.norm(F_l[:, self.state_links], axis=-1) > 1.0)).astype(f32),
                abi.SEG_HEIGHTS: lambda sc_: np.clip(sim["base_pos"][:, 2:3] - f32(T.heights_offset) - sim["measured_heights"], -1, 1) * f32(T.obs_scale_height),
                abi.SEG_FEET_REL_HEIGHTS: lambda sc_: np.clip((fpz[:, :, None] - har).reshape(N, -1), -1.0, 1.0),
                abi.SEG_FEET_HEIGHTS: lambda sc_: har.reshape(N, -1),
                abi.SEG_FEET_NORMALS: lambda sc_: sim["normals"],
                abi.SEG_FOOT_CLEARANCE: lambda sc_: np.clip(fpz - np.mean(har, axis=-1) - f32(T.foot_height_offset), -1, 1),
                abi.SEG_NEXT_STATE: lambda sc_: np.concatenate(head + [self.actions * f32(self.cfg.control.action_scale)], axis=1),
            }

            def run(prog):
                parts = [blocks[prog.kind[i]](prog.scale[i]).astype(f32) for i in range(prog.n_segs)]
                if not parts:
                    return np.zeros((N, 1), f32)
                off = 0
                for i, p_ in enumerate(parts):
                    assert prog.offset[i] == off, (i, prog.offset[i], off)
                    off += p_.shape[1]
                v = np.concatenate(parts, axis=1).astype(f32)
                return np.clip(v, -f32(T.clip_obs), f32(T.clip_obs)).astype(f32) if prog.clip else v
            crit = run(T.priv_prog)
            now = frame
            if T.add_noise:
                now = frame + (f32(2) * R[:, S.noise:S.noise + frame.shape[1]] - f32(1)) * self.noise_vec
            self.obs_hist = np.concatenate([self.obs_hist[:, 1:], now[:, None]], axis=1)
            self.priv_hist = np.concatenate([self.priv_hist[:, 1:], crit[:, None]], axis=1)
            self.obs_buf = self.obs_hist.reshape(N, -1).astype(f32)          # the history itself is not clipped (legged_robot_ts.py:71)
            self.priv_obs_buf = self.priv_hist.reshape(N, -1).astype(f32)
            self.labels_buf = run(T.labels_prog)
        else:
            raise NotImplementedError
        if T.double_shift:                                                   # go2_wtw.py:45-46
            self.llast_actions[:] = self.last_actions
            self.last_actions[:] = self.actions
        self.base_lin_vel, self.base_ang_vel, self.projected_gravity = blv, bav, pg
        return self.obs_buf, self.rew_buf, self.reset_buf

    def update_command_curriculum(self, ids):
        """legged_robot.py:336-348."""
        k = abi.REWARD_ID["tracking_lin_vel"]
        c = self.cfg.commands
        if np.mean(self.episode_sums[k][ids]) / self.task.max_episode_length > c.curriculum_threshold * self.scales[k]:
            self.command_ranges[0] = np.clip(self.command_ranges[0] - 0.5, -c.max_curriculum, 0.)
            self.command_ranges[1] = np.clip(self.command_ranges[1] + 0.5, 0., c.max_curriculum)
