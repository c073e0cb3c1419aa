"""cfg + RobotModel -> the C-ABI structs (LgModelDesc / LgSimOptions / LgTaskCfg).

Each derived constant cites where the reference derives the same number.
"""
from __future__ import annotations

import os

import math

import numpy as np

from . import abi
from . import config as cfgmod


def make_model_desc(model, cfg):
    """Link-index sets by substring match, genesis_simulator.py:333-363."""
    term = model.find_link_indices(cfg.asset.terminate_after_contacts_on)
    pen = model.find_link_indices(cfg.asset.penalize_contacts_on)
    state = model.find_link_indices(cfg.asset.contact_state_link_names) if cfg.asset.obtain_link_contact_states else []
    return abi.model_desc(model, term, pen, state)


def make_sim_options(model, cfg, terrain=None):
    o = abi.LgSimOptions()
    o.dt = cfg.sim.dt
    o.decimation = cfg.control.decimation
    o.gravity_z = -9.81  # engine default, identical to cfg.sim.gravity (legged_robot_config.py:254)
    o.contact_k, o.contact_b = cfg.hip.contact_stiffness, cfg.hip.contact_damping
    o.terrain_friction = cfg.terrain.static_friction  # genesis_simulator.py:276
    o.limit_k, o.limit_b = cfg.hip.joint_limit_stiffness, cfg.hip.joint_limit_damping
    o.contact_iters = cfg.hip.contact_iters
    o.contact_w_every = int(os.environ.get("LG_W_EVERY", getattr(cfg.hip, "contact_w_every", 1)))   # env override: developer timing only
    o.sim_layout = int(os.environ.get("LG_SIM_LAYOUT", getattr(cfg.hip, "sim_layout", 0)))   # env override: developer timing only
    o.contact_margin, o.limit_margin = cfg.hip.contact_margin, cfg.hip.limit_margin
    o.max_base_lin_vel, o.max_base_ang_vel = cfg.hip.max_base_lin_vel, cfg.hip.max_base_ang_vel
    o.joint_vel_clamp = cfg.hip.joint_vel_clamp
    o.action_scale = cfg.control.action_scale
    kp, kd = cfgmod.pd_gains(cfg)
    abi.fill_array(o.kp, kp)
    abi.fill_array(o.kd, kd)
    abi.fill_array(o.default_dof_pos, cfgmod.default_dof_pos(cfg))
    abi.fill_array(o.base_init_pos, cfg.init_state.pos)
    bx, by = cfgmod.terrain_bounds(cfg)
    abi.fill_array(o.bound_x, bx)
    abi.fill_array(o.bound_y, by)
    if terrain is not None:
        o.terrain_rows, o.terrain_cols = int(terrain.tot_rows), int(terrain.tot_cols)
    o.hscale, o.vscale, o.border = cfg.terrain.horizontal_scale, cfg.terrain.vertical_scale, cfg.terrain.border_size
    if cfg.terrain.measure_heights and cfg.terrain.mesh_type != "plane":
        o.n_height_points = len(cfg.terrain.measured_points_x) * len(cfg.terrain.measured_points_y)
        o.feet_terrain_info = int(bool(cfg.terrain.obtain_terrain_info_around_feet))
    return o


def go2_noise_vec(cfg, A=12):
    """go2.py:92-117 for the 9 + 3 A wide frame (45 for Go2; tron1_pf.py:105-128 for A = 6)."""
    ns, lvl, sc = cfg.noise.noise_scales, cfg.noise.noise_level, cfg.normalization.obs_scales
    v = np.zeros(9 + 3 * A, np.float32)
    v[3:6] = ns.gravity * lvl
    v[6:9] = ns.ang_vel * lvl * sc.ang_vel
    v[9:9 + A] = ns.dof_pos * lvl * sc.dof_pos
    v[9 + A:9 + 2 * A] = ns.dof_vel * lvl * sc.dof_vel
    return v


def go2_slots(A=12, obs_frame=45):
    s = abi.LgRandSlots()
    off = 0
    for name, w in (("cb_cmd", 3), ("push", 2), ("reset_cmd", 3), ("reset_dof", A), ("reset_root_xy", 2),
                    ("reset_lin_vel", 3), ("reset_ang_vel", 3), ("dr_friction", 1), ("dr_mass", 1), ("dr_com", 3),
                    ("dr_kp", A), ("dr_kd", A), ("dr_joint", 3), ("terrain_level", 1), ("task_cb", 5), ("task_reset", 5),
                    ("noise", obs_frame)):
        setattr(s, name, off)
        off += w
    s.n_slots = off
    return s


def _span(lo, hi):
    # torch_rand_float (math_utils.py:79-81): (upper - lower) * rand + lower, scalars cast to f32
    return np.float32(lo), np.float32(hi - lo)


def make_task_cfg(model, cfg, seed=None, env_id_offset=0):
    t = abi.LgTaskCfg()
    dt = cfgmod.control_dt(cfg)
    layout = cfg.reset.obs_layout
    A = model.n_dof
    t.obs_layout = {"go2": abi.OBS_GO2, "go2_wtw": abi.OBS_GO2_WTW, "go2_ee": abi.OBS_GO2_EE,
                    "tron1_ee": abi.OBS_TRON1_EE, "program": abi.OBS_PROGRAM}[layout]
    if layout == "go2":
        t.num_obs, t.obs_frame, t.obs_stack = 45, 45, 1
        t.num_priv_obs = t.priv_frame = t.priv_stack = 0
        abi.fill_array(t.noise_vec, go2_noise_vec(cfg))
        t.slots = go2_slots(A, 45)
    elif layout == "go2_wtw":
        e = cfg.env
        t.obs_frame, t.obs_stack, t.num_obs = e.num_single_obs, e.frame_stack, e.num_observations
        t.priv_frame, t.priv_stack, t.num_priv_obs = e.single_num_privileged_obs, e.c_frame_stack, e.num_privileged_obs
        nv = np.zeros(e.num_single_obs, np.float32)          # go2_wtw.py:266-293
        nv[:45] = go2_noise_vec(cfg)
        abi.fill_array(t.noise_vec, nv)
        t.slots = go2_slots(A, e.num_single_obs)
        prf, bp = cfg.rewards.periodic_reward_framework, cfg.rewards.behavior_params_range
        t.gait_mode, t.double_shift = 1, 1
        t.behavior_resample_steps = int(bp.resampling_time / dt)           # go2_wtw.py:260-261
        t.num_gait_max = len(prf.theta_fl_list)
        t.b_swing = prf.b_swing
        for g in range(t.num_gait_max):   # feet_indices order FL, FR, RL, RR (go2_wtw.py:382-407)
            for k, lst in enumerate((prf.theta_fl_list, prf.theta_fr_list, prf.theta_rl_list, prf.theta_rr_list)):
                t.theta_table[g][k] = lst[g]
        t.task_state_width = abi.TASK_STATE_WTW
    elif layout == "go2_ee":
        e = cfg.env
        t.obs_frame, t.obs_stack, t.num_obs = e.num_single_obs, e.frame_stack, e.num_estimator_features
        t.priv_frame, t.priv_stack, t.num_priv_obs = e.single_critic_obs_len, e.c_frame_stack, e.num_privileged_obs
        t.num_labels = e.num_estimator_labels
        abi.fill_array(t.noise_vec, go2_noise_vec(cfg))            # go2_ee.py:100-122
        t.slots = go2_slots(A, e.num_single_obs)
        t.heights_offset, t.heights_clip_scale = 0.5, 1             # go2_ee.py:45-46
    elif layout == "program":
        # go2_ts / go2_cts / go2_dreamwaq / go2_cat: go2's actor frame, stacked `frame_stack` deep; critic frame and the
        # auxiliary per-step output assembled from blocks (config.py GO2TSCfg ...)
        e = cfg.env
        K = len(model.find_link_indices(cfg.asset.contact_state_link_names)) if cfg.asset.obtain_link_contact_states else 0
        P = len(cfg.terrain.measured_points_x) * len(cfg.terrain.measured_points_y) if cfg.terrain.measure_heights else 0
        F = model.n_legs
        FW = 9 + 3 * A
        width = {"frame": FW, "dr": 7 + 2 * A, "dr_joint": 3, "base_lin_vel": 3, "contact_states": K, "heights": P,
                 "feet_rel_heights": 9 * F, "feet_heights": 9 * F, "feet_normals": 3 * F, "foot_clearance": F, "next_state": FW,
                 "last_actions": A, "dr_base": 7, "feet_air_time": F, "kp": A, "kd": A}
        kind = {"frame": abi.SEG_FRAME, "dr": abi.SEG_DR, "dr_joint": abi.SEG_DR_JOINT, "base_lin_vel": abi.SEG_BASE_LIN_VEL,
                "contact_states": abi.SEG_CONTACT_STATES, "heights": abi.SEG_HEIGHTS, "feet_rel_heights": abi.SEG_FEET_REL_HEIGHTS,
                "feet_heights": abi.SEG_FEET_HEIGHTS, "feet_normals": abi.SEG_FEET_NORMALS, "foot_clearance": abi.SEG_FOOT_CLEARANCE,
                "next_state": abi.SEG_NEXT_STATE, "last_actions": abi.SEG_LAST_ACTIONS, "dr_base": abi.SEG_DR_BASE,
                "feet_air_time": abi.SEG_FEET_AIR_TIME, "kp": abi.SEG_KP, "kd": abi.SEG_KD}

        def fill(prog, blocks, clip):
            assert len(blocks) <= abi.MAX_SEGS
            off = 0
            for i, (name, scale) in enumerate(blocks):
                prog.kind[i], prog.offset[i], prog.scale[i] = kind[name], off, scale
                off += width[name]
            prog.n_segs, prog.clip = len(blocks), int(clip)
            return off
        t.obs_frame, t.obs_stack, t.num_obs = FW, e.frame_stack, FW * e.frame_stack
        t.priv_frame = fill(t.priv_prog, cfg.reset.critic_program, cfg.reset.critic_clip)
        t.priv_stack, t.num_priv_obs = e.c_frame_stack, e.c_frame_stack * t.priv_frame
        t.num_labels = fill(t.labels_prog, cfg.reset.aux_program, cfg.reset.aux_clip)
        want = getattr(e, "single_critic_obs_len", None) or e.num_single_privileged_obs
        assert t.priv_frame == want, (t.priv_frame, want)
        abi.fill_array(t.noise_vec, go2_noise_vec(cfg, A))         # go2_ts.py:110-131, tron1_pf.py:105-128
        t.slots = go2_slots(A, FW)
        t.heights_offset, t.heights_clip_scale = 0.5, 1             # go2_ts.py:50-52
        ini = cfg.init_state
        if getattr(ini, "sit_init_percent", 0.0) > 0.0:            # tron1_sf.py:160-166, 235-254: batch-wide coin, no pitch
            t.sit_percent = ini.sit_init_percent
            abi.fill_array(t.sit_pos, ini.sit_pos)
            abi.fill_array(t.sit_quat, cfg.init_state.rot)
            abi.fill_array(t.sit_dof_pos, [ini.sit_joint_angles[n] for n in cfg.asset.dof_names])
    elif layout == "tron1_ee":
        e = cfg.env
        t.obs_frame, t.obs_stack, t.num_obs = e.num_single_obs, e.frame_stack, e.num_estimator_features
        t.priv_frame, t.priv_stack, t.num_priv_obs = e.single_critic_obs_len, e.c_frame_stack, e.num_privileged_obs
        t.num_labels = e.num_estimator_labels
        # tron1_pf_ee.py:322-342 writes the 12-DOF slices into a 31-wide vector: q AND qd get
        # the dof_pos scale, actions and clock get the dof_vel scale
        ns, lvl, sc = cfg.noise.noise_scales, cfg.noise.noise_level, cfg.normalization.obs_scales
        nv = np.zeros(e.num_single_obs, np.float32)
        nv[3:6] = ns.gravity * lvl
        nv[6:9] = ns.ang_vel * lvl * sc.ang_vel
        nv[9:21] = ns.dof_pos * lvl * sc.dof_pos
        nv[21:33] = ns.dof_vel * lvl * sc.dof_vel
        abi.fill_array(t.noise_vec, nv)
        t.slots = go2_slots(A, e.num_single_obs)
        t.heights_offset, t.heights_clip_scale = 0.6, 1           # tron1_pf_ee.py:96-98
        prf, ini = cfg.rewards.periodic_reward_framework, cfg.init_state
        t.gait_mode, t.double_shift = 2, 1
        t.b_swing, t.gait_period_fixed = prf.b_swing, prf.gait_period
        t.theta_table[0][0], t.theta_table[0][1] = prf.theta_left, prf.theta_right
        t.task_state_width = abi.TASK_STATE_BIPED
        t.sit_percent = ini.sit_init_percent
        abi.fill_array(t.sit_pos, ini.sit_pos)
        hp = 0.5 * ini.sit_pitch_angle                           # quat_from_euler_xyz(0, pitch, 0), math_utils.py:112-124
        abi.fill_array(t.sit_quat, [0.0, np.float32(np.sin(np.float32(hp))), 0.0, np.float32(np.cos(np.float32(hp)))])
        abi.fill_array(t.sit_dof_pos, [ini.sit_joint_angles[n] for n in cfg.asset.dof_names])
    else:
        raise NotImplementedError(layout)
    t.control_dt = dt
    t.clip_actions = cfg.normalization.clip_actions
    t.clip_obs = cfg.normalization.clip_observations
    t.max_episode_length = cfgmod.max_episode_length(cfg)
    t.fail_threshold = cfg.env.fail_to_terminal_time_s / dt          # legged_robot.py:90
    t.max_projected_gravity = cfg.rewards.max_projected_gravity
    t.resample_steps = int(cfg.commands.resampling_time / dt)        # legged_robot.py:305
    t.push_interval = int(np.ceil(cfg.domain_rand.push_interval_s / dt)) if cfg.domain_rand.push_robots else 0
    t.max_push_vel_xy = cfg.domain_rand.max_push_vel_xy
    t.heading_command = int(cfg.commands.heading_command)
    abi.fill_array(t.yaw_clip, cfg.commands.ranges.ang_vel_yaw)
    scales = np.zeros(abi.NUM_REWARDS, np.float32)
    for name, s in cfgmod.reward_scales_sorted(cfg).items():
        try:
            k = abi.reward_id(name, model.joints_per_leg)
        except KeyError:
            raise NotImplementedError(f"reward term {name!r} has no kernel implementation") from None
        scales[k] = s * dt                                             # legged_robot.py:421 (python float math)
    abi.fill_array(t.reward_scales, scales)
    soft = cfgmod.soft_dof_limits(model, cfg)
    abi.fill_array(t.soft_dof_lo, soft[:, 0])
    abi.fill_array(t.soft_dof_hi, soft[:, 1])
    r = cfg.rewards
    t.only_positive_rewards = int(r.only_positive_rewards)
    t.tracking_sigma, t.base_height_target = r.tracking_sigma, r.base_height_target
    t.foot_clearance_target, t.foot_height_offset = r.foot_clearance_target, r.foot_height_offset
    t.foot_clearance_sigma = r.foot_clearance_tracking_sigma
    t.about_landing_threshold = getattr(r, "about_landing_threshold", 0.0)
    t.feet_air_time_threshold = cfg.reset.feet_air_time_threshold
    t.base_height_sigma = getattr(r, "base_height_tracking_sigma", 0.01)
    t.euler_sigma = getattr(r, "euler_tracking_sigma", 0.1)
    t.foot_distance_threshold = getattr(r, "foot_distance_threshold", 0.0)
    t.no_fly_contact_threshold = getattr(cfg.reset, "no_fly_contact_threshold", 0.1)     # tron1_pf.py:151-154 | tron1_sf.py:275-278
    t.air_time_cmd_dims = getattr(cfg.reset, "air_time_cmd_dims", 2)                    # legged_robot.py:553 | tron1_sf.py:264
    t.foot_clearance_ref = {"none": 0, "mean": 1, "max": 2}[getattr(cfg.reset, "foot_clearance_ref", "none")]
    sc = cfg.normalization.obs_scales
    t.obs_scale_lin_vel, t.obs_scale_ang_vel = sc.lin_vel, sc.ang_vel
    t.obs_scale_dof_pos, t.obs_scale_dof_vel, t.obs_scale_height = sc.dof_pos, sc.dof_vel, sc.height_measurements
    t.add_noise = int(cfg.noise.add_noise)
    # reset distribution: go2.py:17-37 (additive per joint type), go2.py:119-134
    lo, span = np.zeros(A, np.float32), np.zeros(A, np.float32)
    for i, dn in enumerate(cfg.asset.dof_names):
        for key, rng in cfg.reset.dof_ranges.items():
            if key in dn:
                lo[i], span[i] = _span(-rng, rng)
    for i, rng in enumerate(getattr(cfg.reset, "dof_range_list", [])):   # per-dof half-ranges (tron1_sf.py:213-233)
        lo[i], span[i] = _span(-rng, rng)
    abi.fill_array(t.reset_dof_lo, lo)
    abi.fill_array(t.reset_dof_span, span)
    t.reset_root_xy_lo, t.reset_root_xy_span = _span(-0.5, 0.5)        # legged_robot.py:288
    t.custom_origins = int(cfg.terrain.mesh_type in ("heightfield", "trimesh"))
    rv = cfg.reset.root_vel_range
    t.reset_lin_vel_lo, t.reset_lin_vel_span = _span(-rv, rv)
    t.reset_ang_vel_lo, t.reset_ang_vel_span = _span(-rv, rv)
    abi.fill_array(t.base_init_quat, cfg.init_state.rot)
    d = cfg.domain_rand
    t.dr_friction_on, t.dr_mass_on, t.dr_com_on = int(d.randomize_friction), int(d.randomize_base_mass), int(d.randomize_com_displacement)
    t.dr_pd_on = int(d.randomize_pd_gain)
    t.dr_joint_on = int(d.randomize_joint_armature or d.randomize_joint_friction or d.randomize_joint_damping)
    t.dr_friction_lo, t.dr_friction_span = _span(*d.friction_range)
    t.dr_mass_lo, t.dr_mass_span = _span(*d.added_mass_range)
    for i, rg in enumerate((d.com_pos_x_range, d.com_pos_y_range, d.com_pos_z_range)):
        t.dr_com_lo[i], t.dr_com_span[i] = _span(*rg)
    t.dr_kp_lo, t.dr_kp_span = _span(*d.kp_range)
    t.dr_kd_lo, t.dr_kd_span = _span(*d.kd_range)
    for i, rg in enumerate((d.joint_armature_range, d.joint_friction_range, d.joint_damping_range)):
        t.dr_joint_lo[i], t.dr_joint_span[i] = _span(*rg)
    t.friction_offset = (d.friction_range[0] + d.friction_range[1]) / 2  # legged_robot.py:448-453
    t.kp_offset = (d.kp_range[0] + d.kp_range[1]) / 2
    t.kd_offset = (d.kd_range[0] + d.kd_range[1]) / 2
    t.terrain_curriculum = int(cfg.terrain.curriculum and t.custom_origins)  # legged_robot.py:443-444
    t.max_terrain_level = cfg.terrain.num_rows
    t.terrain_cols_n = cfg.terrain.num_cols
    t.terrain_env_length = cfg.terrain.terrain_length
    t.episode_length_s = cfg.env.episode_length_s
    cst = getattr(cfg, "constraints", None)                          # go2_cat_config.py:24-33
    if cst is not None and cst.enable == "cat":
        t.cat_enable, t.cat_soft_p = 1, cst.soft_p
        t.cat_action_rate, t.cat_min_base_height = cst.limits.action_rate, cst.limits.min_base_height
        t.cat_max_projected_gravity = cst.limits.max_projected_gravity
        t.double_shift = 1                                             # go2_cat.py:119-122: second history shift at the end of the step
    dvl = list(getattr(cfg.asset, "dof_vel_limits", [])) or [1e30] * A
    abi.fill_array(t.dof_vel_limits, np.asarray(dvl, np.float32))
    t.seed = int(cfg.hip.seed if seed is None else seed)
    t.env_id_offset = int(env_id_offset)
    # history stacks: slide a window over rows with slack instead of moving the history every step (288 GB of HBM
    # buys bandwidth: one new frame written per step, one compaction every `obs_history_slack` steps)
    # two copies of the observation outputs, written alternately: the tensors one step() hands out survive the next step()
    # (rsl_rl holds them across env.step(), ppo.py:103-104; the reference's step() returns fresh tensors, legged_robot.py:48-49)
    t.obs_sets = int(os.environ.get("LG_OBS_SETS", getattr(cfg.hip, "obs_sets", 2)))              # env: tests only
    if t.obs_stack > 1 or t.priv_stack > 1:
        slack = int(os.environ.get("LG_OBS_SLACK", getattr(cfg.hip, "obs_history_slack", 64)))   # env: tests / timing only
        t.obs_slack = max(slack, int(t.obs_stack) + (t.obs_sets > 1), int(t.priv_stack) + (t.obs_sets > 1)) if slack > 0 else 0
    return t
