"""Configuration tree with the attribute names the reference's env/simulator code reads.

The reference keeps its configs as nested Python classes instantiated recursively
(legged_gym/envs/base/base_config.py:3-25, legged_robot_config.py, common_cfgs.py,
go2/go2_config.py).  Task code addresses them as ``cfg.<section>.<field>``; this module
offers the same addressing for the tasks in scope, declared with a small ``section`` helper
instead of class bodies.  Values are the reference's (file:line cited per block).

``cfg = GO2Cfg()`` gives an independent instance: every section is copied so that editing
``cfg.env.num_envs`` does not leak into other instances (base_config.py:20-25).
"""
from __future__ import annotations

import copy
import math

import numpy as np


class Section:
    """Bag of attributes; nested sections are Section subclasses."""

    def __init__(self):
        for k in dir(type(self)):
            if k.startswith("__"):
                continue
            v = getattr(type(self), k)
            if isinstance(v, type) and issubclass(v, Section):
                setattr(self, k, v())
            elif isinstance(v, (list, dict)):
                setattr(self, k, copy.deepcopy(v))

    def to_dict(self):
        """Sorted-key dict, nested (reference helpers.py:10-25 iterates dir() => alphabetical)."""
        out = {}
        for k in sorted(dir(self)):
            if k.startswith("_") or k == "to_dict":
                continue
            v = getattr(self, k)
            if callable(v) and not isinstance(v, Section):
                continue
            out[k] = v.to_dict() if isinstance(v, Section) else v
        return out


def section(*bases, **fields):
    bases = tuple(b for b in bases) or (Section,)
    return type("section", bases, dict(fields))


# ---------------------------------------------------------------------------------------------
# legged_robot_config.py:3-262 (defaults) -- only fields the hot path reads
class LeggedRobotCfg(Section):
    seed = 1                       # legged_robot_config.py:275
    env = section(
        num_envs=4096, num_observations=48, num_privileged_obs=None, num_actions=12,
        send_timeouts=True, episode_length_s=20, env_spacing=1.0, fail_to_terminal_time_s=0.1, debug=False)
    terrain = section(
        mesh_type="plane", plane_length=200.0, horizontal_scale=0.1, vertical_scale=0.005,
        border_size=5, border_height=1.0, curriculum=False, static_friction=1.0, dynamic_friction=1.0,
        restitution=0.0, obtain_terrain_info_around_feet=False, measure_heights=False,
        measured_points_x=[round(-0.8 + 0.1 * i, 1) for i in range(17)],
        measured_points_y=[round(-0.5 + 0.1 * i, 1) for i in range(11)],
        selected=False, terrain_kwargs=None, max_init_terrain_level=1,
        terrain_length=6.0, terrain_width=6.0, platform_size=3.0, num_rows=4, num_cols=4,
        terrain_proportions=[0.1, 0.1, 0.35, 0.25, 0.2], slope_treshold=0.75)
    init_state = section(
        pos=[0.0, 0.0, 1.0], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0],
        default_joint_angles={"joint_a": 0.0, "joint_b": 0.0})
    control = section(
        control_type="P", stiffness={"joint_a": 10.0, "joint_b": 15.0}, damping={"joint_a": 1.0, "joint_b": 1.5},
        action_scale=0.5, dt=0.02, decimation=4)
    asset = section(
        name=None, file="", foot_name="", penalize_contacts_on=[], terminate_after_contacts_on=[],
        fix_base_link=False, obtain_link_contact_states=False, contact_state_link_names=["thigh", "calf", "foot"],
        base_link_name="", self_collisions=0, dof_names=["joint_a", "joint_b"], links_to_keep=[],
        dof_vel_limits=[])
    rewards = section(
        scales=section(
            termination=-0.0, tracking_lin_vel=0, tracking_ang_vel=0, lin_vel_z=0, ang_vel_xy=0, orientation=-0.0,
            torques=0, dof_vel=-0.0, dof_acc=0, base_height=-0.0, feet_air_time=0, collision=0, feet_stumble=-0.0,
            action_rate=0, dof_pos_stand_still=-0.0),
        only_positive_rewards=True, tracking_sigma=0.25, soft_dof_pos_limit=1.0, soft_dof_vel_limit=1.0,
        soft_torque_limit=1.0, base_height_target=1.0, foot_clearance_target=0.04, foot_height_offset=0.0,
        foot_clearance_tracking_sigma=0.01, max_projected_gravity=-0.1)
    commands = section(
        curriculum=False, max_curriculum=1.0, num_commands=4, resampling_time=10.0, heading_command=True,
        curriculum_threshold=0.8,
        ranges=section(lin_vel_x=[-1.0, 1.0], lin_vel_y=[-1.0, 1.0], ang_vel_yaw=[-1, 1], heading=[-3.14, 3.14]))
    domain_rand = section(
        randomize_friction=True, friction_range=[0.5, 1.25], randomize_base_mass=True, added_mass_range=[-1.0, 1.0],
        push_robots=True, push_interval_s=15, max_push_vel_xy=1.0,
        randomize_com_displacement=True, com_pos_x_range=[-0.01, 0.01], com_pos_y_range=[-0.01, 0.01],
        com_pos_z_range=[-0.01, 0.01], randomize_ctrl_delay=False, ctrl_delay_step_range=[0, 1],
        randomize_pd_gain=False, kp_range=[0.8, 1.2], kd_range=[0.8, 1.2],
        randomize_joint_armature=False, joint_armature_range=[0.0, 0.05],
        randomize_joint_friction=False, joint_friction_range=[0.0, 0.1],
        randomize_joint_damping=False, joint_damping_range=[0.0, 1.0])
    normalization = section(
        obs_scales=section(lin_vel=1.0, ang_vel=0.25, dof_pos=1.0, dof_vel=0.05, height_measurements=5.0),
        clip_observations=100.0, clip_actions=100.0)
    noise = section(
        add_noise=True, noise_level=1.0,
        noise_scales=section(dof_pos=0.01, dof_vel=0.5, lin_vel=0.1, ang_vel=0.2, gravity=0.05, height_measurements=0.1))
    sensor = section(add_depth=False)
    viewer = section(ref_env=0, pos=[4.0, 4.0, 2.0], lookat=[0.0, 0.0, 0.0])
    sim = section(dt=0.005, substeps=1, max_collision_pairs=100, IK_max_targets=2, gravity=[0.0, 0.0, -9.81])
    # engine constants of this backend (no counterpart in the reference: Genesis' soft-constraint
    # parameters are internal to genesis-world).  See DESIGN.md "Contact model".
    hip = section(contact_stiffness=4.0e4, contact_damping=4.0e2, joint_limit_stiffness=5.0e3,
                  joint_limit_damping=5.0e1, contact_iters=2, contact_w_every=1, sim_layout=0, obs_history_slack=64, obs_sets=2, contact_margin=0.02, limit_margin=0.2,
                  max_base_lin_vel=50.0, max_base_ang_vel=40.0, joint_vel_clamp=2.0, seed=1)


_GO2_LEGS = ("FR", "FL", "RR", "RL")  # policy order, common_cfgs.py:52-65


# common_cfgs.py:10-71 (Go2FlatCommonCfg) + go2/go2_config.py:5-78 (GO2Cfg)
class GO2Cfg(LeggedRobotCfg):
    env = section(LeggedRobotCfg.env, num_envs=4096, num_observations=45, num_privileged_obs=None, num_actions=12)
    terrain = section(LeggedRobotCfg.terrain, mesh_type="plane")
    init_state = section(
        LeggedRobotCfg.init_state, pos=[0.0, 0.0, 0.42],
        default_joint_angles={f"{leg}_{j}_joint": a for leg in ("FL", "RL", "FR", "RR")
                              for j, a in (("hip", 0.0), ("thigh", 0.8), ("calf", -1.5))})
    control = section(LeggedRobotCfg.control, stiffness={"joint": 20.0}, damping={"joint": 0.5},
                      action_scale=0.25, dt=0.02, decimation=4)
    asset = section(
        LeggedRobotCfg.asset, name="go2", file="{LEGGED_GYM_ROOT_DIR}/resources/robots/go2/urdf/go2.urdf",
        foot_name="foot", penalize_contacts_on=["thigh", "calf"], terminate_after_contacts_on=["base", "Head"],
        base_link_name="base", dof_names=[f"{leg}_{j}_joint" for leg in _GO2_LEGS for j in ("hip", "thigh", "calf")],
        links_to_keep=["FL_foot", "FR_foot", "RL_foot", "RR_foot"],
        dof_vel_limits=[30.1, 30.1, 15.7] * 4)
    rewards = section(
        LeggedRobotCfg.rewards, soft_dof_pos_limit=0.9, base_height_target=0.36, foot_clearance_target=0.05,
        foot_height_offset=0.022, foot_clearance_tracking_sigma=0.01, only_positive_rewards=True,
        scales=section(
            LeggedRobotCfg.rewards.scales, dof_pos_limits=-1.0, collision=-1.0, tracking_lin_vel=1.0,
            tracking_ang_vel=0.5, lin_vel_z=-0.5, base_height=-2.0, ang_vel_xy=-0.05, orientation=-1.0,
            dof_vel=-5.0e-4, dof_acc=-2.0e-7, action_rate=-0.01, action_smoothness=-0.01, torques=-2.0e-4,
            feet_air_time=1.0, foot_clearance=0.5))
    commands = section(
        LeggedRobotCfg.commands, curriculum=True, max_curriculum=1.0, num_commands=4, resampling_time=10.0,
        heading_command=True,
        ranges=section(LeggedRobotCfg.commands.ranges, lin_vel_x=[-0.5, 0.5], lin_vel_y=[-1.0, 1.0],
                       ang_vel_yaw=[-1, 1], heading=[-3.14, 3.14]))
    domain_rand = section(
        LeggedRobotCfg.domain_rand, randomize_friction=True, friction_range=[0.5, 1.25],
        randomize_base_mass=True, added_mass_range=[-1.0, 1.0], push_robots=True, push_interval_s=15,
        max_push_vel_xy=1.0, randomize_com_displacement=True, com_pos_x_range=[-0.01, 0.01],
        com_pos_y_range=[-0.01, 0.01], com_pos_z_range=[-0.01, 0.01])
    # task specifics the reference hard-codes in go2.py:17-37,131-133
    reset = section(dof_ranges={"hip": 0.2, "thigh": 0.4, "calf": 0.4}, root_vel_range=0.0, robot="go2",
                    obs_layout="go2", feet_air_time_threshold=0.3)


def class_to_dict(obj):
    """helpers.py:10-25 equivalent for Section trees / plain objects."""
    if isinstance(obj, Section):
        return obj.to_dict()
    if not hasattr(obj, "__dict__"):
        return obj
    return {k: class_to_dict(getattr(obj, k)) for k in sorted(dir(obj)) if not k.startswith("_")}


def reward_scales_sorted(cfg):
    """name -> raw scale (before the x dt of legged_robot.py:416-421), alphabetical, zeros dropped."""
    d = class_to_dict(cfg.rewards.scales)
    return {k: float(v) for k, v in sorted(d.items()) if float(v) != 0.0}


def control_dt(cfg):
    return cfg.sim.dt * cfg.control.decimation


def max_episode_length(cfg):
    return float(np.ceil(cfg.env.episode_length_s / control_dt(cfg)))  # legged_robot.py:446


def pd_gains(cfg):
    """genesis_simulator.py:476-486: first stiffness key that is a substring of the dof name."""
    kp, kd = [], []
    for dn in cfg.asset.dof_names:
        for key in cfg.control.stiffness.keys():
            if key in dn:
                kp.append(cfg.control.stiffness[key])
                kd.append(cfg.control.damping[key])
    return np.array(kp, np.float32), np.array(kd, np.float32)


def default_dof_pos(cfg):
    return np.array([cfg.init_state.default_joint_angles[n] for n in cfg.asset.dof_names], np.float32)


def terrain_bounds(cfg):
    """genesis_simulator.py:278-294."""
    t = cfg.terrain
    if t.mesh_type in ("heightfield", "trimesh"):
        return ((-t.border_size + 1.0, t.border_size + t.num_rows * t.terrain_length - 1.0),
                (-t.border_size + 1.0, t.border_size + t.num_cols * t.terrain_width - 1.0))
    return ((-t.plane_length / 2 + 1, t.plane_length / 2 - 1),) * 2


def soft_dof_limits(model, cfg):
    """genesis_simulator.py:373-382."""
    lo, hi = model.arrays["q_lo"].astype(np.float32), model.arrays["q_hi"].astype(np.float32)
    m = (lo + hi) / 2
    r = hi - lo
    s = np.float32(cfg.rewards.soft_dof_pos_limit)
    return np.stack([m - np.float32(0.5) * r * s, m + np.float32(0.5) * r * s], axis=1).astype(np.float32)


# go2/go2_wtw/go2_wtw_config.py:5-104 (GO2WTWCfg): flat terrain, periodic-gait rewards, PD-gain DR,
# 5-frame observation / critic histories
class GO2WTWCfg(LeggedRobotCfg):
    env = section(LeggedRobotCfg.env, num_envs=4096, num_actions=12, frame_stack=5, c_frame_stack=5,
                  num_single_obs=61, num_observations=61 * 5, single_num_privileged_obs=61 + 38,
                  num_privileged_obs=5 * (61 + 38), env_spacing=1.0)
    terrain = section(GO2Cfg.terrain)
    init_state = section(GO2Cfg.init_state)
    control = section(GO2Cfg.control)
    asset = section(GO2Cfg.asset)
    rewards = section(
        LeggedRobotCfg.rewards, soft_dof_pos_limit=0.9, base_height_tracking_sigma=0.01, foot_height_offset=0.022,
        foot_clearance_tracking_sigma=0.01, euler_tracking_sigma=0.1, about_landing_threshold=0.03,
        only_positive_rewards=True,
        scales=section(
            LeggedRobotCfg.rewards.scales, dof_pos_limits=-10.0, collision=-1.0, tracking_lin_vel=1.0,
            tracking_ang_vel=0.5, tracking_base_height=0.6, tracking_orientation=0.6, tracking_foot_clearance=0.9,
            quad_periodic_gait=1.5, lin_vel_z=-0.5, ang_vel_xy=-0.05, dof_vel=-5.0e-4, dof_acc=-2.0e-7,
            action_rate=-0.01, action_smoothness=-0.01, torques=-2.0e-4, foot_landing_vel=-0.1, hip_pos=-1.0),
        periodic_reward_framework=section(
            gait_function_type="step", kappa=20, b_swing=0.5,
            # trot, pronk, pace, bound
            theta_fl_list=[0.0, 0.0, 0.5, 0.0], theta_fr_list=[0.5, 0.0, 0.0, 0.0],
            theta_rl_list=[0.5, 0.0, 0.5, 0.5], theta_rr_list=[0.0, 0.0, 0.0, 0.5]),
        behavior_params_range=section(
            resampling_time=5.0, gait_period_range=[0.3, 0.6], foot_clearance_target_range=[0.04, 0.12],
            base_height_target_range=[0.2, 0.34], pitch_target_range=[-0.3, 0.3]))
    commands = section(
        LeggedRobotCfg.commands, curriculum=True, max_curriculum=1.0, num_commands=4, resampling_time=8.0,
        heading_command=True,
        ranges=section(LeggedRobotCfg.commands.ranges, lin_vel_x=[-0.5, 0.5], lin_vel_y=[-1.0, 1.0],
                       ang_vel_yaw=[-1, 1], heading=[-3.14, 3.14]))
    domain_rand = section(
        LeggedRobotCfg.domain_rand, randomize_friction=True, friction_range=[0.2, 1.7], randomize_base_mass=True,
        added_mass_range=[-1.0, 1.0], push_robots=True, push_interval_s=15, max_push_vel_xy=1.0,
        randomize_com_displacement=True, com_pos_x_range=[-0.03, 0.03], com_pos_y_range=[-0.03, 0.03],
        com_pos_z_range=[-0.03, 0.03], randomize_pd_gain=True, kp_range=[0.8, 1.2], kd_range=[0.8, 1.2])
    # GO2WTW derives from LeggedRobot, not GO2: base-class reset distribution (legged_robot.py:274-298)
    reset = section(dof_ranges={"joint": 0.2}, root_vel_range=0.5, robot="go2", obs_layout="go2_wtw",
                    feet_air_time_threshold=0.3)


# common_cfgs.py:74-128 (Go2RoughCommonCfg) + go2/go2_ee/go2_ee_config.py:5-66 (Go2EECfg, experiment "go2_rough")
class GO2EECfg(LeggedRobotCfg):
    env = section(LeggedRobotCfg.env, num_envs=4096, num_single_obs=45, frame_stack=20, num_estimator_features=45 * 20,
                  num_estimator_labels=24, c_frame_stack=5, single_critic_obs_len=45 + 31 + 81 + 17,
                  num_privileged_obs=5 * (45 + 31 + 81 + 17), num_observations=45 * 20, num_actions=12, env_spacing=2.0)
    terrain = section(
        LeggedRobotCfg.terrain, mesh_type="heightfield", border_size=20.0, curriculum=True,
        obtain_terrain_info_around_feet=True, measure_heights=True,
        measured_points_x=[-0.4, -0.3, -0.2, -0.1, 0., 0.1, 0.2, 0.3, 0.4],
        measured_points_y=[-0.4, -0.3, -0.2, -0.1, 0., 0.1, 0.2, 0.3, 0.4],
        terrain_length=8.0, terrain_width=8.0, platform_size=4.0, num_rows=10, num_cols=10,
        terrain_proportions=[0.2, 0.1, 0.25, 0.25, 0.2])
    init_state = section(GO2Cfg.init_state)
    control = section(GO2Cfg.control)
    asset = section(GO2Cfg.asset, obtain_link_contact_states=True,
                    contact_state_link_names=["thigh", "calf", "foot", "base", "hip"],
                    penalize_contacts_on=["thigh", "calf", "base", "Head", "hip"], terminate_after_contacts_on=[])
    rewards = section(
        LeggedRobotCfg.rewards, soft_dof_pos_limit=0.9, foot_clearance_target=0.09, foot_height_offset=0.022,
        foot_clearance_tracking_sigma=0.01, only_positive_rewards=True,
        scales=section(
            LeggedRobotCfg.rewards.scales, dof_pos_limits=-2.0, collision=-1.0, tracking_lin_vel=1.0,
            tracking_ang_vel=0.5, lin_vel_z=-2.0, ang_vel_xy=-0.05, dof_power=-2.0e-4, dof_acc=-2.0e-7,
            action_rate=-0.01, action_smoothness=-0.01, feet_air_time=1.0, foot_clearance=0.2, hip_pos=-0.05,
            feet_contact_stand_still=0.5))
    commands = section(
        LeggedRobotCfg.commands, curriculum=True, max_curriculum=1.0, num_commands=4, resampling_time=10.0,
        heading_command=True,
        ranges=section(LeggedRobotCfg.commands.ranges, lin_vel_x=[-0.5, 0.5], lin_vel_y=[-1.0, 1.0],
                       ang_vel_yaw=[-1, 1], heading=[-3.14, 3.14]))
    domain_rand = section(
        LeggedRobotCfg.domain_rand, randomize_friction=True, friction_range=[0.2, 1.7], randomize_base_mass=True,
        added_mass_range=[-1.0, 1.0], push_robots=True, push_interval_s=10, max_push_vel_xy=1.0,
        randomize_com_displacement=True, com_pos_x_range=[-0.03, 0.03], com_pos_y_range=[-0.03, 0.03],
        com_pos_z_range=[-0.03, 0.03], randomize_pd_gain=True, kp_range=[0.8, 1.2], kd_range=[0.8, 1.2],
        randomize_joint_armature=False, randomize_joint_friction=False, randomize_joint_damping=False,
        joint_armature_range=[0.015, 0.025], joint_friction_range=[0.01, 0.02], joint_damping_range=[0.25, 0.3])
    # go2_ee.py:77-98 (per-joint reset ranges), legged_robot.py:283-298 (base-class root reset), :124-134 (air time 0.25)
    reset = section(dof_ranges={"hip": 0.2, "thigh": 0.4, "calf": 0.4}, root_vel_range=0.5, robot="go2",
                    obs_layout="go2_ee", feet_air_time_threshold=0.25, foot_clearance_ref="mean")


# tron1_pf/tron1_pf_config.py:4-122 (TRON1PFCfg, experiment "tron1_pf"): the point-foot biped on the plane, 5-frame actor / critic
# stacks (tron1_pf.py:15-66): actor frame 9 + 3 A = 27; critic frame [v_b 3 | frame 27 | last actions 6 | friction, mass, CoM, push 7 |
# feet air time 2] = 45
class TRON1PFCfg(LeggedRobotCfg):
    env = section(LeggedRobotCfg.env, num_envs=4096, num_single_obs=27, frame_stack=5, c_frame_stack=5, num_observations=27 * 5,
                  num_single_privileged_obs=27 + 18, num_privileged_obs=(27 + 18) * 5, num_actions=6, env_spacing=2.0)
    terrain = section(LeggedRobotCfg.terrain, mesh_type="plane")
    init_state = section(
        LeggedRobotCfg.init_state, pos=[0.0, 0.0, 0.8],
        default_joint_angles={f"{j}_{s}_Joint": 0.0 for s in ("L", "R") for j in ("abad", "hip", "knee", "foot")})
    control = section(LeggedRobotCfg.control, stiffness={"Joint": 42.0}, damping={"Joint": 2.5}, action_scale=0.25, decimation=4)
    asset = section(
        LeggedRobotCfg.asset, name="tron1_pf", file="{LEGGED_GYM_ROOT_DIR}/resources/robots/PF_TRON1A/urdf/robot.urdf",
        foot_name="foot", penalize_contacts_on=["knee", "hip"], terminate_after_contacts_on=["base", "abad"],
        base_link_name="base_Link", dof_names=[f"{j}_{s}_Joint" for s in ("L", "R") for j in ("abad", "hip", "knee")],
        links_to_keep=["foot_L_Link", "foot_R_Link"], dof_vel_limits=[])
    rewards = section(
        LeggedRobotCfg.rewards, soft_dof_pos_limit=0.9, base_height_target=0.68, foot_clearance_target=0.07, foot_height_offset=0.032,
        foot_clearance_tracking_sigma=0.01, foot_distance_threshold=0.115, about_landing_threshold=0.1, only_positive_rewards=False,
        scales=section(
            LeggedRobotCfg.rewards.scales, keep_balance=1.0, dof_pos_limits=-2.0, collision=-1.0, feet_distance=-100.0,
            tracking_lin_vel=1.0, tracking_ang_vel=0.5, lin_vel_z=-0.5, base_height=-2.0, ang_vel_xy=-0.05, orientation=-3.0,
            dof_vel=-5.0e-4, dof_acc=-2.0e-7, action_rate=-0.01, action_smoothness=-0.01, torques=-2.0e-5, feet_air_time=1.0,
            foot_clearance=0.5, no_fly=0.5, foot_landing_vel=-0.15))
    commands = section(
        LeggedRobotCfg.commands, curriculum=True, max_curriculum=1.0, num_commands=4, resampling_time=10.0, heading_command=True,
        ranges=section(LeggedRobotCfg.commands.ranges, lin_vel_x=[-0.5, 0.5], lin_vel_y=[-0.6, 0.6], ang_vel_yaw=[-1, 1],
                       heading=[-3.14, 3.14]))
    domain_rand = section(
        LeggedRobotCfg.domain_rand, randomize_friction=True, friction_range=[0.5, 1.25], randomize_base_mass=True,
        added_mass_range=[-1.0, 1.0], push_robots=True, push_interval_s=10, max_push_vel_xy=1.0, randomize_com_displacement=True,
        com_pos_x_range=[-0.03, 0.03], com_pos_y_range=[-0.03, 0.03], com_pos_z_range=[-0.03, 0.03])
    # TRON1PF derives from LeggedRobot: base-class reset distribution (legged_robot.py:274-298); air time 0.25 (tron1_pf.py:131-141)
    reset = section(dof_ranges={"Joint": 0.2}, root_vel_range=0.5, robot="tron1_pf", obs_layout="program", feet_air_time_threshold=0.25,
                    critic_program=[("base_lin_vel", 1.0), ("frame", 1.0), ("last_actions", 1.0), ("dr_base", 1.0), ("feet_air_time", 1.0)],
                    critic_clip=True, aux_program=[], aux_clip=False)


# tron1_sf/tron1_sf_config.py:5-163 (TRON1SFCfg) + tron1_sf.py: the 8-DOF sole-foot biped on the plane.  Four joints per leg (abad, hip,
# knee, ankle); the foot is the ankle body itself (foot_name "ankle", no kept links) with a 0.2 x 0.06 x 0.03 m box sole.
_SF_JOINTS = [f"{j}_{s}_Joint" for s in ("L", "R") for j in ("abad", "hip", "knee", "ankle")]


class TRON1SFCfg(LeggedRobotCfg):
    env = section(LeggedRobotCfg.env, num_envs=4096, num_single_obs=33, frame_stack=10, c_frame_stack=10, num_observations=33 * 10,
                  num_single_privileged_obs=33 + 39, num_privileged_obs=(33 + 39) * 10, num_actions=8, env_spacing=2.0)
    terrain = section(LeggedRobotCfg.terrain, mesh_type="plane")
    init_state = section(
        LeggedRobotCfg.init_state, pos=[0.0, 0.0, 0.85], default_joint_angles={n: 0.0 for n in _SF_JOINTS},
        sit_pos=[0.0, 0.0, 0.6],
        sit_joint_angles={"abad_L_Joint": 0.0, "hip_L_Joint": 0.58, "knee_L_Joint": 1.35, "ankle_L_Joint": -0.8,
                          "abad_R_Joint": 0.0, "hip_R_Joint": -0.58, "knee_R_Joint": -1.35, "ankle_R_Joint": -0.8},
        sit_pitch_angle=0.0, sit_init_percent=0.5)
    control = section(LeggedRobotCfg.control, stiffness={n: 45.0 for n in _SF_JOINTS},
                      damping={n: (0.8 if "ankle" in n else 1.5) for n in _SF_JOINTS}, action_scale=0.25, decimation=4)
    asset = section(
        LeggedRobotCfg.asset, name="tron1_sf", file="{LEGGED_GYM_ROOT_DIR}/resources/robots/SF_TRON1A/urdf/robot.urdf",
        foot_name="ankle", penalize_contacts_on=["knee", "hip", "base", "abad"], terminate_after_contacts_on=[],
        base_link_name="base_Link", dof_names=list(_SF_JOINTS), links_to_keep=[], dof_vel_limits=[])
    rewards = section(
        LeggedRobotCfg.rewards, soft_dof_pos_limit=0.9, base_height_target=0.75, foot_clearance_target=0.1, foot_height_offset=0.055,
        foot_clearance_tracking_sigma=0.01, foot_distance_threshold=0.115, about_landing_threshold=0.05, max_projected_gravity=-0.4,
        only_positive_rewards=False,
        scales=section(
            LeggedRobotCfg.rewards.scales, keep_balance=1.0, dof_pos_limits=-2.0, collision=-1.0, feet_distance=-100.0,
            tracking_lin_vel=1.0, tracking_ang_vel=1.0, lin_vel_z=-0.5, base_height=-4.0, ang_vel_xy=-0.05, orientation=-5.0,
            dof_power=-2.0e-4, dof_acc=-2.0e-7, action_rate=-0.01, action_smoothness=-0.01, feet_air_time=1.0, no_fly=0.4,
            foot_clearance=0.5, foot_landing_vel=-0.15, hip_pos_zero_command=-10.0, foot_flat=0.3))
    commands = section(
        LeggedRobotCfg.commands, curriculum=True, max_curriculum=1.0, num_commands=4, resampling_time=10.0, heading_command=True,
        ranges=section(LeggedRobotCfg.commands.ranges, lin_vel_x=[-0.5, 0.5], lin_vel_y=[-1.0, 1.0], ang_vel_yaw=[-1, 1],
                       heading=[-3.14, 3.14]))
    domain_rand = section(
        LeggedRobotCfg.domain_rand, randomize_friction=True, friction_range=[0.0, 2.0], randomize_base_mass=True,
        added_mass_range=[-0.5, 1.0], push_robots=True, push_interval_s=10, max_push_vel_xy=1.0, randomize_com_displacement=True,
        com_pos_x_range=[-0.03, 0.03], com_pos_y_range=[-0.03, 0.03], com_pos_z_range=[-0.03, 0.03], randomize_pd_gain=True,
        kp_range=[0.8, 1.2], kd_range=[0.8, 1.2], randomize_joint_armature=True, joint_armature_range=[0.11, 0.13],
        randomize_joint_friction=True, joint_friction_range=[0.00, 0.01], randomize_joint_damping=True, joint_damping_range=[1.4, 1.45])
    # tron1_sf.py:213-233 writes the reset offsets with the 6-DOF index pairs [0,3] [1,4] [2,5] [3,6] into the 8-DOF vector: dof 3 is
    # overwritten by the last pair and dof 7 is never set -- reproduced as the per-dof half-ranges below.  Sit pose: one coin per
    # reset_idx call (:160-166), no pitch.  Air time 0.25 gated by |commands[:, :3]| (:256-267); no_fly counts contacts above 1 N (:275-278)
    reset = section(dof_ranges={}, dof_range_list=[0.05, 0.2, 0.2, 0.2, 0.2, 0.2, 0.2, 0.0], root_vel_range=0.5, robot="tron1_sf",
                    obs_layout="program", feet_air_time_threshold=0.25, air_time_cmd_dims=3, no_fly_contact_threshold=1.0,
                    critic_program=[("base_lin_vel", 1.0), ("frame", 1.0), ("last_actions", 1.0), ("dr_base", 1.0), ("feet_air_time", 1.0),
                                    ("kp", 1.0), ("kd", 1.0), ("dr_joint", 1.0)],
                    critic_clip=True, aux_program=[], aux_clip=False)


# ---- the other Go2-rough task heads (legged_gym/envs/__init__.py:82-86): same robot, terrain, rewards, resets and domain
# randomisation as go2_ee (verified by diffing the reference's instantiated config trees); they differ in how the step's
# outputs are packaged.  Frames are described as observation programs (include/lgsim.h LgObsSeg): (block, extra scale).
# K = 17 contact-state links AS CONFIGURED (Go2RoughCommonCfg.asset.contact_state_link_names, common_cfgs.py:100-101); the
# reference's own size fields (go2_ts_config.py:8-14: 94 = 82 + 12, 172 = 160 + 12) were written for 12 links, so its critic
# deque starts with 172-wide zero frames and fills with 177-wide real ones.  Pinned here: what the reference EMITS per step
# (tests/golden/go2_ts_mdp.npz ...): privileged 99, critic frame 177; the history is 5 x 177 from the first step on.
_TS_CRITIC = [("frame", 1.0), ("dr", 1.0), ("base_lin_vel", 1.0), ("contact_states", 1.0), ("heights", 1.0)]


# go2/go2_ts/go2_ts_config.py:5-92 (Go2TSCfg) + go2_ts.py:5-86
class GO2TSCfg(GO2EECfg):
    env = section(GO2EECfg.env, num_observations=45, num_privileged_obs=31 + 36 + 12 + 3 + 17, frame_stack=20, num_history_obs=45 * 20,
                  num_latent_dims=31 + 36 + 12 + 3 + 17, c_frame_stack=5, single_critic_obs_len=45 + 31 + 3 + 17 + 81,
                  num_critic_obs=5 * (45 + 31 + 3 + 17 + 81), env_spacing=0.5)
    reset = section(GO2EECfg.reset, obs_layout="program", critic_program=_TS_CRITIC, critic_clip=False,
                    aux_program=[("dr", 1.0), ("feet_rel_heights", 1.0), ("feet_normals", 1.0), ("base_lin_vel", 1.0),
                                 ("contact_states", 1.0)], aux_clip=True)


# go2/go2_cts/go2_cts_config.py:5-82 (Go2CTSCfg) + go2_cts.py:11-95: privileged frame carries the raw heights around the feet
class GO2CTSCfg(GO2TSCfg):
    env = section(GO2TSCfg.env, num_teacher=4096 // 4 * 3, env_spacing=1.0)
    domain_rand = section(GO2TSCfg.domain_rand, joint_friction_range=[0.0, 0.1])
    # go2_cts.py:156-170: foot clearance above the MAX of the heights around the foot (TS / EE / Dreamwaq / CaT use the mean)
    reset = section(GO2TSCfg.reset, foot_clearance_ref="max",
                    aux_program=[("dr", 1.0), ("feet_heights", 1.0), ("feet_normals", 1.0), ("base_lin_vel", 1.0), ("contact_states", 1.0)])


# go2/go2_dreamwaq/go2_dreamwaq_config.py:5-87 (Go2DreamwaqCfg) + go2_dreamwaq.py:7-84: the critic stack is returned as the
# privileged observation (clipped), the auxiliary row is [explicit labels 24 | next state 45]
class GO2DreamwaqCfg(GO2EECfg):
    env = section(GO2EECfg.env, num_envs=3000, num_observations=45, frame_stack=20, num_history_obs=45 * 20, num_latent_dims=16,
                  num_explicit_dims=24, num_decoder_output=45, c_frame_stack=5, single_critic_obs_len=45 + 31 + 81 + 17 + 3,
                  num_privileged_obs=5 * (45 + 31 + 81 + 17 + 3), env_spacing=1.0)
    reset = section(GO2EECfg.reset, obs_layout="program",
                    critic_program=[("base_lin_vel", 1.0), ("frame", 1.0), ("dr", 1.0), ("contact_states", 1.0), ("heights", 1.0)],
                    critic_clip=True,
                    aux_program=[("base_lin_vel", 0.5), ("contact_states", 1.0), ("foot_clearance", 1.0), ("next_state", 1.0)],
                    aux_clip=False)


# go2/go2_cat/go2_cat_config.py:4-45 (Go2CaTCfg) + go2_cat.py: constraints as terminations on top of the TS packaging
class GO2CaTCfg(GO2TSCfg):
    env = section(GO2TSCfg.env, num_privileged_obs=34 + 36 + 12 + 17, num_latent_dims=34 + 36 + 12 + 17,
                  single_critic_obs_len=45 + 34 + 17 + 81, num_critic_obs=5 * (45 + 34 + 17 + 81))
    rewards = section(
        GO2TSCfg.rewards, soft_dof_pos_limit=0.9, base_height_target=0.34, foot_clearance_target=0.09, foot_height_offset=0.022,
        foot_clearance_tracking_sigma=0.01, only_positive_rewards=True,
        scales=section(GO2TSCfg.rewards.scales, dof_pos_limits=0.0, collision=0.0, dof_pos_stand_still=0.0, lin_vel_z=-1.0,
                       orientation=-0.5, hip_pos=-0.2, dof_close_to_default=-0.05, foot_clearance=0.2))
    constraints = section(enable="cat", tau_constraint=0.95, soft_p=0.25,
                          limits=section(action_rate=100.0, max_projected_gravity=-0.1, min_base_height=0.25))
    normalization = section(GO2TSCfg.normalization, clip_actions=10.0)
    reset = section(GO2TSCfg.reset,
                    critic_program=[("frame", 1.0), ("dr", 1.0), ("dr_joint", 1.0), ("contact_states", 1.0), ("heights", 1.0)],
                    aux_program=[("dr", 1.0), ("dr_joint", 1.0), ("feet_heights", 1.0), ("feet_normals", 1.0), ("contact_states", 1.0)])


# tron1_pf/tron1_pf_ee/tron1_pf_ee_config.py:4-174 (TRON1PF_EECfg, experiment "tron1_pf_rough"): 6-DOF point-foot biped
class TRON1PFEECfg(LeggedRobotCfg):
    env = section(LeggedRobotCfg.env, num_envs=4096, num_single_obs=31, frame_stack=10, num_estimator_features=310,
                  num_estimator_labels=17, c_frame_stack=10, single_critic_obs_len=31 + 22 + 49 + 6 + 2 + 24,
                  num_privileged_obs=10 * (31 + 22 + 49 + 6 + 2 + 24), num_observations=310, num_actions=6, env_spacing=3.0)
    terrain = section(
        LeggedRobotCfg.terrain, mesh_type="heightfield", border_size=15.0, curriculum=True,
        obtain_terrain_info_around_feet=True, measure_heights=True,
        measured_points_x=[-0.3, -0.2, -0.1, 0., 0.1, 0.2, 0.3], measured_points_y=[-0.3, -0.2, -0.1, 0., 0.1, 0.2, 0.3],
        terrain_length=8.0, terrain_width=8.0, platform_size=4.0, num_rows=10, num_cols=10,
        terrain_proportions=[0.2, 0.2, 0.2, 0.2, 0.2])
    init_state = section(
        LeggedRobotCfg.init_state, pos=[0.0, 0.0, 0.83],
        default_joint_angles={f"{j}_{s}_Joint": 0.0 for s in ("L", "R") for j in ("abad", "hip", "knee", "foot")},
        sit_pos=[0.0, 0.0, 0.55],
        sit_joint_angles={"abad_L_Joint": 0.0, "hip_L_Joint": 0.6, "knee_L_Joint": 1.36, "foot_L_Joint": 0.0,
                          "abad_R_Joint": 0.0, "hip_R_Joint": -0.6, "knee_R_Joint": -1.36, "foot_R_Joint": 0.0},
        sit_pitch_angle=-0.2, sit_init_percent=0.7)
    control = section(LeggedRobotCfg.control, stiffness={"Joint": 42.0}, damping={"Joint": 2.5}, action_scale=0.25,
                      decimation=4, dt=0.02)
    asset = section(
        LeggedRobotCfg.asset, name="tron1_pf", file="{LEGGED_GYM_ROOT_DIR}/resources/robots/PF_TRON1A/urdf/robot.urdf",
        obtain_link_contact_states=True, contact_state_link_names=["hip", "knee", "foot"], foot_name="foot",
        penalize_contacts_on=["knee", "hip"], terminate_after_contacts_on=["base", "abad"], base_link_name="base_Link",
        dof_names=[f"{j}_{s}_Joint" for s in ("L", "R") for j in ("abad", "hip", "knee")],
        links_to_keep=["foot_L_Link", "foot_R_Link"], dof_vel_limits=[])
    rewards = section(
        LeggedRobotCfg.rewards, soft_dof_pos_limit=0.95, base_height_target=0.75, foot_clearance_target=0.06,
        foot_height_offset=0.032, foot_clearance_tracking_sigma=0.01, base_height_tracking_sigma=0.01,
        foot_distance_threshold=0.115, only_positive_rewards=False, max_projected_gravity=-0.2,
        scales=section(
            LeggedRobotCfg.rewards.scales, keep_balance=1.0, dof_pos_limits=-2.0, collision=-1.0, feet_distance=-100.0,
            tracking_lin_vel=1.0, tracking_ang_vel=0.5, tracking_base_height=0.3, lin_vel_z=-0.5, ang_vel_xy=-0.05,
            orientation=-4.0, dof_power=-2.0e-4, dof_acc=-2.0e-7, foot_acc=-1.0e-5, action_rate=-0.01,
            action_smoothness=-0.01, biped_periodic_gait=1.0, foot_clearance=0.5),
        periodic_reward_framework=section(gait_function_type="step", kappa=20, b_swing=0.5, theta_left=0.0,
                                          theta_right=0.5, gait_period=0.5))
    commands = section(
        LeggedRobotCfg.commands, curriculum=True, max_curriculum=0.8, num_commands=4, resampling_time=10.0,
        heading_command=True,
        ranges=section(LeggedRobotCfg.commands.ranges, lin_vel_x=[-0.5, 0.5], lin_vel_y=[-0.6, 0.6],
                       ang_vel_yaw=[-1, 1], heading=[-3.14, 3.14]))
    domain_rand = section(
        LeggedRobotCfg.domain_rand, randomize_friction=True, friction_range=[0.0, 1.7], randomize_base_mass=True,
        added_mass_range=[-1.0, 2.0], push_robots=True, push_interval_s=10, max_push_vel_xy=1.0,
        randomize_com_displacement=True, com_pos_x_range=[-0.03, 0.03], com_pos_y_range=[-0.03, 0.03],
        com_pos_z_range=[-0.03, 0.03], randomize_pd_gain=True, kp_range=[0.8, 1.2], kd_range=[0.8, 1.2],
        randomize_joint_armature=True, joint_armature_range=[0.11, 0.13], randomize_joint_friction=True,
        joint_friction_range=[0.00, 0.01], randomize_joint_damping=True, joint_damping_range=[1.4, 1.45])
    normalization = section(LeggedRobotCfg.normalization, clip_actions=20.0)
    # tron1_pf_ee.py:258-275 (per-joint reset ranges), base-class root reset, sit-pose branch :204-210, 277-310
    reset = section(dof_ranges={"abad": 0.2, "hip": 0.4, "knee": 0.4}, root_vel_range=0.5, robot="tron1_pf",
                    obs_layout="tron1_ee", feet_air_time_threshold=0.3, foot_clearance_ref="max")
