"""ctypes mirror of include/lgsim.h (the C ABI) and the loader of the HIP library.

Field order and types must match the header exactly; tests/test_abi.py compiles a C
probe that prints sizeof/offsetof for every struct and compares them with these classes.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

MAX_JPL = 4
MAX_LEGS = 4
MAX_DOF = 12
MAX_BODIES = 13
MAX_LINKS = 24
MAX_SPHERES = 64
MAX_OBS = 192
NUM_REWARDS = 40
CMD_RANGE_FLOATS = 24
TASK_STATE_WTW = 22
TASK_STATE_BIPED = 12

PHASE_PRE, PHASE_SIM, PHASE_POST, PHASE_RESET, PHASE_ALL = 1, 2, 4, 8, 15
FAIL_NONFINITE = 1 << 20      # LG_FAIL_NONFINITE

# alphabetical evaluation order of the reference (helpers.py:10-25); index = enum LgReward
REWARD_NAMES = [
    "action_rate", "action_smoothness", "ang_vel_xy", "base_height", "biped_periodic_gait",
    "collision", "dof_acc", "dof_close_to_default", "dof_pos_limits", "dof_pos_stand_still",
    "dof_power", "dof_vel", "dof_vel_stand_still", "feet_air_time", "feet_contact_stand_still",
    "feet_distance", "foot_acc", "foot_clearance", "foot_landing_vel", "hip_pos", "keep_balance",
    "lin_vel_z", "no_fly", "orientation", "quad_periodic_gait", "torques", "tracking_ang_vel",
    "tracking_base_height", "tracking_foot_clearance", "tracking_lin_vel", "tracking_orientation",
    "termination",
]
R_COUNT = len(REWARD_NAMES)
REWARD_ID = {n: i for i, n in enumerate(REWARD_NAMES)}
# terms of the 4-joint sole-foot biped (tron1_sf.py:281-308) share the ids of quadruped-only terms (include/lgsim.h LG_R_FOOT_FLAT ...)
REWARD_ALIASES_JPL4 = {"hip_pos_zero_command": "hip_pos", "foot_flat": "quad_periodic_gait", "keep_ankle_pitch_zero_in_air": "tracking_foot_clearance"}


def reward_id(name, joints_per_leg=3):
    """enum LgReward value of a reward-scale name for a robot with that many joints per leg."""
    if joints_per_leg == 4 and name in REWARD_ALIASES_JPL4:
        return REWARD_ID[REWARD_ALIASES_JPL4[name]]
    if joints_per_leg == 4 and name in REWARD_ALIASES_JPL4.values():
        raise KeyError(f"reward term {name} does not exist for four-joint legs")
    return REWARD_ID[name]


def reward_names(joints_per_leg=3):
    """Names by enum LgReward value as they read for that robot (episode-sum keys)."""
    inv = {v: k for k, v in REWARD_ALIASES_JPL4.items()} if joints_per_leg == 4 else {}
    return [inv.get(n, n) for n in REWARD_NAMES]
OBS_GO2, OBS_GO2_WTW, OBS_GO2_EE, OBS_TRON1_EE, OBS_PROGRAM = 0, 1, 2, 3, 4

f32, i32, u32, i64, u64 = C.c_float, C.c_int32, C.c_uint32, C.c_int64, C.c_uint64
fp = C.POINTER(C.c_float)


class LgModelDesc(C.Structure):
    _fields_ = [
        ("n_legs", i32), ("n_bodies", i32), ("n_links", i32), ("n_spheres", i32),
        ("mass", f32 * MAX_BODIES), ("com", f32 * 3 * MAX_BODIES), ("inertia", f32 * 6 * MAX_BODIES),
        ("jpos", f32 * 3 * MAX_BODIES), ("jrot", f32 * 9 * MAX_BODIES), ("axis", f32 * 3 * MAX_BODIES),
        ("q_lo", f32 * MAX_DOF), ("q_hi", f32 * MAX_DOF), ("effort", f32 * MAX_DOF), ("vel_limit", f32 * MAX_DOF),
        ("armature", f32 * MAX_DOF), ("damping", f32 * MAX_DOF), ("frictionloss", f32 * MAX_DOF),
        ("link_body", i32 * MAX_LINKS), ("link_pos", f32 * 3 * MAX_LINKS),
        ("sph_body", i32 * MAX_SPHERES), ("sph_link", i32 * MAX_SPHERES),
        ("sph_pos", f32 * 3 * MAX_SPHERES), ("sph_r", f32 * MAX_SPHERES), ("sph_w", f32 * MAX_SPHERES),
        ("body_sph_start", i32 * (MAX_BODIES + 1)),
        ("foot_link", i32 * MAX_LEGS), ("foot_sphere", i32 * MAX_LEGS),
        ("term_link_mask", u32), ("pen_link_mask", u32), ("state_link_mask", u32),
    ]


class LgSimOptions(C.Structure):
    _fields_ = [
        ("dt", f32), ("decimation", i32), ("gravity_z", f32), ("contact_k", f32), ("contact_b", f32),
        ("terrain_friction", f32), ("limit_k", f32), ("limit_b", f32), ("contact_iters", i32), ("contact_margin", f32), ("limit_margin", f32),
        ("max_base_lin_vel", f32), ("max_base_ang_vel", f32), ("joint_vel_clamp", f32), ("action_scale", f32),
        ("kp", f32 * MAX_DOF), ("kd", f32 * MAX_DOF), ("default_dof_pos", f32 * MAX_DOF),
        ("base_init_pos", f32 * 3), ("bound_x", f32 * 2), ("bound_y", f32 * 2),
        ("terrain_rows", i32), ("terrain_cols", i32), ("hscale", f32), ("vscale", f32), ("border", f32),
        ("n_height_points", i32), ("feet_terrain_info", i32), ("sim_layout", i32), ("contact_w_every", i32),
    ]


class LgRandSlots(C.Structure):
    _fields_ = [(n, i32) for n in (
        "n_slots", "cb_cmd", "push", "reset_cmd", "reset_dof", "reset_root_xy", "reset_lin_vel",
        "reset_ang_vel", "dr_friction", "dr_mass", "dr_com", "dr_kp", "dr_kd", "dr_joint",
        "terrain_level", "task_cb", "task_reset", "noise")]


MAX_SEGS = 8
NUM_CSTR = 9
CR_ANY_FAST = 17
CSTR_NAMES = ["torque", "dof_vel", "action_rate", "base_height", "collision", "feet_stumble", "dof_pos", "base_orientation", "stand_still"]
(SEG_END, SEG_FRAME, SEG_DR, SEG_DR_JOINT, SEG_BASE_LIN_VEL, SEG_CONTACT_STATES, SEG_HEIGHTS, SEG_FEET_REL_HEIGHTS,
 SEG_FEET_HEIGHTS, SEG_FEET_NORMALS, SEG_FOOT_CLEARANCE, SEG_NEXT_STATE, SEG_LAST_ACTIONS, SEG_DR_BASE, SEG_FEET_AIR_TIME, SEG_KP,
 SEG_KD) = range(17)


class LgObsProgram(C.Structure):
    _fields_ = [("n_segs", i32), ("clip", i32), ("kind", i32 * MAX_SEGS), ("offset", i32 * MAX_SEGS), ("scale", f32 * MAX_SEGS)]


class LgTaskCfg(C.Structure):
    _fields_ = [
        ("obs_layout", i32), ("num_obs", i32), ("num_priv_obs", i32), ("obs_frame", i32), ("priv_frame", i32),
        ("obs_stack", i32), ("priv_stack", i32), ("obs_slack", i32), ("obs_sets", i32),
        ("control_dt", f32), ("clip_actions", f32), ("clip_obs", f32), ("max_episode_length", f32),
        ("fail_threshold", f32), ("max_projected_gravity", f32),
        ("resample_steps", i32), ("push_interval", i32), ("max_push_vel_xy", f32), ("heading_command", i32),
        ("yaw_clip", f32 * 2),
        ("reward_scales", f32 * NUM_REWARDS), ("soft_dof_lo", f32 * MAX_DOF), ("soft_dof_hi", f32 * MAX_DOF),
        ("only_positive_rewards", i32),
        ("tracking_sigma", f32), ("base_height_target", f32), ("foot_clearance_target", f32),
        ("foot_height_offset", f32), ("foot_clearance_sigma", f32), ("about_landing_threshold", f32),
        ("feet_air_time_threshold", f32), ("base_height_sigma", f32), ("euler_sigma", f32),
        ("foot_distance_threshold", f32), ("no_fly_contact_threshold", f32), ("air_time_cmd_dims", i32), ("foot_clearance_ref", i32),
        ("obs_scale_lin_vel", f32), ("obs_scale_ang_vel", f32), ("obs_scale_dof_pos", f32),
        ("obs_scale_dof_vel", f32), ("obs_scale_height", f32),
        ("add_noise", i32), ("noise_vec", f32 * MAX_OBS),
        ("reset_dof_lo", f32 * MAX_DOF), ("reset_dof_span", f32 * MAX_DOF),
        ("reset_root_xy_lo", f32), ("reset_root_xy_span", f32), ("custom_origins", i32),
        ("reset_lin_vel_lo", f32), ("reset_lin_vel_span", f32), ("reset_ang_vel_lo", f32), ("reset_ang_vel_span", f32),
        ("base_init_quat", f32 * 4),
        ("dr_friction_on", i32), ("dr_mass_on", i32), ("dr_com_on", i32), ("dr_pd_on", i32), ("dr_joint_on", i32),
        ("dr_friction_lo", f32), ("dr_friction_span", f32), ("dr_mass_lo", f32), ("dr_mass_span", f32),
        ("dr_com_lo", f32 * 3), ("dr_com_span", f32 * 3),
        ("dr_kp_lo", f32), ("dr_kp_span", f32), ("dr_kd_lo", f32), ("dr_kd_span", f32),
        ("dr_joint_lo", f32 * 3), ("dr_joint_span", f32 * 3),
        ("friction_offset", f32), ("kp_offset", f32), ("kd_offset", f32),
        ("terrain_curriculum", i32), ("max_terrain_level", i32), ("terrain_cols_n", i32),
        ("num_labels", i32), ("heights_offset", f32), ("heights_clip_scale", i32),
        ("terrain_env_length", f32), ("episode_length_s", f32),
        ("gait_mode", i32), ("double_shift", i32), ("behavior_resample_steps", i32), ("num_gait_max", i32),
        ("b_swing", f32), ("gait_period_fixed", f32), ("theta_table", f32 * 4 * 4),
        ("sit_percent", f32), ("sit_pos", f32 * 3), ("sit_quat", f32 * 4), ("sit_dof_pos", f32 * MAX_DOF),
        ("task_state_width", i32),
        ("priv_prog", LgObsProgram), ("labels_prog", LgObsProgram),
        ("cat_enable", i32), ("cat_soft_p", f32), ("cat_action_rate", f32), ("cat_min_base_height", f32),
        ("cat_max_projected_gravity", f32), ("dof_vel_limits", f32 * MAX_DOF),
        ("slots", LgRandSlots), ("seed", u64), ("env_id_offset", i64),
    ]


_BUF_FIELDS = [
    ("n_envs", i32),
    *[(n, C.c_void_p) for n in (
        "base_pos", "base_quat", "base_lin_vel_w", "base_ang_vel_w", "dof_pos", "dof_vel",
        "friction_values", "added_base_mass", "base_com_bias", "kp_scale", "kd_scale",
        "joint_armature", "joint_friction", "joint_damping", "rand_push_vels", "env_origins",
        "base_lin_vel", "base_ang_vel", "projected_gravity", "base_euler",
        "last_base_lin_vel", "last_base_ang_vel", "last_dof_vel", "last_feet_vel",
        "torques", "link_contact_forces", "feet_pos", "feet_vel",
        "link_contact_states", "measured_heights", "height_around_feet", "normal_vector_around_feet",
        "height_points", "terrain_levels", "terrain_types", "terrain_origins",
        "actions", "last_actions", "llast_actions", "commands",
        "feet_air_time", "last_contacts", "episode_length_buf", "fail_buf",
        "reset_buf", "time_out_buf",
        "rew_buf", "obs_buf", "priv_obs_buf", "labels_buf", "obs_dirty",
        "episode_sums", "episode_done_sums", "episode_done_step", "cstr_prob", "cstr_sums", "cstr_done_sums",
        "command_ranges", "task_state", "rand_in", "nonfinite_count")],
]


class LgBuffers(C.Structure):
    _fields_ = _BUF_FIELDS


BUFFER_NAMES = [n for n, _ in _BUF_FIELDS[1:]]


def fill_array(dst, src):
    """Copy a numpy array into a (possibly nested) ctypes array field."""
    a = np.ascontiguousarray(np.asarray(src)).reshape(-1)
    n = C.sizeof(dst) // 4
    if a.size > n:
        raise ValueError(f"array of {a.size} does not fit field of {n}")
    leaf = _leaf_type(type(dst))
    np_t = np.float32 if leaf is C.c_float else (np.uint32 if leaf is C.c_uint32 else np.int32)
    buf = np.zeros(n, dtype=np_t)
    buf[:a.size] = a.astype(np_t)
    C.memmove(dst, buf.ctypes.data, n * 4)


def _leaf_type(t):
    while hasattr(t, "_type_") and not isinstance(t._type_, str):
        t = t._type_
    return t


def model_desc(model, term_links=(), pen_links=(), state_links=()):
    """Pack a RobotModel (model_compiler.py) into LgModelDesc."""
    a = model.arrays
    d = LgModelDesc()
    d.n_legs, d.n_bodies = int(a["n_legs"]), int(a["n_bodies"])
    d.n_links, d.n_spheres = int(a["n_links"]), int(a["n_spheres"])
    for k in ("mass", "com", "inertia", "jpos", "jrot", "axis", "q_lo", "q_hi", "effort", "vel_limit", "armature",
              "damping", "frictionloss", "link_body", "link_pos", "sph_body", "sph_link", "sph_pos",
              "sph_r", "sph_w", "body_sph_start", "foot_link", "foot_sphere"):
        fill_array(getattr(d, k), a[k])
    mask = lambda ids: int(sum(1 << int(i) for i in ids))
    d.term_link_mask, d.pen_link_mask, d.state_link_mask = mask(term_links), mask(pen_links), mask(state_links)
    return d


_LIB = None
LIB_NAME = "liblgsim.so"


class HipExtensionMissing(RuntimeError):
    pass


def lib_path():
    # LG_LIB lets a developer point at an experimental build of the same ABI (still a HIP library, never a fallback)
    return os.environ.get("LG_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", LIB_NAME)


def load_lib():
    """dlopen the HIP engine.  There is no CPU fallback: a missing library is an error."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not os.path.exists(p):
        raise HipExtensionMissing(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). The product path has no CPU fallback.")
    lib = C.CDLL(p)
    H = C.c_void_p
    lib.lg_create.argtypes = [C.POINTER(LgModelDesc), C.POINTER(LgSimOptions), C.POINTER(LgTaskCfg), C.POINTER(H)]
    lib.lg_destroy.argtypes = [H]
    lib.lg_set_task.argtypes = [H, C.POINTER(LgTaskCfg)]
    lib.lg_set_terrain.argtypes = [H, C.c_void_p, i32, i32]
    lib.lg_bind.argtypes = [H, C.POINTER(LgBuffers)]
    lib.lg_step.argtypes = [H, u32, C.c_void_p, i64, C.c_void_p]
    lib.lg_time_steps.argtypes = [H, C.c_void_p, i64, i32, C.c_void_p, C.POINTER(C.c_float)]
    lib.lg_obs_window.argtypes = [H, C.POINTER(i32)]
    lib.lg_obs_set.argtypes = [H, C.POINTER(i32)]
    lib.lg_obs_set_select.argtypes = [H, i32]
    lib.lg_obs_window_select.argtypes = [H, i32]
    lib.lg_profile.argtypes = [H, i32]
    lib.lg_profile_read.argtypes = [H, C.POINTER(C.c_float), C.POINTER(i32)]
    lib.lg_philox.argtypes = [C.POINTER(u32 * 4), C.POINTER(u32 * 2), C.POINTER(u32 * 4)]
    lib.lg_philox.restype = C.c_int
    lib.lg_dpp_kat.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.lg_dpp_kat.restype = C.c_int
    lib.lg_stream_copy.argtypes = [C.c_void_p, C.c_void_p, i64, i32, C.c_void_p, C.POINTER(C.c_float)]
    lib.lg_stream_copy.restype = C.c_int
    lib.lg_terrain_generate.argtypes = [C.c_void_p, i32, C.c_void_p, C.c_void_p, C.c_void_p, i32, i32, i32, i32, i32, i32, C.c_double, C.c_void_p, C.c_void_p]
    lib.lg_terrain_generate.restype = C.c_int
    lib.lg_rollout_record.argtypes = [i32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, f32, C.c_void_p, C.c_void_p,
                                      C.POINTER(LgRowCopy), i32, C.c_void_p]
    lib.lg_rollout_gae.argtypes = [i32, i32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, f32, f32, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]
    lib.lg_rollout_record.restype = lib.lg_rollout_gae.restype = C.c_int
    lib.lg_last_error.restype = C.c_char_p
    lib.lg_last_kernel.argtypes = [H]
    lib.lg_last_kernel.restype = C.c_char_p
    lib.lg_abi_version.restype = C.c_int
    for f in ("lg_create", "lg_destroy", "lg_set_task", "lg_set_terrain", "lg_bind", "lg_step", "lg_time_steps", "lg_obs_window", "lg_obs_set", "lg_obs_set_select", "lg_obs_window_select", "lg_profile",
              "lg_profile_read"):
        getattr(lib, f).restype = C.c_int
    _LIB = lib
    return lib


EXPORTS = ["lg_create", "lg_destroy", "lg_set_task", "lg_set_terrain", "lg_bind", "lg_step",
           "lg_time_steps", "lg_obs_window", "lg_obs_set", "lg_obs_set_select", "lg_obs_window_select", "lg_profile", "lg_profile_read", "lg_philox", "lg_dpp_kat", "lg_stream_copy", "lg_terrain_generate", "lg_last_kernel", "lg_last_error",
           "lg_abi_version"]
ROLLOUT_EXPORTS = ["lg_rollout_record", "lg_rollout_gae"]          # include/lgrollout.h
ROLLOUT_MAX_COPIES = 8


TILE_SLOPE, TILE_UNIFORM, TILE_STAIRS, TILE_OBSTACLES = 0, 1, 2, 3


class LgTerrainTile(C.Structure):
    _fields_ = [("kind", i32), ("row", i32), ("col", i32), ("ip", i32 * 5), ("aux_off", i32)]


class LgRowCopy(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("width", i32), ("src_stride", i32)]


def check(rc, lib=None):
    if rc != 0:
        lib = lib or load_lib()
        raise RuntimeError("lgsim: " + lib.lg_last_error().decode())
