/* lgsim.h -- C ABI of the MI355X-native legged-locomotion environment engine.
 *
 * This is the boundary a maintainer of the reference (LeggedGym-Ex,
 * oscar-youngquist/HCR_Genesis_LR_CL) binds with ctypes to obtain a
 * `SIMULATOR=hip` backend.  Every entry point replaces a piece of the
 * reference's Python/Genesis path; the reference file:line each one stands in
 * for is cited next to it.  No torch types cross this boundary: plain device
 * pointers, sizes, and the caller's HIP stream (as void*).
 *
 * Ownership: the caller (PyTorch on the Python side) owns every per-env buffer
 * named in LgBuffers; the library allocates only the small per-handle constant
 * tables (model, options, task config).  All buffers are float32 row-major
 * (N, ...) unless stated otherwise; quaternions are xyzw (reference
 * legged_gym/envs/base/legged_robot_config.py:59-61).
 *
 * Error convention: every call returns 0 on success, non-zero on failure;
 * lg_last_error() returns a thread-local message (reference raises Python
 * exceptions: genesis_simulator.py:271,275,356).  Kernel launch errors surface
 * on the call that enqueued them (hipGetLastError after launch).
 */
#ifndef LGSIM_H
#define LGSIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LG_MAX_JPL 4      /* joints per leg (serial chain): (n_bodies - 1) / n_legs = 3 (go2, TRON1 point foot) or 4 (TRON1 sole foot) */
#define LG_MAX_LEGS 4
#define LG_MAX_DOF 12
#define LG_MAX_BODIES 13  /* floating base + one body per dof */
#define LG_MAX_LINKS 24   /* reported links (bodies + kept fixed links such as feet) */
#define LG_MAX_SPHERES 64
#define LG_MAX_OBS 192    /* widest single observation frame handled in-kernel */
#define LG_NUM_REWARDS 40
#define LG_CMD_RANGE_FLOATS 24 /* [0..7] command ranges; [8..15] wtw behaviour-parameter ranges; [16] num_gaits */
#define LG_TASK_STATE_WTW 22   /* gait_time, phi, gait_period, base_h_tgt, clr_tgt, pitch_tgt, theta[4], clock[8], exp_C_frc[4] */
#define LG_NUM_CSTR 9          /* go2_cat.py:143-205: torque, dof_vel, action_rate, base_height | collision, feet_stumble, dof_pos,
                                 base_orientation | stand_still (soft | hard | style) */
#define LG_CR_ANY_FAST 17      /* command_ranges[17 + (step counter & 1)]: 1 when SOME env of the whole job moves a joint faster than 4 rad/s.
                                 Set by the physics launch of a step (LgTaskCfg.cat_enable), read by its MDP launch, which also clears the
                                 other slot for the next step; a sharded job max-reduces the slot over its ranks between the two launches */
#define LG_TASK_STATE_BIPED 12 /* gait_time, phi, gait_period(unused), pad, theta[2], clock[4], exp_C_frc[2] */

/* ---- robot model (output of hcr_genesis_lr_cl_amd/model_compiler.py) -------------------
 * Stands in for gs.morphs.URDF(merge_fixed_links=True, links_to_keep=feet)
 * (genesis_simulator.py:303-313).  Body 0 is the floating base, body 1+d carries dof d
 * (policy order, cfg.asset.dof_names); with J = (n_bodies - 1) / n_legs joints per leg, leg l owns dofs J l .. J l + J - 1.
 * The foot of a leg is a reported link on the chain's last body: a kept fixed link carrying one sphere (go2, TRON1 point foot)
 * or, for 4-joint legs, the last body itself (TRON1 sole foot, tron1_sf_config.py:82-84) whose foot_sphere is the sole centre. */
typedef struct LgModelDesc {
    int32_t n_legs, n_bodies, n_links, n_spheres;
    float mass[LG_MAX_BODIES];
    float com[LG_MAX_BODIES][3];      /* body frame */
    float inertia[LG_MAX_BODIES][6];  /* about com, body axes: xx yy zz xy xz yz */
    float jpos[LG_MAX_BODIES][3];     /* joint origin in parent body frame */
    float jrot[LG_MAX_BODIES][9];     /* child->parent rotation at q=0, row-major */
    float axis[LG_MAX_BODIES][3];     /* unit joint axis, child frame */
    float q_lo[LG_MAX_DOF], q_hi[LG_MAX_DOF], effort[LG_MAX_DOF], vel_limit[LG_MAX_DOF];
    float armature[LG_MAX_DOF], damping[LG_MAX_DOF], frictionloss[LG_MAX_DOF];
    int32_t link_body[LG_MAX_LINKS];  /* body each reported link moves with */
    float link_pos[LG_MAX_LINKS][3];  /* link frame origin in that body's frame */
    int32_t sph_body[LG_MAX_SPHERES], sph_link[LG_MAX_SPHERES];
    float sph_pos[LG_MAX_SPHERES][3]; /* body frame */
    float sph_r[LG_MAX_SPHERES];
    float sph_w[LG_MAX_SPHERES];      /* conservative point inverse mass [1/kg] */
    int32_t body_sph_start[LG_MAX_BODIES + 1];
    int32_t foot_link[LG_MAX_LEGS], foot_sphere[LG_MAX_LEGS];
    uint32_t term_link_mask;          /* links whose |F|>10 terminates (legged_robot.py:81-83) */
    uint32_t pen_link_mask;           /* links counted by _reward_collision (legged_robot.py:505-512) */
    uint32_t state_link_mask;         /* links reported in link_contact_states (genesis_simulator.py:53-55) */
} LgModelDesc;

/* ---- physics options: gs.options.SimOptions/RigidOptions (genesis_simulator.py:231-257)
 * plus the PD law of _compute_torques (genesis_simulator.py:630-642). */
typedef struct LgSimOptions {
    float dt;              /* physics sub-step, cfg.sim.dt = 0.005 */
    int32_t decimation;    /* cfg.control.decimation = 4 */
    float gravity_z;       /* -9.81 */
    float contact_k;       /* penalty normal stiffness  [N/m] */
    float contact_b;       /* penalty normal damping    [N s/m] */
    float terrain_friction;/* cfg.terrain.static_friction (genesis_simulator.py:276) */
    float limit_k;         /* joint-limit stop stiffness [N m/rad] */
    float limit_b;         /* joint-limit stop damping   [N m s/rad] */
    int32_t contact_iters; /* block-Jacobi sweeps over the feet (>=1) */
    float contact_margin;  /* [m]  gap below which an approaching sphere is already decelerated */
    float limit_margin;    /* [rad] same for joint-limit stops */
    /* velocity sanitisation (the reference's asset options carry max_angular_velocity /
     * max_linear_velocity = 1000, legged_robot_config.py:110-111; here tighter so that
     * omega*dt stays inside the explicit integrator's stable range) */
    float max_base_lin_vel, max_base_ang_vel, joint_vel_clamp; /* joint: multiple of URDF velocity limit */
    float action_scale;    /* cfg.control.action_scale */
    float kp[LG_MAX_DOF], kd[LG_MAX_DOF];
    float default_dof_pos[LG_MAX_DOF];
    float base_init_pos[3];
    float bound_x[2], bound_y[2]; /* out-of-terrain teleport box (genesis_simulator.py:278-294,612-628) */
    /* heightfield (genesis_simulator.py:765-778); mesh_type plane => terrain_rows = 0 */
    int32_t terrain_rows, terrain_cols;
    float hscale, vscale, border;
    /* height sampling grid around the base (genesis_simulator.py:496-507,552-577) */
    int32_t n_height_points;   /* 0 = measure_heights off */
    int32_t feet_terrain_info; /* cfg.terrain.obtain_terrain_info_around_feet */
    int32_t sim_layout;        /* physics kernel layout: 0 auto (by batch size), 1 one leg per lane, 2 one vector component per lane */
    int32_t contact_w_every;   /* the feet's 3x3 operational-space inverse inertias W (three unit-force response sweeps per foot) are
                                * recomputed on sub-steps 0, k, 2k ... of a control step (every foot, in contact or not) and reused as they
                                * are in between; <= 1: every sub-step.  W only preconditions the block-Jacobi sweeps -- their fixed point
                                * (the contact law on the EXACT response of the robot) does not depend on it */
} LgSimOptions;

/* ---- reward term ids, in the alphabetical order the reference evaluates them
 * (legged_gym/utils/helpers.py:10-25 class_to_dict iterates dir() => sorted names;
 *  legged_robot.py:411-434).  scale 0 => term inactive. */
enum LgReward {
    LG_R_ACTION_RATE = 0, LG_R_ACTION_SMOOTHNESS, LG_R_ANG_VEL_XY, LG_R_BASE_HEIGHT,
    LG_R_BIPED_PERIODIC_GAIT, LG_R_COLLISION, LG_R_DOF_ACC, LG_R_DOF_CLOSE_TO_DEFAULT,
    LG_R_DOF_POS_LIMITS, LG_R_DOF_POS_STAND_STILL, LG_R_DOF_POWER, LG_R_DOF_VEL,
    LG_R_DOF_VEL_STAND_STILL, LG_R_FEET_AIR_TIME, LG_R_FEET_CONTACT_STAND_STILL,
    LG_R_FEET_DISTANCE, LG_R_FOOT_ACC, LG_R_FOOT_CLEARANCE, LG_R_FOOT_LANDING_VEL,
    LG_R_HIP_POS, LG_R_KEEP_BALANCE, LG_R_LIN_VEL_Z, LG_R_NO_FLY, LG_R_ORIENTATION,
    LG_R_QUAD_PERIODIC_GAIT, LG_R_TORQUES, LG_R_TRACKING_ANG_VEL, LG_R_TRACKING_BASE_HEIGHT,
    LG_R_TRACKING_FOOT_CLEARANCE, LG_R_TRACKING_LIN_VEL, LG_R_TRACKING_ORIENTATION,
    LG_R_TERMINATION, /* added after the positive clip (legged_robot.py:163-168) */
    LG_R_COUNT
};
/* Terms that only exist for the 4-joint sole-foot biped (tron1_sf.py:281-308) share the ids of terms that only exist for
 * quadrupeds; which meaning an id has follows from the model (joints per leg), the episode-sum / scale arrays are unchanged. */
#define LG_R_HIP_POS_ZERO_COMMAND LG_R_HIP_POS                          /* tron1_sf.py:281-285 */
#define LG_R_FOOT_FLAT LG_R_QUAD_PERIODIC_GAIT                          /* tron1_sf.py:297-308 */
#define LG_R_KEEP_ANKLE_PITCH_ZERO_IN_AIR LG_R_TRACKING_FOOT_CLEARANCE  /* tron1_sf.py:287-295 */

/* observation layouts (reference compute_observations overrides) */
enum LgObsLayout {
    LG_OBS_GO2 = 0,      /* go2.py:40-56        45 */
    LG_OBS_GO2_WTW = 1,  /* go2_wtw.py:53-111   61x5 / 99x5 */
    LG_OBS_GO2_EE = 2,   /* go2_ee.py:10-75     45x20 / 174x5 / 24 labels */
    LG_OBS_TRON1_EE = 3, /* tron1_pf_ee.py:53-141 31x10 / 134x10 / 17 labels */
    LG_OBS_PROGRAM = 4   /* actor frame = go2's 9 + 3 A (45; 27 for tron1_pf) x stack; critic frame and labels assembled by LgTaskCfg.priv_prog / labels_prog */
};

/* Observation programs (obs_layout == LG_OBS_PROGRAM): the critic frame and the auxiliary per-step output ("labels") of the
 * Go2-rough task heads are concatenations of the same few blocks in different orders (go2_ts.py:16-86, go2_cts.py:13-87,
 * go2_dreamwaq.py:7-84, go2_cat.py:19-95).  A program lists the blocks with the offset each one starts at. */
enum LgObsSeg {
    LG_SEG_END = 0,
    LG_SEG_FRAME,         /* the actor frame without noise (obs_frame wide) */
    LG_SEG_DR,            /* friction - offset, added mass, CoM 3, push xy 2, kp - offset A, kd - offset A (go2_ts.py:26-37) */
    LG_SEG_DR_JOINT,      /* joint armature, frictionloss, damping (go2_cat.py:40-42) */
    LG_SEG_BASE_LIN_VEL,  /* body-frame base velocity * obs_scales.lin_vel * scale */
    LG_SEG_CONTACT_STATES,/* link_contact_states (genesis_simulator.py:53-55), K wide */
    LG_SEG_HEIGHTS,       /* clip(base_z - heights_offset - measured_heights, +-1) * obs_scales.height_measurements, P wide */
    LG_SEG_FEET_REL_HEIGHTS, /* clip(foot_z - height_around_feet, +-1), 9 F (go2_ts.py:72) */
    LG_SEG_FEET_HEIGHTS,  /* height_around_feet, 9 F (go2_cts.py:73) */
    LG_SEG_FEET_NORMALS,  /* normal_vector_around_feet, 3 F */
    LG_SEG_FOOT_CLEARANCE,/* clip(foot_z - mean(height_around_feet) - foot_height_offset, +-1), F (go2_dreamwaq.py:79-81) */
    LG_SEG_NEXT_STATE,    /* the actor frame without noise, actions * action_scale (go2_dreamwaq.py:66-74) */
    LG_SEG_LAST_ACTIONS,  /* last_actions, A (tron1_pf.py:36) */
    LG_SEG_DR_BASE,       /* friction - offset, added mass, CoM 3, push xy 2 (tron1_pf.py:37-41) */
    LG_SEG_FEET_AIR_TIME, /* feet_air_time, F (tron1_pf.py:42) */
    LG_SEG_KP,            /* kp_scale - kp_offset, A (tron1_sf.py:40-41) */
    LG_SEG_KD             /* kd_scale - kd_offset, A (tron1_sf.py:42-43) */
};
#define LG_MAX_SEGS 8
typedef struct LgObsProgram {
    int32_t n_segs;
    int32_t clip;                 /* 1: clip every entry to +-clip_obs (what step() does to that output in the reference) */
    int32_t kind[LG_MAX_SEGS];    /* enum LgObsSeg */
    int32_t offset[LG_MAX_SEGS];  /* first column of the block inside the frame */
    float scale[LG_MAX_SEGS];     /* extra factor (LG_SEG_BASE_LIN_VEL: 0.5 for dreamwaq's explicit labels), else 1 */
} LgObsProgram;

/* uniform-draw slot layout: every random number the reference draws per env per step has a
 * fixed slot so that tests can inject the reference's own draws (rand_in) and the Philox
 * stream is independent of which envs reset.  Slots are offsets into a per-env row. */
typedef struct LgRandSlots {
    int32_t n_slots;        /* row width of rand_in */
    int32_t cb_cmd;         /* 3: lin_vel_x, lin_vel_y, heading|yaw  (legged_robot.py:317-330) */
    int32_t push;           /* 2: push vel xy (genesis_simulator.py:150-158) */
    int32_t reset_cmd;      /* 3 */
    int32_t reset_dof;      /* A  (go2.py:17-37) */
    int32_t reset_root_xy;  /* 2  (legged_robot.py:288) */
    int32_t reset_lin_vel;  /* 3 */
    int32_t reset_ang_vel;  /* 3 */
    int32_t dr_friction;    /* 1 (genesis_simulator.py:665-675) */
    int32_t dr_mass;        /* 1 */
    int32_t dr_com;         /* 3 */
    int32_t dr_kp;          /* A */
    int32_t dr_kd;          /* A */
    int32_t dr_joint;       /* 3: armature, frictionloss, damping */
    int32_t terrain_level;  /* 1: randint for solved-last-level envs (genesis_simulator.py:143-146) */
    int32_t task_cb;        /* 5: behaviour params resampled in the callback: gait period, base height, clearance,
                               pitch targets + the batch-wide gait index draw (go2_wtw.py:180-210, 258-263) */
    int32_t task_reset;     /* 5: the same drawn inside reset_idx */
    int32_t noise;          /* obs_frame floats: observation noise (go2.py:62-64) */
} LgRandSlots;

/* ---- MDP constants: everything LeggedRobot._parse_cfg/_init_buffers/_prepare_reward_function
 * derive from the cfg (legged_robot.py:380-455) plus the task overrides. */
typedef struct LgTaskCfg {
    int32_t obs_layout;
    int32_t num_obs, num_priv_obs;   /* total widths written per env (0 = none) */
    int32_t obs_frame, priv_frame;   /* single-frame widths */
    int32_t obs_stack, priv_stack;   /* history lengths */
    int32_t obs_slack;               /* history stacks only: extra frames per row; the stacked observation is a window that
                                      * slides one frame per step over rows of (stack + slack) frames (lg_obs_window), so a
                                      * step writes one frame instead of moving the whole history; 0 = shift in place */
    int32_t obs_sets;                /* copies of obs_buf / priv_obs_buf / labels_buf (1, 2, or more when nothing is stacked).  Consecutive
                                      * launches cycle through the copies, so the observation tensors handed out by one step stay
                                      * intact while the next step runs (the reference's step() returns a fresh tensor,
                                      * legged_robot.py:48-49, and rsl_rl keeps it across env.step(), ppo.py:103-104);
                                      * needs obs_slack >= stack + 1 when obs_slack > 0 */
    float control_dt;                /* dt * decimation */
    float clip_actions, clip_obs;
    float max_episode_length;        /* ceil(episode_length_s / dt) as float (legged_robot.py:446) */
    float fail_threshold;            /* fail_to_terminal_time_s / dt (legged_robot.py:90) */
    float max_projected_gravity;
    int32_t resample_steps;          /* int(resampling_time / dt) (legged_robot.py:305) */
    int32_t push_interval;           /* ceil(push_interval_s / dt); 0 = no pushes */
    float max_push_vel_xy;
    int32_t heading_command;
    float yaw_clip[2];               /* commands.ranges.ang_vel_yaw */
    float reward_scales[LG_NUM_REWARDS]; /* already multiplied by dt (legged_robot.py:416-421) */
    float soft_dof_lo[LG_MAX_DOF], soft_dof_hi[LG_MAX_DOF]; /* dof_pos_limits property (genesis_simulator.py:373-382) */
    int32_t only_positive_rewards;
    float tracking_sigma, base_height_target, foot_clearance_target, foot_height_offset;
    float foot_clearance_sigma, about_landing_threshold, feet_air_time_threshold;
    float base_height_sigma, euler_sigma, foot_distance_threshold;
    float no_fly_contact_threshold;  /* foot force z above which _reward_no_fly counts a contact: 0.1 (tron1_pf.py:151-154), 1.0 (tron1_sf.py:275-278) */
    int32_t air_time_cmd_dims;       /* _reward_feet_air_time is gated by |commands[:, :dims]| > 0.1: 2 (legged_robot.py:553), 3 (tron1_sf.py:264) */
    int32_t foot_clearance_ref;      /* terrain height _reward_foot_clearance measures from: 0 none (legged_robot.py:575-588), 1 mean of the
                                      * 9 heights around the foot (go2_ee.py:136-150, go2_ts.py:146-160), 2 their max (tron1_pf_ee.py:442-456,
                                      * go2_cts.py:156-170) */
    float obs_scale_lin_vel, obs_scale_ang_vel, obs_scale_dof_pos, obs_scale_dof_vel, obs_scale_height;
    int32_t add_noise;
    float noise_vec[LG_MAX_OBS];     /* _get_noise_scale_vec (go2.py:92-117) */
    /* reset distribution */
    float reset_dof_lo[LG_MAX_DOF], reset_dof_span[LG_MAX_DOF]; /* q0 + lo + span*u (go2.py:30-35) */
    float reset_root_xy_lo, reset_root_xy_span; int32_t custom_origins;
    float reset_lin_vel_lo, reset_lin_vel_span, reset_ang_vel_lo, reset_ang_vel_span;
    float base_init_quat[4];
    /* domain randomisation (genesis_simulator.py:62-82, 665-739); span 0 & flag 0 = off */
    int32_t dr_friction_on, dr_mass_on, dr_com_on, dr_pd_on, dr_joint_on;
    float dr_friction_lo, dr_friction_span, dr_mass_lo, dr_mass_span;
    float dr_com_lo[3], dr_com_span[3];
    float dr_kp_lo, dr_kp_span, dr_kd_lo, dr_kd_span;
    float dr_joint_lo[3], dr_joint_span[3];
    float friction_offset, kp_offset, kd_offset; /* privileged-obs centring (legged_robot.py:448-453) */
    /* terrain curriculum (legged_robot.py:254-272, genesis_simulator.py:140-148) */
    int32_t terrain_curriculum, max_terrain_level, terrain_cols_n;
    int32_t num_labels;              /* estimator labels width (legged_robot_ee.py:20-27); 0 = none */
    float heights_offset;            /* base_z - offset - h in the critic heights (0.5 go2_ee.py:45-46, 0.6 tron1) */
    int32_t heights_clip_scale;      /* 1: clip to +-1 and scale by obs_scales.height_measurements */
    float terrain_env_length, episode_length_s;
    /* periodic-gait tasks (go2_wtw.py:29-36, 377-484; tron1_pf_ee.py:28-35, 347-433) */
    int32_t gait_mode;               /* 0 none, 1 quadruped wtw, 2 biped */
    int32_t double_shift;            /* second action-history shift at the end of the step (go2_wtw.py:45-46) */
    int32_t behavior_resample_steps; /* int(behavior_params_range.resampling_time / dt); 0 = off */
    int32_t num_gait_max;
    float b_swing;                   /* fraction of the cycle */
    float gait_period_fixed;         /* biped: constant gait period */
    float theta_table[4][4];         /* [gait][foot slot in feet_indices order] */
    /* tron1_pf_ee sit-pose resets (tron1_pf_ee.py:204-210, 277-310): one batch-wide draw picks the branch */
    float sit_percent;               /* 0 = feature off */
    float sit_pos[3], sit_quat[4], sit_dof_pos[LG_MAX_DOF];
    int32_t task_state_width;
    LgObsProgram priv_prog, labels_prog; /* obs_layout == LG_OBS_PROGRAM */
    /* constraints as terminations (go2_cat.py:143-232, utils/constraint_manager.py:3-106).  Every constraint of the task is a
     * 0/1 violation flag, so the manager's running-max normalisation clamps to 1 and the termination probability of an env is
     * max_p of its worst violated constraint: 1 for a hard one, cat_soft_p for a soft / style one, else 0.  The reward is
     * clip(rew * (1 - p), 0).  The style constraint multiplies a (N,) flag with a (N,1) flag (go2_cat.py:179-180): env e violates
     * it when ITS command is ~0 and ANY env moves a joint faster than 4 rad/s -- reproduced through the job-wide flag in
     * command_ranges[LG_CR_ANY_FAST ..]; the physics and the MDP phases of a CaT step therefore have to be separate launches. */
    int32_t cat_enable;
    float cat_soft_p, cat_action_rate, cat_min_base_height, cat_max_projected_gravity;
    float dof_vel_limits[LG_MAX_DOF];   /* cfg.asset.dof_vel_limits (Simulator.dof_vel_limits) */
    LgRandSlots slots;
    uint64_t seed;
    int64_t env_id_offset;           /* global index of local env 0 (multi-GPU sharding) */
} LgTaskCfg;

/* ---- per-env device buffers.  Names follow the reference attributes they back
 * (genesis_simulator.py:407-494 _init_buffers, legged_robot.py:380-409, base_task.py:29-37). */
typedef struct LgBuffers {
    int32_t n_envs;
    /* engine state (what Genesis keeps internally; set by reset_dofs/reset_root_states :84-133) */
    float *base_pos, *base_quat, *base_lin_vel_w, *base_ang_vel_w, *dof_pos, *dof_vel;
    /* per-env dynamics parameters (genesis_simulator.py:644-663) */
    float *friction_values, *added_base_mass, *base_com_bias, *kp_scale, *kd_scale;
    float *joint_armature, *joint_friction, *joint_damping; /* (N,1) each, may be NULL */
    float *rand_push_vels;   /* (N,3) */
    float *env_origins;      /* (N,3) */
    /* Simulator read-back properties (genesis_simulator.py:35-60) */
    float *base_lin_vel, *base_ang_vel, *projected_gravity, *base_euler;
    float *last_base_lin_vel, *last_base_ang_vel, *last_dof_vel, *last_feet_vel;
    float *torques, *link_contact_forces, *feet_pos, *feet_vel;
    float *link_contact_states;       /* (N,K) float 0/1, may be NULL */
    float *measured_heights;          /* (N,P) may be NULL */
    float *height_around_feet;        /* (N,F,9) may be NULL */
    float *normal_vector_around_feet; /* (N,3F) may be NULL */
    float *height_points;             /* (P,2) body-frame sample offsets, may be NULL */
    int32_t *terrain_levels, *terrain_types; /* (N) may be NULL */
    float *terrain_origins;           /* (rows, cols, 3) may be NULL */
    /* MDP buffers (legged_robot.py:380-409, base_task.py:29-37) */
    float *actions, *last_actions, *llast_actions, *commands;
    float *feet_air_time; uint8_t *last_contacts;
    int32_t *episode_length_buf; int64_t *fail_buf;
    uint8_t *reset_buf, *time_out_buf;
    float *rew_buf;
    float *obs_buf, *priv_obs_buf, *labels_buf; /* (obs_sets, N, row); priv/labels may be NULL */
    uint8_t *obs_dirty;               /* (N) with obs_sets == 2 and history stacks: env was reset at the previous observation
                                         launch, so its history in the other copy still has to be blanked; may be NULL */
    float *episode_sums;              /* (LG_R_COUNT, N) */
    float *episode_done_sums;         /* (LG_R_COUNT, N): each env's episode sums as they stood at its latest reset */
    int32_t *episode_done_step;       /* (N): common_step_counter of that reset; together they back the lazy
                                         extras["episode"] means (legged_robot.py:128-132) without atomics */
    float *cstr_prob;                 /* (N) termination probability of the step (go2_cat.py:205), may be NULL */
    float *cstr_sums;                 /* (LG_NUM_CSTR, N) steps with a violation this episode (constraint_manager.py:91-97), may be NULL */
    float *cstr_done_sums;            /* (LG_NUM_CSTR, N) the same as they stood at the env's latest reset, may be NULL */
    float *command_ranges;            /* (LG_CMD_RANGE_FLOATS): vx lo/hi, vy lo/hi, yaw lo/hi, heading lo/hi */
    float *task_state;                /* task specific per-env block (gait phase ...), may be NULL */
    const float *rand_in;             /* (N, slots.n_slots) injected uniforms, NULL => Philox */
    int32_t *nonfinite_count;         /* (1) may be NULL: envs whose state came out of the physics non-finite (NaN / Inf) since the caller
                                         last cleared it.  Such an env is re-seated at its origin AND its fail_buf is raised to
                                         LG_FAIL_NONFINITE, so that check_termination ends the episode in the same step (reset_buf = 1,
                                         not a time-out): the guard is never silent */
} LgBuffers;
#define LG_FAIL_NONFINITE (1 << 20)   /* fail_buf value of an env re-seated by the non-finite guard (far above any fail threshold) */

/* phases of one LeggedRobot.step (legged_robot.py:37-76) a launch may cover */
#define LG_PHASE_PRE 1    /* _pre_sim_step: clip + action history (legged_robot.py:230-239) */
#define LG_PHASE_SIM 2    /* simulator.step + simulator.post_physics_step (genesis_simulator.py:20-60) */
#define LG_PHASE_POST 4   /* callback, check_termination, compute_reward (legged_robot.py:64-68) */
#define LG_PHASE_RESET 8  /* reset_idx + compute_observations (legged_robot.py:69-73) */
#define LG_PHASE_ALL 15

typedef struct LgEngine *LgHandle;

/* Build the per-handle constant tables on the current HIP device.
 * Replaces Simulator.__init__ -> _create_sim/_create_envs (simulator.py:7-18). */
int lg_create(const LgModelDesc *model, const LgSimOptions *opts, const LgTaskCfg *task, LgHandle *out);
int lg_destroy(LgHandle h);
/* Replace the task constants (reward scales after curriculum edits, seeds ...). */
int lg_set_task(LgHandle h, const LgTaskCfg *task);
/* Register the int16 heightfield already resident on the device
 * (genesis_simulator.py:765-778 _create_heightfield; rows x cols, row-major [x][y]). */
int lg_set_terrain(LgHandle h, const int16_t *height_samples_dev, int32_t rows, int32_t cols);
/* Record the caller-owned device buffers (validated: required pointers non-NULL). */
int lg_bind(LgHandle h, const LgBuffers *bufs);
/* Enqueue one launch covering `phases` of a control step on `stream`.
 * actions: (N,A) device pointer (required with LG_PHASE_PRE or LG_PHASE_SIM);
 * common_step_counter: value AFTER this step's increment (legged_robot.py:61). */
int lg_step(LgHandle h, uint32_t phases, const float *actions, int64_t common_step_counter, void *stream);
/* Timing helper for bench.py: enqueue `count` fused steps reusing `actions`, bracketed by
 * HIP events on `stream`; returns mean kernel-to-kernel milliseconds per step in *ms. */
int lg_time_steps(LgHandle h, const float *actions, int64_t first_counter, int32_t count, void *stream, float *ms);
/* Sampling timer for the roofline line of bench.py: with stride N > 0 every N-th lg_step that contains LG_PHASE_SIM
 * brackets the control step's kernel(s) -- begin of the first launch to end of the last when a step takes two (biped:
 * physics, then the MDP phases) -- with a pair of HIP events on the caller's stream (at most 1024
 * samples are kept); stride 0 switches it off.  lg_profile_read waits for the recorded events and returns the
 * mean kernel duration in microseconds and the sample count, then clears the samples. */
/* First frame of the window that holds the most recent stacked observation: row e of obs_buf starts at
 * obs_buf + e * (obs_stack + obs_slack) * obs_frame + *first_frame * obs_frame (same frame index for priv_obs_buf
 * with its own stack / frame widths).  Always 0 when obs_slack == 0. */
int lg_obs_window(LgHandle h, int32_t *first_frame);
/* Copy (0 or 1; always 0 with obs_sets == 1) of obs_buf / priv_obs_buf / labels_buf that holds the most recent
 * observation: copy s starts s * n_envs * row floats into the allocation. */
int lg_obs_set(LgHandle h, int32_t *set);
/* Declare copy `set` to be the one holding the current observation (the caller has put it there): the next observation
 * launch writes copy (set + 1) % obs_sets.  With obs_sets = T + 1 unstacked copies a rollout storage of T steps lays its
 * observation rows over copies 0 .. T-1 and the env writes every step's observation straight into its row
 * (rsl_rl/storage/rollout_storage.py:92 without the copy). */
int lg_obs_set_select(LgHandle h, int32_t set);
/* Restore the window position of the sliding observation history (checkpoint resume: the buffers are the caller's, the
 * position of the window inside them is the only observation state the handle keeps besides the copy index). */
int lg_obs_window_select(LgHandle h, int32_t first_frame);
int lg_profile(LgHandle h, int32_t stride);
/* Diagnostic: which kernel instantiation(s) the latest lg_step launched, as the launcher expression(s) of lg_host.hip (e.g.
 * "(lg_launch_quad<4, true, PR, 1, 3>)" = quad_sim_kernel<LEGS 4, PRE, POST | RESET in the tail, PROF 1, 3 joints per leg>; two launches
 * are joined by " + ").  bench.py reports it as roofline.kernel; the golden replays assert that they went through the tail they mean to. */
const char *lg_last_kernel(LgHandle h);
int lg_profile_read(LgHandle h, float *mean_us, int32_t *samples);
/* Measurement: the achievable HBM stream rate of THIS device, the denominator bench.py quotes next to the nominal peak (SURVEY 8d:
 * "peak figures must be replaced by a measured stream-copy bandwidth on the box").  Copies `bytes` (a multiple of 16) from src to dst
 * `iters` times with a float4 grid-stride kernel on `stream`, timed with HIP events after one untimed pass; *gbs = (read + write)
 * bytes per second / 1e9.  src / dst: device pointers, 16-byte aligned, not overlapping. */
int lg_stream_copy(const void *src, void *dst, int64_t bytes, int32_t iters, void *stream, float *gbs);
/* ---- heightfield generation on the device (legged_gym/utils/terrain.py:37-203, terrain_utils.py:34-372).
 * One tile of the terrain map.  The host side (hcr_genesis_lr_cl_amd/terrain.py) walks the tiles in the reference's order, takes the
 * numpy draws the reference would take (np.random.choice call order) and describes each tile here; the kernel then evaluates every pixel
 * of every tile: integers for the stairs and the obstacles, the reference's float64 arithmetic for the slopes, and FITPACK's own evaluation
 * recurrences (fpbspl / fpbisp, kx = ky = 1) for the up-sampled random-uniform tiles, whose spline knots and coefficients the host fitted.
 * Result: the int16 grid of the reference sample for sample (tests/golden/terrain_*.npz). */
#define LG_TILE_SLOPE 0      /* terrain_utils.py:128-181  ip: peak, edge, half platform; */
#define LG_TILE_UNIFORM 1    /* terrain_utils.py:34-96    ip: edge, nx (knots), ny, n eval x, n eval y; aux: tx | ty | c | x | y (doubles) */
#define LG_TILE_STAIRS 2     /* terrain_utils.py:330-372  ip: step width px, step height (int16 units, signed), platform px */
#define LG_TILE_OBSTACLES 3  /* terrain_utils.py:204-258  ip: n rectangles, platform px; iaux: (i0, j0, w, l, height) per rectangle, in draw order */
typedef struct LgTerrainTile {
    int32_t kind, row, col;  /* tile (row, col) of the map: rows = levels (x), cols = types (y) */
    int32_t ip[5];
    int32_t aux_off;         /* first double of this tile in `aux` / first int in `iaux` */
} LgTerrainTile;
/* Fill the (rows x cols) int16 heightfield `hf` (device, zeroed here: the border stays flat) and the per-tile origin heights
 * `origin_z` (n_tiles doubles, device: max over the central 2 m x 2 m times vertical_scale, terrain.py:197-202).  tiles / aux / iaux: device
 * copies of the descriptor table and of the injected arrays.  tile_px: pixels per tile edge; border_px: border width; o1, o2: pixel range of
 * the origin window. */
int lg_terrain_generate(const LgTerrainTile *tiles, int32_t n_tiles, const double *aux, const int32_t *iaux, int16_t *hf, int32_t rows,
                        int32_t cols, int32_t tile_px, int32_t border_px, int32_t o1, int32_t o2, double vertical_scale, double *origin_z,
                        void *stream);
/* Diagnostic: one Philox4x32-10 block computed on the device by the kernel's own generator (known-answer tests). */
int lg_philox(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]);
/* Known-answer test of the DPP operand behaviour the build's hazard pass relies on (hcr_genesis_lr_cl_amd/dpp_hazard_pass.py; no
 * reference counterpart).  in: 64 x then 64 y (host); out: 5 x 64 floats (host) -- rot1(x) + x y with the product written in the slot
 * before | x y + rot1(x) y likewise (fused) | rot1(x y) + x behind the two wait states | the same without them (negative control) | x y. */
int lg_dpp_kat(const float *in_host, float *out_host);
const char *lg_last_error(void);
int lg_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* LGSIM_H */
