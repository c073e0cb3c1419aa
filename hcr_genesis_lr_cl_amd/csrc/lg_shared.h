// lg_shared.h -- what the host side (lg_host.hip) and the kernel translation units (lg_inst.hip) have in common: the hot-constant
// block, the kernel argument block and the launchers of the kernel instantiations.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <type_traits>
#include "../../include/lgsim.h"
#include "lg_math.h"
#include <vector>
#include <hip/hip_ext.h>

#define LG_ABI_VERSION 3
#define BLOCK 64
#define MODEL_STG 4   /* uint4 per thread to stage the model table: 4 * 64 * 16 B = 4 KiB >= sizeof(LgModelDesc) */
static_assert(sizeof(LgModelDesc) <= MODEL_STG * BLOCK * 16, "model table does not fit the staging image");

// ---- hot constants: every scalar of LgTaskCfg / LgSimOptions the kernel reads, packed contiguously so that ONE burst of
//      s_load_dwordx16 at kernel start fetches them (scattered `T->x` reads each cost an exposed scalar-cache round trip
//      inside the dependency chain of a lone wave).  Internal to this file, filled on the host by fill_hot().
struct LgHot {
    int32_t obs_layout;
    int32_t num_obs;
    int32_t num_priv_obs;
    int32_t obs_frame;
    int32_t priv_frame;
    int32_t obs_stack;
    int32_t priv_stack;
    int32_t obs_slack;
    int32_t obs_sets;
    int32_t reward_mask;   // bit k: reward term k has a non-zero scale (one scalar test per term instead of an LDS round trip)
    float control_dt;
    float clip_actions;
    float clip_obs;
    float max_episode_length;
    float fail_threshold;
    float max_projected_gravity;
    int32_t resample_steps;
    int32_t push_interval;
    float max_push_vel_xy;
    int32_t heading_command;
    int32_t only_positive_rewards;
    float tracking_sigma;
    float base_height_target;
    float foot_clearance_target;
    float foot_height_offset;
    float foot_clearance_sigma;
    float about_landing_threshold;
    float feet_air_time_threshold;
    float base_height_sigma;
    float euler_sigma;
    float foot_distance_threshold;
    int32_t foot_clearance_ref;
    float obs_scale_lin_vel;
    float obs_scale_ang_vel;
    float obs_scale_dof_pos;
    float obs_scale_dof_vel;
    float obs_scale_height;
    int32_t add_noise;
    float reset_root_xy_lo;
    float reset_root_xy_span;
    int32_t custom_origins;
    float reset_lin_vel_lo;
    float reset_lin_vel_span;
    float reset_ang_vel_lo;
    float reset_ang_vel_span;
    int32_t dr_friction_on;
    int32_t dr_mass_on;
    int32_t dr_com_on;
    int32_t dr_pd_on;
    int32_t dr_joint_on;
    float dr_friction_lo;
    float dr_friction_span;
    float dr_mass_lo;
    float dr_mass_span;
    float dr_kp_lo;
    float dr_kp_span;
    float dr_kd_lo;
    float dr_kd_span;
    float friction_offset;
    float kp_offset;
    float kd_offset;
    int32_t terrain_curriculum;
    int32_t max_terrain_level;
    int32_t terrain_cols_n;
    int32_t num_labels;
    float heights_offset;
    int32_t heights_clip_scale;
    float terrain_env_length;
    float episode_length_s;
    int32_t gait_mode;
    int32_t double_shift;
    int32_t behavior_resample_steps;
    int32_t num_gait_max;
    float b_swing;
    float gait_period_fixed;
    float sit_percent;
    int32_t air_time_cmd_dims;
    int32_t task_state_width;
    float yaw_clip[2];
    float base_init_quat[4];
    float dr_com_lo[3];
    float dr_com_span[3];
    float dr_joint_lo[3];
    float dr_joint_span[3];
    float reward_scales[LG_NUM_REWARDS];
    float noise_lead[6];   // noise_vec[3..8]: gravity + angular velocity entries written by the lead lane
    float noise_act0;      // noise_vec[9 + 2A]: non-zero only for the tron1 layout
    LgRandSlots slots;
    unsigned long long seed; long long env_id_offset;
    int32_t m_n_links, m_foot_link[LG_MAX_LEGS], m_foot_sphere[LG_MAX_LEGS];   // model uniforms needed before the LDS table is up
    float o_dt;
    int32_t o_decimation;
    float o_gravity_z;
    float o_contact_k;
    float o_contact_b;
    float o_terrain_friction;
    float o_limit_k;
    float o_limit_b;
    int32_t o_contact_iters;
    int32_t o_contact_w_every;
    float o_contact_margin;
    float o_limit_margin;
    float o_max_base_lin_vel;
    float o_max_base_ang_vel;
    float o_joint_vel_clamp;
    float o_action_scale;
    int32_t o_terrain_rows;
    int32_t o_terrain_cols;
    float o_hscale;
    float o_vscale;
    float o_border;
    int32_t o_n_height_points;
    int32_t o_feet_terrain_info;
    float o_base_init_pos[3];
    float o_bound_x[2];
    float o_bound_y[2];
};
static void fill_hot(LgHot &H, const LgTaskCfg &t, const LgSimOptions &o, const LgModelDesc &m) {
    const int n_dof = m.n_bodies - 1;
    memset(&H, 0, sizeof(H));
    H.m_n_links = m.n_links;
    for (int i = 0; i < LG_MAX_LEGS; i++) { H.m_foot_link[i] = m.foot_link[i]; H.m_foot_sphere[i] = m.foot_sphere[i]; }
    H.obs_layout = t.obs_layout;
    H.num_obs = t.num_obs;
    H.num_priv_obs = t.num_priv_obs;
    H.obs_frame = t.obs_frame;
    H.priv_frame = t.priv_frame;
    H.obs_stack = t.obs_stack;
    H.priv_stack = t.priv_stack;
    H.obs_slack = t.obs_slack;
    H.obs_sets = t.obs_sets > 1 ? t.obs_sets : 1;
    H.reward_mask = 0;
    for (int i = 0; i < LG_R_COUNT; i++) if (t.reward_scales[i] != 0.f) H.reward_mask |= (int32_t)(1u << i);
    H.control_dt = t.control_dt;
    H.clip_actions = t.clip_actions;
    H.clip_obs = t.clip_obs;
    H.max_episode_length = t.max_episode_length;
    H.fail_threshold = t.fail_threshold;
    H.max_projected_gravity = t.max_projected_gravity;
    H.air_time_cmd_dims = t.air_time_cmd_dims;
    H.resample_steps = t.resample_steps;
    H.push_interval = t.push_interval;
    H.max_push_vel_xy = t.max_push_vel_xy;
    H.heading_command = t.heading_command;
    H.only_positive_rewards = t.only_positive_rewards;
    H.tracking_sigma = t.tracking_sigma;
    H.base_height_target = t.base_height_target;
    H.foot_clearance_target = t.foot_clearance_target;
    H.foot_height_offset = t.foot_height_offset;
    H.foot_clearance_sigma = t.foot_clearance_sigma;
    H.about_landing_threshold = t.about_landing_threshold;
    H.feet_air_time_threshold = t.feet_air_time_threshold;
    H.base_height_sigma = t.base_height_sigma;
    H.euler_sigma = t.euler_sigma;
    H.foot_distance_threshold = t.foot_distance_threshold;
    H.foot_clearance_ref = t.foot_clearance_ref;
    H.obs_scale_lin_vel = t.obs_scale_lin_vel;
    H.obs_scale_ang_vel = t.obs_scale_ang_vel;
    H.obs_scale_dof_pos = t.obs_scale_dof_pos;
    H.obs_scale_dof_vel = t.obs_scale_dof_vel;
    H.obs_scale_height = t.obs_scale_height;
    H.add_noise = t.add_noise;
    H.reset_root_xy_lo = t.reset_root_xy_lo;
    H.reset_root_xy_span = t.reset_root_xy_span;
    H.custom_origins = t.custom_origins;
    H.reset_lin_vel_lo = t.reset_lin_vel_lo;
    H.reset_lin_vel_span = t.reset_lin_vel_span;
    H.reset_ang_vel_lo = t.reset_ang_vel_lo;
    H.reset_ang_vel_span = t.reset_ang_vel_span;
    H.dr_friction_on = t.dr_friction_on;
    H.dr_mass_on = t.dr_mass_on;
    H.dr_com_on = t.dr_com_on;
    H.dr_pd_on = t.dr_pd_on;
    H.dr_joint_on = t.dr_joint_on;
    H.dr_friction_lo = t.dr_friction_lo;
    H.dr_friction_span = t.dr_friction_span;
    H.dr_mass_lo = t.dr_mass_lo;
    H.dr_mass_span = t.dr_mass_span;
    H.dr_kp_lo = t.dr_kp_lo;
    H.dr_kp_span = t.dr_kp_span;
    H.dr_kd_lo = t.dr_kd_lo;
    H.dr_kd_span = t.dr_kd_span;
    H.friction_offset = t.friction_offset;
    H.kp_offset = t.kp_offset;
    H.kd_offset = t.kd_offset;
    H.terrain_curriculum = t.terrain_curriculum;
    H.max_terrain_level = t.max_terrain_level;
    H.terrain_cols_n = t.terrain_cols_n;
    H.num_labels = t.num_labels;
    H.heights_offset = t.heights_offset;
    H.heights_clip_scale = t.heights_clip_scale;
    H.terrain_env_length = t.terrain_env_length;
    H.episode_length_s = t.episode_length_s;
    H.gait_mode = t.gait_mode;
    H.double_shift = t.double_shift;
    H.behavior_resample_steps = t.behavior_resample_steps;
    H.num_gait_max = t.num_gait_max;
    H.b_swing = t.b_swing;
    H.gait_period_fixed = t.gait_period_fixed;
    H.sit_percent = t.sit_percent;
    H.task_state_width = t.task_state_width;
    for (int i = 0; i < 2; i++) H.yaw_clip[i] = t.yaw_clip[i];
    for (int i = 0; i < 4; i++) H.base_init_quat[i] = t.base_init_quat[i];
    for (int i = 0; i < 3; i++) H.dr_com_lo[i] = t.dr_com_lo[i];
    for (int i = 0; i < 3; i++) H.dr_com_span[i] = t.dr_com_span[i];
    for (int i = 0; i < 3; i++) H.dr_joint_lo[i] = t.dr_joint_lo[i];
    for (int i = 0; i < 3; i++) H.dr_joint_span[i] = t.dr_joint_span[i];
    for (int i = 0; i < LG_NUM_REWARDS; i++) H.reward_scales[i] = t.reward_scales[i];
    for (int i = 0; i < 6; i++) H.noise_lead[i] = t.noise_vec[3 + i];
    H.noise_act0 = t.noise_vec[9 + 2 * n_dof];
    H.slots = t.slots; H.seed = t.seed; H.env_id_offset = t.env_id_offset;
    H.o_dt = o.dt;
    H.o_decimation = o.decimation;
    H.o_gravity_z = o.gravity_z;
    H.o_contact_k = o.contact_k;
    H.o_contact_b = o.contact_b;
    H.o_terrain_friction = o.terrain_friction;
    H.o_limit_k = o.limit_k;
    H.o_limit_b = o.limit_b;
    H.o_contact_iters = o.contact_iters;
    H.o_contact_w_every = o.contact_w_every > 1 ? o.contact_w_every : 1;
    H.o_contact_margin = o.contact_margin;
    H.o_limit_margin = o.limit_margin;
    H.o_max_base_lin_vel = o.max_base_lin_vel;
    H.o_max_base_ang_vel = o.max_base_ang_vel;
    H.o_joint_vel_clamp = o.joint_vel_clamp;
    H.o_action_scale = o.action_scale;
    H.o_terrain_rows = o.terrain_rows;
    H.o_terrain_cols = o.terrain_cols;
    H.o_hscale = o.hscale;
    H.o_vscale = o.vscale;
    H.o_border = o.border;
    H.o_n_height_points = o.n_height_points;
    H.o_feet_terrain_info = o.feet_terrain_info;
    for (int i = 0; i < 3; i++) H.o_base_init_pos[i] = o.base_init_pos[i];
    for (int i = 0; i < 2; i++) H.o_bound_x[i] = o.bound_x[i];
    for (int i = 0; i < 2; i++) H.o_bound_y[i] = o.bound_y[i];
}

// ---- lane table of the component-per-lane kernels (lg_quad.h, three-joint legs) ----------------------------------------------------------
// What a lane of type (leg, c) -- c: vector component / joint of the quad's lane -- needs of the model and of the options, row by row, laid out
// by the host whenever either changes (lg_host.hip upload_hot).  The kernel used to derive these ~70 values per lane from the staged model
// table on every launch: dependent LDS look-ups (body -> sphere range -> sphere), selects for the rows of the symmetric inertias, ten IEEE
// divisions -- 628 instructions, 2.5 k of the go2 step's 51 k cycles.  Now the wave stages the sixteen rows (5.3 KB) and a lane reads its own.
#ifndef LG_LT_ROW
#define LG_LT_ROW 84          /* row stride in floats: 4 x odd, so that the sixteen rows' 16-byte reads spread over all LDS banks (stride 80 = 16 x 5
                                 puts them on two bank groups: eight-way conflicts, +0.6 us on the go2 step) */
#endif
#define LG_LT_STG ((16 * LG_LT_ROW * 4 + 1023) / 1024)   /* uint4 per thread to stage 16 rows (64 threads x 16 B each) */
enum LgLaneCol {
    LT_M = 0,        /* 3: mass of the leg's bodies */
    LT_COM = 3,      /* 3: their centre of mass, this lane's component */
    LT_JPOS = 6,     /* 3: joint position in the parent frame, component */
    LT_AX = 9,       /* 3: joint axis, component */
    LT_IC = 12,      /* 9: row c of each body's rotational inertia */
    LT_I0 = 21,      /* 3: row c of the base's */
    LT_MASS0 = 24, LT_COM0 = 25,
    LT_QLO = 26, LT_QHI = 27, LT_EFF = 28, LT_VLIM = 29, LT_ARM = 30, LT_JFRIC = 31, LT_JDAMP = 32,   /* this lane's joint (lane 3 shadows lane 2) */
    LT_FOOT_C = 33, LT_FOOT_R = 34, LT_LINKPOS = 35,
    LT_SLOT = 36,    /* 5 x (x, y, z, radius, 1 / (1 + kappa dt w), 1 / (dt w)): this lane's sphere of each collision slot */
    LT_KP = 66, LT_KD = 67, LT_Q0 = 68,
    LT_TMASK = 69, LT_PMASK = 70, LT_SMASK = 71,   /* link masks (bit patterns) */
    LT_USED = 72
};
static_assert(LT_USED <= LG_LT_ROW && LG_LT_ROW % 4 == 0 && 16 * LG_LT_ROW * 4 <= LG_LT_STG * 64 * 16, "lane table geometry");
inline void lg_fill_lane_table(float *t, const LgModelDesc &m, const LgSimOptions &o) {
    for (int i = 0; i < LG_LT_STG * 256; i++) t[i] = 0.f;
    const int legs = m.n_legs;
    if (m.n_bodies != 1 + 3 * legs) return;       // four-joint legs keep the in-kernel derivation
    const float dt = o.dt, kappa = o.contact_k * dt + o.contact_b;
    auto bits = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
    for (int leg = 0; leg < legs; leg++) for (int c = 0; c < 4; c++) {
        float *r = t + ((leg << 2) | c) * LG_LT_ROW;
        const int cj = c < 2 ? c : 2, b0 = 1 + 3 * leg, d0 = 3 * leg, dj = d0 + cj;
        auto sym_row = [&](const float *s6, float *out) {   // row c (lane 3: row 2) of a symmetric 3x3 stored (xx, yy, zz, xy, xz, yz)
            out[0] = s6[c == 0 ? 0 : (c == 1 ? 3 : 4)]; out[1] = s6[c == 0 ? 3 : (c == 1 ? 1 : 5)]; out[2] = s6[c == 0 ? 4 : (c == 1 ? 5 : 2)];
        };
        for (int j = 0; j < 3; j++) {
            const int b = b0 + j;
            r[LT_M + j] = m.mass[b]; r[LT_COM + j] = m.com[b][cj]; r[LT_JPOS + j] = m.jpos[b][cj]; r[LT_AX + j] = m.axis[b][cj];
            sym_row(m.inertia[b], r + LT_IC + 3 * j);
        }
        sym_row(m.inertia[0], r + LT_I0);
        r[LT_MASS0] = m.mass[0]; r[LT_COM0] = m.com[0][cj];
        r[LT_QLO] = m.q_lo[dj]; r[LT_QHI] = m.q_hi[dj]; r[LT_EFF] = m.effort[dj]; r[LT_VLIM] = o.joint_vel_clamp * m.vel_limit[dj];
        r[LT_ARM] = m.armature[dj]; r[LT_JFRIC] = m.frictionloss[dj]; r[LT_JDAMP] = m.damping[dj];
        const int fs = m.foot_sphere[leg], fl = m.foot_link[leg];
        r[LT_FOOT_C] = m.sph_pos[fs][cj]; r[LT_FOOT_R] = m.sph_r[fs]; r[LT_LINKPOS] = m.link_pos[fl][cj];
        // collision slots (lg_quad.h): 0 hip, 1 thigh, 2-3 calf without the foot sphere, 4 base (four spheres per quad); lane c tests sphere c
        const int a0 = m.body_sph_start[b0], a1 = m.body_sph_start[b0 + 1], a2 = m.body_sph_start[b0 + 2], a3 = m.body_sph_start[b0 + 3];
        const int e0 = m.body_sph_start[0], e1 = m.body_sph_start[1];
        int idx[5];
        idx[0] = a0 + c < a1 ? a0 + c : -1;
        idx[1] = a1 + c < a2 ? a1 + c : -1;
        for (int k = 0; k < 2; k++) { int s = a2 + 4 * k + c; if (s >= fs) s++; idx[2 + k] = s < a3 ? s : -1; }
        idx[4] = e0 + leg * 4 + c < e1 ? e0 + leg * 4 + c : -1;
        for (int k = 0; k < 5; k++) {
            const int s = idx[k] > 0 ? idx[k] : 0;
            float *q = r + LT_SLOT + 6 * k;
            const float wi = m.sph_w[s];
            q[0] = m.sph_pos[s][0]; q[1] = m.sph_pos[s][1]; q[2] = m.sph_pos[s][2];
            q[3] = idx[k] >= 0 ? m.sph_r[s] : -1e30f;        // an empty slot is infinitely far from any surface
            q[4] = 1.f / (1.f + kappa * dt * wi); q[5] = 1.f / (dt * wi);
        }
        r[LT_KP] = o.kp[dj]; r[LT_KD] = o.kd[dj]; r[LT_Q0] = o.default_dof_pos[dj];
        r[LT_TMASK] = bits(m.term_link_mask); r[LT_PMASK] = bits(m.pen_link_mask); r[LT_SMASK] = bits(m.state_link_mask);
    }
}

struct KParams {
    const LgModelDesc *M;
    const LgSimOptions *O;
    const LgTaskCfg *T;
    const LgHot *H;
    const float *LT;     // lane table (above)
    const int16_t *hf;
    LgBuffers B;
    const float *actions;
    long long counter;
    int jrot_identity;   // every joint frame is axis-aligned with its parent at q = 0 (host-checked)
    // kernarg copies of the few constants the start-of-kernel load burst needs for its addresses and predicates (scalar
    // loads that return before anything else): reading them from the hot block would put a full memory round trip in
    // front of the burst
    struct { int m_n_links, m_foot_link[4], obs_layout, o_n_height_points; unsigned reward_mask; float clip_actions;
             int cat_enable;      // LgTaskCfg.cat_enable: tested on every step, so not behind a memory round trip
             int joint_axis[4];   // per joint index: 0/1/2 if that joint's axis is +-e_x/e_y/e_z on every leg, else -1 (lg_quad.h joint_rot)
    } k;
    int obs_win;         // first frame of the observation window this launch writes (sliding history, LgTaskCfg.obs_slack)
    int obs_set;         // copy of obs_buf / priv_obs_buf / labels_buf this launch writes (LgTaskCfg.obs_sets)
};


// ---- launchers.  The kernel templates (lg_kernel.h: env_step_kernel, lg_quad.h: quad_sim_kernel) are instantiated in lg_inst.hip, which is
//      compiled once per instantiation GROUP (-DLG_GROUP=g, in parallel: one translation unit with every instantiation took over three
//      minutes); the host side only sees these declarations.  e0 / e1: begin / end timestamps attached to the dispatch (either may be null).
template <int LEGS, bool PRE, unsigned MPH, int PROF, int JPL>
void lg_launch_quad(dim3 grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const KParams &p);
template <int LEGS, unsigned PH, int PROF, int JPL, bool REPL>
void lg_launch_env(dim3 grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const KParams &p);
// test instantiations of the component-layout tails: PRE | POST | RESET on injected read-backs and uniforms (lg_quad.h, INJ)
template <int LEGS, int PROF>
void lg_launch_quad_inj(dim3 grid, hipStream_t st, const KParams &p);
// The reward sets of the reference's own task configs (bit k: term k of include/lgsim.h LgReward has a non-zero scale), by task profile:
// go2 (legged_gym/envs/go2/go2_config.py), go2_wtw, go2_ee and the go2 rough heads (one set).  A task whose set equals its profile's runs
// the instantiation that has it as a compile-time constant (lg_quad.h RS): whole step (INJ = false) or the golden-replay form (INJ = true).
// (tron1_pf_ee, 0x2cb38577: +-0.1 us, not kept.)
constexpr unsigned lg_default_reward_mask(int prof) { return prof == 1 ? 0x26a2296fu : (prof == 2 ? 0x7f2c0967u : ((prof == 3 || prof == 4) ? 0x242a6567u : 0u)); }
template <int LEGS, int PROF, bool INJ>
void lg_launch_quad_rs(dim3 grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const KParams &p);
