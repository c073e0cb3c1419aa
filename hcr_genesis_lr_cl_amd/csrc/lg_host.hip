// lg_host.hip -- host side of the library: the C ABI of include/lgsim.h (engine handle, launch selection, profiling timers).
// The kernels themselves are instantiated in lg_inst.hip (lg_kernel.h: one leg per lane; lg_quad.h: one vector component per lane) and
// reached through the launchers declared in lg_shared.h.
#include "lg_shared.h"
#include <algorithm>

// ---------------------------------------------------------------------------------------------
// sliding observation history ran out of slack: move the newest stack-1 frames of every row back to frames [1, stack)
__global__ __launch_bounds__(256) void obs_compact_kernel(float *buf, int n_rows, int row, int frame, int stack, int from) {
    const int per = (stack - 1) * frame;
    const long long total = (long long)n_rows * per;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(t / per), i = (int)(t % per);
        buf[(size_t)e * row + frame + i] = buf[(size_t)e * row + (size_t)(from + 1) * frame + i];
    }
}

// =============================== host side: the C ABI ==========================================
static thread_local std::string g_err;
static int fail(const std::string &m) { g_err = m; return 1; }
int lg_fail_msg(const std::string &m) { return fail(m); }   // for the other translation units of the library (lg_rollout.hip)
#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(_e)); } while (0)

struct LgEngine {
    LgModelDesc model; LgSimOptions opts; LgTaskCfg task;
    LgModelDesc *d_model = nullptr; LgSimOptions *d_opts = nullptr; LgTaskCfg *d_task = nullptr; LgHot *d_hot = nullptr;
    float *d_lane = nullptr;
    const int16_t *hf = nullptr;
    LgBuffers bufs; bool bound = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    LgHot hot;           // host copy of the hot block (upload_hot)
    int obs_win = 0;     // window of the latest stacked observation (obs_slack > 0)
    int obs_set = 0;     // copy of the observation buffers the latest observation launch wrote (obs_sets == 2)
    std::string last_kernel;   // launcher instantiation(s) of the latest lg_step (lg_last_kernel)
    // bounded run-ahead: the host never gets more than ~128 lg_step calls ahead of the device (see lg_step)
    hipEvent_t ra_ev[4] = {nullptr, nullptr, nullptr, nullptr}; long long ra_calls = 0;
    // sampling timer of the physics kernel (lg_profile)
    int prof_stride = 0, prof_count = 0; long long prof_seen = 0;
    std::vector<hipEvent_t> prof_ev;
};

extern "C" const char *lg_last_error(void) { return g_err.c_str(); }
extern "C" int lg_abi_version(void) { return LG_ABI_VERSION; }

static int upload_hot(LgEngine *h) {
    LgHot hot;
    fill_hot(hot, h->task, h->opts, h->model);
    h->hot = hot;
    if (!h->d_hot) { hipError_t e = hipMalloc(&h->d_hot, 1024); if (e != hipSuccess) return fail(std::string("hipMalloc(hot): ") + hipGetErrorString(e)); }
    HIPCHK(hipMemset(h->d_hot, 0, 1024));
    HIPCHK(hipMemcpy(h->d_hot, &hot, sizeof(LgHot), hipMemcpyHostToDevice));
    // the lane table of the component-per-lane kernels follows the model and the options (lg_shared.h)
    std::vector<float> lt(LG_LT_STG * 256);
    lg_fill_lane_table(lt.data(), h->model, h->opts);
    if (!h->d_lane) { hipError_t e = hipMalloc(&h->d_lane, lt.size() * sizeof(float)); if (e != hipSuccess) return fail(std::string("hipMalloc(lane table): ") + hipGetErrorString(e)); }
    HIPCHK(hipMemcpy(h->d_lane, lt.data(), lt.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

static int validate_model(const LgModelDesc *m) {
    if (m->n_legs != 2 && m->n_legs != 4) return fail("lg_create: engine supports 2 or 4 legs");
    const int J = m->n_bodies == 1 + 3 * m->n_legs ? 3 : 4;     // joints per leg
    if (m->n_bodies != 1 + J * m->n_legs || (J == 4 && m->n_legs != 2))
        return fail("lg_create: n_bodies must be 1 + 3*n_legs (2 or 4 legs of 3 revolute joints) or 1 + 4*2 (two legs of 4)");
    if (m->n_links < m->n_bodies || m->n_links > LG_MAX_LINKS) return fail("lg_create: bad n_links");
    if (m->n_spheres < m->n_legs || m->n_spheres > LG_MAX_SPHERES) return fail("lg_create: bad n_spheres");
    for (int l = 0; l < m->n_legs; l++) {
        int fl = m->foot_link[l], fsph = m->foot_sphere[l];
        if (fl < 4 || fl >= m->n_links) return fail("lg_create: foot link index out of range");
        if (fsph < 0 || fsph >= m->n_spheres || m->sph_body[fsph] != J + J * l) return fail("lg_create: foot sphere must sit on the last body of its leg");
        // kernel assumes the four link rows of a leg are contiguous: [hip, thigh, calf, foot] (the foot a kept link of the calf body) or,
        // with four joints, the four bodies themselves (the foot is the last one)
        for (int k = 0; k < 3; k++)
            if (m->link_body[fl - 3 + k] != 1 + J * l + k) return fail("lg_create: leg links must be contiguous and end with the foot");
        if (m->link_body[fl] != J + J * l) return fail("lg_create: foot link must move with the last body of its leg");
    }
    if (m->link_body[0] != 0) return fail("lg_create: link 0 must be the base");
    for (int l = 0; l < m->n_legs; l++) {   // register-resident sphere tables of the kernel: 2 / 2 / [3 /] 5 (+foot) per chain body
        const int *st = m->body_sph_start + 1 + J * l;
        if (st[1] - st[0] > 2 || st[2] - st[1] > 2 || (J == 4 && st[3] - st[2] > 3) || st[J] - st[J - 1] > 6)
            return fail("lg_create: too many collision spheres on a leg body (max 2/2/5+foot, or 2/2/3/5+foot with four joints)");
    }
    if ((m->body_sph_start[1] - m->body_sph_start[0] + m->n_legs - 1) / m->n_legs > 4) return fail("lg_create: too many base collision spheres (max 4 per leg lane)");
    for (int b = 0; b <= m->n_bodies; b++)
        if (m->body_sph_start[b] < 0 || m->body_sph_start[b] > m->n_spheres || (b && m->body_sph_start[b] < m->body_sph_start[b - 1]))
            return fail("lg_create: body_sph_start not monotone");
    for (int b = 0; b < m->n_bodies; b++)
        if (!(m->mass[b] > 0.f)) return fail("lg_create: non-positive body mass");
    return 0;
}

extern "C" int lg_create(const LgModelDesc *model, const LgSimOptions *opts, const LgTaskCfg *task, LgHandle *out) {
    if (!model || !opts || !task || !out) return fail("lg_create: null argument");
    if (validate_model(model)) return 1;
    if (!(opts->dt > 0.f) || opts->decimation < 1 || opts->decimation > 64) return fail("lg_create: bad dt/decimation");
    if (opts->contact_iters < 1 || opts->contact_iters > 16) return fail("lg_create: contact_iters must be in [1,16]");
    if (task->obs_slack < 0 || (task->obs_slack > 0 && (task->obs_slack < task->obs_stack || task->obs_slack < task->priv_stack)))
        return fail("lg_create: obs_slack must be 0 or at least as large as the history stacks");
    if (task->obs_sets > 1 && task->obs_slack > 0 && (task->obs_slack <= task->obs_stack || task->obs_slack <= task->priv_stack))
        return fail("lg_create: with two observation sets obs_slack must exceed the history stacks");
    if (task->obs_sets > 2 && (task->obs_stack > 1 || task->priv_stack > 1))
        return fail("lg_create: more than two observation sets (rollout-resident observations) need unstacked observations");
    if (task->obs_sets > 4096) return fail("lg_create: obs_sets out of range");
    LgEngine *h = new LgEngine();
    h->model = *model; h->opts = *opts; h->task = *task;
    memset(&h->bufs, 0, sizeof(h->bufs));
    hipError_t e;
    if ((e = hipMalloc(&h->d_model, MODEL_STG * BLOCK * 16)) != hipSuccess || (e = hipMalloc(&h->d_opts, sizeof(LgSimOptions))) != hipSuccess ||
        (e = hipMalloc(&h->d_task, sizeof(LgTaskCfg))) != hipSuccess) { delete h; return fail(std::string("lg_create: hipMalloc: ") + hipGetErrorString(e)); }
    HIPCHK(hipMemset(h->d_model, 0, MODEL_STG * BLOCK * 16));
    HIPCHK(hipMemcpy(h->d_model, model, sizeof(LgModelDesc), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_opts, opts, sizeof(LgSimOptions), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_task, task, sizeof(LgTaskCfg), hipMemcpyHostToDevice));
    if (upload_hot(h)) return 1;
    HIPCHK(hipEventCreate(&h->ev0));
    HIPCHK(hipEventCreate(&h->ev1));
    *out = h;
    return 0;
}

extern "C" int lg_destroy(LgHandle h) {
    if (!h) return 0;
    (void)hipFree(h->d_model); (void)hipFree(h->d_opts); (void)hipFree(h->d_task); (void)hipFree(h->d_hot); (void)hipFree(h->d_lane);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    for (hipEvent_t ev : h->prof_ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : h->ra_ev) if (ev) (void)hipEventDestroy(ev);
    delete h;
    return 0;
}

extern "C" int lg_set_task(LgHandle h, const LgTaskCfg *task) {
    if (!h || !task) return fail("lg_set_task: null argument");
    h->task = *task;
    HIPCHK(hipMemcpy(h->d_task, task, sizeof(LgTaskCfg), hipMemcpyHostToDevice));
    return upload_hot(h);
}

extern "C" int lg_set_terrain(LgHandle h, const int16_t *hf, int32_t rows, int32_t cols) {
    if (!h) return fail("lg_set_terrain: null handle");
    if (hf && (rows < 2 || cols < 2)) return fail("lg_set_terrain: heightfield needs at least 2x2 samples");
    h->hf = hf;
    h->opts.terrain_rows = hf ? rows : 0;
    h->opts.terrain_cols = hf ? cols : 0;
    HIPCHK(hipMemcpy(h->d_opts, &h->opts, sizeof(LgSimOptions), hipMemcpyHostToDevice));
    return upload_hot(h);
}

extern "C" int lg_bind(LgHandle h, const LgBuffers *b) {
    if (!h || !b) return fail("lg_bind: null argument");
    if (b->n_envs < 1) return fail("lg_bind: n_envs must be positive");
#define REQ(f) if (!b->f) return fail("lg_bind: required buffer " #f " is NULL")
    REQ(base_pos); REQ(base_quat); REQ(base_lin_vel_w); REQ(base_ang_vel_w); REQ(dof_pos); REQ(dof_vel);
    REQ(base_lin_vel); REQ(base_ang_vel); REQ(projected_gravity); REQ(base_euler);
    REQ(last_base_lin_vel); REQ(last_base_ang_vel); REQ(last_dof_vel); REQ(last_feet_vel);
    REQ(torques); REQ(link_contact_forces); REQ(feet_pos); REQ(feet_vel); REQ(env_origins);
#undef REQ
    h->bufs = *b;
    h->bound = true;
    return 0;
}

static int prof_begin(LgEngine *h, hipStream_t st) {
    if (h->prof_stride <= 0 || (h->prof_seen++ % h->prof_stride) != 0 || h->prof_count >= 1024) return -1;
    if ((int)h->prof_ev.size() < 2 * (h->prof_count + 1)) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
        h->prof_ev.push_back(a); h->prof_ev.push_back(b);
    }
    (void)st;
    return h->prof_count++;
}
// launch with the kernel's own begin / end timestamps when this step is sampled (hipExtLaunchKernelGGL attaches the
// events to the dispatch packet itself, so the reading is the kernel's duration, comparable with rocprofv3's).
// `kern`: a launcher instantiation of lg_shared.h, e.g. (lg_launch_quad<4, true, 12u, 1, 3>)
#define LG_LAUNCH(pi, kern, grid_) (h->last_kernel = #kern, kern(grid_, st, (pi) >= 0 ? h->prof_ev[2 * (pi)] : nullptr, (pi) >= 0 ? h->prof_ev[2 * (pi) + 1] : nullptr, p))
// a control step made of two launches (physics, then the MDP phases): begin timestamp of the first, end timestamp of
// the second, so that the sample is the whole step including the gap between the two
#define LG_LAUNCH_FIRST(pi, kern, grid_) (h->last_kernel = #kern, kern(grid_, st, (pi) >= 0 ? h->prof_ev[2 * (pi)] : nullptr, nullptr, p))
#define LG_LAUNCH_LAST(pi, kern, grid_) (h->last_kernel += " + " #kern, kern(grid_, st, nullptr, (pi) >= 0 ? h->prof_ev[2 * (pi) + 1] : nullptr, p))
#define LG_LAUNCH_PLAIN(kern, grid_) (h->last_kernel = #kern, kern(grid_, st, nullptr, nullptr, p))

static bool flat_noise_ok(const LgEngine *h) {   // commands and actions carry no observation noise (go2.py:92-117)
    const int A = h->model.n_bodies - 1;
    for (int i = 0; i < 3; i++) if (h->task.noise_vec[i] != 0.f) return false;
    for (int i = 0; i < A; i++) if (h->task.noise_vec[9 + 2 * A + i] != 0.f) return false;
    return true;
}
// the plain go2-on-a-plane task: every switch the FLAT instantiations hard-wire (env_step_body) really has that value
static bool flat_profile(const LgEngine *h, bool inj = false) {   // inj: the injected-uniform test instantiation (rand_in REQUIRED instead of forbidden)
    const LgTaskCfg &t = h->task;
    const LgSimOptions &o = h->opts;
    const LgBuffers &b = h->bufs;
    return t.obs_layout == LG_OBS_GO2 && t.gait_mode == 0 && t.double_shift == 0 && t.obs_stack == 1 && t.obs_slack == 0 && t.priv_frame == 0 &&
           t.priv_stack <= 1 && t.num_priv_obs == 0 && t.terrain_curriculum == 0 && t.custom_origins == 0 && t.sit_percent == 0.f &&
           t.behavior_resample_steps == 0 && t.num_labels == 0 && t.cat_enable == 0 && t.noise_vec[9 + 6 * h->model.n_legs] == 0.f && t.air_time_cmd_dims != 3 &&
           h->model.n_bodies == 1 + 3 * h->model.n_legs &&
           o.n_height_points == 0 && o.terrain_rows == 0 && o.feet_terrain_info == 0 && !b.link_contact_states && !b.task_state && !h->hf &&
           (inj ? b.rand_in != nullptr : !b.rand_in) && !b.joint_armature && !t.dr_joint_on && !t.dr_pd_on && t.reset_lin_vel_span == 0.f && t.reset_ang_vel_span == 0.f && flat_noise_ok(h) &&
           // reward terms the component-layout tail (lg_quad.h) does not carry: gait clocks, biped and wtw-only terms
           ((unsigned)h->hot.reward_mask & ((1u << LG_R_BIPED_PERIODIC_GAIT) | (1u << LG_R_QUAD_PERIODIC_GAIT) | (1u << LG_R_FEET_DISTANCE) |
                                            (1u << LG_R_TRACKING_BASE_HEIGHT) | (1u << LG_R_TRACKING_FOOT_CLEARANCE) |
                                            (1u << LG_R_TRACKING_ORIENTATION))) == 0;
}

// go2_wtw on the plane (PROF 2)
static bool wtw_profile(const LgEngine *h, bool inj = false) {
    const LgTaskCfg &t = h->task;
    const LgSimOptions &o = h->opts;
    const LgBuffers &b = h->bufs;
    return t.obs_layout == LG_OBS_GO2_WTW && t.gait_mode == 1 && t.double_shift == 1 && t.terrain_curriculum == 0 && t.custom_origins == 0 &&
           t.sit_percent == 0.f && t.num_labels == 0 && t.cat_enable == 0 && t.noise_vec[9 + 6 * h->model.n_legs] == 0.f &&
           o.n_height_points == 0 && o.terrain_rows == 0 && o.feet_terrain_info == 0 && !b.link_contact_states && !h->hf && b.task_state &&
           // what the component-layout tail of lg_quad.h (PROF 2) hard-wires: sliding-window stacks of 61 | 99-wide frames, Philox draws,
           // no per-env joint parameters, no noise on commands / actions
           t.obs_slack > 0 && t.obs_frame == 61 && t.priv_frame == 61 + 10 + 6 * h->model.n_legs + h->model.n_legs && t.obs_stack > 1 && t.priv_stack > 1 &&
           t.obs_sets <= 2 && (t.obs_sets < 2 || b.obs_dirty) && (inj ? b.rand_in != nullptr : !b.rand_in) && !b.joint_armature && !t.dr_joint_on && flat_noise_ok(h) &&
           h->model.n_legs == 4 && h->model.n_bodies == 13 && b.priv_obs_buf && b.rand_push_vels &&
           ((unsigned)h->hot.reward_mask & ((1u << LG_R_BIPED_PERIODIC_GAIT) | (1u << LG_R_FEET_DISTANCE))) == 0;
}

// the Go2-rough family (PROF 3: go2_ee packaging, PROF 4: observation programs -- go2_ts / go2_cts / go2_dreamwaq / go2_cat): no gait
// clock, no sit pose, no noise on actions; terrain, curriculum, stacks and (PROF 4) CaT stay runtime
static int rough_profile(const LgEngine *h, bool inj = false) {
    const LgTaskCfg &t = h->task;
    if (t.gait_mode != 0 || t.sit_percent != 0.f || t.behavior_resample_steps != 0 || t.noise_vec[9 + 6 * h->model.n_legs] != 0.f || h->bufs.task_state)
        return 0;
    if (!h->hf || h->opts.terrain_rows <= 0) return 0;     // the profiles hard-wire "there is a heightfield" (lg_quad.h HFC)
    // PROF 3 / 4 = the component-layout tail of lg_quad.h for the go2_ee family: what it hard-wires
    const LgBuffers &b = h->bufs;
    const int K = __builtin_popcount(h->model.state_link_mask), P = h->opts.n_height_points, A = h->model.n_bodies - 1;
    const bool common = t.double_shift == 0 && t.cat_enable == 0 && t.obs_slack > 0 && t.obs_frame == 9 + 3 * A && t.obs_stack > 1 && t.priv_stack > 1 &&
                        t.obs_sets <= 2 && (t.obs_sets < 2 || b.obs_dirty) && (inj ? b.rand_in != nullptr : !b.rand_in) && !b.joint_armature && !t.dr_joint_on && flat_noise_ok(h) &&
                        h->model.n_legs == 4 && h->model.n_bodies == 13 && P > 0 && P <= 7 * 16 && h->opts.feet_terrain_info && b.priv_obs_buf &&
                        b.rand_push_vels && t.air_time_cmd_dims != 3 && t.num_priv_obs > 0 &&
                        (!t.terrain_curriculum || (b.terrain_levels && b.terrain_types && b.terrain_origins && b.env_origins)) &&
                        ((unsigned)h->hot.reward_mask & ((1u << LG_R_BIPED_PERIODIC_GAIT) | (1u << LG_R_QUAD_PERIODIC_GAIT) | (1u << LG_R_FEET_DISTANCE) |
                                                         (1u << LG_R_BASE_HEIGHT) | (1u << LG_R_TRACKING_BASE_HEIGHT) | (1u << LG_R_TRACKING_FOOT_CLEARANCE) |
                                                         (1u << LG_R_TRACKING_ORIENTATION))) == 0;
    if (!common) return 0;
    if (t.obs_layout == LG_OBS_GO2_EE)
        return (t.priv_frame == 9 + 3 * A + 7 + 2 * A + K + P && t.num_labels == 3 + K + h->model.n_legs && b.labels_buf) ? 3 : 0;
    if (t.obs_layout == LG_OBS_PROGRAM) {
        if (t.num_labels > 0 && !b.labels_buf) return 0;
        for (const LgObsProgram *pr : {&t.priv_prog, &t.labels_prog})
            for (int i = 0; i < pr->n_segs; i++)
                if (pr->kind[i] == LG_SEG_DR_JOINT || pr->kind[i] <= LG_SEG_END || pr->kind[i] > LG_SEG_KD) return 0;
        return 4;
    }
    return 0;
}

// tron1_pf_ee (PROF 6 = the component-layout tail of lg_quad.h for the three-joint biped): what it hard-wires
static bool biped_profile(const LgEngine *h, bool inj = false) {
    const LgTaskCfg &t = h->task;
    const LgBuffers &b = h->bufs;
    if (h->model.n_legs != 2 || h->model.n_bodies != 7 || !h->hf || h->opts.terrain_rows <= 0) return false;
    const int K = __builtin_popcount(h->model.state_link_mask), P = h->opts.n_height_points, A = 6, F = 2;
    return t.obs_layout == LG_OBS_TRON1_EE && t.gait_mode == 2 && t.double_shift == 1 && t.cat_enable == 0 && t.behavior_resample_steps == 0 &&
           t.obs_slack > 0 && t.obs_stack > 1 && t.priv_stack > 1 && t.obs_sets <= 2 && (t.obs_sets < 2 || b.obs_dirty) && (inj ? b.rand_in != nullptr : !b.rand_in) &&
           t.obs_frame == 9 + 3 * A + 2 * F && t.priv_frame == t.obs_frame + 7 + 2 * A + 3 + F + K + P + 3 * F + 9 * F && t.num_labels == 3 + K + F + 3 * F &&
           P > 0 && P <= 7 * 8 && h->opts.feet_terrain_info && b.priv_obs_buf && b.labels_buf && b.task_state && b.rand_push_vels &&
           t.task_state_width == LG_TASK_STATE_BIPED && b.link_contact_states &&
           (!t.terrain_curriculum || (b.terrain_levels && b.terrain_types && b.terrain_origins && b.env_origins)) &&
           (!t.dr_joint_on || (b.joint_armature && b.joint_friction && b.joint_damping)) &&
           (t.slots.reset_root_xy & 3) != 3 &&       // the two root xy draws share a Philox block
           // 32-bit byte offsets into the observation allocations (lg_quad.h)
           (double)b.n_envs * (t.priv_stack + t.obs_slack) * t.priv_frame * 4.0 < 4.0e9 && (double)b.n_envs * (t.obs_stack + t.obs_slack) * t.obs_frame * 4.0 < 4.0e9 &&
           // commands carry no observation noise (tron1_pf_ee.py:322-342)
           t.noise_vec[0] == 0.f && t.noise_vec[1] == 0.f && t.noise_vec[2] == 0.f &&
           ((unsigned)h->hot.reward_mask & ((1u << LG_R_QUAD_PERIODIC_GAIT) | (1u << LG_R_TRACKING_FOOT_CLEARANCE) | (1u << LG_R_TRACKING_ORIENTATION))) == 0;
}

template <int LEGS, int JPL = 3> static int launch(LgEngine *h, uint32_t ph, const float *actions, int64_t counter, hipStream_t st) {
    KParams p;
    p.M = h->d_model; p.O = h->d_opts; p.T = h->d_task; p.H = h->d_hot; p.LT = h->d_lane; p.hf = h->hf; p.B = h->bufs; p.actions = actions; p.counter = counter;
    p.jrot_identity = 1;
    {
        const LgHot &hot = h->hot;   // refreshed by upload_hot
        p.k.m_n_links = hot.m_n_links;
        for (int i = 0; i < 4; i++) p.k.m_foot_link[i] = hot.m_foot_link[i];
        p.k.obs_layout = hot.obs_layout; p.k.o_n_height_points = hot.o_n_height_points;
        p.k.reward_mask = (unsigned)hot.reward_mask; p.k.clip_actions = hot.clip_actions;
        p.k.cat_enable = h->task.cat_enable;
        p.k.joint_axis[3] = -1;
        for (int j = 0; j < JPL; j++) {
            int code = -2;
            for (int l = 0; l < LEGS; l++) {
                const float *ax = h->model.axis[1 + JPL * l + j];
                int c = -1;
                for (int k = 0; k < 3; k++)
                    if (fabsf(fabsf(ax[k]) - 1.f) < 1e-6f && fabsf(ax[(k + 1) % 3]) < 1e-6f && fabsf(ax[(k + 2) % 3]) < 1e-6f) c = k;
                code = (code == -2 || code == c) ? c : -1;
            }
            p.k.joint_axis[j] = code;
        }
    }
    for (int b = 1; b < h->model.n_bodies; b++)
        for (int k = 0; k < 9; k++)
            if (h->model.jrot[b][k] != ((k % 4 == 0) ? 1.f : 0.f)) p.jrot_identity = 0;
    const int threads = h->bufs.n_envs * LEGS;
    dim3 grid((threads + BLOCK - 1) / BLOCK), block(BLOCK);
    p.obs_win = 0;
    p.obs_set = h->obs_set;
    if (ph & LG_PHASE_RESET) {
        const LgTaskCfg &t = h->task;
        const int sets = t.obs_sets > 1 ? t.obs_sets : 1;
        if (sets > 1) p.obs_set = h->obs_set = (h->obs_set + 1) % sets;      // this launch writes a copy the caller is NOT holding
        if (t.obs_slack > 0) {
            // the observation written by this launch lives one frame further; out of slack -> compact first (source and
            // destination ranges are disjoint because lg_create enforces slack >= stack, and with two sets the window the
            // caller still holds, [slack, slack + stack), is clear of everything written here because slack >= stack + 1).
            // Both sets are compacted together: they are one allocation of sets * n rows
            if (h->obs_win >= t.obs_slack) {
                const int n = h->bufs.n_envs * sets;
                if (t.obs_stack > 1)
                    hipLaunchKernelGGL(obs_compact_kernel, dim3(1024), dim3(256), 0, st, h->bufs.obs_buf, n, (t.obs_stack + t.obs_slack) * t.obs_frame, t.obs_frame, t.obs_stack, h->obs_win);
                if (t.num_priv_obs > 0 && t.priv_stack > 1)
                    hipLaunchKernelGGL(obs_compact_kernel, dim3(1024), dim3(256), 0, st, h->bufs.priv_obs_buf, n, (t.priv_stack + t.obs_slack) * t.priv_frame, t.priv_frame, t.priv_stack, h->obs_win);
                h->obs_win = 0;
            }
            p.obs_win = ++h->obs_win;
        }
    }
    // physics layout (lg_quad.h): one vector component per lane while the batch cannot fill the SIMDs with one leg per
    // lane; the MDP phases then follow in a second launch on the same stream
    // auto: component-per-lane while that needs at most two waves per SIMD (1024 SIMDs).  Measured go2, us per step,
    // component vs leg layout: 4096 envs 34.7 / 56.6, 8192: 50.9 / 55.2, 12288: 69.2 / 55.5, 16384: 92 / 60
    // the component-per-lane kernel is specialised for identity joint frames and hip-x / thigh-y / knee-y axes (lg_quad.h)
    // MDP-only launches of small biped batches run replicated (env_step_kernel<..., REPL>) while four times the waves still fit one per
    // SIMD.  Measured at 4096 envs, us per step, plain / replicated: tron1_pf_ee 55.8 / 53.7, tron1_sf 52.3 / 50.7, tron1_pf 43.4 / 43.4;
    // go2_cat (16 k leg-lanes already) 59.4 / 59.6: quadrupeds stay plain.  LG_MDP_REPLICAS=0/1 forces either (read per call: tests flip it).
    const char *repl_s = getenv("LG_MDP_REPLICAS");
    const bool repl = repl_s ? atoi(repl_s) != 0 : (LEGS == 2 && (long long)threads * 4 <= 1024LL * BLOCK);
    const dim3 rgrid((threads + 15) / 16);
    const bool quad_ok = p.jrot_identity && p.k.joint_axis[0] == 0 && p.k.joint_axis[1] == 1 && p.k.joint_axis[2] == 1 && (JPL == 3 || p.k.joint_axis[3] == 1);
    if (h->opts.sim_layout == 2 && !quad_ok) return fail("lg_step: sim_layout 2 needs identity joint frames and x / y / y (/ y) joint axes");
    const int layout = h->opts.sim_layout ? h->opts.sim_layout : ((quad_ok && (long long)threads * 4 <= 2048LL * BLOCK) ? 2 : 1);
    constexpr unsigned PR = LG_PHASE_POST | LG_PHASE_RESET;
    if constexpr (JPL == 4) if (layout == 2 && (ph & LG_PHASE_SIM)) {   // four-joint legs: physics in component layout, then the MDP phases
        dim3 qgrid((threads * 4 + BLOCK - 1) / BLOCK);
        const bool pre = (ph & LG_PHASE_PRE) != 0;
        if (!pre && !actions) p.actions = nullptr;
        const int pi = prof_begin(h, st);
        const uint32_t rest = ph & PR;
        if (pre && rest) LG_LAUNCH_FIRST(pi, (lg_launch_quad<LEGS, true, 0u, 0, 4>), qgrid);
        else if (pre) LG_LAUNCH(pi, (lg_launch_quad<LEGS, true, 0u, 0, 4>), qgrid);
        else if (rest) LG_LAUNCH_FIRST(pi, (lg_launch_quad<LEGS, false, 0u, 0, 4>), qgrid);
        else LG_LAUNCH(pi, (lg_launch_quad<LEGS, false, 0u, 0, 4>), qgrid);
        HIPCHK(hipGetLastError());
        if (rest == PR && repl) LG_LAUNCH_LAST(pi, (lg_launch_env<LEGS, PR, 0, 4, true>), rgrid);
        else if (rest == PR) LG_LAUNCH_LAST(pi, (lg_launch_env<LEGS, PR, 0, 4, false>), grid);
        else if (rest == LG_PHASE_POST) LG_LAUNCH_LAST(pi, (lg_launch_env<LEGS, LG_PHASE_POST, 0, 4, false>), grid);
        else if (rest) return fail("lg_step: unsupported phase combination");
        HIPCHK(hipGetLastError());
        return 0;
    }
    if constexpr (JPL == 3) if (layout == 2 && (ph & LG_PHASE_SIM)) {
        dim3 qgrid((threads * 4 + BLOCK - 1) / BLOCK);
        const bool pre = (ph & LG_PHASE_PRE) != 0;
        if (!pre && !actions) p.actions = nullptr;
        const int pi = prof_begin(h, st);
        const uint32_t rest = ph & PR;
        // MDP phases: in the tail of the same launch for the quadruped (measured 39.5 vs 40.6 us for go2, 69.6 vs 72.5
        // for go2_ee), as a second launch for the biped (84.7 vs 90.6 us for tron1_pf_ee: 8 envs per wave there)
        bool fuse = false;
        const bool hfb = h->hf != nullptr && h->opts.terrain_rows > 0;
        if constexpr (LEGS == 4) {
            fuse = pre && rest != 0;
            // rs(P): the task's reward set is the default one of profile P -> the instantiation that has it as a constant (lg_quad.h RS)
            const char *rse = getenv("LG_REWARD_SET_CONST");
            auto rs = [&](int prof) { return (unsigned)h->hot.reward_mask == lg_default_reward_mask(prof) && !(rse && atoi(rse) == 0); };
            if (fuse && rest == PR && flat_profile(h)) { if (rs(1)) LG_LAUNCH(pi, (lg_launch_quad_rs<4, 1, false>), qgrid); else LG_LAUNCH(pi, (lg_launch_quad<4, true, PR, 1, 3>), qgrid); }
            else if (fuse && rest == PR && wtw_profile(h)) { if (rs(2)) LG_LAUNCH(pi, (lg_launch_quad_rs<4, 2, false>), qgrid); else LG_LAUNCH(pi, (lg_launch_quad<4, true, PR, 2, 3>), qgrid); }
            else if (fuse && rest == PR && rough_profile(h) == 3) { if (rs(3)) LG_LAUNCH(pi, (lg_launch_quad_rs<4, 3, false>), qgrid); else LG_LAUNCH(pi, (lg_launch_quad<4, true, PR, 3, 3>), qgrid); }
            else if (fuse && rest == PR && rough_profile(h) == 4) { if (rs(4)) LG_LAUNCH(pi, (lg_launch_quad_rs<4, 4, false>), qgrid); else LG_LAUNCH(pi, (lg_launch_quad<4, true, PR, 4, 3>), qgrid); }
            else if (fuse && rest == PR) LG_LAUNCH(pi, (lg_launch_quad<4, true, PR, 0, 3>), qgrid);
            else if (fuse && rest == LG_PHASE_POST) LG_LAUNCH(pi, (lg_launch_quad<4, true, LG_PHASE_POST, 0, 3>), qgrid);
        } else {
            // biped (TRON1 point foot): the whole step in one launch too -- the leg-per-lane MDP body in the tail, four replicas of the
            // wave's 16 leg-lanes (8 envs) -- unless the job-wide CaT flag has to pass between the phases.  LG_BIPED_FUSE=0: two launches
            const char *bf = getenv("LG_BIPED_FUSE");
            fuse = pre && rest == PR && !h->task.cat_enable && (bf ? atoi(bf) != 0 : true);
            const char *bt = getenv("LG_BIPED_TAIL");    // 0: the leg-per-lane MDP body in the tail even where the component-layout tail applies
            // PROF 6: workgroups of two waves per group of 8 envs (lg_quad.h DUO: both run the physics, then split the tail)
            if (fuse && hfb && biped_profile(h) && !(bt && atoi(bt) == 0)) LG_LAUNCH(pi, (lg_launch_quad<2, true, PR, 6, 3>), qgrid);
            else if (fuse && hfb) LG_LAUNCH(pi, (lg_launch_quad<2, true, PR, 5, 3>), qgrid);
            else if (fuse) LG_LAUNCH(pi, (lg_launch_quad<2, true, PR, 0, 3>), qgrid);
        }
        // physics-only launches: PROF 3 here only says "a heightfield is bound" (lg_quad.h HFC: no branch in front of the terrain loads)
        if (fuse) {}
        else if (pre && rest && hfb) LG_LAUNCH_FIRST(pi, (lg_launch_quad<LEGS, true, 0u, 3, 3>), qgrid);
        else if (pre && rest) LG_LAUNCH_FIRST(pi, (lg_launch_quad<LEGS, true, 0u, 0, 3>), qgrid);
        else if (pre && hfb) LG_LAUNCH(pi, (lg_launch_quad<LEGS, true, 0u, 3, 3>), qgrid);
        else if (pre) LG_LAUNCH(pi, (lg_launch_quad<LEGS, true, 0u, 0, 3>), qgrid);
        else if (rest && hfb) LG_LAUNCH_FIRST(pi, (lg_launch_quad<LEGS, false, 0u, 3, 3>), qgrid);
        else if (rest) LG_LAUNCH_FIRST(pi, (lg_launch_quad<LEGS, false, 0u, 0, 3>), qgrid);
        else if (hfb) LG_LAUNCH(pi, (lg_launch_quad<LEGS, false, 0u, 3, 3>), qgrid);
        else LG_LAUNCH(pi, (lg_launch_quad<LEGS, false, 0u, 0, 3>), qgrid);
        HIPCHK(hipGetLastError());
        if (!fuse && rest) {
            if (rest == PR && repl) LG_LAUNCH_LAST(pi, (lg_launch_env<LEGS, PR, 0, JPL, true>), rgrid);
            else if (rest == PR) LG_LAUNCH_LAST(pi, (lg_launch_env<LEGS, PR, 0, JPL, false>), grid);
            else if (rest == LG_PHASE_POST) LG_LAUNCH_LAST(pi, (lg_launch_env<LEGS, LG_PHASE_POST, 0, JPL, false>), grid);
            else return fail("lg_step: unsupported phase combination");
        }
        HIPCHK(hipGetLastError());
        return 0;
    }
    if constexpr (JPL == 3) if (ph == (LG_PHASE_PRE | PR) && h->opts.sim_layout == 2 && quad_ok && h->bufs.rand_in) {
        // golden replays through the benchmarked tails (tests/test_gpu_mdp.py, tail = "fused-profile"): the component-layout tail of the
        // task's profile on injected read-backs and uniforms (lg_quad.h INJ); a task outside every profile takes the leg-per-lane launch below
        dim3 qgrid((threads * 4 + BLOCK - 1) / BLOCK);
        bool done = true;
        if constexpr (LEGS == 4) {
            const int rp = rough_profile(h, true);
            // the same choice of instantiation as the whole-step launch: constant reward set where the task's is the profile's default
            const char *rse = getenv("LG_REWARD_SET_CONST");
            auto rs = [&](int prof) { return (unsigned)h->hot.reward_mask == lg_default_reward_mask(prof) && !(rse && atoi(rse) == 0); };
#define LG_INJ(L_, P_) do { if (rs(P_)) { h->last_kernel = "(lg_launch_quad_inj<" #L_ ", " #P_ ">, reward set constant)"; lg_launch_quad_rs<L_, P_, true>(qgrid, st, nullptr, nullptr, p); } \
                            else { h->last_kernel = "(lg_launch_quad_inj<" #L_ ", " #P_ ">)"; lg_launch_quad_inj<L_, P_>(qgrid, st, p); } } while (0)
            if (flat_profile(h, true)) LG_INJ(4, 1);
            else if (wtw_profile(h, true)) LG_INJ(4, 2);
            else if (rp == 3) LG_INJ(4, 3);
            else if (rp == 4) LG_INJ(4, 4);
            else done = false;
#undef LG_INJ
        } else {
            if (biped_profile(h, true)) { h->last_kernel = "(lg_launch_quad_inj<2, 6>)"; lg_launch_quad_inj<2, 6>(qgrid, st, p); }
            else done = false;
        }
        if (done) { HIPCHK(hipGetLastError()); return 0; }
    }
    const int pi = (ph & LG_PHASE_SIM) ? prof_begin(h, st) : -1;
    switch (ph) {
    case LG_PHASE_ALL: {
        // LG_SPLIT_ALL=1 (developer switch): PRE | SIM and POST | RESET as two launches instead of the whole-step kernel, whose 480-512
        // registers per lane spill a few dwords to scratch (tools/register_table.py)
        const char *sp = getenv("LG_SPLIT_ALL");
        if (sp && atoi(sp) != 0) {
            LG_LAUNCH_FIRST(pi, (lg_launch_env<LEGS, LG_PHASE_PRE | LG_PHASE_SIM, 0, JPL, false>), grid);
            LG_LAUNCH_LAST(pi, (lg_launch_env<LEGS, PR, 0, JPL, false>), grid);
            break;
        }
        if constexpr (JPL == 3 && LEGS == 4) { if (flat_profile(h)) { LG_LAUNCH(pi, (lg_launch_env<4, LG_PHASE_ALL, 1, 3, false>), grid); break; } }   // large go2 batches: same FLAT constants
        LG_LAUNCH(pi, (lg_launch_env<LEGS, LG_PHASE_ALL, 0, JPL, false>), grid);
        break;
    }
    case LG_PHASE_SIM: LG_LAUNCH(pi, (lg_launch_env<LEGS, LG_PHASE_SIM, 0, JPL, false>), grid); break;
    case LG_PHASE_PRE | LG_PHASE_POST | LG_PHASE_RESET: LG_LAUNCH_PLAIN((lg_launch_env<LEGS, LG_PHASE_PRE | PR, 0, JPL, false>), grid); break;
    case LG_PHASE_PRE | LG_PHASE_SIM | LG_PHASE_POST: LG_LAUNCH(pi, (lg_launch_env<LEGS, LG_PHASE_PRE | LG_PHASE_SIM | LG_PHASE_POST, 0, JPL, false>), grid); break;
    case LG_PHASE_PRE | LG_PHASE_SIM: LG_LAUNCH(pi, (lg_launch_env<LEGS, LG_PHASE_PRE | LG_PHASE_SIM, 0, JPL, false>), grid); break;
    case LG_PHASE_RESET: LG_LAUNCH_PLAIN((lg_launch_env<LEGS, LG_PHASE_RESET, 0, JPL, false>), grid); break;
    case LG_PHASE_PRE | LG_PHASE_POST: LG_LAUNCH_PLAIN((lg_launch_env<LEGS, LG_PHASE_PRE | LG_PHASE_POST, 0, JPL, false>), grid); break;
    case LG_PHASE_POST | LG_PHASE_RESET:
        if (repl) LG_LAUNCH_PLAIN((lg_launch_env<LEGS, PR, 0, JPL, true>), rgrid);
        else LG_LAUNCH_PLAIN((lg_launch_env<LEGS, PR, 0, JPL, false>), grid);
        break;
    case LG_PHASE_POST: LG_LAUNCH_PLAIN((lg_launch_env<LEGS, LG_PHASE_POST, 0, JPL, false>), grid); break;
    default: return fail("lg_step: unsupported phase combination (ALL, SIM, PRE|SIM, PRE|POST|RESET, PRE|SIM|POST, PRE|POST, POST|RESET, POST, RESET)");
    }
    HIPCHK(hipGetLastError());
    return 0;
}


static int check_mdp_bufs(const LgEngine *h, uint32_t ph) {
    const LgBuffers &b = h->bufs;
    if (ph & (LG_PHASE_PRE | LG_PHASE_POST | LG_PHASE_RESET)) {
#define REQ(f) if (!b.f) return fail("lg_step: MDP phases need buffer " #f)
        REQ(actions); REQ(last_actions); REQ(llast_actions); REQ(commands); REQ(feet_air_time); REQ(last_contacts);
        REQ(episode_length_buf); REQ(fail_buf); REQ(reset_buf); REQ(time_out_buf); REQ(rew_buf); REQ(obs_buf);
        REQ(episode_sums); REQ(episode_done_sums); REQ(episode_done_step); REQ(command_ranges); REQ(rand_push_vels);
        REQ(friction_values); REQ(added_base_mass); REQ(base_com_bias); REQ(kp_scale); REQ(kd_scale);
#undef REQ
        if (h->task.cat_enable && (!b.cstr_prob || !b.cstr_sums || !b.cstr_done_sums)) return fail("lg_step: cat_enable needs buffers cstr_prob, cstr_sums, cstr_done_sums");
        if (h->task.obs_sets > 1 && h->task.obs_slack > 0 && !b.obs_dirty) return fail("lg_step: two observation sets with history stacks need buffer obs_dirty");
    }
    return 0;
}

extern "C" int lg_step(LgHandle h, uint32_t phases, const float *actions, int64_t counter, void *stream) {
    if (!h) return fail("lg_step: null handle");
    if (!h->bound) return fail("lg_step: lg_bind has not been called");
    if ((phases & (LG_PHASE_PRE | LG_PHASE_SIM)) && !actions && (phases & LG_PHASE_PRE || !(phases & LG_PHASE_POST)))
        return fail("lg_step: actions pointer is NULL");
    if (h->opts.terrain_rows > 0 && !h->hf) return fail("lg_step: heightfield options set but lg_set_terrain was not called");
    if (check_mdp_bufs(h, phases)) return 1;
    if (h->task.cat_enable && (phases & LG_PHASE_SIM) && (phases & LG_PHASE_POST))
        return fail("lg_step: with cat_enable the physics and the MDP phases must be separate launches (job-wide constraint flag, LG_CR_ANY_FAST)");
    hipStream_t st = (hipStream_t)stream;
    // Bounded run-ahead.  A host that enqueues thousands of launches ahead of the device (a bench loop without a
    // policy in between) drives the runtime into a slow submission path: measured 90 us per step instead of 42 with
    // ~4000 launches in flight.  Every 32nd call records an event; before a slot is reused (128 calls later) the host
    // waits for it, which is free unless it really is that far ahead.
    if ((h->ra_calls++ & 31) == 0) {
        const int slot = (int)((h->ra_calls >> 5) & 3);
        if (h->ra_ev[slot]) HIPCHK(hipEventSynchronize(h->ra_ev[slot]));
        else HIPCHK(hipEventCreateWithFlags(&h->ra_ev[slot], hipEventDisableTiming));
        HIPCHK(hipEventRecord(h->ra_ev[slot], st));
    }
    if (h->model.n_bodies == 1 + 4 * h->model.n_legs) return launch<2, 4>(h, phases, actions, counter, st);   // validate_model: two legs
    return h->model.n_legs == 4 ? launch<4>(h, phases, actions, counter, st) : launch<2>(h, phases, actions, counter, st);
}

__global__ __launch_bounds__(256) void stream_copy_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    // one 16-byte load and store per lane and iteration: measured 5.0-5.1 TB/s (read + write) on an MI355X; four loads in flight per lane
    // before the stores measured 4.5, torch's elementwise copy_ 4.8
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}

extern "C" int lg_stream_copy(const void *src, void *dst, int64_t bytes, int32_t iters, void *stream, float *gbs) {
    if (!src || !dst || !gbs || bytes < 16 || (bytes & 15) || iters < 1) return fail("lg_stream_copy: bad argument");
    if (((uintptr_t)src | (uintptr_t)dst) & 15) return fail("lg_stream_copy: pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const long long n4 = bytes / 16;
    // 256 CUs x 8 workgroups of 256 threads: every SIMD holds two waves, each streaming 16 B per lane per iteration
    const dim3 grid((unsigned)std::min<long long>((n4 + 255) / 256, 256 * 8)), block(256);
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(stream_copy_kernel, grid, block, 0, st, (const float4 *)src, (float4 *)dst, n4);
    HIPCHK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL(stream_copy_kernel, grid, block, 0, st, (const float4 *)src, (float4 *)dst, n4);
    HIPCHK(hipEventRecord(e1, st));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    HIPCHK(hipGetLastError());
    *gbs = ms > 0.f ? (float)(2.0 * (double)bytes * iters / (ms * 1e-3) / 1e9) : 0.f;
    return 0;
}

// ---- heightfield generation (include/lgsim.h lg_terrain_generate) ---------------------------------------------------------------------
// One thread per pixel of a tile.  Everything float is float64 with the reference's (numpy / FITPACK) operation order and NO contraction
// into fused multiply-adds (__dmul_rn / __dadd_rn / __ddiv_rn): the grid has to come out sample for sample.
__global__ __launch_bounds__(256) void terrain_tile_kernel(const LgTerrainTile *__restrict__ tiles, const double *__restrict__ aux,
                                                           const int32_t *__restrict__ iaux, int16_t *__restrict__ hf, int cols, int W, int border) {
    const LgTerrainTile T = tiles[blockIdx.y];
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= W * W) return;
    const int x = pix / W, y = pix % W;
    int v = 0;
    if (T.kind == LG_TILE_SLOPE) {
        // (peak * ramp_x * ramp_y).astype(int16) inside the edge, then clip to [min(ref, 0), max(ref, 0)] (terrain_utils.py:160-181)
        const int peak = T.ip[0], edge = T.ip[1], half = T.ip[2], c = W / 2;
        auto val = [&](int px, int py) {
            if (px < edge || px >= W - edge || py < edge || py >= W - edge) return 0;
            const double rx = __ddiv_rn((double)(c - abs(c - px)), (double)c), ry = __ddiv_rn((double)(c - abs(c - py)), (double)c);
            return (int)(int16_t)__dmul_rn(__dmul_rn((double)peak, rx), ry);
        };
        const int ref = val(W / 2 - half, W / 2 - half);
        v = min(max(val(x, y), min(ref, 0)), max(ref, 0));
    } else if (T.kind == LG_TILE_STAIRS) {
        // nested squares [r w, W - r w)^2 at height r h while the square is wider than the platform (terrain_utils.py:361-371)
        const int w = T.ip[0], h = T.ip[1], plat = T.ip[2];
        int rings = 0;
        for (int lo = 0, hi = W; hi - lo > plat; lo += w, hi -= w) rings++;
        const int d = min(min(x, W - 1 - x), min(y, W - 1 - y));
        v = rings > 0 ? min(d / w, rings - 1) * h : 0;
    } else if (T.kind == LG_TILE_OBSTACLES) {
        const int n = T.ip[0], plat = T.ip[1];
        const int32_t *r = iaux + T.aux_off;
        for (int k = 0; k < n; k++)          // later rectangles overwrite earlier ones
            if (x >= r[5 * k] && x < r[5 * k] + r[5 * k + 2] && y >= r[5 * k + 1] && y < r[5 * k + 1] + r[5 * k + 3]) v = r[5 * k + 4];
        const int a = (W - plat) / 2, b = (W + plat) / 2;
        if (x >= a && x < b && y >= a && y < b) v = 0;
    } else if (T.kind == LG_TILE_UNIFORM) {
        // np.rint(RectBivariateSpline(kx = ky = 1)(xs, ys)) inside the edge: FITPACK fpbisp / fpbspl for degree 1
        const int edge = T.ip[0], nx = T.ip[1], ny = T.ip[2], mx = T.ip[3], my = T.ip[4];
        const int i = x - edge, j = y - edge;
        if (i >= 0 && i < mx && j >= 0 && j < my) {
            const double *tx = aux + T.aux_off, *ty = tx + nx, *cf = ty + ny, *xs = cf + (nx - 2) * (ny - 2), *ys = xs + mx;
            auto basis = [](const double *t, int n, double arg, int &l, double &h1, double &h2) {
                const int k1 = 2, nk1 = n - k1;                       // fpbisp: clamp, then the knot interval (1-based l: t(l) <= arg < t(l+1))
                const double tb = t[k1 - 1], te = t[nk1];
                if (arg < tb) arg = tb;
                if (arg > te) arg = te;
                l = k1;
                while (!(arg < t[l]) && l != nk1) l++;                // t(l1) with l1 = l + 1 is t[l] in 0-based terms
                const double f = __ddiv_rn(1.0, __dadd_rn(t[l], -t[l - 1]));      // fpbspl, k = 1
                h1 = __dmul_rn(f, __dadd_rn(t[l], -arg));
                h2 = __dmul_rn(f, __dadd_rn(arg, -t[l - 1]));
            };
            int lx, ly;
            double wx1, wx2, wy1, wy2;
            basis(tx, nx, xs[i], lx, wx1, wx2);
            basis(ty, ny, ys[j], ly, wy1, wy2);
            const int nky1 = ny - 2;
            int l1 = (lx - 2) * nky1 + (ly - 2);
            double sp = 0.0;
            sp = __dadd_rn(sp, __dmul_rn(__dmul_rn(cf[l1], wx1), wy1));
            sp = __dadd_rn(sp, __dmul_rn(__dmul_rn(cf[l1 + 1], wx1), wy2));
            l1 += nky1;
            sp = __dadd_rn(sp, __dmul_rn(__dmul_rn(cf[l1], wx2), wy1));
            sp = __dadd_rn(sp, __dmul_rn(__dmul_rn(cf[l1 + 1], wx2), wy2));
            v = (int)(int16_t)rint(sp);                               // np.rint: half to even
        }
    }
    hf[(size_t)(border + T.row * W + x) * cols + border + T.col * W + y] = (int16_t)v;
}

__global__ __launch_bounds__(256) void terrain_origin_kernel(const LgTerrainTile *__restrict__ tiles, const int16_t *__restrict__ hf, int cols, int W,
                                                             int border, int o1, int o2, double vscale, double *__restrict__ origin_z) {
    __shared__ int smax[256];
    const LgTerrainTile T = tiles[blockIdx.x];
    const int n = o2 - o1;
    int m = -32768;
    for (int k = threadIdx.x; k < n * n; k += blockDim.x)
        m = max(m, (int)hf[(size_t)(border + T.row * W + o1 + k / n) * cols + border + T.col * W + o1 + k % n]);
    smax[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) smax[threadIdx.x] = max(smax[threadIdx.x], smax[threadIdx.x + s]); __syncthreads(); }
    if (threadIdx.x == 0) origin_z[blockIdx.x] = __dmul_rn((double)smax[0], vscale);
}

extern "C" int lg_terrain_generate(const LgTerrainTile *tiles, int32_t n_tiles, const double *aux, const int32_t *iaux, int16_t *hf, int32_t rows,
                                   int32_t cols, int32_t tile_px, int32_t border_px, int32_t o1, int32_t o2, double vertical_scale, double *origin_z,
                                   void *stream) {
    if (!tiles || !hf || !origin_z || n_tiles < 1 || tile_px < 1 || rows < 2 || cols < 2 || o1 < 0 || o2 <= o1 || o2 > tile_px)
        return fail("lg_terrain_generate: bad argument");
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(hf, 0, (size_t)rows * cols * sizeof(int16_t), st));
    hipLaunchKernelGGL(terrain_tile_kernel, dim3((tile_px * tile_px + 255) / 256, n_tiles), dim3(256), 0, st, tiles, aux, iaux, hf, cols, tile_px, border_px);
    hipLaunchKernelGGL(terrain_origin_kernel, dim3(n_tiles), dim3(256), 0, st, tiles, hf, cols, tile_px, border_px, o1, o2, vertical_scale, origin_z);
    HIPCHK(hipGetLastError());
    return 0;
}

__global__ void philox_kat_kernel(U4 c, unsigned k0, unsigned k1, unsigned *out) {
    const U4 r = philox4x32_10(c, k0, k1);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

extern "C" int lg_philox(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]) {
    if (!counter || !key || !out) return fail("lg_philox: null argument");
    unsigned *d = nullptr;
    HIPCHK(hipMalloc(&d, 16));
    const U4 c = {counter[0], counter[1], counter[2], counter[3]};
    hipLaunchKernelGGL(philox_kat_kernel, dim3(1), dim3(1), 0, 0, c, key[0], key[1], d);
    hipError_t e = hipMemcpy(out, d, 16, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(std::string("lg_philox: ") + hipGetErrorString(e));
    return 0;
}

// DPP operand known-answer test (tests/test_gpu_dpp_operands.py): the behaviour of THIS chip that hcr_genesis_lr_cl_amd/dpp_hazard_pass.py builds on.
// The rule "VALU writes a VGPR -> DPP read: two wait states" concerns the DPP-routed source (src0) only; a plain operand (src1) or the
// accumulator of a DPP multiply-add written in the slot before is forwarded like for any VALU instruction -- cases 0 and 1, no wait state, must
// be exact.  Case 2: a fresh DPP source behind `s_nop 1` -- exact.  Case 3: the same without the nop -- the negative control, reported, not
// required either way (it reads whatever the register held before: here a NaN).  This file is compiled with plain hipcc: the pass never sees it.
__global__ void dpp_kat_kernel(const float *in, float *out) {
    const int t = threadIdx.x;
    const float x = in[t], y = in[64 + t];
    float a, r0, r1, r2, r3;
    asm volatile(
        "s_nop 4\n\t"
        "v_mul_f32_e32 %0, %5, %6\n\t"
        "v_add_f32_dpp %1, %5, %0 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_e32 %2, %5, %6\n\t"
        "v_fmac_f32_dpp %2, %5, %6 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_e32 %3, %5, %6\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %3, %3, %5 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_e32 %4, 0x7fc00000\n\t"
        "s_nop 4\n\t"
        "v_mul_f32_e32 %4, %5, %6\n\t"
        "v_add_f32_dpp %4, %4, %5 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 4"
        : "=&v"(a), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(x), "v"(y));
    out[t] = r0; out[64 + t] = r1; out[128 + t] = r2; out[192 + t] = r3; out[256 + t] = a;
}

extern "C" int lg_dpp_kat(const float *in_host, float *out_host) {
    if (!in_host || !out_host) return fail("lg_dpp_kat: null argument");
    float *d = nullptr;
    HIPCHK(hipMalloc(&d, (128 + 320) * sizeof(float)));
    HIPCHK(hipMemcpy(d, in_host, 128 * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(dpp_kat_kernel, dim3(1), dim3(64), 0, 0, d, d + 128);
    hipError_t e = hipMemcpy(out_host, d + 128, 320 * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(std::string("lg_dpp_kat: ") + hipGetErrorString(e));
    return 0;
}

extern "C" const char *lg_last_kernel(LgHandle h) { return h ? h->last_kernel.c_str() : ""; }

extern "C" int lg_obs_window(LgHandle h, int32_t *first_frame) {
    if (!h || !first_frame) return fail("lg_obs_window: null argument");
    *first_frame = h->task.obs_slack > 0 ? h->obs_win : 0;
    return 0;
}

extern "C" int lg_obs_set(LgHandle h, int32_t *set) {
    if (!h || !set) return fail("lg_obs_set: null argument");
    *set = h->task.obs_sets > 1 ? h->obs_set : 0;
    return 0;
}

extern "C" int lg_obs_set_select(LgHandle h, int32_t set) {
    if (!h) return fail("lg_obs_set_select: null handle");
    const int sets = h->task.obs_sets > 1 ? h->task.obs_sets : 1;
    if (set < 0 || set >= sets) return fail("lg_obs_set_select: set out of range");
    h->obs_set = set;
    return 0;
}

extern "C" int lg_obs_window_select(LgHandle h, int32_t first_frame) {
    if (!h) return fail("lg_obs_window_select: null handle");
    if (first_frame < 0 || first_frame > h->task.obs_slack) return fail("lg_obs_window_select: window out of range");
    h->obs_win = h->task.obs_slack > 0 ? first_frame : 0;
    return 0;
}

extern "C" int lg_profile(LgHandle h, int32_t stride) {
    if (!h || stride < 0) return fail("lg_profile: bad argument");
    h->prof_stride = stride; h->prof_seen = 0;
    return 0;
}

extern "C" int lg_profile_read(LgHandle h, float *mean_us, int32_t *samples) {
    if (!h || !mean_us || !samples) return fail("lg_profile_read: null argument");
    double sum = 0.0;
    for (int i = 0; i < h->prof_count; i++) {
        float ms = 0.f;
        HIPCHK(hipEventSynchronize(h->prof_ev[2 * i + 1]));
        HIPCHK(hipEventElapsedTime(&ms, h->prof_ev[2 * i], h->prof_ev[2 * i + 1]));
        sum += ms;
    }
    *samples = h->prof_count;
    *mean_us = h->prof_count ? (float)(sum / h->prof_count * 1e3) : 0.f;
    h->prof_count = 0;
    return 0;
}

extern "C" int lg_time_steps(LgHandle h, const float *actions, int64_t first_counter, int32_t count, void *stream, float *ms) {
    if (!h || !ms || count < 1) return fail("lg_time_steps: bad argument");
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipEventRecord(h->ev0, st));
    for (int i = 0; i < count; i++)
        if (lg_step(h, LG_PHASE_ALL, actions, first_counter + i, stream)) return 1;
    HIPCHK(hipEventRecord(h->ev1, st));
    HIPCHK(hipEventSynchronize(h->ev1));
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, h->ev0, h->ev1));
    *ms = t / count;
    return 0;
}

