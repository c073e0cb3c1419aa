// lg_kernel.h -- the env-step kernel for gfx950 (MI355X), one leg per lane (instantiated by lg_inst.hip; C ABI: lg_host.hip, include/lgsim.h).
//
// Work decomposition: ONE LEG PER LANE.  A Go2 env is a quad of 4 consecutive lanes, a TRON1 env
// a pair; a wave64 carries 16 (resp. 32) envs.  Each lane owns the serial 3-joint chain of its
// leg (FK, articulated inertias, its foot's contact solve, its joints' integration); the
// floating base is replicated in the lanes of the quad and everything that couples legs
// (articulated inertia / bias force of the base, contact responses, per-env reward sums) is
// combined with DPP quad-permute adds -- no LDS, no atomics, no inter-workgroup traffic.
// At the headline size (4096 envs) one-env-per-thread would occupy 64 waves on a 1024-SIMD
// chip; leg-per-lane gives 256 waves = one per CU, and shortens the serial chain 4x.
//
// Dynamics formulation: Featherstone ABA with every spatial quantity expressed in WORLD-ALIGNED
// axes about the base origin O.  With a common reference point all Pluecker transforms are the
// identity, so child-to-parent accumulation of articulated inertias is a plain add, and the
// base (6x6 Cholesky) is solved redundantly per lane.  Reference call sites this replaces:
// legged_gym/simulator/genesis_simulator.py:20-60 (step + post_physics_step) and the MDP stack
// legged_gym/envs/base/legged_robot.py:37-168, 230-348, go2/go2.py:17-134.
//
// The CPU oracle (oracle/lg_oracle.c) implements the same physics in body-fixed coordinates
// with dense 6x6 algebra; the two share no code.
#pragma once
#include "lg_shared.h"

static_assert(LG_R_COUNT <= 32, "reward mask is one dword");
static_assert(sizeof(LgHot) <= 256 * 4, "hot block must fit four dwords per lane");
template <typename T> struct HotGet;
template <> struct HotGet<float> { static LG_DEV float get(int v0, int v1, int v2, int v3, int i) {
    const int v = i < 64 ? v0 : (i < 128 ? v1 : (i < 192 ? v2 : v3)); return __int_as_float(__builtin_amdgcn_readlane(v, i & 63)); } };
template <> struct HotGet<int32_t> { static LG_DEV int get(int v0, int v1, int v2, int v3, int i) {
    const int v = i < 64 ? v0 : (i < 128 ? v1 : (i < 192 ? v2 : v3)); return __builtin_amdgcn_readlane(v, i & 63); } };
template <> struct HotGet<unsigned long long> { static LG_DEV unsigned long long get(int v0, int v1, int v2, int v3, int i) {
    return (unsigned long long)(unsigned)HotGet<int32_t>::get(v0, v1, v2, v3, i) | ((unsigned long long)(unsigned)HotGet<int32_t>::get(v0, v1, v2, v3, i + 1) << 32); } };
template <> struct HotGet<long long> { static LG_DEV long long get(int v0, int v1, int v2, int v3, int i) {
    return (long long)HotGet<unsigned long long>::get(v0, v1, v2, v3, i); } };
#define HOT_OFF(f) ((int)(((size_t) & reinterpret_cast<char const volatile &>(((LgHot *)0)->f)) / 4))
// After the prologue the block lives in LDS (sHot): a field is one broadcast ds_read.  (Holding it in VGPR lanes for the
// whole kernel is NOT safe: register copies made under a partial exec mask drop the inactive lanes' entries.)  Integers
// steer control flow, so they are made scalar.
template <typename T> struct HotLds;
template <> struct HotLds<float> { static LG_DEV float get(const int *s, int i) { return __int_as_float(s[i]); } };
template <> struct HotLds<int32_t> { static LG_DEV int get(const int *s, int i) { return __builtin_amdgcn_readfirstlane(s[i]); } };
template <> struct HotLds<unsigned long long> { static LG_DEV unsigned long long get(const int *s, int i) {
    return (unsigned long long)(unsigned)HotLds<int32_t>::get(s, i) | ((unsigned long long)(unsigned)HotLds<int32_t>::get(s, i + 1) << 32); } };
template <> struct HotLds<long long> { static LG_DEV long long get(const int *s, int i) { return (long long)HotLds<unsigned long long>::get(s, i); } };
#define HOT_T(f) typename std::remove_cv<typename std::remove_reference<decltype(((LgHot *)0)->f)>::type>::type
#define CR(k) (__int_as_float(sHot[256 + (k)]))
// observation programs (LgTaskCfg.priv_prog / labels_prog, 26 dwords each: n_segs, clip, kind[8], offset[8], scale[8]) staged behind the
// command ranges: `which` 0 = priv_prog, 1 = labels_prog.  Read from T-> they were ~25 dependent scalar loads per launch.
#define PRG_I(which, dw) (__builtin_amdgcn_readfirstlane(sHot[256 + BLOCK + 26 * (which) + (dw)]))
#define PRG_F(which, dw) (__int_as_float(sHot[256 + BLOCK + 26 * (which) + (dw)]))
static_assert(sizeof(LgObsProgram) == 26 * 4, "program staging assumes 26 dwords");
#define HOT0(f) (HotGet<HOT_T(f)>::get(hv0, hv1, hv2, hv3, HOT_OFF(f)))   // prologue only (full exec, fresh registers)
#define HOT(f) (HotLds<HOT_T(f)>::get(sHot, HOT_OFF(f)))
// prologue accessor of env_step_body: lane registers in a stand-alone launch, LDS when fused behind the physics kernel
#define HOTB(f) (FUSED ? HOT(f) : HOT0(f))
#define RON(id) ((rmask >> (id)) & 1u)   // reward term `id` active (rmask: scalar copy of LgHot.reward_mask)

#ifdef LG_DBG_STAMPS
// diagnostic build only: shader-clock stamps of workgroup 0 / lane 0 into episode_done_sums[0..15]
#define STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { unsigned long long _t; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); \
    p.B.episode_done_sums[i] = (float)(_t - _stamp0); if (i == 0) _stamp0 = _t; } } while (0)
// per-workgroup stamp (every block): slot base + blockIdx.x
#define STAMPB(base) do { if (threadIdx.x == 0) { unsigned long long _t; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); \
    p.B.episode_done_sums[(base) + blockIdx.x] = (float)(_t & 0xFFFFFFull); } } while (0)
#else
#define STAMP(i) do { } while (0)
#define STAMPB(base) do { } while (0)
#endif

LG_DEV V3 ld3(const float *p) { return v3(p[0], p[1], p[2]); }
LG_DEV void st3(float *p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
LG_DEV float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
LG_DEV float rcp(float x) { return __builtin_amdgcn_rcpf(x); }  // 1 ulp; physics only, never the MDP arithmetic

// math_utils.py:64-76
LG_DEV V3 quat_rotate_inverse(float qx, float qy, float qz, float qw, V3 v) {
    V3 q = v3(qx, qy, qz);
    V3 a = v * (2.0f * qw * qw - 1.0f);
    V3 b = cross(q, v) * (2.0f * qw);
    V3 c = q * (2.0f * dot(q, v));
    return a - b + c;
}

// terrain height and unit normal at world (x, y); plane when rows == 0
LG_DEV void terrain_at(const LgSimOptions *O, const int16_t *hf, float x, float y, float &h, V3 &n) {
    if (O->terrain_rows <= 0) { h = 0.f; n = v3(0.f, 0.f, 1.f); return; }
    float gx = (x + O->border) / O->hscale, gy = (y + O->border) / O->hscale;
    int ix = (int)floorf(gx), iy = (int)floorf(gy);
    ix = min(max(ix, 0), O->terrain_rows - 2);
    iy = min(max(iy, 0), O->terrain_cols - 2);
    float fx = clampf(gx - ix, 0.f, 1.f), fy = clampf(gy - iy, 0.f, 1.f);
    int C = O->terrain_cols;
    float h00 = hf[ix * C + iy] * O->vscale, h10 = hf[(ix + 1) * C + iy] * O->vscale;
    float h01 = hf[ix * C + iy + 1] * O->vscale, h11 = hf[(ix + 1) * C + iy + 1] * O->vscale;
    h = (h00 * (1 - fx) + h10 * fx) * (1 - fy) + (h01 * (1 - fx) + h11 * fx) * fy;
    float hx = ((h10 - h00) * (1 - fy) + (h11 - h01) * fy) / O->hscale;
    float hy = ((h01 - h00) * (1 - fx) + (h11 - h10) * fx) / O->hscale;
    float inv = rsqrtf(hx * hx + hy * hy + 1.f);
    n = v3(-hx * inv, -hy * inv, inv);
}

// genesis_simulator.py:565-575: cell index by truncation, clipped, min over three neighbours
LG_DEV float sample_min3(const LgSimOptions *O, const int16_t *hf, float wx, float wy) {
    int px = (int)((wx + O->border) / O->hscale), py = (int)((wy + O->border) / O->hscale);
    px = min(max(px, 0), O->terrain_rows - 2);
    py = min(max(py, 0), O->terrain_cols - 2);
    const int C = O->terrain_cols;
    const int h1 = hf[px * C + py], h2 = hf[(px + 1) * C + py], h3 = hf[px * C + py + 1];
    return (float)min(min(h1, h2), h3) * O->vscale;
}

// one joint of the lane's chain, everything pass 3 / the response passes need
struct Joint {
    V6 S;       // motion subspace [s; P x s] about O, world axes
    V6 U;       // IA S
    V6 c;       // velocity-product acceleration
    float dinv; // 1 / (S^T IA S + armature)
    float u;    // tau - S^T pA
};

// per-substep world-frame kinematics of one body of the lane's chain
struct BodyKin {
    M3 R;   // body -> world
    V3 P;   // body origin relative to O (world axes)
    V6 V;   // spatial velocity about O
};

// response of the lane's chain + base to forces/torques that only this quad applies:
//   fspat: spatial force [r x f; f] on the distal body; tl[NJ]: joint torques.
// Up-sweep to the base contribution dp0 (to be quad-summed by the caller), then after the
// base solve the down-sweep gives joint accelerations and the distal body's acceleration.
template <int NJ> LG_DEV V6 resp_up(const Joint (&J)[NJ], const V6 &fspat, const float (&tl)[NJ], float (&du)[NJ]) {
    V6 dp = {-fspat.a, -fspat.l};
#pragma unroll
    for (int j = NJ - 1; j >= 0; j--) {
        du[j] = tl[j] - dot(J[j].S, dp);
        dp = dp + J[j].U * (du[j] * J[j].dinv);
    }
    return dp;
}
template <int NJ> LG_DEV V6 resp_down(const Joint (&J)[NJ], const V6 &a0, const float (&du)[NJ], float (&dqdd)[NJ]) {
    V6 a = a0;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        dqdd[j] = (du[j] - dot(J[j].U, a)) * J[j].dinv;
        a = a + J[j].S * dqdd[j];
    }
    return a;
}

// N consecutive floats with as few store instructions as possible (global_store_dwordx4 / x3 / x2 take dword-aligned addresses).
// A wave's store costs one cache-line transaction per distinct line it touches; in the leg-per-lane layout every lane writes into a
// different row, so the observation section's time is its number of store instructions (measured: 18 single-dword puts of the actor
// frame = 6.7 k cycles of the biped's MDP launch)
template <int N> LG_DEV void stv(float *p, const float *c) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef float v3f __attribute__((ext_vector_type(3)));
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef v4f __attribute__((aligned(4))) u4f;
    typedef v3f __attribute__((aligned(4))) u3f;
    typedef v2f __attribute__((aligned(4))) u2f;
    if constexpr (N >= 4) { const v4f t = {c[0], c[1], c[2], c[3]}; *reinterpret_cast<u4f *>(p) = t; if constexpr (N > 4) stv<N - 4>(p + 4, c + 4); }
    else if constexpr (N == 3) { const v3f t = {c[0], c[1], c[2]}; *reinterpret_cast<u3f *>(p) = t; }
    else if constexpr (N == 2) { const v2f t = {c[0], c[1]}; *reinterpret_cast<u2f *>(p) = t; }
    else if constexpr (N == 1) p[0] = c[0];
}

struct RandSrc {
    const float *in;  // injected row or nullptr
    unsigned k0, k1, e_lo, e_hi, step;
    LG_DEV float draw(int slot) const {
#ifdef LG_DBG_NO_DRAWS
        return 0.5f;
#endif
        if (in) return in[slot];
        U4 c = {e_lo, e_hi, step, (unsigned)(slot >> 2)};
        U4 r = philox4x32_10(c, k0, k1);
        unsigned v = (slot & 3) == 0 ? r.x : ((slot & 3) == 1 ? r.y : ((slot & 3) == 2 ? r.z : r.w));
        return u01(v);
    }
    // three consecutive slots from ONE Philox call (integer multiplies are quarter-rate: a call is ~800 cycles);
    // injected mode still reads the slots one by one
    LG_DEV void draw3(int slot, float &a, float &b, float &c) const {
#ifdef LG_DBG_NO_DRAWS
        a = b = c = 0.5f; return;
#endif
        if (in) { a = in[slot]; b = in[slot + 1]; c = in[slot + 2]; return; }
        U4 ctr = {e_lo, e_hi, step, 0x40000000u + (unsigned)slot};
        U4 r = philox4x32_10(ctr, k0, k1);
        a = u01(r.x); b = u01(r.y); c = u01(r.z);
    }
    LG_DEV void draw4(int slot, float &a, float &b, float &c, float &d) const {   // four-joint legs: the block's fourth output too
#ifdef LG_DBG_NO_DRAWS
        a = b = c = d = 0.5f; return;
#endif
        if (in) { a = in[slot]; b = in[slot + 1]; c = in[slot + 2]; d = in[slot + 3]; return; }
        U4 ctr = {e_lo, e_hi, step, 0x40000000u + (unsigned)slot};
        U4 r = philox4x32_10(ctr, k0, k1);
        a = u01(r.x); b = u01(r.y); c = u01(r.z); d = u01(r.w);
    }
    // a whole Philox block (4 uniforms) from a counter space disjoint from the slot space; used
    // where a lane needs several draws per step (observation noise)
    LG_DEV void block3(int id, float &a, float &b, float &c) const {
        U4 ctr = {e_lo, e_hi, step, 0x80000000u + (unsigned)id};
        U4 r = philox4x32_10(ctr, k0, k1);
        a = u01(r.x); b = u01(r.y); c = u01(r.z);
    }
    LG_DEV void block4(int id, float &a, float &b, float &c, float &d) const {
        U4 ctr = {e_lo, e_hi, step, 0x80000000u + (unsigned)id};
        U4 r = philox4x32_10(ctr, k0, k1);
        a = u01(r.x); b = u01(r.y); c = u01(r.z); d = u01(r.w);
    }
};

// math_utils.py:50-53 as TorchScript executes it: `angles %= 2*pi` lowers to aten::fmod_
// (C remainder, sign of the dividend), so negative angles are NOT wrapped; only m > pi shifts.
LG_DEV float wrap_to_pi(float a) {
    const float two_pi = 6.283185307179586f;
    float m = fmodf(a, two_pi);
    if (m > 3.141592653589793f) m -= two_pi;
    return m;
}

// Hand-off of the physics results from quad_sim_kernel to its MDP tail through LDS: value k of leg-lane l sits at
// sX[k * 16 + l] (per-env values are written once per leg).  Indices:
enum { XA = 0, XLA = 3, XLLA = 6, XQ = 9, XQD = 12, XLQD = 15, XTQ = 18, XFL = 21, XFP = 33, XFV = 36, XLFV = 39, XPOS = 42, XQUAT = 45,
       XVW = 49, XWW = 52, XBLV = 55, XBAV = 58, XPG = 61, XEUL = 64, XFB = 67, XBAD = 70, NX = 71 };   // XBAD: the non-finite guard fired (one value)

// The body of the leg-per-lane step.  FUSED = called from the tail of quad_sim_kernel (lg_quad.h) by the first 16 lanes of
// the wave, one per leg of the wave's envs: the model table, the hot constants and the command ranges are already in
// LDS, `vtid` is the leg-lane index and nothing is staged here.
// FLAT: the launch is known (host-checked, `flat_profile`) to be the plain go2-on-a-plane task -- one unstacked 45-wide
// observation, no privileged output, no gait clock, no terrain, no task extras: those switches become compile-time constants
// and every other task's code drops out of the instantiation (fewer scalar registers, a shorter tail).
// PROF 2 (host-checked, `wtw_profile`): the go2_wtw task on the plane -- gait clock and 5-frame stacks stay, terrain and the other
// tasks' packaging drop out.
// JPL: joints per leg of the serial chains (3: go2, TRON1 point foot; 4: TRON1 sole foot, whose last body is the foot itself).
template <int LEGS, unsigned PH, bool FUSED, int PROF = 0, int JPL = 3, bool REPL = false>
LG_DEV void env_step_body(const KParams &p, uint4 *sMraw, int *sHot, float *sStF, const float *sX, const int vtid, const int vlane) {
    constexpr bool FLAT = PROF == 1, WTWP = PROF == 2, PLANE = FLAT || WTWP, EEP = PROF == 3, PRGP = PROF == 4, ROUGHQ = EEP || PRGP;
    constexpr bool DO_PRE = (PH & LG_PHASE_PRE) != 0, DO_SIM = (PH & LG_PHASE_SIM) != 0;
    constexpr bool DO_POST = (PH & LG_PHASE_POST) != 0, DO_RESET = (PH & LG_PHASE_RESET) != 0;
    constexpr int A = LEGS * JPL;
    static_assert(JPL == 3 || (JPL == 4 && !FUSED && LEGS == 2), "four-joint legs: leg-per-lane biped only");
    // model table -> LDS once per workgroup: per-lane (leg-indexed) reads then cost an LDS access
    // instead of an L2 round trip with a single wave per SIMD to hide it
    unsigned long long _stamp0 = 0; (void)_stamp0;
    STAMP(0);
    // model table: device copy and LDS image are padded to MODEL_STG * BLOCK uint4 so the staging needs no predicate
    uint4 stg0 = {0, 0, 0, 0}, stg1 = stg0, stg2 = stg0, stg3 = stg0;
    if (!FUSED) {   // global -> registers now (same batch as every other start-of-kernel load); registers -> LDS after the barrier
        const uint4 *src = reinterpret_cast<const uint4 *>(p.M) + threadIdx.x;
        stg0 = src[0]; stg1 = src[BLOCK]; stg2 = src[2 * BLOCK]; stg3 = src[3 * BLOCK];
    }
    const LgModelDesc *M = reinterpret_cast<const LgModelDesc *>(sMraw);
    const LgSimOptions *__restrict__ O = p.O;
    const LgTaskCfg *__restrict__ T = p.T;
    // hot constants: lane i of hv[k] holds dword 64*k + i of the LgHot block (three coalesced loads, one wait);
    // a field is then one v_readlane with a constant lane -- no scalar-cache round trips, no SGPR pressure
    int hv0 = 0, hv1 = 0, hv2 = 0, hv3 = 0;
    if (!FUSED) {
        const int *hp = reinterpret_cast<const int *>(p.H) + vlane;
        hv0 = hp[0]; hv1 = hp[64]; hv2 = hp[128]; hv3 = hp[192];
    }
    const LgBuffers &B = p.B;

    const int tid = vtid;
    const int leg = tid % LEGS;
    int e = tid / LEGS;
    // FUSED: the wave runs four replicas of its 16 leg-lanes (lg_quad.h); replica `sub` > 0 computes the same values and stores only
    // its share of the observation outputs
    constexpr bool RP = FUSED || REPL;   // REPL: the same replication in a launch of its own (MDP phases after component-layout physics)
    const int sub = RP ? ((int)threadIdx.x >> 4) : 0;
    const int glane = RP ? ((int)threadIdx.x & 15) : vlane;   // lane inside the replica
    // replicated launches share Philox blocks between the replicas of a leg-lane: `from(w, r)` = word `w` (selected the same way in
    // every lane) as replica r's lane of this leg holds it
    auto from = [&](unsigned w, int r) { return (unsigned)__builtin_amdgcn_ds_bpermute(((r << 4) | glane) << 2, (int)w); };
    auto word = [](const U4 &m, int k) { return k == 0 ? m.x : (k == 1 ? m.y : (k == 2 ? m.z : m.w)); };
    (void)from; (void)word;
    const bool live_all = e < B.n_envs;
    const bool live = live_all && sub == 0;
    if (!live_all) e = B.n_envs - 1;  // dead lanes shadow the last env (keeps DPP quads uniform), never store
    const bool lead = live && leg == 0;
    const int L = p.k.m_n_links, F = LEGS;
    const int b0 = 1 + JPL * leg;          // first body of this lane's chain
    const int d0 = JPL * leg;              // first dof
    const int foot_link = leg == 0 ? p.k.m_foot_link[0] : (leg == 1 ? p.k.m_foot_link[1] : (leg == 2 ? p.k.m_foot_link[2] : p.k.m_foot_link[3]));
    int foot_slot = 0;                     // rank of this foot among feet in link order (feet_indices)
#pragma unroll
    for (int k = 0; k < LEGS; k++) foot_slot += (p.k.m_foot_link[k] < foot_link) ? 1 : 0;

    // ---------------- PRE: clip + action history (legged_robot.py:230-239) -----------------
    float act[JPL], last_act[JPL], llast_act[JPL];
    if (DO_PRE) {
        const float ca = p.k.clip_actions;
#pragma unroll
        for (int j = 0; j < JPL; j++) {
            float prev = B.actions[e * A + d0 + j], prev2 = B.last_actions[e * A + d0 + j];
            act[j] = clampf(p.actions[e * A + d0 + j], -ca, ca);
            last_act[j] = prev;
            llast_act[j] = prev2;
        }   // the three history stores are issued after every start-of-kernel load (see "prologue stores" below)
    } else if (DO_SIM && !DO_POST) {  // Simulator.step(actions): actions come pre-clipped from the env
#pragma unroll
        for (int j = 0; j < JPL; j++) { act[j] = p.actions[e * A + d0 + j]; last_act[j] = llast_act[j] = 0.f; }
    } else if (FUSED) {   // handed over by the physics phase of the same launch through LDS
#pragma unroll
        for (int j = 0; j < JPL; j++) { act[j] = sX[(XA + j) * 16 + vlane]; last_act[j] = sX[(XLA + j) * 16 + vlane]; llast_act[j] = sX[(XLLA + j) * 16 + vlane]; }
    } else {
#pragma unroll
        for (int j = 0; j < JPL; j++) {
            act[j] = B.actions[e * A + d0 + j];
            last_act[j] = B.last_actions[e * A + d0 + j];
            llast_act[j] = B.llast_actions[e * A + d0 + j];
        }
    }

    // ---------------- state ----------------------------------------------------------------
    V3 pos, vw, ww;
    float qx, qy, qz, qw, q[JPL], qd[JPL];
    if (FUSED) {
        auto X3 = [&](int k) { return v3(sX[k * 16 + vlane], sX[(k + 1) * 16 + vlane], sX[(k + 2) * 16 + vlane]); };
        pos = X3(XPOS); vw = X3(XVW); ww = X3(XWW);
        qx = sX[XQUAT * 16 + vlane]; qy = sX[(XQUAT + 1) * 16 + vlane]; qz = sX[(XQUAT + 2) * 16 + vlane]; qw = sX[(XQUAT + 3) * 16 + vlane];
#pragma unroll
        for (int j = 0; j < JPL; j++) { q[j] = sX[(XQ + j) * 16 + vlane]; qd[j] = sX[(XQD + j) * 16 + vlane]; }
    } else {
        pos = ld3(B.base_pos + 3 * e); vw = ld3(B.base_lin_vel_w + 3 * e); ww = ld3(B.base_ang_vel_w + 3 * e);
        qx = B.base_quat[4 * e]; qy = B.base_quat[4 * e + 1]; qz = B.base_quat[4 * e + 2]; qw = B.base_quat[4 * e + 3];
#pragma unroll
        for (int j = 0; j < JPL; j++) { q[j] = B.dof_pos[e * A + d0 + j]; qd[j] = B.dof_vel[e * A + d0 + j]; }
    }

    // ---- MDP working set, fetched NOW so that the round trips overlap the physics below instead of being
    //      exposed one by one behind it (a lone wave per SIMD has nothing else to switch to) -----------------
    constexpr bool DO_MDP = DO_POST || DO_RESET;
    const unsigned rmask0 = (unsigned)p.k.reward_mask, rmask = rmask0;   // scalar for the whole kernel
    const int N = B.n_envs;
    float q0l[JPL], soft_lo[JPL] = {0.f, 0.f, 0.f}, soft_hi[JPL] = {0.f, 0.f, 0.f}, rdof_lo[JPL] = {0.f, 0.f, 0.f}, rdof_span[JPL] = {0.f, 0.f, 0.f};
    float nv_q[JPL] = {0.f, 0.f, 0.f}, nv_qd[JPL] = {0.f, 0.f, 0.f}, nv_act[JPL] = {0.f, 0.f, 0.f}, nv_clk[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < JPL; j++) q0l[j] = O->default_dof_pos[d0 + j];
    float cmd0 = 0.f, cmd1 = 0.f, cmd2 = 0.f, cmd3 = 0.f, air = 0.f;
    int ep_len = 0, last_contact = 0, dirty_prev = 0;
    long long fail_buf = 0;
    bool guard_bad = false;   // the non-finite guard of the physics phase fired for this env in this step
    float es[LG_R_COUNT];
#pragma unroll
    for (int k = 0; k < LG_R_COUNT; k++) es[k] = 0.f;
    int crv = 0;   // command / behaviour ranges (runtime buffer): lane k holds entry k, read with v_readlane
    int prw = 0;   // observation programs, one dword per lane (lanes 0-25 priv_prog, 26-51 labels_prog)
    V3 origin_pre = v3(0, 0, 0);
    // physics-side start-of-kernel loads: snapshot sources and per-env dynamics parameters
    V3 snap_fv = v3(0, 0, 0), snap_blv = v3(0, 0, 0), snap_bav = v3(0, 0, 0), dr_com = v3(0, 0, 0), dr_joint = v3(0, 0, 0);
    float dr_mass = 0.f, dr_fric = 1.f, dr_kp[JPL], dr_kd[JPL], gain_p[JPL] = {0.f, 0.f, 0.f}, gain_d[JPL] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < JPL; j++) dr_kp[j] = dr_kd[j] = 1.f;
    if (DO_SIM) {
        snap_fv = ld3(B.feet_vel + (e * F + foot_slot) * 3);
        snap_blv = ld3(B.base_lin_vel + 3 * e); snap_bav = ld3(B.base_ang_vel + 3 * e);
        if (B.added_base_mass) dr_mass = B.added_base_mass[e];
        if (B.base_com_bias) dr_com = ld3(B.base_com_bias + 3 * e);
        if (B.friction_values) dr_fric = B.friction_values[e];
#pragma unroll
        for (int j = 0; j < JPL; j++) {
            if (B.kp_scale) dr_kp[j] = B.kp_scale[e * A + d0 + j];
            if (B.kd_scale) dr_kd[j] = B.kd_scale[e * A + d0 + j];
            gain_p[j] = O->kp[d0 + j]; gain_d[j] = O->kd[d0 + j];
        }
        if (B.joint_armature) dr_joint = v3(B.joint_armature[e], B.joint_friction[e], B.joint_damping[e]);
    }
    if (DO_MDP && !FUSED) {
#pragma unroll
        for (int j = 0; j < JPL; j++) {
            soft_lo[j] = T->soft_dof_lo[d0 + j]; soft_hi[j] = T->soft_dof_hi[d0 + j];
            rdof_lo[j] = T->reset_dof_lo[d0 + j]; rdof_span[j] = T->reset_dof_span[d0 + j];
            nv_q[j] = T->noise_vec[9 + d0 + j]; nv_qd[j] = T->noise_vec[9 + A + d0 + j]; nv_act[j] = T->noise_vec[9 + 2 * A + d0 + j];
        }
        if (!PLANE && !ROUGHQ && p.k.obs_layout == LG_OBS_TRON1_EE) { nv_clk[0] = T->noise_vec[9 + 3 * A + foot_slot]; nv_clk[1] = T->noise_vec[9 + 3 * A + LEGS + foot_slot]; }
        cmd0 = B.commands[4 * e]; cmd1 = B.commands[4 * e + 1]; cmd2 = B.commands[4 * e + 2]; cmd3 = B.commands[4 * e + 3];
        ep_len = B.episode_length_buf[e];
        fail_buf = B.fail_buf[e];
        air = B.feet_air_time[e * F + foot_slot];
        last_contact = B.last_contacts[e * F + foot_slot];
        if (DO_RESET && B.obs_dirty) dirty_prev = B.obs_dirty[e];
        origin_pre = ld3(B.env_origins + 3 * e);
        if (!FUSED) crv = reinterpret_cast<const int *>(B.command_ranges)[min(vlane, LG_CMD_RANGE_FLOATS - 1)];
        if (!FUSED && DO_RESET && !PLANE && !EEP && p.k.obs_layout == LG_OBS_PROGRAM)
            prw = reinterpret_cast<const int *>(vlane < 26 ? &T->priv_prog : &T->labels_prog)[vlane < 26 ? vlane : min(vlane - 26, 25)];
        if (lead) {
#pragma unroll
            for (int k = 0; k < LG_R_COUNT; k++) es[k] = ((rmask0 >> k) & 1u) ? B.episode_sums[(size_t)k * N + e] : 0.f;
        }
    }

    // read-back quantities handed from SIM to POST/RESET (registers when fused, HBM otherwise)
    V3 blv, bav, pg, eul;            // body-frame lin/ang vel, projected gravity, euler
    float last_qd[JPL], torque[JPL];
    V3 f_link[4];                    // net contact force on this leg's hip, thigh, calf, foot (world)
    V3 f_base = v3(0, 0, 0);         // net contact force on the base (whole env)
    V3 foot_p, foot_v, last_foot_v = v3(0, 0, 0);
    float foot_zx = 0.f, foot_zy = 0.f;   // x, y of the world z axis seen from the foot body (four-joint legs: _reward_foot_flat)
    // this lane's share of the sampled terrain heights (k = leg + i * LEGS), kept in registers from the moment they are
    // sampled / loaded until the observation is written: one batch of independent loads instead of a dependent
    // load -> store chain per entry in each consumer
    constexpr int HMAX = 26;
    float hts[HMAX];
#pragma unroll
    for (int i = 0; i < HMAX; i++) hts[i] = 0.f;
    const int P = PLANE ? 0 : p.k.o_n_height_points;
    const bool hreg = P <= HMAX * LEGS;
    float foot_hmean = 0.f, foot_hmax = 0.f;  // mean / max of the 9 terrain heights around this lane's foot (a8)
    float mean_height = 0.f;         // mean over the height-sample grid of (base_z - h) is formed from this (a7)

    // compiler-level memory barrier: every load above is issued before anything below (LLVM otherwise sinks each one
    // next to its consumer, where its full round trip is exposed); no hardware wait is emitted here
    asm volatile("" ::: "memory");
    if (!FUSED) {
        sHot[threadIdx.x] = hv0; sHot[threadIdx.x + 64] = hv1; sHot[threadIdx.x + 128] = hv2; sHot[threadIdx.x + 192] = hv3; sHot[threadIdx.x + 256] = crv; sHot[threadIdx.x + 256 + BLOCK] = prw;
        sMraw[threadIdx.x] = stg0; sMraw[threadIdx.x + BLOCK] = stg1; sMraw[threadIdx.x + 2 * BLOCK] = stg2; sMraw[threadIdx.x + 3 * BLOCK] = stg3;
    }
    // MDP working set -> LDS for the duration of the physics (it arrived in the same load burst; parking it here keeps
    // ~70 registers per lane free in the sub-step loop).  Layout [value][lane]: conflict-free.
    constexpr bool STASH = DO_MDP && DO_SIM;
    constexpr int NST = LG_R_COUNT + 7 * JPL + 15;
    __shared__ float sSt[STASH ? NST : 1][BLOCK];
    __shared__ float sW[DO_SIM ? 9 * BLOCK : 1];   // dt * W of each lane's foot between refreshes (LgSimOptions.contact_w_every > 1)
    if (STASH) {
        const int t = threadIdx.x;
#pragma unroll
        for (int k = 0; k < LG_R_COUNT; k++) sSt[k][t] = es[k];
        int c = LG_R_COUNT;
#pragma unroll
        for (int j = 0; j < JPL; j++) {
            sSt[c++][t] = soft_lo[j]; sSt[c++][t] = soft_hi[j]; sSt[c++][t] = rdof_lo[j]; sSt[c++][t] = rdof_span[j];
            sSt[c++][t] = nv_q[j]; sSt[c++][t] = nv_qd[j]; sSt[c++][t] = nv_act[j];
        }
        sSt[c++][t] = nv_clk[0]; sSt[c++][t] = nv_clk[1];
        sSt[c++][t] = cmd0; sSt[c++][t] = cmd1; sSt[c++][t] = cmd2; sSt[c++][t] = cmd3;
        sSt[c++][t] = __int_as_float(ep_len); sSt[c++][t] = __int_as_float((int)fail_buf); sSt[c++][t] = air;
        sSt[c++][t] = __int_as_float(last_contact);
        sSt[c++][t] = origin_pre.x; sSt[c++][t] = origin_pre.y; sSt[c++][t] = origin_pre.z;
        sSt[c++][t] = __int_as_float(dirty_prev);
    }
    __syncthreads();
    STAMP(2);
    // ---- prologue stores: only now, after every start-of-kernel load has been issued and awaited.  vmcnt counts
    //      loads and stores together in issue order, so a load queued behind stores waits for the full HBM write
    //      round trip of each; issued here the stores drain under the physics instead.
    if (DO_PRE && live) {
#pragma unroll
        for (int j = 0; j < JPL; j++) {
            B.llast_actions[e * A + d0 + j] = llast_act[j];
            B.last_actions[e * A + d0 + j] = last_act[j];
            B.actions[e * A + d0 + j] = act[j];
        }
    }
    if (DO_SIM) {
        // "last" snapshots (genesis_simulator.py:21-24)
#pragma unroll
        for (int j = 0; j < JPL; j++) last_qd[j] = qd[j];
        last_foot_v = snap_fv;
        if (live) {
#pragma unroll
            for (int j = 0; j < JPL; j++) B.last_dof_vel[e * A + d0 + j] = qd[j];
            st3(B.last_feet_vel + (e * F + foot_slot) * 3, snap_fv);
            if (lead) {
                st3(B.last_base_lin_vel + 3 * e, snap_blv);
                st3(B.last_base_ang_vel + 3 * e, snap_bav);
            }
        }
        // per-env dynamics parameters (genesis_simulator.py:665-739), fetched in the prologue batch
        const float mass0 = M->mass[0] + dr_mass;
        V3 com0 = ld3(M->com[0]) + dr_com;
        const float mu = HOT(o_terrain_friction) * dr_fric;
        float kps[JPL], kds[JPL], arm[JPL], jdamp[JPL], jfric[JPL];
#pragma unroll
        for (int j = 0; j < JPL; j++) {
            kps[j] = dr_kp[j] * gain_p[j];
            kds[j] = dr_kd[j] * gain_d[j];
            arm[j] = B.joint_armature ? dr_joint.x : M->armature[d0 + j];
            jfric[j] = B.joint_friction ? dr_joint.y : M->frictionloss[d0 + j];
            jdamp[j] = B.joint_damping ? dr_joint.z : M->damping[d0 + j];
        }
        const float dt = HOT(o_dt), kc = HOT(o_contact_k), kappa = kc * dt + HOT(o_contact_b), margin = HOT(o_contact_margin);
        const float kl = HOT(o_limit_k), kapl = kl * dt + HOT(o_limit_b);
        const V3 grav = v3(0.f, 0.f, HOT(o_gravity_z));
        const S3 I0 = {M->inertia[0][0], M->inertia[0][1], M->inertia[0][2], M->inertia[0][3], M->inertia[0][4], M->inertia[0][5]};
        const int fs = leg == 0 ? HOT(m_foot_sphere[0]) : (leg == 1 ? HOT(m_foot_sphere[1]) : (leg == 2 ? HOT(m_foot_sphere[2]) : HOT(m_foot_sphere[3])));
        const V3 foot_c_loc = ld3(M->sph_pos[fs]);
        const float foot_r = M->sph_r[fs];

        // ---- this lane's model constants and collision spheres, read from the LDS table ONCE per launch and kept
        //      in registers across the four sub-steps (a single wave per SIMD cannot hide ~100-cycle LDS round
        //      trips issued from inside the dependency chain: they were 40 % of the wave's cycles)
        float Lm[JPL], Lqlo[JPL], Lqhi[JPL], Leff[JPL], Lvlim[JPL];
        V3 Lcom[JPL], Ljpos[JPL], Lax[JPL];
        S3 LIc[JPL];
#pragma unroll
        for (int j = 0; j < JPL; j++) {
            const int b = b0 + j;
            Lm[j] = M->mass[b]; Lcom[j] = ld3(M->com[b]); Ljpos[j] = ld3(M->jpos[b]); Lax[j] = ld3(M->axis[b]);
            const S3 t = {M->inertia[b][0], M->inertia[b][1], M->inertia[b][2], M->inertia[b][3], M->inertia[b][4], M->inertia[b][5]};
            LIc[j] = t;
            Lqlo[j] = M->q_lo[d0 + j]; Lqhi[j] = M->q_hi[d0 + j]; Leff[j] = M->effort[d0 + j];
            Lvlim[j] = HOT(o_joint_vel_clamp) * M->vel_limit[d0 + j];
        }
        // spheres: up to SPH0/SPH1/SPH2 on the first, second and LAST chain body (foot sphere excluded), SPHM on the third
        // of a four-joint chain, and SPHB of the base spheres dealt round-robin to the lanes of the env; counts are
        // validated by lg_create
        constexpr int SPH0 = 2, SPH1 = 2, SPH2 = 5, SPHM = JPL == 4 ? 3 : 1, SPHB = 4;
        struct Sph { V3 p; float r, w; };
        Sph S0[SPH0], S1[SPH1], S2[SPH2], SM[SPHM], SB[SPHB];
        int n0 = 0, n1 = 0, n2 = 0, nM = 0, nB = 0;
        {
            auto ld = [&](int s) { Sph t = {ld3(M->sph_pos[s]), M->sph_r[s], M->sph_w[s]}; return t; };
            const int a0 = M->body_sph_start[b0], a1 = M->body_sph_start[b0 + 1];
            const int a2 = M->body_sph_start[b0 + JPL - 1], a3 = M->body_sph_start[b0 + JPL];   // last body of the chain
            const int a1e = M->body_sph_start[b0 + 2];                                           // end of the second body's spheres
            n0 = a1 - a0; n1 = a1e - a1;
#pragma unroll
            for (int k = 0; k < SPH0; k++) S0[k] = ld(min(a0 + k, a1 - 1 >= a0 ? a1 - 1 : a0));
#pragma unroll
            for (int k = 0; k < SPH1; k++) S1[k] = ld(min(a1 + k, a1e - 1 >= a1 ? a1e - 1 : a1));
            if (JPL == 4) {
                nM = a2 - a1e;
#pragma unroll
                for (int k = 0; k < SPHM; k++) SM[k] = ld(min(a1e + k, a2 - 1 >= a1e ? a2 - 1 : a1e));
            }
            int cnt = 0;
#pragma unroll
            for (int k = 0; k < SPH2 + 1; k++) {       // skip the foot sphere (solved implicitly in stage 2)
                const int s = a2 + k;
                if (s < a3 && s != fs && cnt < SPH2) {
                    const Sph t = ld(s);
#pragma unroll
                    for (int q2 = 0; q2 < SPH2; q2++) if (q2 == cnt) S2[q2] = t;
                    cnt++;
                }
            }
            n2 = cnt;
            const int e0 = M->body_sph_start[0], e1 = M->body_sph_start[1];
#pragma unroll
            for (int k = 0; k < SPHB; k++) {
                const int s = e0 + leg + LEGS * k;
                SB[k] = ld(min(s, e1 - 1));
                if (s < e1) nB = k + 1;
            }
        }

        STAMP(3);
        const bool jrot_identity = p.jrot_identity != 0;

        float nf_poison = 0.f;   // becomes NaN when an acceleration or a joint rate of any sub-step was not finite (read-back guard)
        for (int sub = 0; sub < HOT(o_decimation); sub++) {
            const M3 Rb = quat_to_mat(qx, qy, qz, qw);
            // ---- forward kinematics + velocities of the chain (root -> leaf) -----------------
            BodyKin K[JPL];
            Joint J[JPL];
#pragma unroll
            for (int j = 0; j < JPL; j++) f_link[j] = v3(0, 0, 0);
            f_link[3] = v3(0, 0, 0);
            V3 fb_acc = v3(0, 0, 0);       // this lane's share of base-sphere forces
            V6 pbase_ext = {v3(0, 0, 0), v3(0, 0, 0)};  // spatial force of that share about O
            {
                M3 Rp = Rb;
                V3 Pp = v3(0, 0, 0);
                V6 Vp = {ww, vw};
#pragma unroll
                for (int j = 0; j < JPL; j++) {
                    const int b = b0 + j;
                    const V3 jp = Ljpos[j], ax = Lax[j];
                    K[j].P = Pp + mul(Rp, jp);
                    M3 Rfix = Rp;
                    if (!jrot_identity) {
                        const M3 jr = {M->jrot[b][0], M->jrot[b][1], M->jrot[b][2], M->jrot[b][3], M->jrot[b][4],
                                       M->jrot[b][5], M->jrot[b][6], M->jrot[b][7], M->jrot[b][8]};
                        Rfix = mul(Rp, jr);
                    }
                    const V3 s = mul(Rfix, ax);
                    K[j].R = mul(Rfix, axis_angle(ax, q[j]));
                    J[j].S.a = s;
                    J[j].S.l = cross(K[j].P, s);
                    K[j].V.a = Vp.a + s * qd[j];
                    K[j].V.l = Vp.l + J[j].S.l * qd[j];
                    // c = V x (S qd)
                    J[j].c.a = cross(K[j].V.a, s) * qd[j];
                    J[j].c.l = (cross(K[j].V.a, J[j].S.l) + cross(K[j].V.l, s)) * qd[j];
                    Rp = K[j].R; Pp = K[j].P; Vp = K[j].V;
                }
            }
            // ---- body (non-foot) collision spheres: penalty made implicit with the conservative
            //      point inverse mass sph_w; lane handles its chain + every LEGS-th base sphere
            // Sole corners (four-joint legs, the spheres of the foot body besides the foot sphere): they hold the foot flat, the load
            // itself is carried by the foot sphere at the sole centre (stage 2, exact).  `ank` = the chain's last joint with
            // dinv = 1 / (S^T I_foot S + armature); a corner's inverse mass is twice the ankle's own compliance at the contact point,
            // 2 (n . (s x (c - P)))^2 dinv (the two corners of an edge share that rotation), plus the model's sph_w; its penetration and
            // approach velocity are measured relative to the sole centre's (dref, vref) while that one is in the ground
            auto sphere_contact = [&](const Sph &sp, const M3 &R, V3 P, const V6 &V, V3 &fsum, V6 &pacc, const Joint *ank = nullptr,
                                      const float dref = 0.f, const float vref = 0.f) {
                const V3 r = P + mul(R, sp.p);
                const float rad = sp.r;
                float h; V3 n;
                if (PLANE) { h = 0.f; n = v3(0.f, 0.f, 1.f); } else terrain_at(O, p.hf, pos.x + r.x, pos.y + r.y, h, n);
                const float depth = rad - (pos.z + r.z - h) * n.z - dref;
                if (depth > -margin) {
                    const V3 v = V.l + cross(V.a, r);
                    float wi = sp.w;
                    if (JPL == 4 && ank) { const float g = dot(n, ank->S.l + cross(ank->S.a, r - n * rad)); wi += 2.f * g * g * ank->dinv; }
                    const float vnf = dot(v, n), vn = vnf - vref;
                    const float fn = (kc * depth - kappa * vn) * rcp(1.f + kappa * dt * wi);
                    if (fn > 0.f) {
                        const V3 vt = v - n * vnf;
                        const float vtn = norm(vt);
                        const float ft = fminf(vtn * rcp(dt * wi), mu * fn);
                        V3 f = n * fn;
                        if (vtn > 1e-9f) f -= vt * (ft * rcp(vtn));
                        const V3 cp = r - n * rad;
                        pacc.a += cross(cp, f);
                        pacc.l += f;
                        fsum += f;
                    }
                }
            };
            const V6 V0 = {ww, vw};
#pragma unroll
            for (int k = 0; k < SPHB; k++)
                if (k < nB) sphere_contact(SB[k], Rb, v3(0, 0, 0), V0, fb_acc, pbase_ext);

            // ---- actuation (genesis_simulator.py:630-642): PD, torque reported unclipped ----
            float tau[JPL];
#pragma unroll
            for (int j = 0; j < JPL; j++) {
                const float t = kps[j] * (act[j] * HOT(o_action_scale) + q0l[j] - q[j]) - kds[j] * qd[j];
                torque[j] = t;
                const float lim = Leff[j];
                tau[j] = clampf(t, -lim, lim) - jdamp[j] * qd[j] - jfric[j] * clampf(qd[j] * 20.f, -1.f, 1.f);
            }
            // ---- ABA pass 2 up the chain (leaf -> root): rigid inertia + bias force of each body
            //      are formed right where they are consumed ------------------------------------
            I6 IAacc;
            V6 pacc;
#pragma unroll
            for (int j = JPL - 1; j >= 0; j--) {
                {
                    const float m = Lm[j];
                    const V3 cw = K[j].P + mul(K[j].R, Lcom[j]);
                    const S3 Icw = rot_sym(K[j].R, LIc[j]);
                    // bias force: V x* (I V) - gravity - contacts
                    const V3 vc = K[j].V.l + cross(K[j].V.a, cw);
                    const V3 Pm = vc * m;
                    const V3 Lm = mul(Icw, K[j].V.a) + cross(cw, Pm);
                    const V3 fg = grav * m;
                    V6 pb;
                    pb.a = cross(K[j].V.a, Lm) + cross(K[j].V.l, Pm) - cross(cw, fg);
                    pb.l = cross(K[j].V.a, Pm) - fg;
                    V6 ext = {v3(0, 0, 0), v3(0, 0, 0)};
                    if (j == 0) {
#pragma unroll
                        for (int k = 0; k < SPH0; k++) if (k < n0) sphere_contact(S0[k], K[j].R, K[j].P, K[j].V, f_link[j], ext);
                    } else if (j == 1) {
#pragma unroll
                        for (int k = 0; k < SPH1; k++) if (k < n1) sphere_contact(S1[k], K[j].R, K[j].P, K[j].V, f_link[j], ext);
                    } else if (j < JPL - 1) {
#pragma unroll
                        for (int k = 0; k < SPHM; k++) if (k < nM) sphere_contact(SM[k], K[j].R, K[j].P, K[j].V, f_link[j], ext);
                    } else if (JPL == 4) {   // the foot body: spheres see the ankle's compliance (see sphere_contact)
                        const I6 Ifoot = {Icw + parallel_axis(m, cw), skew(cw * m), S3{m, m, m, 0.f, 0.f, 0.f}};
                        Joint ank = J[j];
                        ank.dinv = rcp(dot(J[j].S, mul(Ifoot, J[j].S)) + arm[j]);
                        float dref = 0.f, vref = 0.f;
                        {
                            const V3 rc = K[j].P + mul(K[j].R, foot_c_loc);
                            float hc; V3 nc;
                            if (PLANE) { hc = 0.f; nc = v3(0.f, 0.f, 1.f); } else terrain_at(O, p.hf, pos.x + rc.x, pos.y + rc.y, hc, nc);
                            const float dc = foot_r - (pos.z + rc.z - hc) * nc.z;
                            if (dc > 0.f) { dref = dc; vref = dot(K[j].V.l + cross(K[j].V.a, rc), nc); }
                        }
#pragma unroll
                        for (int k = 0; k < SPH2; k++) if (k < n2) sphere_contact(S2[k], K[j].R, K[j].P, K[j].V, f_link[j], ext, &ank, dref, vref);
                    } else {
#pragma unroll
                        for (int k = 0; k < SPH2; k++) if (k < n2) sphere_contact(S2[k], K[j].R, K[j].P, K[j].V, f_link[j], ext);
                    }
                    pb = pb - ext;
                    const S3 Ab = Icw + parallel_axis(m, cw);
                    const M3 Bb = skew(cw * m);
                    if (j == JPL - 1) {
                        IAacc.A = Ab; IAacc.B = Bb;
                        const S3 Cm = {m, m, m, 0.f, 0.f, 0.f};
                        IAacc.C = Cm;
                        pacc = pb;
                    } else {
                        IAacc.A = IAacc.A + Ab; IAacc.B = IAacc.B + Bb;
                        IAacc.C.xx += m; IAacc.C.yy += m; IAacc.C.zz += m;
                        pacc = pacc + pb;
                    }
                }
                J[j].U = mul(IAacc, J[j].S);
                J[j].dinv = rcp(dot(J[j].S, J[j].U) + arm[j]);
                J[j].u = tau[j] - dot(J[j].S, pacc);
                const V6 Ic6 = mul(IAacc, J[j].c);
                const float k = (J[j].u - dot(J[j].U, J[j].c)) * J[j].dinv;
                pacc = pacc + Ic6 + J[j].U * k;
                IAacc = rank1_down(IAacc, J[j].U, J[j].dinv);
            }
            // ---- base: own inertia + bias, plus the quad-sum of the legs' contributions ----
            I6 IA0 = quad_sum<LEGS>(IAacc);
            V6 p0 = quad_sum<LEGS>(pacc - pbase_ext);
            {
                const V3 cw = mul(Rb, com0);
                const S3 Icw = rot_sym(Rb, I0);
                IA0.A = IA0.A + Icw + parallel_axis(mass0, cw);
                IA0.B = IA0.B + skew(cw * mass0);
                IA0.C.xx += mass0; IA0.C.yy += mass0; IA0.C.zz += mass0;
                const V3 vc = vw + cross(ww, cw);
                const V3 Pm = vc * mass0;
                const V3 Lm = mul(Icw, ww) + cross(cw, Pm);
                const V3 fg = grav * mass0;
                p0.a += cross(ww, Lm) + cross(vw, Pm) - cross(cw, fg);
                p0.l += cross(ww, Pm) - fg;
            }
            const Chol6 ch = chol6(IA0);
            V6 a0 = chol6_solve(ch, V6{-p0.a, -p0.l});
            // ---- pass 3 down the chain ------------------------------------------------------
            float qdd[JPL];
            V6 a_calf;
            {
                V6 a = a0;
#pragma unroll
                for (int j = 0; j < JPL; j++) {
                    a = a + J[j].c;
                    qdd[j] = (J[j].u - dot(J[j].U, a)) * J[j].dinv;
                    a = a + J[j].S * qdd[j];
                }
                a_calf = a;
            }
            // ---- stage 2: foot contact (exact 3x3 W) + joint-limit stops, block-Jacobi --------
            V3 fc = v3(0, 0, 0);          // foot force in the contact frame (n, t1, t2)
            V3 cn = v3(0, 0, 1), ct1 = v3(1, 0, 0), ct2 = v3(0, 1, 0), cp = v3(0, 0, 0), vfree = v3(0, 0, 0);
            M3 Ac = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // dt * W in the contact frame
            const int w_every = HOT(o_contact_w_every);
            const bool w_refresh = w_every <= 1 || (sub % w_every) == 0;   // uniform: W of every foot is recomputed on these sub-steps
            float depth = -1.f;
            bool fact = false;
            {
                const V3 r = K[JPL - 1].P + mul(K[JPL - 1].R, foot_c_loc);
                float h;
                if (PLANE) { h = 0.f; cn = v3(0.f, 0.f, 1.f); } else terrain_at(O, p.hf, pos.x + r.x, pos.y + r.y, h, cn);
                depth = foot_r - (pos.z + r.z - h) * cn.z;
                fact = depth > -margin;
                cp = r - cn * foot_r;
            }
            float lim_e[JPL], lim_s[JPL], lim_T[JPL] = {0.f, 0.f, 0.f};
            bool lact = false;
#pragma unroll
            for (int j = 0; j < JPL; j++) {
                const float lo = Lqlo[j], hi = Lqhi[j];
                lim_s[j] = 0.f; lim_e[j] = 0.f;
                if (q[j] < lo + HOT(o_limit_margin)) { lim_s[j] = 1.f; lim_e[j] = lo - q[j]; lact = true; }
                else if (q[j] > hi - HOT(o_limit_margin)) { lim_s[j] = -1.f; lim_e[j] = q[j] - hi; lact = true; }
            }
            const int any = quad_or<LEGS>((fact || lact) ? 1 : 0) | (w_refresh && w_every > 1 ? 1 : 0);
            float dqdd[JPL] = {0.f, 0.f, 0.f};
            V6 da0 = {v3(0, 0, 0), v3(0, 0, 0)};
            if (any) {  // uniform across the quad: DPP inside is safe
                // W columns: response of the contact point to unit forces (this lane's chain + base)
                {
                    if (cn.z < 0.999999f) {
                        const float d = cn.x;
                        ct1 = v3(1.f - d * cn.x, -d * cn.y, -d * cn.z);
                        ct1 = ct1 * rsqrtf(dot(ct1, ct1));
                        ct2 = cross(cn, ct1);
                    }
                    const V3 axs[3] = {cn, ct1, ct2};
                    V3 col[3];
                    const float zero3[JPL] = {0.f, 0.f, 0.f};
                    if (w_refresh) {
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        float du[JPL], dq[JPL];
                        const V6 fsp = {cross(cp, axs[k]), axs[k]};
                        const V6 dp = resp_up(J, fsp, zero3, du);
                        const V6 ab = chol6_solve(ch, V6{-dp.a, -dp.l});
                        const V6 ac = resp_down(J, ab, du, dq);
                        const V3 ra = ac.l + cross(ac.a, cp);
                        col[k] = v3(dot(cn, ra), dot(ct1, ra), dot(ct2, ra));
                    }
                    Ac.xx = dt * col[0].x; Ac.xy = dt * col[1].x; Ac.xz = dt * col[2].x;
                    Ac.yx = dt * col[0].y; Ac.yy = dt * col[1].y; Ac.yz = dt * col[2].y;
                    Ac.zx = dt * col[0].z; Ac.zy = dt * col[1].z; Ac.zz = dt * col[2].z;
                    if (w_every > 1) {   // kept for the sub-steps in between (LDS: this kernel has no registers to spare)
                        float *w = sW + threadIdx.x;
                        w[0] = Ac.xx; w[BLOCK] = Ac.xy; w[2 * BLOCK] = Ac.xz; w[3 * BLOCK] = Ac.yx; w[4 * BLOCK] = Ac.yy; w[5 * BLOCK] = Ac.yz;
                        w[6 * BLOCK] = Ac.zx; w[7 * BLOCK] = Ac.zy; w[8 * BLOCK] = Ac.zz;
                    }
                    } else {
                        const float *w = sW + threadIdx.x;
                        Ac.xx = w[0]; Ac.xy = w[BLOCK]; Ac.xz = w[2 * BLOCK]; Ac.yx = w[3 * BLOCK]; Ac.yy = w[4 * BLOCK]; Ac.yz = w[5 * BLOCK];
                        Ac.zx = w[6 * BLOCK]; Ac.zy = w[7 * BLOCK]; Ac.zz = w[8 * BLOCK];
                    }
                    const V3 vpt = K[JPL - 1].V.l + cross(K[JPL - 1].V.a, cp);
                    const V3 apt = a_calf.l + cross(a_calf.a, cp) + cross(K[JPL - 1].V.a, vpt);
                    const V3 vf = vpt + apt * dt;
                    vfree = v3(dot(cn, vf), dot(ct1, vf), dot(ct2, vf));
                }
                V3 resp_c = v3(0, 0, 0);  // contact-frame acceleration response to the current force set
                for (int it = 0; it < HOT(o_contact_iters); it++) {
                    // foot: velocity it would have without its own force, then the local law
                    V3 fnew = v3(0, 0, 0);
                    if (fact) {
                        const V3 vo = vfree + resp_c * dt - mul(Ac, fc);
                        const M3 M3x = {1.f + kappa * Ac.xx, kappa * Ac.xy, kappa * Ac.xz, Ac.yx, Ac.yy, Ac.yz, Ac.zx, Ac.zy, Ac.zz};
                        const float rn = kc * depth - kappa * vo.x;
                        V3 fsol;
                        if (solve3(M3x, v3(rn, -vo.y, -vo.z), fsol) && fsol.x > 0.f) {
                            const float ftn = sqrtf(fsol.y * fsol.y + fsol.z * fsol.z);
                            if (ftn > mu * fsol.x) {
                                const float iftn = rcp(ftn), e1 = fsol.y * iftn, e2 = fsol.z * iftn;
                                // friction coupling limited to 3/4 of A_nn: frictional jamming otherwise lets f_n = rn / (1 + kappa a_eff) grow
                                // without bound (a_eff -> 0 for large mu on a stretched leg; see oracle/lg_oracle.c)
                                const float aeff = fmaxf(Ac.xx + mu * (Ac.xy * e1 + Ac.xz * e2), 0.25f * Ac.xx);
                                const float fn = rn * rcp(1.f + kappa * aeff);
                                fsol = fn > 0.f ? v3(fn, mu * fn * e1, mu * fn * e2) : v3(0, 0, 0);
                            }
                            fnew = fsol;
                        }
                    }
                    fc = fnew;
                    float tl[JPL];
#pragma unroll
                    for (int j = 0; j < JPL; j++) {
                        tl[j] = 0.f;
                        if (lim_s[j] != 0.f) {
                            // compliance: 1/d on the first sweep, then the response measured in the previous one
                            float C = J[j].dinv;
                            if (it > 0 && lim_T[j] > 0.f) C = fminf(fmaxf(lim_s[j] * dqdd[j] * rcp(lim_T[j]), C), 8.f * C);
                            const float vin = -lim_s[j] * (qd[j] + dt * (qdd[j] + dqdd[j])) + dt * lim_T[j] * C;
                            const float Tn = (kl * lim_e[j] + kapl * vin) * rcp(1.f + kapl * dt * C);
                            lim_T[j] = fmaxf(Tn, 0.f);
                            tl[j] = lim_s[j] * lim_T[j];
                        }
                    }
                    // exact response of the whole robot to the current force set
                    const V3 fw = cn * fc.x + ct1 * fc.y + ct2 * fc.z;
                    const V6 fsp = {cross(cp, fw), fw};
                    float du[JPL];
                    const V6 dp = quad_sum<LEGS>(resp_up(J, fsp, tl, du));
                    da0 = chol6_solve(ch, V6{-dp.a, -dp.l});
                    const V6 ac = resp_down(J, da0, du, dqdd);
                    const V3 ra = ac.l + cross(ac.a, cp);
                    resp_c = v3(dot(cn, ra), dot(ct1, ra), dot(ct2, ra));
                }
                if (JPL == 4) f_link[3] += cn * fc.x + ct1 * fc.y + ct2 * fc.z;   // the foot is the last body's own link: add to its other spheres
                else f_link[3] = cn * fc.x + ct1 * fc.y + ct2 * fc.z;
            }
            // ---- semi-implicit Euler ------------------------------------------------------------
            {
                const V3 alpha = a0.a + da0.a;
                const V3 alin = a0.l + da0.l + cross(ww, vw);   // spatial -> classical at O
                const float mv = HOT(o_max_base_lin_vel), mw = HOT(o_max_base_ang_vel);
                // poison for the non-finite guard: the clamps below turn NaN into a limit value (fminf / fmaxf drop it), so what goes into
                // them is watched -- 0 * x is NaN for x = NaN or Inf, 0 otherwise
                nf_poison = fmaf(alin.x + alin.y + alin.z + alpha.x + alpha.y + alpha.z, 0.f, nf_poison);
                vw = v3(clampf(vw.x + dt * alin.x, -mv, mv), clampf(vw.y + dt * alin.y, -mv, mv), clampf(vw.z + dt * alin.z, -mv, mv));
                ww = v3(clampf(ww.x + dt * alpha.x, -mw, mw), clampf(ww.y + dt * alpha.y, -mw, mw), clampf(ww.z + dt * alpha.z, -mw, mw));
#pragma unroll
                for (int j = 0; j < JPL; j++) {
                    const float vl = Lvlim[j];
                    nf_poison = fmaf(qd[j] + qdd[j] + dqdd[j], 0.f, nf_poison);
                    qd[j] = clampf(qd[j] + dt * (qdd[j] + dqdd[j]), -vl, vl);
                    q[j] += dt * qd[j];
                }
                pos += vw * dt;
                const float wn = norm(ww), half = 0.5f * wn * dt;
                float sh, chh;
                __sincosf(half, &sh, &chh);
                const float sc = wn > 1e-12f ? sh * rcp(wn) : 0.5f * dt;
                const float dx = ww.x * sc, dy = ww.y * sc, dz = ww.z * sc, dw = chh;
                const float nw = dw * qw - dx * qx - dy * qy - dz * qz;
                const float nx = dw * qx + qw * dx + dy * qz - dz * qy;
                const float ny = dw * qy + qw * dy + dz * qx - dx * qz;
                const float nz = dw * qz + qw * dz + dx * qy - dy * qx;
                const float inv = rsqrtf(nx * nx + ny * ny + nz * nz + nw * nw);
                qx = nx * inv; qy = ny * inv; qz = nz * inv; qw = nw * inv;
            }
            f_base = quad_sum<LEGS>(fb_acc);
        }  // sub-steps

        STAMP(4);
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the start-of-kernel loads have landed; nothing queues behind the read-back stores (see lg_quad.h)
        if (!PLANE && p.k.cat_enable) {   // go2_cat's job-wide "some joint faster than 4 rad/s" flag (LG_CR_ANY_FAST): every wave that sees one raises it
            bool fast = false;
#pragma unroll
            for (int j = 0; j < JPL; j++) fast = fast || fabsf(qd[j]) > 4.0f;
            fast = fast && live;
            if (__builtin_amdgcn_ballot_w64(fast) != 0ull && threadIdx.x == 0) B.command_ranges[LG_CR_ANY_FAST + (int)(p.counter & 1)] = 1.0f;
        }
        // ---- read-back (genesis_simulator.py:35-60) ----------------------------------------------
        {   // non-finite guard: re-seat the robot (see oracle for the rationale)
            float chk = pos.x + pos.y + pos.z + qx + qy + qz + qw + vw.x + vw.y + vw.z + ww.x + ww.y + ww.z + nf_poison;
#pragma unroll
            for (int j = 0; j < JPL; j++) chk += q[j] + qd[j];
            const int bad = quad_or<LEGS>(isfinite(chk) ? 0 : 1);
            if (bad) {
                pos = v3(HOT(o_base_init_pos[0]), HOT(o_base_init_pos[1]), HOT(o_base_init_pos[2]));
                if (B.env_origins) pos += ld3(B.env_origins + 3 * e);
                vw = ww = v3(0, 0, 0);
                qx = qy = qz = 0.f; qw = 1.f;
#pragma unroll
                for (int j = 0; j < JPL; j++) { q[j] = q0l[j]; qd[j] = 0.f; torque[j] = 0.f; f_link[j] = v3(0, 0, 0); }
                f_link[3] = v3(0, 0, 0); f_base = v3(0, 0, 0);
                // never silent: counted, and the env terminates at this step's check_termination (fail_buf over the threshold)
                if (lead && B.nonfinite_count) atomicAdd(B.nonfinite_count, 1);
                guard_bad = true;
                if (!DO_POST && lead && B.fail_buf) B.fail_buf[e] = LG_FAIL_NONFINITE;
                // the start-of-step snapshots may be what was not finite: this step's rate terms (dof_acc, foot_acc) read zeros instead
                last_foot_v = v3(0, 0, 0);
                if (live) st3(B.last_feet_vel + (e * F + foot_slot) * 3, v3(0, 0, 0));
#pragma unroll
                for (int j = 0; j < JPL; j++) { last_qd[j] = 0.f; if (live) B.last_dof_vel[e * A + d0 + j] = 0.f; }
                if (lead) { st3(B.last_base_lin_vel + 3 * e, v3(0, 0, 0)); st3(B.last_base_ang_vel + 3 * e, v3(0, 0, 0)); }
            }
        }
        // out-of-terrain teleport (genesis_simulator.py:612-628)
        if (pos.x >= HOT(o_bound_x[1]) || pos.x <= HOT(o_bound_x[0]) || pos.y >= HOT(o_bound_y[1]) || pos.y <= HOT(o_bound_y[0])) {
            pos = v3(HOT(o_base_init_pos[0]), HOT(o_base_init_pos[1]), HOT(o_base_init_pos[2]));
            if (B.env_origins) pos += ld3(B.env_origins + 3 * e);
        }
        eul.x = atan2f(2.f * (qw * qx + qy * qz), qw * qw - qx * qx - qy * qy + qz * qz);
        {
            const float sinp = 2.f * (qw * qy - qz * qx);
            eul.y = fabsf(sinp) >= 1.f ? copysignf(1.5707963267948966f, sinp) : asinf(sinp);
        }
        eul.z = atan2f(2.f * (qw * qz + qx * qy), qw * qw + qx * qx - qy * qy - qz * qz);
        blv = quat_rotate_inverse(qx, qy, qz, qw, vw);
        bav = quat_rotate_inverse(qx, qy, qz, qw, ww);
        pg = quat_rotate_inverse(qx, qy, qz, qw, v3(0.f, 0.f, -1.f));
        {   // foot frame at the final state
            const M3 Rb = quat_to_mat(qx, qy, qz, qw);
            M3 Rp = Rb; V3 Pp = v3(0, 0, 0); V6 Vp = {ww, vw};
#pragma unroll
            for (int j = 0; j < JPL; j++) {
                const int b = b0 + j;
                const V3 jp = ld3(M->jpos[b]), ax = ld3(M->axis[b]);
                const M3 jr = {M->jrot[b][0], M->jrot[b][1], M->jrot[b][2], M->jrot[b][3], M->jrot[b][4],
                               M->jrot[b][5], M->jrot[b][6], M->jrot[b][7], M->jrot[b][8]};
                const V3 P = Pp + mul(Rp, jp);
                const M3 Rfix = mul(Rp, jr);
                const V3 s = mul(Rfix, ax);
                Vp.a = Vp.a + s * qd[j];
                Vp.l = Vp.l + cross(P, s) * qd[j];
                Rp = mul(Rfix, axis_angle(ax, q[j]));
                Pp = P;
            }
            const V3 r = Pp + mul(Rp, ld3(M->link_pos[foot_link]));
            foot_p = pos + r;
            foot_v = Vp.l + cross(Vp.a, r);
            foot_zx = Rp.zx; foot_zy = Rp.zy;
        }
        if (live) {
#pragma unroll
            for (int j = 0; j < JPL; j++) {
                B.dof_pos[e * A + d0 + j] = q[j];
                B.dof_vel[e * A + d0 + j] = qd[j];
                B.torques[e * A + d0 + j] = torque[j];
            }
            // link rows of this chain: body links then the foot link (contiguous by construction)
            const int l0 = foot_link - 3;
#pragma unroll
            for (int k = 0; k < 4; k++) st3(B.link_contact_forces + (e * L + l0 + k) * 3, f_link[k]);
            st3(B.feet_pos + (e * F + foot_slot) * 3, foot_p);
            st3(B.feet_vel + (e * F + foot_slot) * 3, foot_v);
            if (lead) {
                st3(B.base_pos + 3 * e, pos);
                B.base_quat[4 * e] = qx; B.base_quat[4 * e + 1] = qy; B.base_quat[4 * e + 2] = qz; B.base_quat[4 * e + 3] = qw;
                st3(B.base_lin_vel_w + 3 * e, vw);
                st3(B.base_ang_vel_w + 3 * e, ww);
                st3(B.base_lin_vel + 3 * e, blv);
                st3(B.base_ang_vel + 3 * e, bav);
                st3(B.projected_gravity + 3 * e, pg);
                st3(B.base_euler + 3 * e, eul);
                st3(B.link_contact_forces + (e * L) * 3, f_base);
            }
        }
        // ---- terrain sampling around the base and the feet (genesis_simulator.py:552-610) ----
        if (P > 0) {
            // yaw-only quaternion (math_utils.py:43-47) applied to the body-frame sample grid
            const float yn = rcp(fmaxf(sqrtf(qz * qz + qw * qw), 1e-9f));
            const float yz = qz * yn, yw = qw * yn;
            float acc = 0.f;
            auto sample_k = [&](int k) {
                const float vx = B.height_points[2 * k], vy = B.height_points[2 * k + 1];
                // quat_apply with xyz = (0, 0, yz): t = 2 xyz x v; r = v + w t + xyz x t
                const float tx = -2.f * yz * vy, ty = 2.f * yz * vx;
                const float rx = vx + yw * tx - yz * ty, ry = vy + yw * ty + yz * tx;
                return sample_min3(O, p.hf, rx + pos.x, ry + pos.y);
            };
            if (hreg) {
#pragma unroll
                for (int i = 0; i < HMAX; i++) {
                    const int k = leg + i * LEGS;
                    if (k < P) {
                        const float h = sample_k(k);
                        hts[i] = h;
                        acc += pos.z - h;
                        if (live) B.measured_heights[(size_t)e * P + k] = h;
                    }
                }
            } else {
                for (int k = leg; k < P; k += LEGS) {
                    const float h = sample_k(k);
                    acc += pos.z - h;
                    if (live) B.measured_heights[(size_t)e * P + k] = h;
                }
            }
            mean_height = quad_sum<LEGS>(acc) / (float)P;
            if (HOT(o_feet_terrain_info)) {
                int px = (int)((foot_p.x + HOT(o_border)) / HOT(o_hscale)), py = (int)((foot_p.y + HOT(o_border)) / HOT(o_hscale));
                px = min(max(px, 0), HOT(o_terrain_rows) - 2);
                py = min(max(py, 0), HOT(o_terrain_cols) - 2);
                const int C = HOT(o_terrain_cols), xm = max(px - 1, 0), ym = max(py - 1, 0);
                const int16_t *hf = p.hf;
                // order of genesis_simulator.py:591-599
                const int hh[9] = {hf[xm * C + py], hf[(px + 1) * C + py], hf[px * C + ym], hf[px * C + py + 1], hf[px * C + py],
                                   hf[xm * C + ym], hf[(px + 1) * C + py + 1], hf[xm * C + py + 1], hf[(px + 1) * C + ym]};
                float sum = 0.f, mx = -1e30f;
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    const float hv = (float)hh[k] * HOT(o_vscale);
                    sum += hv; mx = fmaxf(mx, hv);
                    if (live) B.height_around_feet[((size_t)e * F + foot_slot) * 9 + k] = hv;
                }
                foot_hmean = sum / 9.f; foot_hmax = mx;
                // normal from RAW int16 differences over 2*hscale -- the reference does not apply the
                // vertical scale here (genesis_simulator.py:601-606); reproduced
                const float dx = (float)(hh[1] - hh[0]) / (HOT(o_hscale) * 2.f), dy = (float)(hh[3] - hh[2]) / (HOT(o_hscale) * 2.f);
                const float nn = sqrtf(dx * dx + dy * dy + 1.f);
                if (live) st3(B.normal_vector_around_feet + ((size_t)e * F + foot_slot) * 3, v3(dx / nn, dy / nn, -1.f / nn));
            }
        }
        if (B.link_contact_states && live) {   // genesis_simulator.py:53-55
            const int l0 = foot_link - 3;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int l = l0 + k;
                if ((M->state_link_mask >> l) & 1u) {
                    const int idx = __popc(M->state_link_mask & ((1u << l) - 1u));
                    B.link_contact_states[(size_t)e * __popc(M->state_link_mask) + idx] = norm(f_link[k]) > 1.f ? 1.f : 0.f;
                }
            }
            if (lead && (M->state_link_mask & 1u)) B.link_contact_states[(size_t)e * __popc(M->state_link_mask)] = norm(f_base) > 1.f ? 1.f : 0.f;
        }
    } else {
        // SIM not in this launch: pick the read-back up from LDS (fused behind the physics) or from HBM
        const int l0 = foot_link - 3;
        if (FUSED) {
            auto X3 = [&](int k) { return v3(sX[k * 16 + vlane], sX[(k + 1) * 16 + vlane], sX[(k + 2) * 16 + vlane]); };
            blv = X3(XBLV); bav = X3(XBAV); pg = X3(XPG); eul = X3(XEUL);
#pragma unroll
            for (int k = 0; k < 4; k++) f_link[k] = X3(XFL + 3 * k);
            f_base = X3(XFB); foot_p = X3(XFP); foot_v = X3(XFV); last_foot_v = X3(XLFV);
#pragma unroll
            for (int j = 0; j < JPL; j++) { last_qd[j] = sX[(XLQD + j) * 16 + vlane]; torque[j] = sX[(XTQ + j) * 16 + vlane]; }
            guard_bad = sX[XBAD * 16 + vlane] != 0.f;   // the physics' non-finite guard re-seated this env
        } else {
            blv = ld3(B.base_lin_vel + 3 * e); bav = ld3(B.base_ang_vel + 3 * e);
            pg = ld3(B.projected_gravity + 3 * e); eul = ld3(B.base_euler + 3 * e);
#pragma unroll
            for (int k = 0; k < 4; k++) f_link[k] = ld3(B.link_contact_forces + (e * L + l0 + k) * 3);
            f_base = ld3(B.link_contact_forces + (e * L) * 3);
            foot_p = ld3(B.feet_pos + (e * F + foot_slot) * 3);
            foot_v = ld3(B.feet_vel + (e * F + foot_slot) * 3);
            last_foot_v = ld3(B.last_feet_vel + (e * F + foot_slot) * 3);
#pragma unroll
            for (int j = 0; j < JPL; j++) { last_qd[j] = B.last_dof_vel[e * A + d0 + j]; torque[j] = B.torques[e * A + d0 + j]; }
        }
        if (JPL == 4 && DO_POST) {   // orientation of the foot body (rigid_body_states[:, feet, 3:7] of tron1_sf.py:300) from the joint angles
            M3 Rp = quat_to_mat(qx, qy, qz, qw);
#pragma unroll
            for (int j = 0; j < JPL; j++) {
                const int b = b0 + j;
                const M3 jr = {M->jrot[b][0], M->jrot[b][1], M->jrot[b][2], M->jrot[b][3], M->jrot[b][4],
                               M->jrot[b][5], M->jrot[b][6], M->jrot[b][7], M->jrot[b][8]};
                Rp = mul(mul(Rp, jr), axis_angle(ld3(M->axis[b]), q[j]));
            }
            foot_zx = Rp.zx; foot_zy = Rp.zy;
        }
        if (P > 0) {
            float acc = 0.f;
            if (hreg) {
                const float *__restrict__ mh = B.measured_heights + (size_t)e * P;
#pragma unroll
                for (int i = 0; i < HMAX; i++) hts[i] = mh[min(leg + i * LEGS, P - 1)];      // independent loads, clamped index
#pragma unroll
                for (int i = 0; i < HMAX; i++) acc += (leg + i * LEGS < P) ? pos.z - hts[i] : 0.f;
            } else {
                for (int k = leg; k < P; k += LEGS) acc += pos.z - B.measured_heights[(size_t)e * P + k];
            }
            mean_height = quad_sum<LEGS>(acc) / (float)P;
            if (HOT(o_feet_terrain_info)) {
                float sum = 0.f, mx = -1e30f;
#pragma unroll
                for (int k = 0; k < 9; k++) { const float hv = B.height_around_feet[((size_t)e * F + foot_slot) * 9 + k]; sum += hv; mx = fmaxf(mx, hv); }
                foot_hmean = sum / 9.f; foot_hmax = mx;
            }
        }
    }

    STAMP(5);
    if (!DO_POST && !DO_RESET) return;
    // hot constants of the MDP phases, read from LDS in ONE burst (a read at the point of use costs an exposed LDS round
    // trip each time: every reward term and observation group sits in its own basic block, and there is one wave per SIMD)
    const auto hc_control_dt = HOT(control_dt);
    const auto hc_tracking_sigma = HOT(tracking_sigma);
    const auto hc_base_height_target = HOT(base_height_target);
    const auto hc_feet_air_time_threshold = HOT(feet_air_time_threshold);
    const auto hc_foot_clearance_target = HOT(foot_clearance_target);
    const auto hc_foot_height_offset = HOT(foot_height_offset);
    const auto hc_foot_clearance_sigma = HOT(foot_clearance_sigma);
    const auto hc_max_projected_gravity = HOT(max_projected_gravity);
    const auto hc_max_episode_length = HOT(max_episode_length);
    const auto hc_fail_threshold = HOT(fail_threshold);
    const auto hc_obs_scale_lin_vel = HOT(obs_scale_lin_vel);
    const auto hc_obs_scale_ang_vel = HOT(obs_scale_ang_vel);
    const auto hc_obs_scale_dof_pos = HOT(obs_scale_dof_pos);
    const auto hc_obs_scale_dof_vel = HOT(obs_scale_dof_vel);
    const auto hc_clip_obs = HOT(clip_obs);
    const auto hc_only_positive_rewards = HOT(only_positive_rewards);
    const auto hc_resample_steps = HOT(resample_steps);
    const auto hc_heading_command = HOT(heading_command);
    const auto hc_push_interval = HOT(push_interval);
    const auto hc_obs_layout = FLAT ? (HOT_T(obs_layout))(LG_OBS_GO2) : (WTWP ? (HOT_T(obs_layout))(LG_OBS_GO2_WTW) : (EEP ? (HOT_T(obs_layout))(LG_OBS_GO2_EE) : (PRGP ? (HOT_T(obs_layout))(LG_OBS_PROGRAM) : HOT(obs_layout))));
    const auto hc_gait_mode = (FLAT || ROUGHQ) ? (HOT_T(gait_mode))(0) : (WTWP ? (HOT_T(gait_mode))(1) : HOT(gait_mode));
    const auto hc_add_noise = HOT(add_noise);
    const auto hc_double_shift = (FLAT || EEP) ? (HOT_T(double_shift))(0) : (WTWP ? (HOT_T(double_shift))(1) : HOT(double_shift));
    const auto hc_obs_frame = HOT(obs_frame);
    const auto hc_obs_stack = FLAT ? (HOT_T(obs_stack))(1) : HOT(obs_stack);
    const auto hc_obs_slack = FLAT ? (HOT_T(obs_slack))(0) : HOT(obs_slack);
    const auto hc_obs_sets = HOT(obs_sets);
    const auto hc_priv_frame = FLAT ? (HOT_T(priv_frame))(0) : HOT(priv_frame);
    const auto hc_priv_stack = FLAT ? (HOT_T(priv_stack))(0) : HOT(priv_stack);
    const auto hc_num_obs = HOT(num_obs);
    const auto hc_num_priv_obs = FLAT ? (HOT_T(num_priv_obs))(0) : HOT(num_priv_obs);
    const auto hc_max_push_vel_xy = HOT(max_push_vel_xy);
    const auto hc_episode_length_s = HOT(episode_length_s);
    const auto hc_seed = HOT(seed);
    const auto hc_env_id_offset = HOT(env_id_offset);
    const auto hc_heights_offset = HOT(heights_offset);
    const auto hc_obs_scale_height = HOT(obs_scale_height);
    const auto hc_noise_act0 = (PLANE || ROUGHQ) ? (HOT_T(noise_act0))(0) : HOT(noise_act0);
    const auto hc_about_landing_threshold = HOT(about_landing_threshold);
    const auto hc_terrain_curriculum = PLANE ? (HOT_T(terrain_curriculum))(0) : HOT(terrain_curriculum);
    const auto hc_custom_origins = PLANE ? (HOT_T(custom_origins))(0) : HOT(custom_origins);
    const auto hc_sit_percent = (PLANE || ROUGHQ) ? (HOT_T(sit_percent))(0) : HOT(sit_percent);
    const auto hc_behavior_resample_steps = (FLAT || ROUGHQ) ? (HOT_T(behavior_resample_steps))(0) : HOT(behavior_resample_steps);
    const auto hc_heights_clip_scale = HOT(heights_clip_scale);
    const auto hc_num_labels = PLANE ? (HOT_T(num_labels))(0) : HOT(num_labels);
    asm volatile("" ::: "memory");
    if (STASH || FUSED) {   // bring the MDP working set back from LDS (fused: prefetched by quad_sim_kernel's prologue)
        const int t = threadIdx.x;
        auto SS = [&](int k) { return FUSED ? sStF[k * 16 + vlane] : sSt[k][t]; };
#pragma unroll
        for (int k = 0; k < LG_R_COUNT; k++) es[k] = SS(k);
        int c = LG_R_COUNT;
#pragma unroll
        for (int j = 0; j < JPL; j++) {
            soft_lo[j] = SS(c++); soft_hi[j] = SS(c++); rdof_lo[j] = SS(c++); rdof_span[j] = SS(c++);
            nv_q[j] = SS(c++); nv_qd[j] = SS(c++); nv_act[j] = SS(c++);
        }
        nv_clk[0] = SS(c++); nv_clk[1] = SS(c++);
        cmd0 = SS(c++); cmd1 = SS(c++); cmd2 = SS(c++); cmd3 = SS(c++);
        ep_len = __float_as_int(SS(c++)); fail_buf = (long long)__float_as_int(SS(c++)); air = SS(c++);
        last_contact = __float_as_int(SS(c++));
        origin_pre.x = SS(c++); origin_pre.y = SS(c++); origin_pre.z = SS(c++);
        dirty_prev = __float_as_int(SS(c++));
    }

    // Everything the privileged frames (observation section) read back from the state arrays -- DR parameters, terrain around the
    // feet -- is loaded HERE, in one batch at the start of the MDP phases: the round trips pass under the reward terms, and nothing
    // is loaded behind the reset block's and the observation section's stores (vmcnt retires loads and stores in order: a load next
    // to each `putp` cost a store round trip apiece -- tron1: 25 of them, 17 k of the observation section's 24 k cycles).  A push or
    // a reset updates these copies when it redraws the parameters.
    float ld_kp[JPL], ld_kd[JPL], ld_nv3[3] = {0.f, 0.f, 0.f}, ld_haf[9];
#pragma unroll
    for (int j = 0; j < JPL; j++) ld_kp[j] = ld_kd[j] = 1.f;
    float ld_fric = 1.f, ld_mass = 0.f, ld_com[3] = {0.f, 0.f, 0.f}, ld_push[2] = {0.f, 0.f}, ld_jnt[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 9; k++) ld_haf[k] = 0.f;
    if (DO_RESET && hc_obs_layout != LG_OBS_GO2) {   // ahead of the reset block's stores too; a reset updates these copies
#pragma unroll
        for (int j = 0; j < JPL; j++) { ld_kp[j] = B.kp_scale[e * A + d0 + j]; ld_kd[j] = B.kd_scale[e * A + d0 + j]; }
        if (P > 0 && HOT(o_feet_terrain_info)) {
#pragma unroll
            for (int k = 0; k < 3; k++) ld_nv3[k] = B.normal_vector_around_feet[((size_t)e * F + foot_slot) * 3 + k];
#pragma unroll
            for (int k = 0; k < 9; k++) ld_haf[k] = B.height_around_feet[((size_t)e * F + foot_slot) * 9 + k];
        }
        ld_fric = B.friction_values[e]; ld_mass = B.added_base_mass[e];
#pragma unroll
        for (int k = 0; k < 3; k++) ld_com[k] = B.base_com_bias[3 * e + k];
        ld_push[0] = B.rand_push_vels[3 * e]; ld_push[1] = B.rand_push_vels[3 * e + 1];
        if (B.joint_armature) { ld_jnt[0] = B.joint_armature[e]; ld_jnt[1] = B.joint_friction[e]; ld_jnt[2] = B.joint_damping[e]; }
        asm volatile("" ::: "memory");
    }
    // go2_cat: the task's constraint limits and this env's violation counters, in the same batch (read at the point of use each was an
    // exposed round trip, the nine counters a load -> add -> store chain behind the stores before it: 7 k of the launch's cycles)
    float cat_lim[JPL], cat_tql[JPL], cat_ar = 0.f, cat_minh = 0.f, cat_maxg = 0.f, cat_sp = 0.f, cat_cnt[LG_NUM_CSTR];
#pragma unroll
    for (int j = 0; j < JPL; j++) cat_lim[j] = cat_tql[j] = 0.f;
#pragma unroll
    for (int k = 0; k < LG_NUM_CSTR; k++) cat_cnt[k] = 0.f;
    if (DO_POST && !PLANE && !EEP && p.k.cat_enable) {
#pragma unroll
        for (int j = 0; j < JPL; j++) { cat_lim[j] = T->dof_vel_limits[d0 + j]; cat_tql[j] = M->effort[d0 + j]; }
        cat_ar = T->cat_action_rate; cat_minh = T->cat_min_base_height; cat_maxg = T->cat_max_projected_gravity; cat_sp = T->cat_soft_p;
#pragma unroll
        for (int k = 0; k < LG_NUM_CSTR; k++) cat_cnt[k] = B.cstr_sums[(size_t)k * N + e];
        asm volatile("" ::: "memory");
    }
    // ======================= MDP: legged_robot.py:55-168, 300-334 ============================
    RandSrc rs;
    rs.in = B.rand_in ? B.rand_in + (size_t)e * HOT(slots.n_slots) : nullptr;
    {
        const unsigned long long gid = (unsigned long long)(hc_env_id_offset + e);
        rs.k0 = (unsigned)(hc_seed & 0xFFFFFFFFu); rs.k1 = (unsigned)(hc_seed >> 32);
        rs.e_lo = (unsigned)(gid & 0xFFFFFFFFu); rs.e_hi = (unsigned)(gid >> 32);
        rs.step = (unsigned)p.counter;
    }
    const float cdt = hc_control_dt;
    bool reset = false, time_out = false;
    float total = 0.f;
    float *esum = B.episode_sums;

    // ---- periodic-gait task state (go2_wtw.py:295-346): per-env scalars + this lane's foot entries
    const bool WTW = hc_gait_mode == 1, BIPED = hc_gait_mode == 2, GAIT = WTW || BIPED;
    float *ts = (!FLAT && B.task_state) ? B.task_state + (size_t)e * HOT(task_state_width) : nullptr;
    float gait_time = 0.f, phi = 0.f, gait_period = 1.f, bh_tgt = 0.f, fc_tgt = 0.f, pitch_tgt = 0.f, theta = 0.f, expC = 0.f;
    if (WTW) {
        gait_time = ts[0]; phi = ts[1]; gait_period = ts[2]; bh_tgt = ts[3]; fc_tgt = ts[4]; pitch_tgt = ts[5];
        theta = ts[6 + foot_slot]; expC = ts[18 + foot_slot];
    }
    if (BIPED) {   // layout LG_TASK_STATE_BIPED (tron1_pf_ee.py:167-184)
        gait_time = ts[0]; phi = ts[1]; gait_period = HOT(gait_period_fixed);
        theta = ts[4 + foot_slot]; expC = ts[10 + foot_slot];
    }
    auto resample_behavior = [&](int slot) {   // go2_wtw.py:180-218
        gait_period = (CR(9) - CR(8)) * rs.draw(slot) + CR(8);
        bh_tgt = (CR(11) - CR(10)) * rs.draw(slot + 1) + CR(10);
        fc_tgt = (CR(13) - CR(12)) * rs.draw(slot + 2) + CR(12);
        pitch_tgt = (CR(15) - CR(14)) * rs.draw(slot + 3) + CR(14);
        // one gait index per call for the whole batch (quirk 6): env-independent Philox counter
        float ug;
        if (rs.in) ug = rs.in[slot + 4];
        else { RandSrc g = rs; g.e_lo = 0xFFFFFFFFu; g.e_hi = 0xFFFFFFFFu; ug = g.draw(slot + 4); }
        const int ng = (int)CR(16);
        const int sel = min((int)floorf(ug * (float)ng), ng - 1);
        theta = T->theta_table[sel][foot_slot];
        // pronk / bound gaits keep the lowest clearance target (go2_wtw.py:212-218); the lower bound
        // never moves, so clamping at the env's own resample is equivalent to the reference's all-env pass
        const float t0 = T->theta_table[sel][0], t1 = T->theta_table[sel][1], t2 = T->theta_table[sel][2], t3 = T->theta_table[sel][3];
        if (t0 == 0.f && t1 == 0.f && ((t2 == 0.f && t3 == 0.f) || (t2 == 0.5f && t3 == 0.5f))) fc_tgt = CR(12);
    };

    auto resample_commands = [&](int slot, const float *pre = nullptr) {  // legged_robot.py:317-334
        float u0, u1, u2;
        if (pre) { u0 = pre[0]; u1 = pre[1]; u2 = pre[2]; }
        else rs.draw3(slot, u0, u1, u2);
        cmd0 = (CR(1) - CR(0)) * u0 + CR(0);
        cmd1 = (CR(3) - CR(2)) * u1 + CR(2);
        if (hc_heading_command) cmd3 = (CR(7) - CR(6)) * u2 + CR(6);
        else cmd2 = (CR(5) - CR(4)) * u2 + CR(4);
        const float keep = sqrtf(cmd0 * cmd0 + cmd1 * cmd1 + cmd2 * cmd2) > 0.2f ? 1.f : 0.f;
        cmd0 *= keep; cmd1 *= keep; cmd2 *= keep;
    };

    if (DO_POST) {
        ep_len += 1;                                                   // legged_robot.py:60
        // ---- _post_physics_step_callback (legged_robot.py:300-315) ----
        if (ep_len % hc_resample_steps == 0) resample_commands(HOT(slots.cb_cmd));
        if (hc_heading_command) {
            // forward = quat_apply(base_quat, [1,0,0]) (math_utils.py:34-40)
            const V3 xyz = v3(qx, qy, qz), bvec = v3(1.f, 0.f, 0.f);
            const V3 t = cross(xyz, bvec) * 2.f;
            const V3 fwd = bvec + t * qw + cross(xyz, t);
            const float heading = atan2f(fwd.y, fwd.x);
            cmd2 = clampf(0.5f * wrap_to_pi(cmd3 - heading), HOT(yaw_clip[0]), HOT(yaw_clip[1]));
        }
        if (hc_push_interval > 0 && (p.counter % hc_push_interval) == 0) {   // genesis_simulator.py:150-158
            const float m = hc_max_push_vel_xy;
            const float px = (m + m) * rs.draw(HOT(slots.push)) - m, py = (m + m) * rs.draw(HOT(slots.push) + 1) - m;
            vw.x += px; vw.y += py;
            ld_push[0] = px; ld_push[1] = py;
            if (lead) {
                B.rand_push_vels[3 * e] = px; B.rand_push_vels[3 * e + 1] = py;
                B.base_lin_vel_w[3 * e] = vw.x; B.base_lin_vel_w[3 * e + 1] = vw.y;
            }
        }
        if (WTW && hc_behavior_resample_steps > 0 && ep_len % hc_behavior_resample_steps == 0)   // go2_wtw.py:258-263
            resample_behavior(HOT(slots.task_cb));
#ifdef LG_DBG_RET_CALLBACK
        if (p.counter >= 0) { if (lead) B.rew_buf[e] = cmd2 + (float)ep_len; return; }
#endif
        STAMP(6);
        // ---- check_termination (legged_robot.py:78-92) ----
        const int l0 = foot_link - 3;
        int fail = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if ((M->term_link_mask >> (l0 + k)) & 1u) fail |= norm(f_link[k]) > 10.0f ? 1 : 0;
        fail = quad_or<LEGS>(fail);
        if (M->term_link_mask & 1u) fail |= norm(f_base) > 10.0f ? 1 : 0;
        fail |= pg.z > hc_max_projected_gravity ? 1 : 0;
        if (guard_bad) fail_buf = LG_FAIL_NONFINITE;   // a re-seated env ends its episode here (lgsim.h)
        fail_buf += fail;
        time_out = (float)ep_len > hc_max_episode_length;
        reset = ((float)fail_buf > hc_fail_threshold) || time_out;

#ifdef LG_DBG_RET_TERM
        if (p.counter >= 0) { if (lead) B.rew_buf[e] = cmd2 + (float)fail_buf + (reset ? 1.f : 0.f); return; }
#endif
        // ---- constraints as terminations (go2_cat.py:143-205): nine 0/1 violation flags; p = 1 for a violated hard constraint,
        //      soft_p for a violated soft / style one (constraint_manager.py:25-74 with binary inputs), per-episode violation counts
        float cat_keep = 1.f;     // (1 - p), applied to the reward before the positive clip (go2_cat.py:219-223)
        float cstr_p = 0.f;
        if (!PLANE && !EEP && p.k.cat_enable) {
            int c_tq = 0, c_qd = 0, c_ar = 0, lo_any = 0, hi_any = 0, c_fast = 0;
#pragma unroll
            for (int j = 0; j < JPL; j++) {
                c_tq |= fabsf(torque[j]) > cat_tql[j] ? 1 : 0;
                c_qd |= fabsf(qd[j]) > cat_lim[j] ? 1 : 0;
                c_ar |= fabsf(act[j] - last_act[j]) / cdt > cat_ar ? 1 : 0;
                lo_any |= q[j] < soft_lo[j] ? 1 : 0;
                hi_any |= q[j] > soft_hi[j] ? 1 : 0;
                c_fast |= fabsf(qd[j]) > 4.0f ? 1 : 0;
            }
            int c_col = 0;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if ((M->pen_link_mask >> (l0 + k)) & 1u) c_col |= norm(f_link[k]) > 10.0f ? 1 : 0;
            const int c_stb = norm(f_link[3]) > 4.f * fabsf(f_link[3].z) ? 1 : 0;
            // one packed OR over the env's lanes
            int bits = c_tq | (c_qd << 1) | (c_ar << 2) | (c_col << 4) | (c_stb << 5) | (lo_any << 9) | (hi_any << 10) | (c_fast << 11);
            bits = quad_or<LEGS>(bits);
            if (M->pen_link_mask & 1u) bits |= norm(f_base) > 10.0f ? (1 << 4) : 0;
            if (((P > 0 ? mean_height : pos.z)) < cat_minh) bits |= 1 << 3;
            if ((bits & (1 << 9)) && (bits & (1 << 10))) bits |= 1 << 6;                 // any(q < lo) * any(q > hi)  (go2_cat.py:170-171)
            if (pg.z > cat_maxg) bits |= 1 << 7;
            const float cmdn = sqrtf(cmd0 * cmd0 + cmd1 * cmd1 + cmd2 * cmd2);
            if (CR(LG_CR_ANY_FAST + (int)(p.counter & 1)) != 0.f && cmdn < 0.1f) bits |= 1 << 8;   // job-wide motion flag x own zero command
            if (blockIdx.x == 0 && threadIdx.x == 0) B.command_ranges[LG_CR_ANY_FAST + (int)((p.counter + 1) & 1)] = 0.f;   // next step's slot
            bits &= 0x1FF;
            cstr_p = (bits & 0xF0) ? 1.f : ((bits & 0x10F) ? cat_sp : 0.f);
            cat_keep = 1.0f - cstr_p;
            if (lead) {
                B.cstr_prob[e] = cstr_p;
#pragma unroll
                for (int k = 0; k < LG_NUM_CSTR; k++) { cat_cnt[k] += (float)((bits >> k) & 1); B.cstr_sums[(size_t)k * N + e] = cat_cnt[k]; }
            }
        }
        // ---- compute_reward (legged_robot.py:150-168), alphabetical order ----
        float scl[LG_R_COUNT];   // one burst of broadcast LDS reads instead of one exposed round trip per active term
#pragma unroll
        for (int k = 0; k < LG_R_COUNT; k++) scl[k] = HOT(reward_scales[k]);
        auto add = [&](int id, float r) {
            const float rew = r * scl[id];
            total += rew;
            es[id] += rew;
        };
        const float cmd_xy = sqrtf(cmd0 * cmd0 + cmd1 * cmd1);
        const float cmd_xyz = sqrtf(cmd0 * cmd0 + cmd1 * cmd1 + cmd2 * cmd2);
        float dq0[JPL];
#pragma unroll
        for (int j = 0; j < JPL; j++) dq0[j] = q[j] - q0l[j];
        auto gait_reward = [&]() {   // "step" indicator of go2_wtw.py:377-470 / tron1_pf_ee.py:347-424
            const float two_pi = 6.283185307179586f;
            float ph = phi + theta;
            ph = (ph - floorf(ph)) * two_pi;                             // (phi + theta) % 1.0, operands >= 0
            const float b_sw = HOT(b_swing) * two_pi;
            const float c_frc = (ph >= 0.f && ph < b_sw) ? -1.f : 0.f;
            const float c_spd = (ph >= b_sw && ph < two_pi) ? -1.f : 0.f;
            expC = c_frc;
            return __expf(quad_sum<LEGS>(c_spd * norm(foot_v) + c_frc * norm(f_link[3])));
        };
        if (RON(LG_R_ACTION_RATE)) {                              // :495-497
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < JPL; j++) { const float d = last_act[j] - act[j]; s += d * d; }
            add(LG_R_ACTION_RATE, quad_sum<LEGS>(s));
        }
        if (RON(LG_R_ACTION_SMOOTHNESS)) {                        // :499-503
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < JPL; j++) { const float d = act[j] - 2.f * last_act[j] + llast_act[j]; s += d * d; }
            add(LG_R_ACTION_SMOOTHNESS, quad_sum<LEGS>(s));
        }
        if (RON(LG_R_ANG_VEL_XY)) add(LG_R_ANG_VEL_XY, bav.x * bav.x + bav.y * bav.y);   // :462-464
        if (RON(LG_R_BASE_HEIGHT)) {                              // :470-476
            // plane: measured_heights is the all-zero buffer of genesis_simulator.py:494 -> base z
            const float d = (P > 0 ? mean_height : pos.z) - hc_base_height_target;
            add(LG_R_BASE_HEIGHT, d * d);
        }
        if (RON(LG_R_BIPED_PERIODIC_GAIT)) add(LG_R_BIPED_PERIODIC_GAIT, gait_reward());  // tron1_pf_ee.py:426-433
        if (RON(LG_R_COLLISION)) {                                // :505-512
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if ((M->pen_link_mask >> (l0 + k)) & 1u) s += norm(f_link[k]) > 0.1f ? 1.f : 0.f;
            s = quad_sum<LEGS>(s);
            if (M->pen_link_mask & 1u) s += norm(f_base) > 0.1f ? 1.f : 0.f;
            add(LG_R_COLLISION, s);
        }
        if (RON(LG_R_DOF_ACC)) {                                  // :490-493
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < JPL; j++) { const float d = (last_qd[j] - qd[j]) / cdt; s += d * d; }
            add(LG_R_DOF_ACC, quad_sum<LEGS>(s));
        }
        if (RON(LG_R_DOF_CLOSE_TO_DEFAULT)) {                     // :571-573
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < JPL; j++) s += dq0[j] * dq0[j];
            add(LG_R_DOF_CLOSE_TO_DEFAULT, quad_sum<LEGS>(s));
        }
        if (RON(LG_R_DOF_POS_LIMITS)) {                           // :518-522
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < JPL; j++) {
                s += -fminf(q[j] - soft_lo[j], 0.f);
                s += fmaxf(q[j] - soft_hi[j], 0.f);
            }
            add(LG_R_DOF_POS_LIMITS, quad_sum<LEGS>(s));
        }
        if (RON(LG_R_DOF_POS_STAND_STILL)) {                      // :561-563
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < JPL; j++) s += dq0[j] * dq0[j];
            add(LG_R_DOF_POS_STAND_STILL, quad_sum<LEGS>(s) * (cmd_xyz < 0.1f ? 1.f : 0.f));
        }
        if (RON(LG_R_DOF_POWER)) {                                // :486-488
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < JPL; j++) s += fabsf(torque[j] * qd[j]);
            add(LG_R_DOF_POWER, quad_sum<LEGS>(s));
        }
        if (RON(LG_R_DOF_VEL)) {                                  // :482-484
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < JPL; j++) s += qd[j] * qd[j];
            add(LG_R_DOF_VEL, quad_sum<LEGS>(s));
        }
        if (RON(LG_R_DOF_VEL_STAND_STILL)) {                      // :557-559
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < JPL; j++) s += fabsf(qd[j]);
            add(LG_R_DOF_VEL_STAND_STILL, quad_sum<LEGS>(s) * (cmd_xyz < 0.1f ? 1.f : 0.f));
        }
        if (RON(LG_R_FEET_AIR_TIME)) {                            // :545-555 (stateful)
            const int contact = f_link[3].z > 1.0f ? 1 : 0;
            const int filt = contact | last_contact;
            last_contact = contact;
            const float first = (air > 0.f ? 1.f : 0.f) * (float)filt;
            air += cdt;
            float r = quad_sum<LEGS>((air - hc_feet_air_time_threshold) * first);
            r *= ((!FLAT && HOT(air_time_cmd_dims) == 3) ? cmd_xyz : cmd_xy) > 0.1f ? 1.f : 0.f;
            air *= filt ? 0.f : 1.f;
            add(LG_R_FEET_AIR_TIME, r);
        }
        if (RON(LG_R_FEET_CONTACT_STAND_STILL)) {                 // :565-569
            const float cnt = quad_sum<LEGS>(f_link[3].z > 0.1f ? 1.f : 0.f);
            add(LG_R_FEET_CONTACT_STAND_STILL, (cnt == (float)LEGS ? 1.f : 0.f) * (cmd_xyz < 0.1f ? 1.f : 0.f));
        }
        if (RON(LG_R_FEET_DISTANCE)) {                            // tron1_pf_ee.py:458-463 (two feet: lane pair)
            const float ox = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(foot_p.x), 0xB1, 0xF, 0xF, false));
            const float oy = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(foot_p.y), 0xB1, 0xF, 0xF, false));
            const float dxy = sqrtf((foot_p.x - ox) * (foot_p.x - ox) + (foot_p.y - oy) * (foot_p.y - oy));
            add(LG_R_FEET_DISTANCE, fmaxf(0.f, HOT(foot_distance_threshold) - dxy));
        }
        if (RON(LG_R_FOOT_ACC)) {                                 // :605-608
            const V3 a = (foot_v - last_foot_v) * (1.f / cdt);
            add(LG_R_FOOT_ACC, quad_sum<LEGS>(dot(a, a)));
        }
        if (RON(LG_R_FOOT_CLEARANCE)) {                           // :575-588
            const float vxy = sqrtf(foot_v.x * foot_v.x + foot_v.y * foot_v.y);
            // go2_ee.py:136-150 measures the clearance above the mean terrain height around the foot, tron1_pf_ee.py:442-456 and
            // go2_cts.py:156-170 above the max (LgTaskCfg.foot_clearance_ref)
            const float d = foot_p.z - (HOT(foot_clearance_ref) == 1 ? foot_hmean : (HOT(foot_clearance_ref) == 2 ? foot_hmax : 0.f))
                            - hc_foot_clearance_target - hc_foot_height_offset;   // tron1_pf_ee.py:442-456 uses the max
            const float err = quad_sum<LEGS>(vxy * (d * d));
            add(LG_R_FOOT_CLEARANCE, __expf(-err / hc_foot_clearance_sigma));
        }
        if (RON(LG_R_FOOT_LANDING_VEL)) {                         // :590-599
            const bool c01 = f_link[3].z > 0.1f;
            const bool land = ((foot_p.z - hc_foot_height_offset) < hc_about_landing_threshold) && !c01 && (foot_v.z < 0.f);
            const float vz = land ? foot_v.z : 0.f;
            add(LG_R_FOOT_LANDING_VEL, quad_sum<LEGS>(vz * vz));
        }
        if (RON(LG_R_HIP_POS)) {
            if constexpr (JPL == 4)   // tron1_sf.py:281-285 (LG_R_HIP_POS_ZERO_COMMAND): dofs [1, 5] = the second joint of each leg, at ~zero command
                add(LG_R_HIP_POS_ZERO_COMMAND, quad_sum<LEGS>(dq0[1] * dq0[1]) * (cmd_xyz < 0.1f ? 1.f : 0.f));
            else add(LG_R_HIP_POS, quad_sum<LEGS>(dq0[0] * dq0[0]));   // go2_ee.py:152-159
        }
        if (RON(LG_R_KEEP_BALANCE)) add(LG_R_KEEP_BALANCE, 1.f);                      // :601-603
        if (RON(LG_R_LIN_VEL_Z)) add(LG_R_LIN_VEL_Z, blv.z * blv.z);                  // :458-460
        if (RON(LG_R_NO_FLY)) {                                   // tron1_pf.py:151-154: exactly one foot on the ground
            const float cnt = quad_sum<LEGS>(f_link[3].z > T->no_fly_contact_threshold ? 1.f : 0.f);
            add(LG_R_NO_FLY, cnt == 1.f ? 1.f : 0.f);
        }
        if (RON(LG_R_ORIENTATION)) add(LG_R_ORIENTATION, pg.x * pg.x + pg.y * pg.y);  // :466-468
        if (RON(LG_R_QUAD_PERIODIC_GAIT)) {
            if constexpr (JPL == 4) {   // tron1_sf.py:297-308 (LG_R_FOOT_FLAT): world z in the foot frame = third row of the foot's rotation
                const float tilt = fabsf(foot_zx) + fabsf(foot_zy);
                add(LG_R_FOOT_FLAT, quad_sum<LEGS>(__expf(-tilt / 0.1f)));
            } else add(LG_R_QUAD_PERIODIC_GAIT, gait_reward());   // go2_wtw.py:472-484
        }
        if (RON(LG_R_TORQUES)) {                                  // :478-480
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < JPL; j++) s += torque[j] * torque[j];
            add(LG_R_TORQUES, quad_sum<LEGS>(s));
        }
        if (RON(LG_R_TRACKING_ANG_VEL)) {                         // :539-543
            const float d = cmd2 - bav.z;
            add(LG_R_TRACKING_ANG_VEL, __expf(-(d * d) / hc_tracking_sigma));
        }
        if (RON(LG_R_TRACKING_BASE_HEIGHT)) {                     // go2_wtw.py:495-500 (plane: heights are zero)
            // wtw: per-env target on the plane; tron1_pf_ee.py:435-440: fixed target, mean over the height samples
            const float d = WTW ? pos.z - bh_tgt : (P > 0 ? mean_height : pos.z) - hc_base_height_target;
            add(LG_R_TRACKING_BASE_HEIGHT, __expf(-(d * d) / HOT(base_height_sigma)));
        }
        if (JPL == 4 && RON(LG_R_TRACKING_FOOT_CLEARANCE)) {      // tron1_sf.py:287-295 (LG_R_KEEP_ANKLE_PITCH_ZERO_IN_AIR): |ankle angle| of the feet in the air
            const float s = quad_sum<LEGS>(f_link[3].z > 1.0f ? 0.f : fabsf(q[JPL - 1]));
            add(LG_R_KEEP_ANKLE_PITCH_ZERO_IN_AIR, __expf(-fabsf(s) / 0.2f));
        }
        if (JPL != 4 && RON(LG_R_TRACKING_FOOT_CLEARANCE)) {      // go2_wtw.py:507-519
            const float vxy = sqrtf(foot_v.x * foot_v.x + foot_v.y * foot_v.y);
            const float d = foot_p.z - fc_tgt - hc_foot_height_offset;
            add(LG_R_TRACKING_FOOT_CLEARANCE, __expf(-quad_sum<LEGS>(vxy * (d * d)) / hc_foot_clearance_sigma));
        }
        if (RON(LG_R_TRACKING_LIN_VEL)) {                         // :533-537
            const float dx = cmd0 - blv.x, dy = cmd1 - blv.y;
            add(LG_R_TRACKING_LIN_VEL, __expf(-(dx * dx + dy * dy) / hc_tracking_sigma));
        }
        if (RON(LG_R_TRACKING_ORIENTATION)) {                     // go2_wtw.py:502-505
            const float dp = eul.y - pitch_tgt;
            add(LG_R_TRACKING_ORIENTATION, __expf(-(eul.x * eul.x + dp * dp) / HOT(euler_sigma)));
        }
#ifdef LG_DBG_RET_REW
        if (p.counter >= 0) { if (lead) B.rew_buf[e] = total + es[0] + es[5] + es[28]; return; }
#endif
        STAMP(7);
        if (hc_only_positive_rewards) total = fmaxf(total * cat_keep, 0.f);        // :161-162; go2_cat.py:219-223 scales by (1 - p) first
        if (RON(LG_R_TERMINATION)) add(LG_R_TERMINATION, (reset && !time_out) ? 1.f : 0.f);  // :163-168
        if (GAIT) {  // gait clock (go2_wtw.py:29-36, tron1_pf_ee.py:28-35).  The reference additionally restarts env 0's clock whenever
                     // ANY env wraps (index-flatten bug); that grid-wide coupling is deliberately not reproduced.
            gait_time += cdt;
            if (gait_time >= gait_period - cdt / 2.f) gait_time = 0.f;
            phi = gait_time / gait_period;
        }
    } else {
        reset = B.reset_buf[e] != 0;
        time_out = B.time_out_buf[e] != 0;
        total = B.rew_buf[e];
    }

    STAMP(8);
    // Every load issued so far has long landed; say so.  vmcnt retires loads and stores in issue order, and the compiler's
    // scoreboard keeps a prologue load "pending" until some wait formally covers it: the first use of such a value after the
    // conditional stores below (joins with path-dependent store counts) then became `s_waitcnt vmcnt(0..2)`, i.e. a drain of the
    // stores just issued -- five store round trips in the actor-frame section alone (~4 k cycles of the biped's MDP launch).
    if (DO_RESET) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
    // ---- reset_idx (legged_robot.py:94-148) + simulator.reset_idx (genesis_simulator.py:62-82) ----
    if (DO_RESET) {
        V3 pos_origin_override = v3(0, 0, 0);
        bool have_origin = false;
        if (reset) {
            // env-level uniforms of a reset (commands, friction, mass, CoM, root twist): ONE Philox block per lane of the env,
            // shared inside the quad, instead of one call per quantity on the lead lane (a call is ~800 cycles and the launch
            // ends with its slowest wave, which is always one that resets).  Injected draws keep their slots.
            constexpr int NB = LEGS == 4 ? 1 : 3;          // blocks per lane: 16 uniforms per quadruped env, 24 per biped env
            float eu[4 * LEGS * NB];
            const bool bundle = !rs.in;
            // Replicated launch: the Philox blocks a resetting leg-lane needs (env-level bundle, _reset_dofs, kp, kd, root xy, terrain
            // level, the sit coin: 7-9 calls of ~800 cycles) are evaluated ONE PER REPLICA in two rounds and fetched from the replica
            // that holds them (same counters, same words: the values are those of the one-by-one calls).
            const bool spread = RP && !rs.in;
            U4 R1 = {0u, 0u, 0u, 0u}, R2 = R1;
            float s_ud[4] = {0.f, 0.f, 0.f, 0.f}, s_up[4] = {0.f, 0.f, 0.f, 0.f}, s_kd[4] = {0.f, 0.f, 0.f, 0.f}, s_xy[2] = {0.f, 0.f}, s_tl = 0.f, s_sit = 0.f;
            if (spread) {
                const unsigned c_dof = 0x40000000u + (unsigned)(HOT(slots.reset_dof) + d0), c_kp = 0x40000000u + (unsigned)(HOT(slots.dr_kp) + d0),
                               c_kd = 0x40000000u + (unsigned)(HOT(slots.dr_kd) + d0), c_bun = 0x80000000u + (unsigned)(0x200 + leg * NB);
                const int sxy = HOT(slots.reset_root_xy), stl = HOT(slots.terrain_level), ssit = HOT(slots.task_reset);
                // round 1: biped  r0-r2 bundle blocks, r3 _reset_dofs; quadruped  r0 bundle, r1 _reset_dofs, r2 kp, r3 kd
                const unsigned c1 = LEGS == 2 ? (sub < 3 ? c_bun + (unsigned)sub : c_dof) : (sub == 0 ? c_bun : (sub == 1 ? c_dof : (sub == 2 ? c_kp : c_kd)));
                R1 = philox4x32_10(U4{rs.e_lo, rs.e_hi, rs.step, c1}, rs.k0, rs.k1);
                // round 2: biped  r0 kp, r1 kd, r2 root xy, r3 sit coin (job-wide: env words all ones); quadruped  r0 root xy, r1 terrain level
                const bool coin = LEGS == 2 && sub == 3;
                const unsigned c2 = LEGS == 2 ? (sub == 0 ? c_kp : (sub == 1 ? c_kd : (sub == 2 ? (unsigned)(sxy >> 2) : (unsigned)(ssit >> 2))))
                                              : (sub == 0 ? (unsigned)(sxy >> 2) : (unsigned)(stl >> 2));
                R2 = philox4x32_10(U4{coin ? 0xFFFFFFFFu : rs.e_lo, coin ? 0xFFFFFFFFu : rs.e_hi, rs.step, c2}, rs.k0, rs.k1);
                const int rdof = LEGS == 2 ? 3 : 1;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    s_ud[k] = u01(from(word(R1, k), rdof));
                    s_up[k] = u01(LEGS == 2 ? from(word(R2, k), 0) : from(word(R1, k), 2));
                    s_kd[k] = u01(LEGS == 2 ? from(word(R2, k), 1) : from(word(R1, k), 3));
                }
                const int rxy = LEGS == 2 ? 2 : 0;
                s_xy[0] = u01(from(word(R2, sxy & 3), rxy));
                // the second coordinate sits in the next block when the slot pair straddles one (never with the slot tables in use)
                s_xy[1] = (sxy & 3) == 3 ? rs.draw(sxy + 1) : u01(from(word(R2, (sxy + 1) & 3), rxy));
                if (LEGS == 4) s_tl = u01(from(word(R2, stl & 3), 1));
                else s_sit = u01(from(word(R2, ssit & 3), 3));
            }
            if (bundle) {
#pragma unroll
                for (int b_ = 0; b_ < NB; b_++) {
                    float b[4];
                    if (spread) {
#pragma unroll
                        for (int k = 0; k < 4; k++) b[k] = u01(from(word(R1, k), b_));
                    } else rs.block4(0x200 + leg * NB + b_, b[0], b[1], b[2], b[3]);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int x = __float_as_int(b[k]);
                        if (LEGS == 4) {
                            eu[0 + k] = __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x00, 0xF, 0xF, false));
                            eu[4 + k] = __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x55, 0xF, 0xF, false));
                            eu[8 + k] = __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0xAA, 0xF, 0xF, false));
                            eu[12 + k] = __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0xFF, 0xF, 0xF, false));
                        } else {   // lane pairs (0,1) and (2,3) of the quad are two envs
                            eu[4 * b_ + k] = __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0xA0, 0xF, 0xF, false));               // quad_perm [0,0,2,2]
                            eu[4 * (NB + b_) + k] = __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0xF5, 0xF, 0xF, false));        // quad_perm [1,1,3,3]
                        }
                    }
                }
            }
            // slots of eu[]: 0-2 commands, 3 friction, 4-6 CoM, 7 mass, 8-10 root lin vel, 12-14 root ang vel; biped only:
            // 16-18 joint armature / friction / damping, 19-20 gait phase offsets, 21 terrain level
            constexpr bool EU2 = LEGS == 2;
            if (hc_terrain_curriculum && p.counter > 0) {
                // legged_robot.py:254-272 + genesis_simulator.py:140-148 (skipped on the construction-time reset,
                // where the reference returns early because init_done is False)
                const V3 org = origin_pre;
                const float dx = pos.x - org.x, dy = pos.y - org.y;
                const float dist = sqrtf(dx * dx + dy * dy);
                const bool up = dist > HOT(terrain_env_length) / 2.f;
                const bool down = (dist < sqrtf(cmd0 * cmd0 + cmd1 * cmd1) * hc_episode_length_s * 0.5f) && !up;
                int lvl = B.terrain_levels[e] + (up ? 1 : 0) - (down ? 1 : 0);
                if (lvl >= HOT(max_terrain_level)) lvl = min((int)floorf(((bundle && EU2) ? eu[EU2 ? 21 : 0] : (spread ? s_tl : rs.draw(HOT(slots.terrain_level)))) * (float)HOT(max_terrain_level)), HOT(max_terrain_level) - 1);
                else lvl = max(lvl, 0);
                const V3 norg = ld3(B.terrain_origins + ((size_t)lvl * HOT(terrain_cols_n) + B.terrain_types[e]) * 3);
                if (lead) { B.terrain_levels[e] = lvl; st3(B.env_origins + 3 * e, norg); }
                pos_origin_override = norg; have_origin = true;
            }
            if (WTW) { resample_behavior(HOT(slots.task_reset)); gait_time = 0.f; phi = 0.f; }   // go2_wtw.py:124-142
            resample_commands(HOT(slots.reset_cmd), bundle ? eu : nullptr);
            // tron1_pf_ee.py:204-210: ONE coin per reset_idx call sends the whole batch to the sit pose (quirk 11)
            bool sit = false;
            if (hc_sit_percent > 0.f) {
                float us;
                if (rs.in) us = rs.in[HOT(slots.task_reset)];
                else if (spread && LEGS == 2) us = s_sit;
                else { RandSrc g = rs; g.e_lo = 0xFFFFFFFFu; g.e_hi = 0xFFFFFFFFu; us = g.draw(HOT(slots.task_reset)); }
                sit = us < hc_sit_percent;
            }
            // _reset_dofs (go2.py:17-37): default + U(range) per joint, zero velocity
            float ud[JPL];
            if (spread) {
#pragma unroll
                for (int j = 0; j < JPL; j++) ud[j] = s_ud[j];
            } else if constexpr (JPL == 4) rs.draw4(HOT(slots.reset_dof) + d0, ud[0], ud[1], ud[2], ud[3]);
            else rs.draw3(HOT(slots.reset_dof) + d0, ud[0], ud[1], ud[2]);
#pragma unroll
            for (int j = 0; j < JPL; j++) {
                q[j] = sit ? T->sit_dof_pos[d0 + j]
                           : q0l[j] + (rdof_span[j] * ud[j] + rdof_lo[j]);
                qd[j] = 0.f;
                last_qd[j] = 0.f;
                act[j] = last_act[j] = llast_act[j] = 0.f;
            }
            // _reset_root_states (go2.py:119-134)
            pos = (sit ? ld3(T->sit_pos) : v3(HOT(o_base_init_pos[0]), HOT(o_base_init_pos[1]), HOT(o_base_init_pos[2]))) + (have_origin ? pos_origin_override : origin_pre);
            if (hc_custom_origins) {
                pos.x += HOT(reset_root_xy_span) * (spread ? s_xy[0] : rs.draw(HOT(slots.reset_root_xy))) + HOT(reset_root_xy_lo);
                pos.y += HOT(reset_root_xy_span) * (spread ? s_xy[1] : rs.draw(HOT(slots.reset_root_xy) + 1)) + HOT(reset_root_xy_lo);
            }
            qx = sit ? T->sit_quat[0] : HOT(base_init_quat[0]); qy = sit ? T->sit_quat[1] : HOT(base_init_quat[1]);
            qz = sit ? T->sit_quat[2] : HOT(base_init_quat[2]); qw = sit ? T->sit_quat[3] : HOT(base_init_quat[3]);
            if (HOT(reset_lin_vel_span) != 0.f || HOT(reset_ang_vel_span) != 0.f || rs.in) {   // go2.py:131-133 draws U(0,0): skip
                float a0, a1, a2, b0_, b1_, b2_;
                if (bundle) { a0 = eu[8]; a1 = eu[9]; a2 = eu[10]; b0_ = eu[12]; b1_ = eu[13]; b2_ = eu[14]; }
                else { rs.draw3(HOT(slots.reset_lin_vel), a0, a1, a2); rs.draw3(HOT(slots.reset_ang_vel), b0_, b1_, b2_); }
                vw = v3(HOT(reset_lin_vel_span) * a0 + HOT(reset_lin_vel_lo), HOT(reset_lin_vel_span) * a1 + HOT(reset_lin_vel_lo),
                        HOT(reset_lin_vel_span) * a2 + HOT(reset_lin_vel_lo));
                ww = v3(HOT(reset_ang_vel_span) * b0_ + HOT(reset_ang_vel_lo), HOT(reset_ang_vel_span) * b1_ + HOT(reset_ang_vel_lo),
                        HOT(reset_ang_vel_span) * b2_ + HOT(reset_ang_vel_lo));
            } else {
                vw = v3(HOT(reset_lin_vel_lo), HOT(reset_lin_vel_lo), HOT(reset_lin_vel_lo));
                ww = v3(HOT(reset_ang_vel_lo), HOT(reset_ang_vel_lo), HOT(reset_ang_vel_lo));
            }
            if (sit) { vw = v3(0, 0, 0); ww = v3(0, 0, 0); }        // tron1_pf_ee.py:304-309
            if (BIPED) {                                             // tron1_pf_ee.py:220-226
                const float th0 = T->theta_table[0][0] + ((bundle && EU2) ? eu[EU2 ? 19 : 0] : rs.draw(HOT(slots.task_reset) + 1));
                theta = foot_slot == 0 ? th0 : th0 + (T->theta_table[0][1] - T->theta_table[0][0]);
                gait_time = ((bundle && EU2) ? eu[EU2 ? 20 : 0] : rs.draw(HOT(slots.task_reset) + 2)) * gait_period;
                phi = gait_time / gait_period;
            }
            // the reference stores the commanded reset twist verbatim in the body-frame properties
            // (genesis_simulator.py:128-129) and refreshes projected gravity (:125)
            blv = vw; bav = ww;
            pg = quat_rotate_inverse(qx, qy, qz, qw, v3(0.f, 0.f, -1.f));
            air = 0.f; ep_len = 0; fail_buf = 0; last_foot_v = v3(0, 0, 0);
            if (live) {
#pragma unroll
                for (int j = 0; j < JPL; j++) {
                    B.dof_pos[e * A + d0 + j] = q[j];
                    B.dof_vel[e * A + d0 + j] = 0.f;
                    B.last_dof_vel[e * A + d0 + j] = 0.f;
                    B.actions[e * A + d0 + j] = 0.f; B.last_actions[e * A + d0 + j] = 0.f; B.llast_actions[e * A + d0 + j] = 0.f;
                }
                st3(B.last_feet_vel + (e * F + foot_slot) * 3, v3(0, 0, 0));
            }
            // the redrawn parameters are formed in every replica (each writes its share of the privileged frame from them); replica 0 stores
            if ((RP || live) && HOT(dr_pd_on)) {   // genesis_simulator.py:735-739
                float up[JPL], ud2[JPL];
                if (spread) {
#pragma unroll
                    for (int j = 0; j < JPL; j++) { up[j] = s_up[j]; ud2[j] = s_kd[j]; }
                } else if constexpr (JPL == 4) {
                    rs.draw4(HOT(slots.dr_kp) + d0, up[0], up[1], up[2], up[3]);
                    rs.draw4(HOT(slots.dr_kd) + d0, ud2[0], ud2[1], ud2[2], ud2[3]);
                } else {
                    rs.draw3(HOT(slots.dr_kp) + d0, up[0], up[1], up[2]);
                    rs.draw3(HOT(slots.dr_kd) + d0, ud2[0], ud2[1], ud2[2]);
                }
#pragma unroll
                for (int j = 0; j < JPL; j++) {
                    ld_kp[j] = HOT(dr_kp_span) * up[j] + HOT(dr_kp_lo);
                    ld_kd[j] = HOT(dr_kd_span) * ud2[j] + HOT(dr_kd_lo);
                    if (live) { B.kp_scale[e * A + d0 + j] = ld_kp[j]; B.kd_scale[e * A + d0 + j] = ld_kd[j]; }
                }
            }
            if (lead) {
                st3(B.base_pos + 3 * e, pos);
                B.base_quat[4 * e] = qx; B.base_quat[4 * e + 1] = qy; B.base_quat[4 * e + 2] = qz; B.base_quat[4 * e + 3] = qw;
                st3(B.base_lin_vel_w + 3 * e, vw); st3(B.base_ang_vel_w + 3 * e, ww);
                st3(B.base_lin_vel + 3 * e, blv); st3(B.base_ang_vel + 3 * e, bav);
                st3(B.projected_gravity + 3 * e, pg);
                st3(B.last_base_lin_vel + 3 * e, v3(0, 0, 0)); st3(B.last_base_ang_vel + 3 * e, v3(0, 0, 0));
            }
            if (RP ? leg == 0 : lead) {   // every replica's lead lane forms the values (shared bundle / injected row); replica 0 stores
                // domain randomisation (genesis_simulator.py:62-77, 665-739)
                if (HOT(dr_friction_on)) { ld_fric = HOT(dr_friction_span) * (bundle ? eu[3] : rs.draw(HOT(slots.dr_friction))) + HOT(dr_friction_lo); if (lead) B.friction_values[e] = ld_fric; }
                if (HOT(dr_mass_on)) { ld_mass = HOT(dr_mass_span) * (bundle ? eu[7] : rs.draw(HOT(slots.dr_mass))) + HOT(dr_mass_lo); if (lead) B.added_base_mass[e] = ld_mass; }
                if (HOT(dr_com_on)) {
                    float uc[3];
                    if (bundle) { uc[0] = eu[4]; uc[1] = eu[5]; uc[2] = eu[6]; }
                    else rs.draw3(HOT(slots.dr_com), uc[0], uc[1], uc[2]);
#pragma unroll
                    for (int k = 0; k < 3; k++) { ld_com[k] = HOT(dr_com_span[k]) * uc[k] + HOT(dr_com_lo[k]); if (lead) B.base_com_bias[3 * e + k] = ld_com[k]; }
                }
                if (HOT(dr_joint_on) && B.joint_armature) {
                    float uj[3];
                    if (bundle && EU2) { uj[0] = eu[EU2 ? 16 : 0]; uj[1] = eu[EU2 ? 17 : 0]; uj[2] = eu[EU2 ? 18 : 0]; }
                    else rs.draw3(HOT(slots.dr_joint), uj[0], uj[1], uj[2]);
                    ld_jnt[0] = HOT(dr_joint_span[0]) * uj[0] + HOT(dr_joint_lo[0]);
                    ld_jnt[1] = HOT(dr_joint_span[1]) * uj[1] + HOT(dr_joint_lo[1]);
                    ld_jnt[2] = HOT(dr_joint_span[2]) * uj[2] + HOT(dr_joint_lo[2]);
                    if (lead) { B.joint_armature[e] = ld_jnt[0]; B.joint_friction[e] = ld_jnt[1]; B.joint_damping[e] = ld_jnt[2]; }
                }
            }
            if (lead) {
                // extras["episode"] (legged_robot.py:128-132): snapshot this env's sums + the step it reset at; the
                // host forms the per-step means lazily from these.  (A first version used one float atomic per term
                // per reset on a shared accumulator: ~10 us per launch of same-line atomic latency.)
#pragma unroll
                for (int k = 0; k < LG_R_COUNT; k++) {
                    if (RON(k)) { B.episode_done_sums[(size_t)k * N + e] = es[k]; es[k] = 0.f; }
                }
                B.episode_done_step[e] = (int)p.counter;
                if (B.cstr_sums) {      // the constraint counters live in episode_sums in the reference and are logged / zeroed with them
#pragma unroll
                    for (int k = 0; k < LG_NUM_CSTR; k++) {
                        // POST in this launch: its counters are in registers; RESET alone: read them back
                        B.cstr_done_sums[(size_t)k * N + e] = (DO_POST && p.k.cat_enable) ? cat_cnt[k] : B.cstr_sums[(size_t)k * N + e];
                        B.cstr_sums[(size_t)k * N + e] = 0.f;
                    }
                }
            }
        }
    }
    if (DO_RESET) {
        STAMP(9);
        // ---- compute_observations + clip (legged_robot.py:48-49).  Layouts: go2.py:40-64 (45),
        //      go2_wtw.py:53-111 (61x5 | 99x5), go2_ee.py:10-75 (45x20 | 174x5 | 24 labels).  Histories are
        //      kept oldest -> newest inside obs_buf / priv_obs_buf themselves and shifted in place.
        const int FR = hc_obs_frame, PF = hc_priv_frame, ST = hc_obs_stack, PST = hc_priv_stack;
        const int SL = hc_obs_slack;
        // Two copies ("sets") of the observation buffers, written alternately (LgTaskCfg.obs_sets): what the previous
        // launch handed to the caller is not touched by this one.  `cs` = this launch's set, `xs` = the other one.
        const size_t orow = (size_t)(hc_num_obs + SL * FR), prow = (size_t)(hc_num_priv_obs + SL * PF);
        const bool two = hc_obs_sets > 1;
        const int cs = two ? p.obs_set : 0, xs = (two && cs < 2) ? 1 - cs : cs;   // the other copy (two sets); with more sets nothing is stacked and xs is unused
        float *const oset_c = B.obs_buf + (size_t)cs * N * orow, *const oset_x = B.obs_buf + (size_t)xs * N * orow;
        float *const pset_c = hc_num_priv_obs > 0 ? B.priv_obs_buf + (size_t)cs * N * prow : nullptr;
        float *const pset_x = hc_num_priv_obs > 0 ? B.priv_obs_buf + (size_t)xs * N * prow : nullptr;
        float *o = oset_c + (size_t)e * orow + (size_t)p.obs_win * FR;
        float *pv = pset_c ? pset_c + (size_t)e * prow + (size_t)p.obs_win * PF : nullptr;
        const float co = hc_clip_obs;
        if (SL > 0) {
            // sliding window: the previous frames are already where this window expects them; only an env that was just
            // reset blanks its history (go2_wtw.py:174-178).  All active lanes of the wave blank each such env together:
            // left to the env's own 2-4 lanes that is up to 750 stores per lane (tron1: 9 x 31 + 9 x 134 floats), and the
            // biped resets often.  With two sets an env reset at the PREVIOUS observation launch still carries its old
            // history in this set (that launch blanked the other one): everything older than the frame that launch
            // wrote goes now (`dirty`).
            const int nl = BLOCK, wl = (int)threadIdx.x & (BLOCK - 1);   // every lane of the wave (all replicas when FUSED)
            auto blank = [&](unsigned long long rm, int keep) {   // zero all but the newest `keep` frames of the window
                while (rm) {
                    const int bit = __builtin_ctzll(rm);
                    rm &= rm - 1;
                    const int er = (vtid - glane + bit) / LEGS;
                    float *orw = oset_c + (size_t)er * orow + (size_t)p.obs_win * FR;
                    for (int i = wl; i < (ST - keep) * FR; i += nl) orw[i] = 0.f;
                    if (pset_c) {
                        float *prw = pset_c + (size_t)er * prow + (size_t)p.obs_win * PF;
                        for (int i = wl; i < (PST - keep) * PF; i += nl) prw[i] = 0.f;
                    }
                }
            };
            blank(__builtin_amdgcn_ballot_w64(live && reset && leg == 0), 1);
            if (two) blank(__builtin_amdgcn_ballot_w64(live && !reset && dirty_prev != 0 && leg == 0), 2);
        } else if (live) {   // this lane's columns move one frame towards the past (zeros after a reset); with two sets the
                             // previous observation is read from the other one
            const float *so = oset_x + (size_t)e * orow;
            for (int f = 0; f + 1 < ST; f++)
                for (int i = leg; i < FR; i += LEGS) o[f * FR + i] = reset ? 0.f : so[(f + 1) * FR + i];
            if (pv) {
                const float *sp = pset_x + (size_t)e * prow;
                for (int f = 0; f + 1 < PST; f++)
                    for (int i = leg; i < PF; i += LEGS) pv[f * PF + i] = reset ? 0.f : sp[(f + 1) * PF + i];
            }
        }
        STAMP(25);
        // the frame written now also goes into the other set's row (same frame index): that set's window covers it at the
        // next launch
        float *on2 = (two && SL > 0 && ST > 1) ? oset_x + (size_t)e * orow + (size_t)p.obs_win * FR + (size_t)(ST - 1) * FR : nullptr;
        float *pn2 = (two && SL > 0 && PST > 1 && pset_x) ? pset_x + (size_t)e * prow + (size_t)p.obs_win * PF + (size_t)(PST - 1) * PF : nullptr;
        float *on = o + (ST - 1) * FR, *pn = pv ? pv + (PST - 1) * PF : nullptr;
        // wave-uniform copies of "is there a privileged frame / a second copy to write" (the pointers themselves are per lane:
        // a test on them costs a compare and an EXEC update per store)
        const bool has_pn = hc_num_priv_obs > 0, has_on2 = two && SL > 0 && ST > 1, has_pn2 = two && SL > 0 && PST > 1 && hc_num_priv_obs > 0;
        const bool nz = hc_add_noise != 0;
        const int ns = HOT(slots.noise);
        // uniforms for the noisy entries only (commands and actions carry zero noise scale): q, qd per lane,
        // gravity + ang vel on the lead lane
        float uq[JPL], uqd[JPL], uact[JPL], ub[6] = {0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f};
#pragma unroll
        for (int j = 0; j < JPL; j++) uq[j] = uqd[j] = uact[j] = 0.5f;
        if (nz) {
            if (rs.in) {
#pragma unroll
                for (int j = 0; j < JPL; j++) { uq[j] = rs.in[ns + 9 + d0 + j]; uqd[j] = rs.in[ns + 9 + A + d0 + j]; }
                if (lead) {
#pragma unroll
                    for (int k = 0; k < 6; k++) ub[k] = rs.in[ns + 3 + k];
                }
            } else {
                // two Philox blocks per lane, all four outputs used: the fourth ones of the env's lanes are the base's
                // six uniforms (a call is ~800 cycles and a lead-only call stalls the whole wave)
                float sp0 = 0.5f, sp1 = 0.5f;
                U4 N1 = {0u, 0u, 0u, 0u};
                if constexpr (RP) {   // one block per replica: 2 leg | 2 leg + 1 | 2 LEGS | 2 LEGS + 1, fetched where needed
                    const unsigned idn = sub == 0 ? 2u * leg : (sub == 1 ? 2u * leg + 1u : (sub == 2 ? 2u * LEGS : 2u * LEGS + 1u));
                    N1 = philox4x32_10(U4{rs.e_lo, rs.e_hi, rs.step, 0x80000000u + idn}, rs.k0, rs.k1);
#pragma unroll
                    for (int j = 0; j < JPL; j++) { uq[j] = u01(from(word(N1, j), 0)); uqd[j] = u01(from(word(N1, j), 1)); }
                    if constexpr (JPL == 3) { sp0 = u01(from(N1.w, 0)); sp1 = u01(from(N1.w, 1)); }
                } else if constexpr (JPL == 4) {   // all four outputs go to the joints; the base's six uniforms come from two more blocks
                    rs.block4(2 * leg, uq[0], uq[1], uq[2], uq[3]);
                    rs.block4(2 * leg + 1, uqd[0], uqd[1], uqd[2], uqd[3]);
                } else {
                    rs.block4(2 * leg, uq[0], uq[1], uq[2], sp0);
                    rs.block4(2 * leg + 1, uqd[0], uqd[1], uqd[2], sp1);
                }
                auto bcq = [](float v, int k) {   // lane k of the quad (the env's lanes), k compile-time after unrolling
                    const int x = __float_as_int(v);
                    return __int_as_float(k == 0 ? __builtin_amdgcn_update_dpp(0, x, 0x00, 0xF, 0xF, false)
                                          : (k == 1 ? __builtin_amdgcn_update_dpp(0, x, 0x55, 0xF, 0xF, false)
                                                    : __builtin_amdgcn_update_dpp(0, x, 0xAA, 0xF, 0xF, false)));
                };
                if (LEGS == 4) {
                    ub[0] = bcq(sp0, 0); ub[1] = bcq(sp1, 0); ub[2] = bcq(sp0, 1); ub[3] = bcq(sp1, 1); ub[4] = bcq(sp0, 2); ub[5] = bcq(sp1, 2);
                } else {   // two lanes per env: one more block, same id on both lanes of the pair (only the lead's copy is used)
                    if constexpr (RP) {
                        if constexpr (JPL == 4) { sp0 = u01(from(N1.x, 3)); sp1 = u01(from(N1.y, 3)); }
                        ub[0] = sp0; ub[1] = sp1;
#pragma unroll
                        for (int k = 0; k < 4; k++) ub[2 + k] = u01(from(word(N1, k), 2));
                    } else {
                        if constexpr (JPL == 4) { float d0_, d1_; rs.block4(2 * LEGS + 1, sp0, sp1, d0_, d1_); }
                        ub[0] = sp0; ub[1] = sp1;
                        rs.block4(2 * LEGS, ub[2], ub[3], ub[4], ub[5]);
                    }
                }
            }
        }
        float uclk[2] = {0.5f, 0.5f};
        if (nz && hc_noise_act0 != 0.f) {  // uniform index: scalar load   // tron1_pf_ee.py:338-342 (quirk 4): actions and clock are noisy too
            if (rs.in) {
#pragma unroll
                for (int j = 0; j < JPL; j++) uact[j] = rs.in[ns + 9 + 2 * A + d0 + j];
                uclk[0] = rs.in[ns + 9 + 3 * A + foot_slot]; uclk[1] = rs.in[ns + 9 + 3 * A + LEGS + foot_slot];
            } else {
                if constexpr (RP) {   // replica 0: the action block, the others: the clock block
                    const unsigned idn = sub == 0 ? 2u * LEGS + 2u + leg : 3u * LEGS + 2u + leg;
                    const U4 N2 = philox4x32_10(U4{rs.e_lo, rs.e_hi, rs.step, 0x80000000u + idn}, rs.k0, rs.k1);
#pragma unroll
                    for (int j = 0; j < JPL; j++) uact[j] = u01(from(word(N2, j), 0));
                    uclk[0] = u01(from(N2.x, 1)); uclk[1] = u01(from(N2.y, 1));
                } else {
                    float dummy;
                    if constexpr (JPL == 4) rs.block4(2 * LEGS + 2 + leg, uact[0], uact[1], uact[2], uact[3]);
                    else rs.block3(2 * LEGS + 2 + leg, uact[0], uact[1], uact[2]);
                    rs.block3(3 * LEGS + 2 + leg, uclk[0], uclk[1], dummy);
                }
            }
        }
        STAMP(26);
        // observation programs: where the noise-free actor frame sits inside the critic frame (0 in the fixed layouts, -1: not at
        // all) and where the "next state" copy goes in the labels row (-1: none)
        int pfo = 0, nxo = -1;
        float *labp = nullptr;
        if (hc_obs_layout == LG_OBS_PROGRAM) {
            pfo = -1;
            for (int s_ = 0; s_ < PRG_I(0, 0); s_++) if (PRG_I(0, 2 + s_) == LG_SEG_FRAME) pfo = PRG_I(0, 10 + s_);
            for (int s_ = 0; s_ < PRG_I(1, 0); s_++) if (PRG_I(1, 2 + s_) == LG_SEG_NEXT_STATE) nxo = PRG_I(1, 10 + s_);
            if (B.labels_buf) labp = B.labels_buf + ((size_t)cs * N + e) * hc_num_labels;
        }
        const float pclip = (hc_obs_layout == LG_OBS_PROGRAM && !PRG_I(0, 1)) ? 3.0e38f : co;
        // NV consecutive entries of the actor frame starting at idx0: noise + clip into this launch's window (and the other set's),
        // the noise-free copy into the critic frame (and the "next state" labels); one store instruction per destination (stv)
        auto putv = [&](auto nv_, const int idx0, const float *v, const float *u, const float *nscale) {
            constexpr int NV = decltype(nv_)::value;
            float cn[NV];
            if (has_pn && pfo >= 0) {   // critic copy of the frame is noise-free
#pragma unroll
                for (int k = 0; k < NV; k++) cn[k] = clampf(v[k], -pclip, pclip);
                stv<NV>(pn + pfo + idx0, cn);
                if (has_pn2) stv<NV>(pn2 + pfo + idx0, cn);
            }
            if (nxo >= 0) {             // go2_dreamwaq.py:66-74, not clipped; the action entries scaled
                const float sc = idx0 >= 9 + 2 * A ? HOT(o_action_scale) : 1.f;
#pragma unroll
                for (int k = 0; k < NV; k++) cn[k] = idx0 >= 9 + 2 * A ? v[k] * sc : v[k];
                stv<NV>(labp + nxo + idx0, cn);
            }
#pragma unroll
            for (int k = 0; k < NV; k++) {
                float x = v[k];
                if (nz) x += (2.f * u[k] - 1.f) * nscale[k];
                cn[k] = clampf(x, -co, co);
            }
            stv<NV>(on + idx0, cn);
            if (has_on2) stv<NV>(on2 + idx0, cn);
        };
        // NV consecutive entries of the critic frame (clipped like the reference's step() output)
        auto putpv = [&](auto nv_, const int idx0, const float *v) {
            constexpr int NV = decltype(nv_)::value;
            float cn[NV];
#pragma unroll
            for (int k = 0; k < NV; k++) cn[k] = clampf(v[k], -co, co);
            stv<NV>(pn + idx0, cn);
            if (has_pn2) stv<NV>(pn2 + idx0, cn);
        };
        auto put = [&](int idx, float v, float u, float nscale) { putv(std::integral_constant<int, 1>{}, idx, &v, &u, &nscale); };
        auto putp = [&](int idx, float v) { putpv(std::integral_constant<int, 1>{}, idx, &v); };
        constexpr std::integral_constant<int, JPL> NJ{};
        constexpr std::integral_constant<int, 3> N3{};
        {   // The actor frame: 3 x JPL joint entries per lane, 9 base entries on the lead lane.  Every value is formed FIRST, then
            // the stores go out back to back, destination by destination: a store's data registers stay locked until the store has
            // completed (the compiler waits on vmcnt before overwriting them), so computing the next group into registers the previous
            // group had just stored from cost a store round trip (~1 k cycles) per group.
            float vj[3][JPL], nj[3][JPL], fj[3][JPL], vl[9], nl[9], fl[9];
#pragma unroll
            for (int j = 0; j < JPL; j++) { vj[0][j] = (q[j] - q0l[j]) * hc_obs_scale_dof_pos; vj[1][j] = qd[j] * hc_obs_scale_dof_vel; vj[2][j] = act[j]; }
#pragma unroll
            for (int j = 0; j < JPL; j++) {
                const float un[3] = {uq[j], uqd[j], uact[j]}, sn[3] = {nv_q[j], nv_qd[j], nv_act[j]};
#pragma unroll
                for (int g = 0; g < 3; g++) {
                    fj[g][j] = clampf(vj[g][j], -pclip, pclip);
                    nj[g][j] = clampf(nz ? vj[g][j] + (2.f * un[g] - 1.f) * sn[g] : vj[g][j], -co, co);
                }
            }
            vl[0] = cmd0 * hc_obs_scale_lin_vel; vl[1] = cmd1 * hc_obs_scale_lin_vel; vl[2] = cmd2 * hc_obs_scale_ang_vel;
            vl[3] = pg.x; vl[4] = pg.y; vl[5] = pg.z;
            vl[6] = bav.x * hc_obs_scale_ang_vel; vl[7] = bav.y * hc_obs_scale_ang_vel; vl[8] = bav.z * hc_obs_scale_ang_vel;
#pragma unroll
            for (int k = 0; k < 9; k++) {
                fl[k] = clampf(vl[k], -pclip, pclip);
                const float x = (nz && k >= 3) ? vl[k] + (2.f * ub[k - 3] - 1.f) * HOT(noise_lead[k - 3]) : vl[k];
                nl[k] = clampf(x, -co, co);
            }
            const bool crit = has_pn && pfo >= 0;
            if constexpr (RP) {   // replica 0: this window, 1: the other set's, 2: the critic frame's copy, 3: the other set's critic copy
                float *const dst = sub == 0 ? on : (sub == 1 ? on2 : (sub == 2 ? pn + pfo : pn2 + pfo));
                const bool en = sub == 0 || (sub == 1 && has_on2) || (sub == 2 && crit) || (sub == 3 && crit && has_pn2);
                const bool noisy = sub < 2;
                if (live_all && en) {
#pragma unroll
                    for (int g = 0; g < 3; g++) {
                        float w[JPL];
#pragma unroll
                        for (int j = 0; j < JPL; j++) w[j] = noisy ? nj[g][j] : fj[g][j];
                        stv<JPL>(dst + 9 + g * A + d0, w);
                    }
                    if (leg == 0) {
                        float w[9];
#pragma unroll
                        for (int k = 0; k < 9; k++) w[k] = noisy ? nl[k] : fl[k];
                        stv<9>(dst, w);
                    }
                }
                if (nxo >= 0) {   // go2_dreamwaq.py:66-74: the "next state" labels, not clipped, actions scaled (replica 0)
                    float sa[JPL];
#pragma unroll
                    for (int j = 0; j < JPL; j++) sa[j] = vj[2][j] * HOT(o_action_scale);
                    if (live) { stv<JPL>(labp + nxo + 9 + d0, vj[0]); stv<JPL>(labp + nxo + 9 + A + d0, vj[1]); stv<JPL>(labp + nxo + 9 + 2 * A + d0, sa); }
                    if (lead) stv<9>(labp + nxo, vl);
                }
            } else {
            if (live) {
#pragma unroll
                for (int g = 0; g < 3; g++) stv<JPL>(on + 9 + g * A + d0, nj[g]);
                if (has_on2) {
#pragma unroll
                    for (int g = 0; g < 3; g++) stv<JPL>(on2 + 9 + g * A + d0, nj[g]);
                }
                if (crit) {   // critic copy of the frame is noise-free
#pragma unroll
                    for (int g = 0; g < 3; g++) stv<JPL>(pn + pfo + 9 + g * A + d0, fj[g]);
                    if (has_pn2) {
#pragma unroll
                        for (int g = 0; g < 3; g++) stv<JPL>(pn2 + pfo + 9 + g * A + d0, fj[g]);
                    }
                }
                if (nxo >= 0) {   // go2_dreamwaq.py:66-74: the "next state" labels, not clipped, actions scaled
                    float sa[JPL];
#pragma unroll
                    for (int j = 0; j < JPL; j++) sa[j] = vj[2][j] * HOT(o_action_scale);
                    stv<JPL>(labp + nxo + 9 + d0, vj[0]); stv<JPL>(labp + nxo + 9 + A + d0, vj[1]); stv<JPL>(labp + nxo + 9 + 2 * A + d0, sa);
                }
            }
            if (lead) {
                stv<9>(on, nl);
                if (has_on2) stv<9>(on2, nl);
                if (crit) { stv<9>(pn + pfo, fl); if (has_pn2) stv<9>(pn2 + pfo, fl); }
                if (nxo >= 0) stv<9>(labp + nxo, vl);
            }
            }
        }
        STAMP(27);
        if (hc_obs_layout == LG_OBS_GO2_WTW) {
            const float ang = 6.283185307179586f * (phi + theta);      // clock inputs (go2_wtw.py:251-256)
            if (live) {
                const float sn = sinf(ang), cs = cosf(ang);
                put(45 + foot_slot, sn, 0.5f, 0.f);
                put(49 + foot_slot, cs, 0.5f, 0.f);
                put(57 + foot_slot, theta, 0.5f, 0.f);
                putpv(NJ, FR + 10 + d0, ld_kp);
                putpv(NJ, FR + 10 + A + d0, ld_kd);
                putp(FR + 10 + 2 * A + foot_slot, expC);
                ts[6 + foot_slot] = theta; ts[10 + foot_slot] = sn; ts[14 + foot_slot] = cs; ts[18 + foot_slot] = expC;
            }
            if (lead) {
                const float vb[4] = {gait_period, bh_tgt, fc_tgt, pitch_tgt}, hb[4] = {0.5f, 0.5f, 0.5f, 0.5f}, zb[4] = {0.f, 0.f, 0.f, 0.f};
                putv(std::integral_constant<int, 4>{}, 53, vb, hb, zb);
                const float vp[10] = {blv.x * hc_obs_scale_lin_vel, blv.y * hc_obs_scale_lin_vel, blv.z * hc_obs_scale_lin_vel, ld_push[0], ld_push[1],
                                      ld_mass, ld_fric, ld_com[0], ld_com[1], ld_com[2]};
                putpv(std::integral_constant<int, 10>{}, FR, vp);
            }
        } else if (hc_obs_layout == LG_OBS_GO2_EE) {
            // critic frame (go2_ee.py:21-48): obs 45 | DR 31 | contact states K | heights P
            const int K = __popc(M->state_link_mask);
            const int l0 = foot_link - 3;
            float *lab = B.labels_buf + ((size_t)cs * N + e) * hc_num_labels;
            // FUSED: the critic entries are dealt over the replicas -- replica r writes the copy r & 1 (this set's / the other set's
            // critic frame); the heights, the long block, are also halved by r >> 1.  The labels row belongs to replica 0.
            const bool lw = RP ? live_all && ((sub & 1) == 0 || has_pn2) : live;
            const bool lw01 = RP ? lw && sub < 2 : live;
            float *const pd = (RP && (sub & 1)) ? pn2 : pn;
            auto wp = [&](int idx, float v) { const float c = clampf(v, -co, co); pd[idx] = c; if (!RP && has_pn2) pn2[idx] = c; };
            auto wpv = [&](auto nv_, const int idx0, const float *v) {
                constexpr int NV = decltype(nv_)::value;
                float cn[NV];
#pragma unroll
                for (int k = 0; k < NV; k++) cn[k] = clampf(v[k], -co, co);
                stv<NV>(pd + idx0, cn);
                if (!RP && has_pn2) stv<NV>(pn2 + idx0, cn);
            };
            if (lw01) {
                float vkp[JPL], vkd[JPL];
#pragma unroll
                for (int j = 0; j < JPL; j++) { vkp[j] = ld_kp[j] - HOT(kp_offset); vkd[j] = ld_kd[j] - HOT(kd_offset); }
                wpv(NJ, FR + 7 + d0, vkp);
                wpv(NJ, FR + 7 + A + d0, vkd);
                // contact states are those of the physics read-back (stale for a just-reset env, as in the reference)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int l = l0 + k;
                    if ((M->state_link_mask >> l) & 1u) {
                        const int idx = __popc(M->state_link_mask & ((1u << l) - 1u));
                        const float cs = norm(f_link[k]) > 1.f ? 1.f : 0.f;
                        wp(FR + 7 + 2 * A + idx, cs);
                        if (live) lab[3 + idx] = cs;
                    }
                }
            }
            if (lw) {
                if (hreg) {
#pragma unroll
                    for (int i = 0; i < HMAX; i++) {
                        const int k = leg + i * LEGS;
                        float hv = pos.z - hc_heights_offset - hts[i];
                        if (hc_heights_clip_scale) hv = clampf(hv, -1.f, 1.f) * hc_obs_scale_height;
                        if (k < P && (!RP || (i & 1) == (sub >> 1))) wp(FR + 7 + 2 * A + K + k, hv);
                    }
                } else if (!RP || sub < 2) {
                    for (int k = leg; k < P; k += LEGS) {
                        float hv = pos.z - hc_heights_offset - B.measured_heights[(size_t)e * P + k];
                        if (hc_heights_clip_scale) hv = clampf(hv, -1.f, 1.f) * hc_obs_scale_height;
                        wp(FR + 7 + 2 * A + K + k, hv);
                    }
                }
            }
            // labels (go2_ee.py:69-75): v_b 3 | contact states K | foot height above the local terrain mean F
            if (live) lab[3 + K + foot_slot] = clampf(foot_p.z - foot_hmean - hc_foot_height_offset, -1.f, 1.f);
            if (lw01 && leg == 0) {
                const float vdr[7] = {ld_fric - HOT(friction_offset), ld_mass, ld_com[0], ld_com[1], ld_com[2], ld_push[0], ld_push[1]};
                wpv(std::integral_constant<int, 7>{}, FR, vdr);
                if (M->state_link_mask & 1u) { const float cs = norm(f_base) > 1.f ? 1.f : 0.f; wp(FR + 7 + 2 * A, cs); if (lead) lab[3] = cs; }
            }
            if (lead) {
                const float vl[3] = {blv.x * hc_obs_scale_lin_vel, blv.y * hc_obs_scale_lin_vel, blv.z * hc_obs_scale_lin_vel};
                stv<3>(lab, vl);
            }
        } else if (hc_obs_layout == LG_OBS_TRON1_EE) {
            // tron1_pf_ee.py:53-141.  actor frame: 9 + 3A + clock 2F.  critic frame: frame | DR (7 + 2A + 3) |
            // gait F | contact states K | heights P | normals 3F | clip(foot_z - h9) 9F.  labels: v_b 3 | K | F | 3F
            const int K = __popc(M->state_link_mask);
            const int l0 = foot_link - 3;
            float *lab = B.labels_buf + ((size_t)cs * N + e) * hc_num_labels;
            const float ang = 6.283185307179586f * (phi + theta);
            const int oDR = FR, oG = FR + 7 + 2 * A + 3, oK = oG + F, oH = oK + K, oN = oH + P, oR = oN + 3 * F;
            // Replicated launch: the critic entries are dealt over the replicas -- replica r writes the copy r & 1 (this set's / the other
            // set's critic frame); r >> 1 halves the heights and splits the other blocks (0: gains, gait, contact states, DR; 1: normals,
            // foot clearances).  The actor-frame clock entries, the task state and the labels row belong to replica 0.
            const bool lwc = RP ? live_all && ((sub & 1) == 0 || has_pn2) : live;
            const bool gA = !RP || (sub >> 1) == 0, gB = !RP || (sub >> 1) == 1;
            float *const pd = (RP && (sub & 1)) ? pn2 : pn;
            auto wpv = [&](auto nv_, const int idx0, const float *v) {
                constexpr int NV = decltype(nv_)::value;
                float cn[NV];
#pragma unroll
                for (int k = 0; k < NV; k++) cn[k] = clampf(v[k], -co, co);
                stv<NV>(pd + idx0, cn);
                if (!RP && has_pn2) stv<NV>(pn2 + idx0, cn);
            };
            auto wp = [&](int idx, float v) { wpv(std::integral_constant<int, 1>{}, idx, &v); };
            if (live) {
                const float sn = sinf(ang), cs = cosf(ang);
                put(9 + 3 * A + foot_slot, sn, uclk[0], nv_clk[0]);
                put(9 + 3 * A + F + foot_slot, cs, uclk[1], nv_clk[1]);
                ts[4 + foot_slot] = theta; ts[6 + foot_slot] = sn; ts[6 + F + foot_slot] = cs; ts[10 + foot_slot] = expC;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int l = l0 + k;
                    if ((M->state_link_mask >> l) & 1u) lab[3 + __popc(M->state_link_mask & ((1u << l) - 1u))] = norm(f_link[k]) > 1.f ? 1.f : 0.f;
                }
                stv<3>(lab + 3 + K + F + 3 * foot_slot, ld_nv3);
                lab[3 + K + foot_slot] = clampf(foot_p.z - foot_hmax - hc_foot_height_offset, -1.f, 1.f);
            }
            if (lwc && gA) {
                float vkp[JPL], vkd[JPL];
#pragma unroll
                for (int j = 0; j < JPL; j++) { vkp[j] = ld_kp[j] - HOT(kp_offset); vkd[j] = ld_kd[j] - HOT(kd_offset); }
                wpv(NJ, oDR + 7 + d0, vkp);
                wpv(NJ, oDR + 7 + A + d0, vkd);
                wp(oG + foot_slot, expC);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int l = l0 + k;
                    if ((M->state_link_mask >> l) & 1u) wp(oK + __popc(M->state_link_mask & ((1u << l) - 1u)), norm(f_link[k]) > 1.f ? 1.f : 0.f);
                }
            }
            if (lwc) {
                if (hreg) {
#pragma unroll
                    for (int i = 0; i < HMAX; i++) {
                        const int k = leg + i * LEGS;
                        float hv = pos.z - hc_heights_offset - hts[i];
                        if (hc_heights_clip_scale) hv = clampf(hv, -1.f, 1.f) * hc_obs_scale_height;
                        if (k < P && (!RP || (i & 1) == (sub >> 1))) wp(oH + k, hv);
                    }
                } else {
                    for (int k = leg + (RP ? (sub >> 1) * LEGS : 0); k < P; k += RP ? 2 * LEGS : LEGS) {
                        float hv = pos.z - hc_heights_offset - B.measured_heights[(size_t)e * P + k];
                        if (hc_heights_clip_scale) hv = clampf(hv, -1.f, 1.f) * hc_obs_scale_height;
                        wp(oH + k, hv);
                    }
                }
            }
            if (lwc && gB) {
                wpv(N3, oN + 3 * foot_slot, ld_nv3);
                float vrel[9];
#pragma unroll
                for (int k = 0; k < 9; k++) vrel[k] = clampf(foot_p.z - ld_haf[k], -1.f, 1.f);
                wpv(std::integral_constant<int, 9>{}, oR + 9 * foot_slot, vrel);
            }
            if (lwc && gA && leg == 0) {
                const float vdr[7] = {ld_fric - HOT(friction_offset), ld_mass, ld_com[0], ld_com[1], ld_com[2], ld_push[0], ld_push[1]};
                wpv(std::integral_constant<int, 7>{}, oDR, vdr);
                wpv(N3, oDR + 7 + 2 * A, ld_jnt);
                if (M->state_link_mask & 1u) wp(oK, norm(f_base) > 1.f ? 1.f : 0.f);
            }
            if (lead) {
                if (M->state_link_mask & 1u) lab[3] = norm(f_base) > 1.f ? 1.f : 0.f;
                const float vl[3] = {blv.x * hc_obs_scale_lin_vel, blv.y * hc_obs_scale_lin_vel, blv.z * hc_obs_scale_lin_vel};
                stv<3>(lab, vl);
            }
        } else if (hc_obs_layout == LG_OBS_PROGRAM) {
            // go2_ts / go2_cts / go2_dreamwaq / go2_cat: actor frame = go2's 45 (written above); the critic frame and the
            // auxiliary output are concatenations of the blocks of LgObsSeg at the offsets the task's programs give
            const int K = __popc(M->state_link_mask);
            const int l0 = foot_link - 3;
            // block values that do not depend on the program, formed once (each HOT() is an LDS round trip)
            const float vdr[7] = {ld_fric - HOT(friction_offset), ld_mass, ld_com[0], ld_com[1], ld_com[2], ld_push[0], ld_push[1]};
            float vkp[JPL], vkd[JPL];
            {
                const float kpo = HOT(kp_offset), kdo = HOT(kd_offset);
#pragma unroll
                for (int j = 0; j < JPL; j++) { vkp[j] = ld_kp[j] - kpo; vkd[j] = ld_kd[j] - kdo; }
            }
            const unsigned slm = M->state_link_mask;
            // `which`: 0 = critic frame (priv_prog), 1 = auxiliary row (labels_prog); the program words come from LDS (PRG_I / PRG_F)
            auto run = [&](const int which, const bool to_lab) {
                const float cl = PRG_I(which, 1) ? co : 3.0e38f;
                auto WrV = [&](auto nv_, const int idx0, const float *v) {     // NV consecutive entries, one store per destination
                    constexpr int NV = decltype(nv_)::value;
                    float cn[NV];
#pragma unroll
                    for (int k = 0; k < NV; k++) cn[k] = clampf(v[k], -cl, cl);
                    if (to_lab) stv<NV>(labp + idx0, cn);
                    else if (RP) stv<NV>((sub == 0 ? pn : pn2) + idx0, cn);      // replica 0 / 1: this set's / the other set's critic frame
                    else { stv<NV>(pn + idx0, cn); if (has_pn2) stv<NV>(pn2 + idx0, cn); }
                };
                // who writes: the labels row replica 0; the critic frame replicas 0 and (second copy) 1; the heights block, the long one, is
                // halved once more: replica r writes copy r & 1, entries with (i & 1) == r >> 1
                const bool live = to_lab ? live_all && sub == 0 : (RP ? live_all && (sub == 0 || (sub == 1 && has_pn2)) : live_all);
                const bool lead = live && leg == 0;
                const bool liveh = (RP && !to_lab) ? live_all && ((sub & 1) == 0 || has_pn2) : live;
                auto Wr = [&](int idx, float v) { WrV(std::integral_constant<int, 1>{}, idx, &v); };
                constexpr std::integral_constant<int, 7> N7{};
                constexpr std::integral_constant<int, 9> N9{};
                const int n_segs = PRG_I(which, 0);
                for (int s_ = 0; s_ < n_segs; s_++) {
                    const int kind = PRG_I(which, 2 + s_), off = PRG_I(which, 10 + s_);
                    const float sc = PRG_F(which, 18 + s_);
                    if (kind == LG_SEG_DR || kind == LG_SEG_KP || kind == LG_SEG_KD) {
                        if (live) {
                            if (kind == LG_SEG_DR) { WrV(NJ, off + 7 + d0, vkp); WrV(NJ, off + 7 + A + d0, vkd); }
                            else {
                                float vs[JPL];
#pragma unroll
                                for (int j = 0; j < JPL; j++) vs[j] = kind == LG_SEG_KP ? vkp[j] : vkd[j];
                                WrV(NJ, off + d0, vs);
                            }
                        }
                        if (lead && kind == LG_SEG_DR) WrV(N7, off, vdr);
                    } else if (kind == LG_SEG_DR_JOINT) {
                        if (lead) WrV(N3, off, ld_jnt);
                    } else if (kind == LG_SEG_BASE_LIN_VEL) {
                        const float vb[3] = {blv.x * hc_obs_scale_lin_vel * sc, blv.y * hc_obs_scale_lin_vel * sc, blv.z * hc_obs_scale_lin_vel * sc};
                        if (lead) WrV(N3, off, vb);
                    } else if (kind == LG_SEG_CONTACT_STATES) {
                        // contact states are those of the physics read-back (stale for a just-reset env, as in the reference)
                        if (live) {
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                const int l = l0 + k;
                                if ((slm >> l) & 1u) Wr(off + __popc(slm & ((1u << l) - 1u)), norm(f_link[k]) > 1.f ? 1.f : 0.f);
                            }
                        }
                        if (lead && (slm & 1u)) Wr(off, norm(f_base) > 1.f ? 1.f : 0.f);
                    } else if (kind == LG_SEG_HEIGHTS) {
                        if (hreg && liveh) {
                            float *const hd = to_lab ? labp : ((RP && (sub & 1)) ? pn2 : pn);
#pragma unroll
                            for (int i = 0; i < HMAX; i++) {
                                const int k = leg + i * LEGS;
                                float hv = pos.z - hc_heights_offset - hts[i];
                                if (hc_heights_clip_scale) hv = clampf(hv, -1.f, 1.f) * hc_obs_scale_height;
                                hv = clampf(hv, -cl, cl);
                                if (k < P && (!RP || to_lab || (i & 1) == (sub >> 1))) { hd[off + k] = hv; if (!RP && !to_lab && has_pn2) pn2[off + k] = hv; }
                            }
                        }
                        if (!hreg && live) {
                            {
                                for (int k = leg; k < P; k += LEGS) {
                                    float hv = pos.z - hc_heights_offset - B.measured_heights[(size_t)e * P + k];
                                    if (hc_heights_clip_scale) hv = clampf(hv, -1.f, 1.f) * hc_obs_scale_height;
                                    Wr(off + k, hv);
                                }
                            }
                        }
                    } else if (kind == LG_SEG_FEET_REL_HEIGHTS || kind == LG_SEG_FEET_HEIGHTS) {
                        if (live) {
                            float vh[9];
#pragma unroll
                            for (int k = 0; k < 9; k++) vh[k] = kind == LG_SEG_FEET_HEIGHTS ? ld_haf[k] : clampf(foot_p.z - ld_haf[k], -1.f, 1.f);
                            WrV(N9, off + 9 * foot_slot, vh);
                        }
                    } else if (kind == LG_SEG_FEET_NORMALS) {
                        if (live) WrV(N3, off + 3 * foot_slot, ld_nv3);
                    } else if (kind == LG_SEG_LAST_ACTIONS) {     // a_{t-1}: zero for an env that was just reset
                        if (live) WrV(NJ, off + d0, last_act);
                    } else if (kind == LG_SEG_DR_BASE) {
                        if (lead) WrV(N7, off, vdr);
                    } else if (kind == LG_SEG_FEET_AIR_TIME) {
                        if (live) Wr(off + foot_slot, air);
                    } else if (kind == LG_SEG_FOOT_CLEARANCE) {
                        if (live) Wr(off + foot_slot, clampf(foot_p.z - foot_hmean - hc_foot_height_offset, -1.f, 1.f));
                    }   // LG_SEG_FRAME / LG_SEG_NEXT_STATE: written group by group in putv()
                }
            };
            if (pn) run(0, false);
            if (labp) run(1, true);
        }
    }
    if (WTW && lead) {
        ts[0] = gait_time; ts[1] = phi; ts[2] = gait_period; ts[3] = bh_tgt; ts[4] = fc_tgt; ts[5] = pitch_tgt;
    }
    if (BIPED && lead) { ts[0] = gait_time; ts[1] = phi; ts[2] = gait_period; }
    if (BIPED && live && !DO_RESET) { ts[4 + foot_slot] = theta; ts[10 + foot_slot] = expC; }
    if (WTW && live && !(DO_RESET && hc_obs_layout == LG_OBS_GO2_WTW)) { ts[6 + foot_slot] = theta; ts[18 + foot_slot] = expC; }
    // second action-history shift of the wtw / tron1_ee tasks (go2_wtw.py:45-46): afterwards
    // last == llast == a_t, which makes action_smoothness == action_rate
    if (DO_RESET && hc_double_shift && live) {
#pragma unroll
        for (int j = 0; j < JPL; j++) {
            B.llast_actions[e * A + d0 + j] = reset ? 0.f : last_act[j];
            B.last_actions[e * A + d0 + j] = act[j];
        }
    }
    STAMP(10);
    // ---- persistent MDP state ----
    if (live) {
        B.feet_air_time[e * F + foot_slot] = air;
        B.last_contacts[e * F + foot_slot] = (uint8_t)last_contact;
    }
    if (lead) {
        B.commands[4 * e] = cmd0; B.commands[4 * e + 1] = cmd1; B.commands[4 * e + 2] = cmd2; B.commands[4 * e + 3] = cmd3;
        B.episode_length_buf[e] = ep_len;
        B.fail_buf[e] = fail_buf;
        B.reset_buf[e] = reset ? 1 : 0;
        if (DO_RESET && B.obs_dirty) B.obs_dirty[e] = reset ? 1 : 0;
        B.time_out_buf[e] = time_out ? 1 : 0;
        B.rew_buf[e] = total;
        if (DO_POST || (DO_RESET && reset)) {
#pragma unroll
            for (int k = 0; k < LG_R_COUNT; k++)
                if (RON(k)) esum[(size_t)k * N + e] = es[k];
        }
    }
    STAMP(11);
}

// XCD-aware workgroup index: blocks b, b + 8, b + 16 ... are dealt to the same XCD (its own 4 MB L2), so THEY get consecutive env groups.
// With the plain index a wave's 16-byte piece of a (term, N) row or its 48 bytes of an (N, 3) array shares its 128-byte line with waves on
// the seven other XCDs: every L2 fetched (and wrote back) the whole line.  Bijective for any grid size.
LG_DEV int lg_wg() {
    const int b = (int)blockIdx.x, G = (int)gridDim.x, q = G >> 3, r = G & 7, x = b & 7;
    return x * q + (x < r ? x : r) + (b >> 3);
}

// REPL (MDP phases only, small batches): a wave carries 16 leg-lanes in four replicas like the fused tail of quad_sim_kernel -- every
// replica computes the same values, replica 0 owns the state stores, the observation stores are dealt over the replicas -- so that
// 4096 biped envs are 512 waves with a quarter of the stores each instead of 128 waves on 1024 SIMDs.
template <int LEGS, unsigned PH, int PROF = 0, int JPL = 3, bool REPL = false>
__global__ __launch_bounds__(BLOCK) void env_step_kernel(KParams p) {
    __shared__ __attribute__((aligned(16))) uint4 sMraw[MODEL_STG * BLOCK];
    __shared__ int sHot[256 + 2 * BLOCK];
    static_assert(!REPL || (PH & (LG_PHASE_PRE | LG_PHASE_SIM)) == 0, "replicated launch: MDP phases only");
    const int vtid = REPL ? (int)(lg_wg() * 16 + (threadIdx.x & 15)) : (int)(lg_wg() * BLOCK + threadIdx.x);
    env_step_body<LEGS, PH, false, PROF, JPL, REPL>(p, sMraw, sHot, nullptr, nullptr, vtid, threadIdx.x & 63);
}

