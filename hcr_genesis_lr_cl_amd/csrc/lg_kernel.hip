// lg_kernel.hip -- the env-step kernel for gfx950 (MI355X) and its C ABI (include/lgsim.h).
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
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include "../../include/lgsim.h"
#include "lg_math.h"

#define LG_ABI_VERSION 1
#define BLOCK 64

struct KParams {
    const LgModelDesc *M;
    const LgSimOptions *O;
    const LgTaskCfg *T;
    const int16_t *hf;
    LgBuffers B;
    const float *actions;
    long long counter;
};

template <int LEGS> struct LegCtx;

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
//   fspat: spatial force [r x f; f] on the distal body; tl[3]: joint torques.
// Up-sweep to the base contribution dp0 (to be quad-summed by the caller), then after the
// base solve the down-sweep gives joint accelerations and the distal body's acceleration.
LG_DEV V6 resp_up(const Joint (&J)[3], const V6 &fspat, const float (&tl)[3], float (&du)[3]) {
    V6 dp = {-fspat.a, -fspat.l};
#pragma unroll
    for (int j = 2; j >= 0; j--) {
        du[j] = tl[j] - dot(J[j].S, dp);
        dp = dp + J[j].U * (du[j] * J[j].dinv);
    }
    return dp;
}
LG_DEV V6 resp_down(const Joint (&J)[3], const V6 &a0, const float (&du)[3], float (&dqdd)[3]) {
    V6 a = a0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        dqdd[j] = (du[j] - dot(J[j].U, a)) * J[j].dinv;
        a = a + J[j].S * dqdd[j];
    }
    return a;
}

struct RandSrc {
    const float *in;  // injected row or nullptr
    unsigned k0, k1, e_lo, e_hi, step;
    LG_DEV float draw(int slot) const {
        if (in) return in[slot];
        U4 c = {e_lo, e_hi, step, (unsigned)(slot >> 2)};
        U4 r = philox4x32_10(c, k0, k1);
        unsigned v = (slot & 3) == 0 ? r.x : ((slot & 3) == 1 ? r.y : ((slot & 3) == 2 ? r.z : r.w));
        return u01(v);
    }
    // a whole Philox block (4 uniforms) from a counter space disjoint from the slot space; used
    // where a lane needs several draws per step (observation noise)
    LG_DEV void block3(int id, float &a, float &b, float &c) const {
        U4 ctr = {e_lo, e_hi, step, 0x80000000u + (unsigned)id};
        U4 r = philox4x32_10(ctr, k0, k1);
        a = u01(r.x); b = u01(r.y); c = u01(r.z);
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

// ---------------------------------------------------------------------------------------------
template <int LEGS, unsigned PH>
__global__ __launch_bounds__(BLOCK) void env_step_kernel(KParams p) {
    constexpr bool DO_PRE = (PH & LG_PHASE_PRE) != 0, DO_SIM = (PH & LG_PHASE_SIM) != 0;
    constexpr bool DO_POST = (PH & LG_PHASE_POST) != 0, DO_RESET = (PH & LG_PHASE_RESET) != 0;
    constexpr int A = LEGS * 3;
    // model table -> LDS once per workgroup: per-lane (leg-indexed) reads then cost an LDS access
    // instead of an L2 round trip with a single wave per SIMD to hide it
    __shared__ LgModelDesc sM;
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(p.M);
        uint32_t *dst = reinterpret_cast<uint32_t *>(&sM);
        for (int i = threadIdx.x; i < (int)(sizeof(LgModelDesc) / 4); i += BLOCK) dst[i] = src[i];
    }
    __syncthreads();
    const LgModelDesc *M = &sM;
    const LgSimOptions *__restrict__ O = p.O;
    const LgTaskCfg *__restrict__ T = p.T;
    const LgBuffers &B = p.B;

    const int tid = blockIdx.x * BLOCK + threadIdx.x;
    const int leg = tid % LEGS;
    int e = tid / LEGS;
    const bool live = e < B.n_envs;
    if (!live) e = B.n_envs - 1;  // dead lanes shadow the last env (keeps DPP quads uniform), never store
    const bool lead = live && leg == 0;
    const int L = M->n_links, F = LEGS;
    const int b0 = 1 + 3 * leg;            // first body of this lane's chain
    const int d0 = 3 * leg;                // first dof
    const int foot_link = M->foot_link[leg];
    int foot_slot = 0;                     // rank of this foot among feet in link order (feet_indices)
#pragma unroll
    for (int k = 0; k < LEGS; k++) foot_slot += (M->foot_link[k] < foot_link) ? 1 : 0;

    // ---------------- PRE: clip + action history (legged_robot.py:230-239) -----------------
    float act[3], last_act[3], llast_act[3];
    if (DO_PRE) {
        const float ca = T->clip_actions;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            float prev = B.actions[e * A + d0 + j], prev2 = B.last_actions[e * A + d0 + j];
            act[j] = clampf(p.actions[e * A + d0 + j], -ca, ca);
            last_act[j] = prev;
            llast_act[j] = prev2;
            if (live) {
                B.llast_actions[e * A + d0 + j] = prev2;
                B.last_actions[e * A + d0 + j] = prev;
                B.actions[e * A + d0 + j] = act[j];
            }
        }
    } else if (DO_SIM && !DO_POST) {  // Simulator.step(actions): actions come pre-clipped from the env
#pragma unroll
        for (int j = 0; j < 3; j++) { act[j] = p.actions[e * A + d0 + j]; last_act[j] = llast_act[j] = 0.f; }
    } else {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            act[j] = B.actions[e * A + d0 + j];
            last_act[j] = B.last_actions[e * A + d0 + j];
            llast_act[j] = B.llast_actions[e * A + d0 + j];
        }
    }

    // ---------------- state ----------------------------------------------------------------
    V3 pos = ld3(B.base_pos + 3 * e), vw = ld3(B.base_lin_vel_w + 3 * e), ww = ld3(B.base_ang_vel_w + 3 * e);
    float qx = B.base_quat[4 * e], qy = B.base_quat[4 * e + 1], qz = B.base_quat[4 * e + 2], qw = B.base_quat[4 * e + 3];
    float q[3], qd[3];
#pragma unroll
    for (int j = 0; j < 3; j++) { q[j] = B.dof_pos[e * A + d0 + j]; qd[j] = B.dof_vel[e * A + d0 + j]; }

    // read-back quantities handed from SIM to POST/RESET (registers when fused, HBM otherwise)
    V3 blv, bav, pg, eul;            // body-frame lin/ang vel, projected gravity, euler
    float last_qd[3], torque[3];
    V3 f_link[4];                    // net contact force on this leg's hip, thigh, calf, foot (world)
    V3 f_base = v3(0, 0, 0);         // net contact force on the base (whole env)
    V3 foot_p, foot_v, last_foot_v = v3(0, 0, 0);
    float foot_hmean = 0.f, foot_hmax = 0.f;  // mean / max of the 9 terrain heights around this lane's foot (a8)
    float mean_height = 0.f;         // mean over the height-sample grid of (base_z - h) is formed from this (a7)
    const int P = O->n_height_points;

    if (DO_SIM) {
        // "last" snapshots (genesis_simulator.py:21-24)
#pragma unroll
        for (int j = 0; j < 3; j++) last_qd[j] = qd[j];
        if (live) {
#pragma unroll
            for (int j = 0; j < 3; j++) B.last_dof_vel[e * A + d0 + j] = qd[j];
            last_foot_v = ld3(B.feet_vel + (e * F + foot_slot) * 3);
            st3(B.last_feet_vel + (e * F + foot_slot) * 3, last_foot_v);
            if (lead) {
                st3(B.last_base_lin_vel + 3 * e, ld3(B.base_lin_vel + 3 * e));
                st3(B.last_base_ang_vel + 3 * e, ld3(B.base_ang_vel + 3 * e));
            }
        }
        // per-env dynamics parameters (genesis_simulator.py:665-739)
        const float mass0 = M->mass[0] + (B.added_base_mass ? B.added_base_mass[e] : 0.f);
        V3 com0 = ld3(M->com[0]);
        if (B.base_com_bias) com0 += ld3(B.base_com_bias + 3 * e);
        const float mu = O->terrain_friction * (B.friction_values ? B.friction_values[e] : 1.f);
        float kps[3], kds[3], arm[3], jdamp[3], jfric[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            kps[j] = (B.kp_scale ? B.kp_scale[e * A + d0 + j] : 1.f) * O->kp[d0 + j];
            kds[j] = (B.kd_scale ? B.kd_scale[e * A + d0 + j] : 1.f) * O->kd[d0 + j];
            arm[j] = B.joint_armature ? B.joint_armature[e] : M->armature[d0 + j];
            jdamp[j] = B.joint_damping ? B.joint_damping[e] : M->damping[d0 + j];
            jfric[j] = B.joint_friction ? B.joint_friction[e] : M->frictionloss[d0 + j];
        }
        const float dt = O->dt, kc = O->contact_k, kappa = kc * dt + O->contact_b, margin = O->contact_margin;
        const float kl = O->limit_k, kapl = kl * dt + O->limit_b;
        const V3 grav = v3(0.f, 0.f, O->gravity_z);
        const S3 I0 = {M->inertia[0][0], M->inertia[0][1], M->inertia[0][2], M->inertia[0][3], M->inertia[0][4], M->inertia[0][5]};
        const int fs = M->foot_sphere[leg];
        const V3 foot_c_loc = ld3(M->sph_pos[fs]);
        const float foot_r = M->sph_r[fs];

        bool jrot_identity = true;
        for (int b = 1; b < M->n_bodies; b++)
            jrot_identity = jrot_identity && M->jrot[b][0] == 1.f && M->jrot[b][4] == 1.f && M->jrot[b][8] == 1.f;

        for (int sub = 0; sub < O->decimation; sub++) {
            const M3 Rb = quat_to_mat(qx, qy, qz, qw);
            // ---- forward kinematics + velocities of the chain (root -> leaf) -----------------
            asm volatile("" ::: "memory");  // keep model-table loads inside the sub-step (register pressure)
            BodyKin K[3];
            Joint J[3];
#pragma unroll
            for (int j = 0; j < 3; j++) f_link[j] = v3(0, 0, 0);
            f_link[3] = v3(0, 0, 0);
            V3 fb_acc = v3(0, 0, 0);       // this lane's share of base-sphere forces
            V6 pbase_ext = {v3(0, 0, 0), v3(0, 0, 0)};  // spatial force of that share about O
            {
                M3 Rp = Rb;
                V3 Pp = v3(0, 0, 0);
                V6 Vp = {ww, vw};
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const int b = b0 + j;
                    const V3 jp = ld3(M->jpos[b]), ax = ld3(M->axis[b]);
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
            auto sphere_contact = [&](int s, const M3 &R, V3 P, const V6 &V, V3 &fsum, V6 &pacc) {
                const V3 r = P + mul(R, ld3(M->sph_pos[s]));
                const float rad = M->sph_r[s];
                float h; V3 n;
                terrain_at(O, p.hf, pos.x + r.x, pos.y + r.y, h, n);
                const float depth = rad - (pos.z + r.z - h) * n.z;
                if (depth > -margin) {
                    const V3 v = V.l + cross(V.a, r);
                    const float vn = dot(v, n), wi = M->sph_w[s];
                    const float fn = (kc * depth - kappa * vn) * rcp(1.f + kappa * dt * wi);
                    if (fn > 0.f) {
                        const V3 vt = v - n * vn;
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
            for (int s = M->body_sph_start[0] + leg; s < M->body_sph_start[1]; s += LEGS)
                sphere_contact(s, Rb, v3(0, 0, 0), V0, fb_acc, pbase_ext);

            // ---- actuation (genesis_simulator.py:630-642): PD, torque reported unclipped ----
            float tau[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const float t = kps[j] * (act[j] * O->action_scale + O->default_dof_pos[d0 + j] - q[j]) - kds[j] * qd[j];
                torque[j] = t;
                const float lim = M->effort[d0 + j];
                tau[j] = clampf(t, -lim, lim) - jdamp[j] * qd[j] - jfric[j] * clampf(qd[j] * 20.f, -1.f, 1.f);
            }
            // ---- ABA pass 2 up the chain (leaf -> root): rigid inertia + bias force of each body
            //      are formed right where they are consumed ------------------------------------
            I6 IAacc;
            V6 pacc;
#pragma unroll
            for (int j = 2; j >= 0; j--) {
                const int b = b0 + j;
                {
                    const float m = M->mass[b];
                    const V3 cw = K[j].P + mul(K[j].R, ld3(M->com[b]));
                    const S3 Ic = {M->inertia[b][0], M->inertia[b][1], M->inertia[b][2], M->inertia[b][3], M->inertia[b][4], M->inertia[b][5]};
                    const S3 Icw = rot_sym(K[j].R, Ic);
                    // bias force: V x* (I V) - gravity - contacts
                    const V3 vc = K[j].V.l + cross(K[j].V.a, cw);
                    const V3 Pm = vc * m;
                    const V3 Lm = mul(Icw, K[j].V.a) + cross(cw, Pm);
                    const V3 fg = grav * m;
                    V6 pb;
                    pb.a = cross(K[j].V.a, Lm) + cross(K[j].V.l, Pm) - cross(cw, fg);
                    pb.l = cross(K[j].V.a, Pm) - fg;
                    V6 ext = {v3(0, 0, 0), v3(0, 0, 0)};
                    for (int s = M->body_sph_start[b]; s < M->body_sph_start[b + 1]; s++) {
                        if (s == fs) continue;   // the foot sphere is solved implicitly in stage 2
                        sphere_contact(s, K[j].R, K[j].P, K[j].V, f_link[j], ext);
                    }
                    pb = pb - ext;
                    const S3 Ab = Icw + parallel_axis(m, cw);
                    const M3 Bb = skew(cw * m);
                    if (j == 2) {
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
            float qdd[3];
            V6 a_calf;
            {
                V6 a = a0;
#pragma unroll
                for (int j = 0; j < 3; j++) {
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
            float depth = -1.f;
            bool fact = false;
            {
                const V3 r = K[2].P + mul(K[2].R, foot_c_loc);
                float h;
                terrain_at(O, p.hf, pos.x + r.x, pos.y + r.y, h, cn);
                depth = foot_r - (pos.z + r.z - h) * cn.z;
                fact = depth > -margin;
                cp = r - cn * foot_r;
            }
            float lim_e[3], lim_s[3], lim_T[3] = {0.f, 0.f, 0.f};
            bool lact = false;
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const float lo = M->q_lo[d0 + j], hi = M->q_hi[d0 + j];
                lim_s[j] = 0.f; lim_e[j] = 0.f;
                if (q[j] < lo + O->limit_margin) { lim_s[j] = 1.f; lim_e[j] = lo - q[j]; lact = true; }
                else if (q[j] > hi - O->limit_margin) { lim_s[j] = -1.f; lim_e[j] = q[j] - hi; lact = true; }
            }
            const int any = quad_or<LEGS>((fact || lact) ? 1 : 0);
            float dqdd[3] = {0.f, 0.f, 0.f};
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
                    const float zero3[3] = {0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        float du[3], dq[3];
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
                    const V3 vpt = K[2].V.l + cross(K[2].V.a, cp);
                    const V3 apt = a_calf.l + cross(a_calf.a, cp) + cross(K[2].V.a, vpt);
                    const V3 vf = vpt + apt * dt;
                    vfree = v3(dot(cn, vf), dot(ct1, vf), dot(ct2, vf));
                }
                V3 resp_c = v3(0, 0, 0);  // contact-frame acceleration response to the current force set
                for (int it = 0; it < O->contact_iters; it++) {
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
                                const float fn = rn * rcp(1.f + kappa * (Ac.xx + mu * (Ac.xy * e1 + Ac.xz * e2)));
                                fsol = fn > 0.f ? v3(fn, mu * fn * e1, mu * fn * e2) : v3(0, 0, 0);
                            }
                            fnew = fsol;
                        }
                    }
                    fc = fnew;
                    float tl[3];
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        tl[j] = 0.f;
                        if (lim_s[j] != 0.f) {
                            const float vin = -lim_s[j] * (qd[j] + dt * (qdd[j] + dqdd[j])) + dt * lim_T[j] * J[j].dinv;
                            const float Tn = (kl * lim_e[j] + kapl * vin) * rcp(1.f + kapl * dt * J[j].dinv);
                            lim_T[j] = fmaxf(Tn, 0.f);
                            tl[j] = lim_s[j] * lim_T[j];
                        }
                    }
                    // exact response of the whole robot to the current force set
                    const V3 fw = cn * fc.x + ct1 * fc.y + ct2 * fc.z;
                    const V6 fsp = {cross(cp, fw), fw};
                    float du[3];
                    const V6 dp = quad_sum<LEGS>(resp_up(J, fsp, tl, du));
                    da0 = chol6_solve(ch, V6{-dp.a, -dp.l});
                    const V6 ac = resp_down(J, da0, du, dqdd);
                    const V3 ra = ac.l + cross(ac.a, cp);
                    resp_c = v3(dot(cn, ra), dot(ct1, ra), dot(ct2, ra));
                }
                f_link[3] = cn * fc.x + ct1 * fc.y + ct2 * fc.z;
            }
            // ---- semi-implicit Euler ------------------------------------------------------------
            {
                const V3 alpha = a0.a + da0.a;
                const V3 alin = a0.l + da0.l + cross(ww, vw);   // spatial -> classical at O
                const float mv = O->max_base_lin_vel, mw = O->max_base_ang_vel;
                vw = v3(clampf(vw.x + dt * alin.x, -mv, mv), clampf(vw.y + dt * alin.y, -mv, mv), clampf(vw.z + dt * alin.z, -mv, mv));
                ww = v3(clampf(ww.x + dt * alpha.x, -mw, mw), clampf(ww.y + dt * alpha.y, -mw, mw), clampf(ww.z + dt * alpha.z, -mw, mw));
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const float vl = O->joint_vel_clamp * M->vel_limit[d0 + j];
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

        // ---- read-back (genesis_simulator.py:35-60) ----------------------------------------------
        {   // non-finite guard: re-seat the robot (see oracle for the rationale)
            float chk = pos.x + pos.y + pos.z + qx + qy + qz + qw + vw.x + vw.y + vw.z + ww.x + ww.y + ww.z;
#pragma unroll
            for (int j = 0; j < 3; j++) chk += q[j] + qd[j];
            const int bad = quad_or<LEGS>(isfinite(chk) ? 0 : 1);
            if (bad) {
                pos = ld3(O->base_init_pos);
                if (B.env_origins) pos += ld3(B.env_origins + 3 * e);
                vw = ww = v3(0, 0, 0);
                qx = qy = qz = 0.f; qw = 1.f;
#pragma unroll
                for (int j = 0; j < 3; j++) { q[j] = O->default_dof_pos[d0 + j]; qd[j] = 0.f; torque[j] = 0.f; f_link[j] = v3(0, 0, 0); }
                f_link[3] = v3(0, 0, 0); f_base = v3(0, 0, 0);
            }
        }
        // out-of-terrain teleport (genesis_simulator.py:612-628)
        if (pos.x >= O->bound_x[1] || pos.x <= O->bound_x[0] || pos.y >= O->bound_y[1] || pos.y <= O->bound_y[0]) {
            pos = ld3(O->base_init_pos);
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
            for (int j = 0; j < 3; j++) {
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
        }
        if (live) {
#pragma unroll
            for (int j = 0; j < 3; j++) {
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
            for (int k = leg; k < P; k += LEGS) {
                const float vx = B.height_points[2 * k], vy = B.height_points[2 * k + 1];
                // quat_apply with xyz = (0, 0, yz): t = 2 xyz x v; r = v + w t + xyz x t
                const float tx = -2.f * yz * vy, ty = 2.f * yz * vx;
                const float rx = vx + yw * tx - yz * ty, ry = vy + yw * ty + yz * tx;
                const float h = sample_min3(O, p.hf, rx + pos.x, ry + pos.y);
                acc += pos.z - h;
                if (live) B.measured_heights[(size_t)e * P + k] = h;
            }
            mean_height = quad_sum<LEGS>(acc) / (float)P;
            if (O->feet_terrain_info) {
                int px = (int)((foot_p.x + O->border) / O->hscale), py = (int)((foot_p.y + O->border) / O->hscale);
                px = min(max(px, 0), O->terrain_rows - 2);
                py = min(max(py, 0), O->terrain_cols - 2);
                const int C = O->terrain_cols, xm = max(px - 1, 0), ym = max(py - 1, 0);
                const int16_t *hf = p.hf;
                // order of genesis_simulator.py:591-599
                const int hh[9] = {hf[xm * C + py], hf[(px + 1) * C + py], hf[px * C + ym], hf[px * C + py + 1], hf[px * C + py],
                                   hf[xm * C + ym], hf[(px + 1) * C + py + 1], hf[xm * C + py + 1], hf[(px + 1) * C + ym]};
                float sum = 0.f, mx = -1e30f;
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    const float hv = (float)hh[k] * O->vscale;
                    sum += hv; mx = fmaxf(mx, hv);
                    if (live) B.height_around_feet[((size_t)e * F + foot_slot) * 9 + k] = hv;
                }
                foot_hmean = sum / 9.f; foot_hmax = mx;
                // normal from RAW int16 differences over 2*hscale -- the reference does not apply the
                // vertical scale here (genesis_simulator.py:601-606); reproduced
                const float dx = (float)(hh[1] - hh[0]) / (O->hscale * 2.f), dy = (float)(hh[3] - hh[2]) / (O->hscale * 2.f);
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
        // SIM not in this launch: pick the read-back up from HBM
        blv = ld3(B.base_lin_vel + 3 * e); bav = ld3(B.base_ang_vel + 3 * e);
        pg = ld3(B.projected_gravity + 3 * e); eul = ld3(B.base_euler + 3 * e);
        const int l0 = foot_link - 3;
#pragma unroll
        for (int k = 0; k < 4; k++) f_link[k] = ld3(B.link_contact_forces + (e * L + l0 + k) * 3);
        f_base = ld3(B.link_contact_forces + (e * L) * 3);
        foot_p = ld3(B.feet_pos + (e * F + foot_slot) * 3);
        foot_v = ld3(B.feet_vel + (e * F + foot_slot) * 3);
        last_foot_v = ld3(B.last_feet_vel + (e * F + foot_slot) * 3);
#pragma unroll
        for (int j = 0; j < 3; j++) { last_qd[j] = B.last_dof_vel[e * A + d0 + j]; torque[j] = B.torques[e * A + d0 + j]; }
        if (P > 0) {
            float acc = 0.f;
            for (int k = leg; k < P; k += LEGS) acc += pos.z - B.measured_heights[(size_t)e * P + k];
            mean_height = quad_sum<LEGS>(acc) / (float)P;
            if (O->feet_terrain_info) {
                float sum = 0.f, mx = -1e30f;
#pragma unroll
                for (int k = 0; k < 9; k++) { const float hv = B.height_around_feet[((size_t)e * F + foot_slot) * 9 + k]; sum += hv; mx = fmaxf(mx, hv); }
                foot_hmean = sum / 9.f; foot_hmax = mx;
            }
        }
    }

    if (!DO_POST && !DO_RESET) return;

    // ======================= MDP: legged_robot.py:55-168, 300-334 ============================
    RandSrc rs;
    rs.in = B.rand_in ? B.rand_in + (size_t)e * T->slots.n_slots : nullptr;
    {
        const unsigned long long gid = (unsigned long long)(T->env_id_offset + e);
        rs.k0 = (unsigned)(T->seed & 0xFFFFFFFFu); rs.k1 = (unsigned)(T->seed >> 32);
        rs.e_lo = (unsigned)(gid & 0xFFFFFFFFu); rs.e_hi = (unsigned)(gid >> 32);
        rs.step = (unsigned)p.counter;
    }
    const int N = B.n_envs;
    const float cdt = T->control_dt;
    float cmd0 = B.commands[4 * e], cmd1 = B.commands[4 * e + 1], cmd2 = B.commands[4 * e + 2], cmd3 = B.commands[4 * e + 3];
    int ep_len = B.episode_length_buf[e];
    long long fail_buf = B.fail_buf[e];
    bool reset = false, time_out = false;
    float air = B.feet_air_time[e * F + foot_slot];
    int last_contact = B.last_contacts[e * F + foot_slot];
    float total = 0.f;
    const float *cr = B.command_ranges;
    float *esum = B.episode_sums;

    // ---- periodic-gait task state (go2_wtw.py:295-346): per-env scalars + this lane's foot entries
    const bool WTW = T->gait_mode == 1, BIPED = T->gait_mode == 2, GAIT = WTW || BIPED;
    float *ts = B.task_state ? B.task_state + (size_t)e * T->task_state_width : nullptr;
    float gait_time = 0.f, phi = 0.f, gait_period = 1.f, bh_tgt = 0.f, fc_tgt = 0.f, pitch_tgt = 0.f, theta = 0.f, expC = 0.f;
    if (WTW) {
        gait_time = ts[0]; phi = ts[1]; gait_period = ts[2]; bh_tgt = ts[3]; fc_tgt = ts[4]; pitch_tgt = ts[5];
        theta = ts[6 + foot_slot]; expC = ts[18 + foot_slot];
    }
    if (BIPED) {   // layout LG_TASK_STATE_BIPED (tron1_pf_ee.py:167-184)
        gait_time = ts[0]; phi = ts[1]; gait_period = T->gait_period_fixed;
        theta = ts[4 + foot_slot]; expC = ts[10 + foot_slot];
    }
    auto resample_behavior = [&](int slot) {   // go2_wtw.py:180-218
        gait_period = (cr[9] - cr[8]) * rs.draw(slot) + cr[8];
        bh_tgt = (cr[11] - cr[10]) * rs.draw(slot + 1) + cr[10];
        fc_tgt = (cr[13] - cr[12]) * rs.draw(slot + 2) + cr[12];
        pitch_tgt = (cr[15] - cr[14]) * rs.draw(slot + 3) + cr[14];
        // one gait index per call for the whole batch (quirk 6): env-independent Philox counter
        float ug;
        if (rs.in) ug = rs.in[slot + 4];
        else { RandSrc g = rs; g.e_lo = 0xFFFFFFFFu; g.e_hi = 0xFFFFFFFFu; ug = g.draw(slot + 4); }
        const int ng = (int)cr[16];
        const int sel = min((int)floorf(ug * (float)ng), ng - 1);
        theta = T->theta_table[sel][foot_slot];
        // pronk / bound gaits keep the lowest clearance target (go2_wtw.py:212-218); the lower bound
        // never moves, so clamping at the env's own resample is equivalent to the reference's all-env pass
        const float t0 = T->theta_table[sel][0], t1 = T->theta_table[sel][1], t2 = T->theta_table[sel][2], t3 = T->theta_table[sel][3];
        if (t0 == 0.f && t1 == 0.f && ((t2 == 0.f && t3 == 0.f) || (t2 == 0.5f && t3 == 0.5f))) fc_tgt = cr[12];
    };

    auto resample_commands = [&](int slot) {  // legged_robot.py:317-334
        cmd0 = (cr[1] - cr[0]) * rs.draw(slot) + cr[0];
        cmd1 = (cr[3] - cr[2]) * rs.draw(slot + 1) + cr[2];
        if (T->heading_command) cmd3 = (cr[7] - cr[6]) * rs.draw(slot + 2) + cr[6];
        else cmd2 = (cr[5] - cr[4]) * rs.draw(slot + 2) + cr[4];
        const float keep = sqrtf(cmd0 * cmd0 + cmd1 * cmd1 + cmd2 * cmd2) > 0.2f ? 1.f : 0.f;
        cmd0 *= keep; cmd1 *= keep; cmd2 *= keep;
    };

    if (DO_POST) {
        ep_len += 1;                                                   // legged_robot.py:60
        // ---- _post_physics_step_callback (legged_robot.py:300-315) ----
        if (ep_len % T->resample_steps == 0) resample_commands(T->slots.cb_cmd);
        if (T->heading_command) {
            // forward = quat_apply(base_quat, [1,0,0]) (math_utils.py:34-40)
            const V3 xyz = v3(qx, qy, qz), bvec = v3(1.f, 0.f, 0.f);
            const V3 t = cross(xyz, bvec) * 2.f;
            const V3 fwd = bvec + t * qw + cross(xyz, t);
            const float heading = atan2f(fwd.y, fwd.x);
            cmd2 = clampf(0.5f * wrap_to_pi(cmd3 - heading), T->yaw_clip[0], T->yaw_clip[1]);
        }
        if (T->push_interval > 0 && (p.counter % T->push_interval) == 0) {   // genesis_simulator.py:150-158
            const float m = T->max_push_vel_xy;
            const float px = (m + m) * rs.draw(T->slots.push) - m, py = (m + m) * rs.draw(T->slots.push + 1) - m;
            vw.x += px; vw.y += py;
            if (lead) {
                B.rand_push_vels[3 * e] = px; B.rand_push_vels[3 * e + 1] = py;
                B.base_lin_vel_w[3 * e] = vw.x; B.base_lin_vel_w[3 * e + 1] = vw.y;
            }
        }
        if (WTW && T->behavior_resample_steps > 0 && ep_len % T->behavior_resample_steps == 0)   // go2_wtw.py:258-263
            resample_behavior(T->slots.task_cb);
        // ---- check_termination (legged_robot.py:78-92) ----
        const int l0 = foot_link - 3;
        int fail = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if ((M->term_link_mask >> (l0 + k)) & 1u) fail |= norm(f_link[k]) > 10.0f ? 1 : 0;
        fail = quad_or<LEGS>(fail);
        if (M->term_link_mask & 1u) fail |= norm(f_base) > 10.0f ? 1 : 0;
        fail |= pg.z > T->max_projected_gravity ? 1 : 0;
        fail_buf += fail;
        time_out = (float)ep_len > T->max_episode_length;
        reset = ((float)fail_buf > T->fail_threshold) || time_out;

        // ---- compute_reward (legged_robot.py:150-168), alphabetical order ----
        const float *sc = T->reward_scales;
        auto add = [&](int id, float r) {
            const float rew = r * sc[id];
            total += rew;
            if (lead) esum[(size_t)id * N + e] += rew;
        };
        const float cmd_xy = sqrtf(cmd0 * cmd0 + cmd1 * cmd1);
        const float cmd_xyz = sqrtf(cmd0 * cmd0 + cmd1 * cmd1 + cmd2 * cmd2);
        float dq0[3];
#pragma unroll
        for (int j = 0; j < 3; j++) dq0[j] = q[j] - O->default_dof_pos[d0 + j];
        auto gait_reward = [&]() {   // "step" indicator of go2_wtw.py:377-470 / tron1_pf_ee.py:347-424
            const float two_pi = 6.283185307179586f;
            float ph = phi + theta;
            ph = (ph - floorf(ph)) * two_pi;                             // (phi + theta) % 1.0, operands >= 0
            const float b_sw = T->b_swing * two_pi;
            const float c_frc = (ph >= 0.f && ph < b_sw) ? -1.f : 0.f;
            const float c_spd = (ph >= b_sw && ph < two_pi) ? -1.f : 0.f;
            expC = c_frc;
            return expf(quad_sum<LEGS>(c_spd * norm(foot_v) + c_frc * norm(f_link[3])));
        };
        if (sc[LG_R_ACTION_RATE] != 0.f) {                              // :495-497
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 3; j++) { const float d = last_act[j] - act[j]; s += d * d; }
            add(LG_R_ACTION_RATE, quad_sum<LEGS>(s));
        }
        if (sc[LG_R_ACTION_SMOOTHNESS] != 0.f) {                        // :499-503
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 3; j++) { const float d = act[j] - 2.f * last_act[j] + llast_act[j]; s += d * d; }
            add(LG_R_ACTION_SMOOTHNESS, quad_sum<LEGS>(s));
        }
        if (sc[LG_R_ANG_VEL_XY] != 0.f) add(LG_R_ANG_VEL_XY, bav.x * bav.x + bav.y * bav.y);   // :462-464
        if (sc[LG_R_BASE_HEIGHT] != 0.f) {                              // :470-476
            // plane: measured_heights is the all-zero buffer of genesis_simulator.py:494 -> base z
            const float d = (P > 0 ? mean_height : pos.z) - T->base_height_target;
            add(LG_R_BASE_HEIGHT, d * d);
        }
        if (sc[LG_R_BIPED_PERIODIC_GAIT] != 0.f) add(LG_R_BIPED_PERIODIC_GAIT, gait_reward());  // tron1_pf_ee.py:426-433
        if (sc[LG_R_COLLISION] != 0.f) {                                // :505-512
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if ((M->pen_link_mask >> (l0 + k)) & 1u) s += norm(f_link[k]) > 0.1f ? 1.f : 0.f;
            s = quad_sum<LEGS>(s);
            if (M->pen_link_mask & 1u) s += norm(f_base) > 0.1f ? 1.f : 0.f;
            add(LG_R_COLLISION, s);
        }
        if (sc[LG_R_DOF_ACC] != 0.f) {                                  // :490-493
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 3; j++) { const float d = (last_qd[j] - qd[j]) / cdt; s += d * d; }
            add(LG_R_DOF_ACC, quad_sum<LEGS>(s));
        }
        if (sc[LG_R_DOF_CLOSE_TO_DEFAULT] != 0.f) {                     // :571-573
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 3; j++) s += dq0[j] * dq0[j];
            add(LG_R_DOF_CLOSE_TO_DEFAULT, quad_sum<LEGS>(s));
        }
        if (sc[LG_R_DOF_POS_LIMITS] != 0.f) {                           // :518-522
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 3; j++) {
                s += -fminf(q[j] - T->soft_dof_lo[d0 + j], 0.f);
                s += fmaxf(q[j] - T->soft_dof_hi[d0 + j], 0.f);
            }
            add(LG_R_DOF_POS_LIMITS, quad_sum<LEGS>(s));
        }
        if (sc[LG_R_DOF_POS_STAND_STILL] != 0.f) {                      // :561-563
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 3; j++) s += dq0[j] * dq0[j];
            add(LG_R_DOF_POS_STAND_STILL, quad_sum<LEGS>(s) * (cmd_xyz < 0.1f ? 1.f : 0.f));
        }
        if (sc[LG_R_DOF_POWER] != 0.f) {                                // :486-488
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 3; j++) s += fabsf(torque[j] * qd[j]);
            add(LG_R_DOF_POWER, quad_sum<LEGS>(s));
        }
        if (sc[LG_R_DOF_VEL] != 0.f) {                                  // :482-484
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 3; j++) s += qd[j] * qd[j];
            add(LG_R_DOF_VEL, quad_sum<LEGS>(s));
        }
        if (sc[LG_R_DOF_VEL_STAND_STILL] != 0.f) {                      // :557-559
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 3; j++) s += fabsf(qd[j]);
            add(LG_R_DOF_VEL_STAND_STILL, quad_sum<LEGS>(s) * (cmd_xyz < 0.1f ? 1.f : 0.f));
        }
        if (sc[LG_R_FEET_AIR_TIME] != 0.f) {                            // :545-555 (stateful)
            const int contact = f_link[3].z > 1.0f ? 1 : 0;
            const int filt = contact | last_contact;
            last_contact = contact;
            const float first = (air > 0.f ? 1.f : 0.f) * (float)filt;
            air += cdt;
            float r = quad_sum<LEGS>((air - T->feet_air_time_threshold) * first);
            r *= cmd_xy > 0.1f ? 1.f : 0.f;
            air *= filt ? 0.f : 1.f;
            add(LG_R_FEET_AIR_TIME, r);
        }
        if (sc[LG_R_FEET_CONTACT_STAND_STILL] != 0.f) {                 // :565-569
            const float cnt = quad_sum<LEGS>(f_link[3].z > 0.1f ? 1.f : 0.f);
            add(LG_R_FEET_CONTACT_STAND_STILL, (cnt == (float)LEGS ? 1.f : 0.f) * (cmd_xyz < 0.1f ? 1.f : 0.f));
        }
        if (sc[LG_R_FEET_DISTANCE] != 0.f) {                            // tron1_pf_ee.py:458-463 (two feet: lane pair)
            const float ox = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(foot_p.x), 0xB1, 0xF, 0xF, false));
            const float oy = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(foot_p.y), 0xB1, 0xF, 0xF, false));
            const float dxy = sqrtf((foot_p.x - ox) * (foot_p.x - ox) + (foot_p.y - oy) * (foot_p.y - oy));
            add(LG_R_FEET_DISTANCE, fmaxf(0.f, T->foot_distance_threshold - dxy));
        }
        if (sc[LG_R_FOOT_ACC] != 0.f) {                                 // :605-608
            const V3 a = (foot_v - last_foot_v) * (1.f / cdt);
            add(LG_R_FOOT_ACC, quad_sum<LEGS>(dot(a, a)));
        }
        if (sc[LG_R_FOOT_CLEARANCE] != 0.f) {                           // :575-588
            const float vxy = sqrtf(foot_v.x * foot_v.x + foot_v.y * foot_v.y);
            // go2_ee.py:136-150 measures the clearance above the mean terrain height around the foot
            const float d = foot_p.z - (T->obs_layout == LG_OBS_GO2_EE ? foot_hmean : (T->obs_layout == LG_OBS_TRON1_EE ? foot_hmax : 0.f))
                            - T->foot_clearance_target - T->foot_height_offset;   // tron1_pf_ee.py:442-456 uses the max
            const float err = quad_sum<LEGS>(vxy * (d * d));
            add(LG_R_FOOT_CLEARANCE, expf(-err / T->foot_clearance_sigma));
        }
        if (sc[LG_R_FOOT_LANDING_VEL] != 0.f) {                         // :590-599
            const bool c01 = f_link[3].z > 0.1f;
            const bool land = ((foot_p.z - T->foot_height_offset) < T->about_landing_threshold) && !c01 && (foot_v.z < 0.f);
            const float vz = land ? foot_v.z : 0.f;
            add(LG_R_FOOT_LANDING_VEL, quad_sum<LEGS>(vz * vz));
        }
        if (sc[LG_R_HIP_POS] != 0.f) add(LG_R_HIP_POS, quad_sum<LEGS>(dq0[0] * dq0[0]));   // go2_ee.py:152-159
        if (sc[LG_R_KEEP_BALANCE] != 0.f) add(LG_R_KEEP_BALANCE, 1.f);                      // :601-603
        if (sc[LG_R_LIN_VEL_Z] != 0.f) add(LG_R_LIN_VEL_Z, blv.z * blv.z);                  // :458-460
        if (sc[LG_R_ORIENTATION] != 0.f) add(LG_R_ORIENTATION, pg.x * pg.x + pg.y * pg.y);  // :466-468
        if (sc[LG_R_QUAD_PERIODIC_GAIT] != 0.f) add(LG_R_QUAD_PERIODIC_GAIT, gait_reward());   // go2_wtw.py:472-484
        if (sc[LG_R_TORQUES] != 0.f) {                                  // :478-480
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 3; j++) s += torque[j] * torque[j];
            add(LG_R_TORQUES, quad_sum<LEGS>(s));
        }
        if (sc[LG_R_TRACKING_ANG_VEL] != 0.f) {                         // :539-543
            const float d = cmd2 - bav.z;
            add(LG_R_TRACKING_ANG_VEL, expf(-(d * d) / T->tracking_sigma));
        }
        if (sc[LG_R_TRACKING_BASE_HEIGHT] != 0.f) {                     // go2_wtw.py:495-500 (plane: heights are zero)
            // wtw: per-env target on the plane; tron1_pf_ee.py:435-440: fixed target, mean over the height samples
            const float d = WTW ? pos.z - bh_tgt : (P > 0 ? mean_height : pos.z) - T->base_height_target;
            add(LG_R_TRACKING_BASE_HEIGHT, expf(-(d * d) / T->base_height_sigma));
        }
        if (sc[LG_R_TRACKING_FOOT_CLEARANCE] != 0.f) {                  // go2_wtw.py:507-519
            const float vxy = sqrtf(foot_v.x * foot_v.x + foot_v.y * foot_v.y);
            const float d = foot_p.z - fc_tgt - T->foot_height_offset;
            add(LG_R_TRACKING_FOOT_CLEARANCE, expf(-quad_sum<LEGS>(vxy * (d * d)) / T->foot_clearance_sigma));
        }
        if (sc[LG_R_TRACKING_LIN_VEL] != 0.f) {                         // :533-537
            const float dx = cmd0 - blv.x, dy = cmd1 - blv.y;
            add(LG_R_TRACKING_LIN_VEL, expf(-(dx * dx + dy * dy) / T->tracking_sigma));
        }
        if (sc[LG_R_TRACKING_ORIENTATION] != 0.f) {                     // go2_wtw.py:502-505
            const float dp = eul.y - pitch_tgt;
            add(LG_R_TRACKING_ORIENTATION, expf(-(eul.x * eul.x + dp * dp) / T->euler_sigma));
        }
        if (T->only_positive_rewards) total = fmaxf(total, 0.f);        // :161-162
        if (sc[LG_R_TERMINATION] != 0.f) add(LG_R_TERMINATION, (reset && !time_out) ? 1.f : 0.f);  // :163-168
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

    // ---- reset_idx (legged_robot.py:94-148) + simulator.reset_idx (genesis_simulator.py:62-82) ----
    if (DO_RESET) {
        V3 pos_origin_override = v3(0, 0, 0);
        bool have_origin = false;
        if (lead && e == 0) {  // clear the per-step reset accumulator the NEXT launch will use
            float *nxt = B.episode_done_sums + ((p.counter + 1) % LG_DONE_RING) * (LG_R_COUNT + 2);
            for (int k = 0; k < LG_R_COUNT + 2; k++) nxt[k] = 0.f;
        }
        if (reset) {
            if (T->terrain_curriculum && p.counter > 0) {
                // legged_robot.py:254-272 + genesis_simulator.py:140-148 (skipped on the construction-time reset,
                // where the reference returns early because init_done is False)
                const V3 org = ld3(B.env_origins + 3 * e);
                const float dx = pos.x - org.x, dy = pos.y - org.y;
                const float dist = sqrtf(dx * dx + dy * dy);
                const bool up = dist > T->terrain_env_length / 2.f;
                const bool down = (dist < sqrtf(cmd0 * cmd0 + cmd1 * cmd1) * T->episode_length_s * 0.5f) && !up;
                int lvl = B.terrain_levels[e] + (up ? 1 : 0) - (down ? 1 : 0);
                if (lvl >= T->max_terrain_level) lvl = min((int)floorf(rs.draw(T->slots.terrain_level) * (float)T->max_terrain_level), T->max_terrain_level - 1);
                else lvl = max(lvl, 0);
                const V3 norg = ld3(B.terrain_origins + ((size_t)lvl * T->terrain_cols_n + B.terrain_types[e]) * 3);
                if (lead) { B.terrain_levels[e] = lvl; st3(B.env_origins + 3 * e, norg); }
                pos_origin_override = norg; have_origin = true;
            }
            if (WTW) { resample_behavior(T->slots.task_reset); gait_time = 0.f; phi = 0.f; }   // go2_wtw.py:124-142
            resample_commands(T->slots.reset_cmd);
            // tron1_pf_ee.py:204-210: ONE coin per reset_idx call sends the whole batch to the sit pose (quirk 11)
            bool sit = false;
            if (T->sit_percent > 0.f) {
                float us;
                if (rs.in) us = rs.in[T->slots.task_reset];
                else { RandSrc g = rs; g.e_lo = 0xFFFFFFFFu; g.e_hi = 0xFFFFFFFFu; us = g.draw(T->slots.task_reset); }
                sit = us < T->sit_percent;
            }
            // _reset_dofs (go2.py:17-37): default + U(range) per joint, zero velocity
#pragma unroll
            for (int j = 0; j < 3; j++) {
                q[j] = sit ? T->sit_dof_pos[d0 + j]
                           : O->default_dof_pos[d0 + j] + (T->reset_dof_span[d0 + j] * rs.draw(T->slots.reset_dof + d0 + j) + T->reset_dof_lo[d0 + j]);
                qd[j] = 0.f;
                last_qd[j] = 0.f;
                act[j] = last_act[j] = llast_act[j] = 0.f;
            }
            // _reset_root_states (go2.py:119-134)
            pos = (sit ? ld3(T->sit_pos) : ld3(O->base_init_pos)) + (have_origin ? pos_origin_override : ld3(B.env_origins + 3 * e));
            if (T->custom_origins) {
                pos.x += T->reset_root_xy_span * rs.draw(T->slots.reset_root_xy) + T->reset_root_xy_lo;
                pos.y += T->reset_root_xy_span * rs.draw(T->slots.reset_root_xy + 1) + T->reset_root_xy_lo;
            }
            const float *iq = sit ? T->sit_quat : T->base_init_quat;
            qx = iq[0]; qy = iq[1]; qz = iq[2]; qw = iq[3];
            vw = v3(T->reset_lin_vel_span * rs.draw(T->slots.reset_lin_vel) + T->reset_lin_vel_lo,
                    T->reset_lin_vel_span * rs.draw(T->slots.reset_lin_vel + 1) + T->reset_lin_vel_lo,
                    T->reset_lin_vel_span * rs.draw(T->slots.reset_lin_vel + 2) + T->reset_lin_vel_lo);
            ww = v3(T->reset_ang_vel_span * rs.draw(T->slots.reset_ang_vel) + T->reset_ang_vel_lo,
                    T->reset_ang_vel_span * rs.draw(T->slots.reset_ang_vel + 1) + T->reset_ang_vel_lo,
                    T->reset_ang_vel_span * rs.draw(T->slots.reset_ang_vel + 2) + T->reset_ang_vel_lo);
            if (sit) { vw = v3(0, 0, 0); ww = v3(0, 0, 0); }        // tron1_pf_ee.py:304-309
            if (BIPED) {                                             // tron1_pf_ee.py:220-226
                const float th0 = T->theta_table[0][0] + rs.draw(T->slots.task_reset + 1);
                theta = foot_slot == 0 ? th0 : th0 + (T->theta_table[0][1] - T->theta_table[0][0]);
                gait_time = rs.draw(T->slots.task_reset + 2) * gait_period;
                phi = gait_time / gait_period;
            }
            // the reference stores the commanded reset twist verbatim in the body-frame properties
            // (genesis_simulator.py:128-129) and refreshes projected gravity (:125)
            blv = vw; bav = ww;
            pg = quat_rotate_inverse(qx, qy, qz, qw, v3(0.f, 0.f, -1.f));
            air = 0.f; ep_len = 0; fail_buf = 0; last_foot_v = v3(0, 0, 0);
            if (live) {
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    B.dof_pos[e * A + d0 + j] = q[j];
                    B.dof_vel[e * A + d0 + j] = 0.f;
                    B.last_dof_vel[e * A + d0 + j] = 0.f;
                    B.actions[e * A + d0 + j] = 0.f; B.last_actions[e * A + d0 + j] = 0.f; B.llast_actions[e * A + d0 + j] = 0.f;
                }
                st3(B.last_feet_vel + (e * F + foot_slot) * 3, v3(0, 0, 0));
                if (T->dr_pd_on) {   // genesis_simulator.py:735-739
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        B.kp_scale[e * A + d0 + j] = T->dr_kp_span * rs.draw(T->slots.dr_kp + d0 + j) + T->dr_kp_lo;
                        B.kd_scale[e * A + d0 + j] = T->dr_kd_span * rs.draw(T->slots.dr_kd + d0 + j) + T->dr_kd_lo;
                    }
                }
            }
            if (lead) {
                st3(B.base_pos + 3 * e, pos);
                B.base_quat[4 * e] = qx; B.base_quat[4 * e + 1] = qy; B.base_quat[4 * e + 2] = qz; B.base_quat[4 * e + 3] = qw;
                st3(B.base_lin_vel_w + 3 * e, vw); st3(B.base_ang_vel_w + 3 * e, ww);
                st3(B.base_lin_vel + 3 * e, blv); st3(B.base_ang_vel + 3 * e, bav);
                st3(B.projected_gravity + 3 * e, pg);
                st3(B.last_base_lin_vel + 3 * e, v3(0, 0, 0)); st3(B.last_base_ang_vel + 3 * e, v3(0, 0, 0));
                // domain randomisation (genesis_simulator.py:62-77, 665-739)
                if (T->dr_friction_on) B.friction_values[e] = T->dr_friction_span * rs.draw(T->slots.dr_friction) + T->dr_friction_lo;
                if (T->dr_mass_on) B.added_base_mass[e] = T->dr_mass_span * rs.draw(T->slots.dr_mass) + T->dr_mass_lo;
                if (T->dr_com_on) {
#pragma unroll
                    for (int k = 0; k < 3; k++)
                        B.base_com_bias[3 * e + k] = T->dr_com_span[k] * rs.draw(T->slots.dr_com + k) + T->dr_com_lo[k];
                }
                if (T->dr_joint_on && B.joint_armature) {
                    B.joint_armature[e] = T->dr_joint_span[0] * rs.draw(T->slots.dr_joint) + T->dr_joint_lo[0];
                    B.joint_friction[e] = T->dr_joint_span[1] * rs.draw(T->slots.dr_joint + 1) + T->dr_joint_lo[1];
                    B.joint_damping[e] = T->dr_joint_span[2] * rs.draw(T->slots.dr_joint + 2) + T->dr_joint_lo[2];
                }
                // extras["episode"] sums (legged_robot.py:128-132): one atomic per active term per reset
                float *row = B.episode_done_sums + (p.counter % LG_DONE_RING) * (LG_R_COUNT + 2);
                for (int k = 0; k < LG_R_COUNT; k++) {
                    if (T->reward_scales[k] != 0.f) {
                        atomicAdd(row + k, esum[(size_t)k * N + e]);
                        esum[(size_t)k * N + e] = 0.f;
                    }
                }
                atomicAdd(row + LG_R_COUNT, 1.f);
            }
        }
    }
    if (DO_RESET) {
        // ---- compute_observations + clip (legged_robot.py:48-49).  Layouts: go2.py:40-64 (45),
        //      go2_wtw.py:53-111 (61x5 | 99x5), go2_ee.py:10-75 (45x20 | 174x5 | 24 labels).  Histories are
        //      kept oldest -> newest inside obs_buf / priv_obs_buf themselves and shifted in place.
        const int FR = T->obs_frame, PF = T->priv_frame, ST = T->obs_stack, PST = T->priv_stack;
        float *o = B.obs_buf + (size_t)e * T->num_obs;
        float *pv = T->num_priv_obs > 0 ? B.priv_obs_buf + (size_t)e * T->num_priv_obs : nullptr;
        const float co = T->clip_obs;
        if (live) {   // this lane's columns move one frame towards the past (zeros after a reset: go2_wtw.py:174-178)
            for (int f = 0; f + 1 < ST; f++)
                for (int i = leg; i < FR; i += LEGS) o[f * FR + i] = reset ? 0.f : o[(f + 1) * FR + i];
            if (pv)
                for (int f = 0; f + 1 < PST; f++)
                    for (int i = leg; i < PF; i += LEGS) pv[f * PF + i] = reset ? 0.f : pv[(f + 1) * PF + i];
        }
        float *on = o + (ST - 1) * FR, *pn = pv ? pv + (PST - 1) * PF : nullptr;
        const bool nz = T->add_noise != 0;
        const int ns = T->slots.noise;
        // uniforms for the noisy entries only (commands and actions carry zero noise scale): q, qd per lane,
        // gravity + ang vel on the lead lane
        float uq[3] = {0.5f, 0.5f, 0.5f}, uqd[3] = {0.5f, 0.5f, 0.5f}, ub[6] = {0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f};
        if (nz) {
            if (rs.in) {
#pragma unroll
                for (int j = 0; j < 3; j++) { uq[j] = rs.in[ns + 9 + d0 + j]; uqd[j] = rs.in[ns + 9 + A + d0 + j]; }
                if (lead) {
#pragma unroll
                    for (int k = 0; k < 6; k++) ub[k] = rs.in[ns + 3 + k];
                }
            } else {
                rs.block3(2 * leg, uq[0], uq[1], uq[2]);
                rs.block3(2 * leg + 1, uqd[0], uqd[1], uqd[2]);
                if (lead) { rs.block3(2 * LEGS, ub[0], ub[1], ub[2]); rs.block3(2 * LEGS + 1, ub[3], ub[4], ub[5]); }
            }
        }
        float uact[3] = {0.5f, 0.5f, 0.5f}, uclk[2] = {0.5f, 0.5f};
        if (nz && T->noise_vec[9 + 2 * A] != 0.f) {   // tron1_pf_ee.py:338-342 (quirk 4): actions and clock are noisy too
            if (rs.in) {
#pragma unroll
                for (int j = 0; j < 3; j++) uact[j] = rs.in[ns + 9 + 2 * A + d0 + j];
                uclk[0] = rs.in[ns + 9 + 3 * A + foot_slot]; uclk[1] = rs.in[ns + 9 + 3 * A + LEGS + foot_slot];
            } else {
                float dummy;
                rs.block3(2 * LEGS + 2 + leg, uact[0], uact[1], uact[2]);
                rs.block3(3 * LEGS + 2 + leg, uclk[0], uclk[1], dummy);
            }
        }
        auto put = [&](int idx, float v, float u) {        // critic copy of the frame is noise-free
            if (pn) pn[idx] = clampf(v, -co, co);
            if (nz) v += (2.f * u - 1.f) * T->noise_vec[idx];
            on[idx] = clampf(v, -co, co);
        };
        auto putp = [&](int idx, float v) { pn[idx] = clampf(v, -co, co); };
        if (live) {
#pragma unroll
            for (int j = 0; j < 3; j++) {
                put(9 + d0 + j, (q[j] - O->default_dof_pos[d0 + j]) * T->obs_scale_dof_pos, uq[j]);
                put(9 + A + d0 + j, qd[j] * T->obs_scale_dof_vel, uqd[j]);
                put(9 + 2 * A + d0 + j, act[j], uact[j]);
            }
        }
        if (lead) {
            put(0, cmd0 * T->obs_scale_lin_vel, 0.5f); put(1, cmd1 * T->obs_scale_lin_vel, 0.5f); put(2, cmd2 * T->obs_scale_ang_vel, 0.5f);
            put(3, pg.x, ub[0]); put(4, pg.y, ub[1]); put(5, pg.z, ub[2]);
            put(6, bav.x * T->obs_scale_ang_vel, ub[3]); put(7, bav.y * T->obs_scale_ang_vel, ub[4]); put(8, bav.z * T->obs_scale_ang_vel, ub[5]);
        }
        if (T->obs_layout == LG_OBS_GO2_WTW) {
            const float ang = 6.283185307179586f * (phi + theta);      // clock inputs (go2_wtw.py:251-256)
            if (live) {
                const float sn = sinf(ang), cs = cosf(ang);
                put(45 + foot_slot, sn, 0.5f);
                put(49 + foot_slot, cs, 0.5f);
                put(57 + foot_slot, theta, 0.5f);
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    putp(FR + 10 + d0 + j, B.kp_scale[e * A + d0 + j]);
                    putp(FR + 10 + A + d0 + j, B.kd_scale[e * A + d0 + j]);
                }
                putp(FR + 10 + 2 * A + foot_slot, expC);
                ts[6 + foot_slot] = theta; ts[10 + foot_slot] = sn; ts[14 + foot_slot] = cs; ts[18 + foot_slot] = expC;
            }
            if (lead) {
                put(53, gait_period, 0.5f); put(54, bh_tgt, 0.5f); put(55, fc_tgt, 0.5f); put(56, pitch_tgt, 0.5f);
                putp(FR + 0, blv.x * T->obs_scale_lin_vel); putp(FR + 1, blv.y * T->obs_scale_lin_vel); putp(FR + 2, blv.z * T->obs_scale_lin_vel);
                putp(FR + 3, B.rand_push_vels[3 * e]); putp(FR + 4, B.rand_push_vels[3 * e + 1]);
                putp(FR + 5, B.added_base_mass[e]); putp(FR + 6, B.friction_values[e]);
                putp(FR + 7, B.base_com_bias[3 * e]); putp(FR + 8, B.base_com_bias[3 * e + 1]); putp(FR + 9, B.base_com_bias[3 * e + 2]);
            }
        } else if (T->obs_layout == LG_OBS_GO2_EE) {
            // critic frame (go2_ee.py:21-48): obs 45 | DR 31 | contact states K | heights P
            const int K = __popc(M->state_link_mask);
            const int l0 = foot_link - 3;
            float *lab = B.labels_buf + (size_t)e * T->num_labels;
            if (live) {
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    putp(FR + 7 + d0 + j, B.kp_scale[e * A + d0 + j] - T->kp_offset);
                    putp(FR + 7 + A + d0 + j, B.kd_scale[e * A + d0 + j] - T->kd_offset);
                }
                // contact states are those of the physics read-back (stale for a just-reset env, as in the reference)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int l = l0 + k;
                    if ((M->state_link_mask >> l) & 1u) {
                        const int idx = __popc(M->state_link_mask & ((1u << l) - 1u));
                        const float cs = norm(f_link[k]) > 1.f ? 1.f : 0.f;
                        putp(FR + 7 + 2 * A + idx, cs);
                        lab[3 + idx] = cs;
                    }
                }
                for (int k = leg; k < P; k += LEGS) {
                    float hv = pos.z - T->heights_offset - B.measured_heights[(size_t)e * P + k];
                    if (T->heights_clip_scale) hv = clampf(hv, -1.f, 1.f) * T->obs_scale_height;
                    putp(FR + 7 + 2 * A + K + k, hv);
                }
                // labels (go2_ee.py:69-75): v_b 3 | contact states K | foot height above the local terrain mean F
                lab[3 + K + foot_slot] = clampf(foot_p.z - foot_hmean - T->foot_height_offset, -1.f, 1.f);
            }
            if (lead) {
                putp(FR + 0, B.friction_values[e] - T->friction_offset); putp(FR + 1, B.added_base_mass[e]);
                putp(FR + 2, B.base_com_bias[3 * e]); putp(FR + 3, B.base_com_bias[3 * e + 1]); putp(FR + 4, B.base_com_bias[3 * e + 2]);
                putp(FR + 5, B.rand_push_vels[3 * e]); putp(FR + 6, B.rand_push_vels[3 * e + 1]);
                if (M->state_link_mask & 1u) { const float cs = norm(f_base) > 1.f ? 1.f : 0.f; putp(FR + 7 + 2 * A, cs); lab[3] = cs; }
                lab[0] = blv.x * T->obs_scale_lin_vel; lab[1] = blv.y * T->obs_scale_lin_vel; lab[2] = blv.z * T->obs_scale_lin_vel;
            }
        } else if (T->obs_layout == LG_OBS_TRON1_EE) {
            // tron1_pf_ee.py:53-141.  actor frame: 9 + 3A + clock 2F.  critic frame: frame | DR (7 + 2A + 3) |
            // gait F | contact states K | heights P | normals 3F | clip(foot_z - h9) 9F.  labels: v_b 3 | K | F | 3F
            const int K = __popc(M->state_link_mask);
            const int l0 = foot_link - 3;
            float *lab = B.labels_buf + (size_t)e * T->num_labels;
            const float ang = 6.283185307179586f * (phi + theta);
            const int oDR = FR, oG = FR + 7 + 2 * A + 3, oK = oG + F, oH = oK + K, oN = oH + P, oR = oN + 3 * F;
            if (live) {
                const float sn = sinf(ang), cs = cosf(ang);
                put(9 + 3 * A + foot_slot, sn, uclk[0]);
                put(9 + 3 * A + F + foot_slot, cs, uclk[1]);
                ts[4 + foot_slot] = theta; ts[6 + foot_slot] = sn; ts[6 + F + foot_slot] = cs; ts[10 + foot_slot] = expC;
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    putp(oDR + 7 + d0 + j, B.kp_scale[e * A + d0 + j] - T->kp_offset);
                    putp(oDR + 7 + A + d0 + j, B.kd_scale[e * A + d0 + j] - T->kd_offset);
                }
                putp(oG + foot_slot, expC);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int l = l0 + k;
                    if ((M->state_link_mask >> l) & 1u) {
                        const int idx = __popc(M->state_link_mask & ((1u << l) - 1u));
                        const float cst = norm(f_link[k]) > 1.f ? 1.f : 0.f;
                        putp(oK + idx, cst);
                        lab[3 + idx] = cst;
                    }
                }
                for (int k = leg; k < P; k += LEGS) {
                    float hv = pos.z - T->heights_offset - B.measured_heights[(size_t)e * P + k];
                    if (T->heights_clip_scale) hv = clampf(hv, -1.f, 1.f) * T->obs_scale_height;
                    putp(oH + k, hv);
                }
                const float *nv3 = B.normal_vector_around_feet + ((size_t)e * F + foot_slot) * 3;
#pragma unroll
                for (int k = 0; k < 3; k++) { putp(oN + 3 * foot_slot + k, nv3[k]); lab[3 + K + F + 3 * foot_slot + k] = nv3[k]; }
#pragma unroll
                for (int k = 0; k < 9; k++)
                    putp(oR + 9 * foot_slot + k, clampf(foot_p.z - B.height_around_feet[((size_t)e * F + foot_slot) * 9 + k], -1.f, 1.f));
                lab[3 + K + foot_slot] = clampf(foot_p.z - foot_hmax - T->foot_height_offset, -1.f, 1.f);
            }
            if (lead) {
                putp(oDR + 0, B.friction_values[e] - T->friction_offset); putp(oDR + 1, B.added_base_mass[e]);
                putp(oDR + 2, B.base_com_bias[3 * e]); putp(oDR + 3, B.base_com_bias[3 * e + 1]); putp(oDR + 4, B.base_com_bias[3 * e + 2]);
                putp(oDR + 5, B.rand_push_vels[3 * e]); putp(oDR + 6, B.rand_push_vels[3 * e + 1]);
                putp(oDR + 7 + 2 * A, B.joint_armature ? B.joint_armature[e] : 0.f);
                putp(oDR + 8 + 2 * A, B.joint_friction ? B.joint_friction[e] : 0.f);
                putp(oDR + 9 + 2 * A, B.joint_damping ? B.joint_damping[e] : 0.f);
                if (M->state_link_mask & 1u) { const float cst = norm(f_base) > 1.f ? 1.f : 0.f; putp(oK, cst); lab[3] = cst; }
                lab[0] = blv.x * T->obs_scale_lin_vel; lab[1] = blv.y * T->obs_scale_lin_vel; lab[2] = blv.z * T->obs_scale_lin_vel;
            }
        }
    }
    if (WTW && lead) {
        ts[0] = gait_time; ts[1] = phi; ts[2] = gait_period; ts[3] = bh_tgt; ts[4] = fc_tgt; ts[5] = pitch_tgt;
    }
    if (BIPED && lead) { ts[0] = gait_time; ts[1] = phi; ts[2] = gait_period; }
    if (BIPED && live && !DO_RESET) { ts[4 + foot_slot] = theta; ts[10 + foot_slot] = expC; }
    if (WTW && live && !(DO_RESET && T->obs_layout == LG_OBS_GO2_WTW)) { ts[6 + foot_slot] = theta; ts[18 + foot_slot] = expC; }
    // second action-history shift of the wtw / tron1_ee tasks (go2_wtw.py:45-46): afterwards
    // last == llast == a_t, which makes action_smoothness == action_rate
    if (DO_RESET && T->double_shift && live) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            B.llast_actions[e * A + d0 + j] = reset ? 0.f : last_act[j];
            B.last_actions[e * A + d0 + j] = act[j];
        }
    }
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
        B.time_out_buf[e] = time_out ? 1 : 0;
        B.rew_buf[e] = total;
    }
}

// =============================== host side: the C ABI ==========================================
static thread_local std::string g_err;
static int fail(const std::string &m) { g_err = m; return 1; }
#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(_e)); } while (0)

struct LgEngine {
    LgModelDesc model; LgSimOptions opts; LgTaskCfg task;
    LgModelDesc *d_model = nullptr; LgSimOptions *d_opts = nullptr; LgTaskCfg *d_task = nullptr;
    const int16_t *hf = nullptr;
    LgBuffers bufs; bool bound = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

extern "C" const char *lg_last_error(void) { return g_err.c_str(); }
extern "C" int lg_abi_version(void) { return LG_ABI_VERSION; }

static int validate_model(const LgModelDesc *m) {
    if (m->n_legs != 2 && m->n_legs != 4) return fail("lg_create: engine supports 2 or 4 legs of 3 revolute joints");
    if (m->n_bodies != 1 + 3 * m->n_legs) return fail("lg_create: n_bodies must be 1 + 3*n_legs");
    if (m->n_links < m->n_bodies || m->n_links > LG_MAX_LINKS) return fail("lg_create: bad n_links");
    if (m->n_spheres < m->n_legs || m->n_spheres > LG_MAX_SPHERES) return fail("lg_create: bad n_spheres");
    for (int l = 0; l < m->n_legs; l++) {
        int fl = m->foot_link[l], fsph = m->foot_sphere[l];
        if (fl < 4 || fl >= m->n_links) return fail("lg_create: foot link index out of range");
        if (fsph < 0 || fsph >= m->n_spheres || m->sph_body[fsph] != 3 + 3 * l) return fail("lg_create: foot sphere must sit on the last body of its leg");
        // kernel assumes link rows [hip, thigh, calf, foot] of a leg are contiguous
        for (int k = 0; k < 3; k++)
            if (m->link_body[fl - 3 + k] != 1 + 3 * l + k) return fail("lg_create: leg links must be contiguous and end with the foot");
        if (m->link_body[fl] != 3 + 3 * l) return fail("lg_create: foot link must move with the last body of its leg");
    }
    if (m->link_body[0] != 0) return fail("lg_create: link 0 must be the base");
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
    LgEngine *h = new LgEngine();
    h->model = *model; h->opts = *opts; h->task = *task;
    memset(&h->bufs, 0, sizeof(h->bufs));
    hipError_t e;
    if ((e = hipMalloc(&h->d_model, sizeof(LgModelDesc))) != hipSuccess || (e = hipMalloc(&h->d_opts, sizeof(LgSimOptions))) != hipSuccess ||
        (e = hipMalloc(&h->d_task, sizeof(LgTaskCfg))) != hipSuccess) { delete h; return fail(std::string("lg_create: hipMalloc: ") + hipGetErrorString(e)); }
    HIPCHK(hipMemcpy(h->d_model, model, sizeof(LgModelDesc), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_opts, opts, sizeof(LgSimOptions), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_task, task, sizeof(LgTaskCfg), hipMemcpyHostToDevice));
    HIPCHK(hipEventCreate(&h->ev0));
    HIPCHK(hipEventCreate(&h->ev1));
    *out = h;
    return 0;
}

extern "C" int lg_destroy(LgHandle h) {
    if (!h) return 0;
    (void)hipFree(h->d_model); (void)hipFree(h->d_opts); (void)hipFree(h->d_task);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    delete h;
    return 0;
}

extern "C" int lg_set_task(LgHandle h, const LgTaskCfg *task) {
    if (!h || !task) return fail("lg_set_task: null argument");
    h->task = *task;
    HIPCHK(hipMemcpy(h->d_task, task, sizeof(LgTaskCfg), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int lg_set_terrain(LgHandle h, const int16_t *hf, int32_t rows, int32_t cols) {
    if (!h) return fail("lg_set_terrain: null handle");
    if (hf && (rows < 2 || cols < 2)) return fail("lg_set_terrain: heightfield needs at least 2x2 samples");
    h->hf = hf;
    h->opts.terrain_rows = hf ? rows : 0;
    h->opts.terrain_cols = hf ? cols : 0;
    HIPCHK(hipMemcpy(h->d_opts, &h->opts, sizeof(LgSimOptions), hipMemcpyHostToDevice));
    return 0;
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

template <int LEGS> static int launch(LgEngine *h, uint32_t ph, const float *actions, int64_t counter, hipStream_t st) {
    KParams p;
    p.M = h->d_model; p.O = h->d_opts; p.T = h->d_task; p.hf = h->hf; p.B = h->bufs; p.actions = actions; p.counter = counter;
    const int threads = h->bufs.n_envs * LEGS;
    dim3 grid((threads + BLOCK - 1) / BLOCK), block(BLOCK);
    switch (ph) {
    case LG_PHASE_ALL: hipLaunchKernelGGL((env_step_kernel<LEGS, LG_PHASE_ALL>), grid, block, 0, st, p); break;
    case LG_PHASE_SIM: hipLaunchKernelGGL((env_step_kernel<LEGS, LG_PHASE_SIM>), grid, block, 0, st, p); break;
    case LG_PHASE_PRE | LG_PHASE_POST | LG_PHASE_RESET:
        hipLaunchKernelGGL((env_step_kernel<LEGS, LG_PHASE_PRE | LG_PHASE_POST | LG_PHASE_RESET>), grid, block, 0, st, p); break;
    case LG_PHASE_PRE | LG_PHASE_SIM | LG_PHASE_POST:
        hipLaunchKernelGGL((env_step_kernel<LEGS, LG_PHASE_PRE | LG_PHASE_SIM | LG_PHASE_POST>), grid, block, 0, st, p); break;
    case LG_PHASE_RESET: hipLaunchKernelGGL((env_step_kernel<LEGS, LG_PHASE_RESET>), grid, block, 0, st, p); break;
    case LG_PHASE_PRE | LG_PHASE_POST:
        hipLaunchKernelGGL((env_step_kernel<LEGS, LG_PHASE_PRE | LG_PHASE_POST>), grid, block, 0, st, p); break;
    default: return fail("lg_step: unsupported phase combination (ALL, SIM, PRE|POST|RESET, PRE|SIM|POST, PRE|POST, RESET)");
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
        REQ(episode_sums); REQ(episode_done_sums); REQ(command_ranges); REQ(rand_push_vels);
        REQ(friction_values); REQ(added_base_mass); REQ(base_com_bias); REQ(kp_scale); REQ(kd_scale);
#undef REQ
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
    hipStream_t st = (hipStream_t)stream;
    return h->model.n_legs == 4 ? launch<4>(h, phases, actions, counter, st) : launch<2>(h, phases, actions, counter, st);
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

