/* lg_oracle.c -- CPU restatement of the physics part of one control step.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under hcr_genesis_lr_cl_amd/ may import, link or call
 * this file; it is the checker for the HIP kernels (tests/, __graft_entry__.smoke(),
 * bench.py's cpu_baseline leg).
 *
 * PARITY UNPINNED at the Genesis boundary: the reference delegates rigid-body dynamics to
 * genesis-world==0.3.11 (pyproject.toml:40), which is neither in /root/reference nor
 * installable here, and the reference holds no test or golden vector for it.  What this file
 * restates is therefore the *published algorithm class* named by BASELINE.json's north_star
 * (Featherstone articulated-body algorithm, PD actuation, semi-implicit Euler, penalty contact
 * with Coulomb friction) applied at the reference's own call sites:
 *   - sub-stepping / actuation order ............ legged_gym/simulator/genesis_simulator.py:20-33
 *   - PD law, unclipped torque reported ......... genesis_simulator.py:630-642
 *   - options (dt, joint limits, collisions) .... genesis_simulator.py:229-257
 *   - state conventions (world-frame base twist in dofs 0-5, xyzw at the boundary)
 *                                                 genesis_simulator.py:40-46, 104-133
 *   - read-back quantities ...................... genesis_simulator.py:35-60
 *   - out-of-bounds teleport .................... genesis_simulator.py:612-628
 *   - quaternion helpers ........................ legged_gym/utils/math_utils.py:64-76, 90-109
 * It is pinned by physics invariants (tests/test_oracle_physics.py): free fall closed form,
 * momentum conservation, ABA == dense inverse-dynamics solve, static-stance force balance.
 *
 * Formulation (deliberately different from the HIP kernel, which works in world-aligned
 * axes about the base origin): textbook ABA in body-fixed link coordinates with 6x6 spatial
 * matrices (Featherstone, "Rigid Body Dynamics Algorithms", Table 7.1 + floating base 9.4).
 *
 * Precision: compile with -DLGO_REAL=double for an f64 reference; I/O buffers stay float32.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../include/lgsim.h"

#ifndef LGO_REAL
#define LGO_REAL float
#endif
typedef LGO_REAL real;

#define NB LG_MAX_BODIES
#define ND LG_MAX_DOF

/* ---------- small linear algebra ---------------------------------------------------- */
static void cross3(const real *a, const real *b, real *c) {
    real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    c[0] = x; c[1] = y; c[2] = z;
}
static real dot3(const real *a, const real *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void matvec3(const real *M, const real *v, real *o) { /* row-major */
    real x = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
    real y = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
    real z = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static void matTvec3(const real *M, const real *v, real *o) {
    real x = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
    real y = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
    real z = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static void matmul3(const real *A, const real *B, real *C) {
    real T[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(C, T, sizeof(T));
}
static void transpose3(const real *A, real *T) {
    real t[9] = {A[0], A[3], A[6], A[1], A[4], A[7], A[2], A[5], A[8]};
    memcpy(T, t, sizeof(t));
}
static void axis_angle(const real *u, real q, real *R) { /* Rodrigues */
    real c = cos(q), s = sin(q), t = 1 - c;
    R[0] = c + t * u[0] * u[0];        R[1] = t * u[0] * u[1] - s * u[2]; R[2] = t * u[0] * u[2] + s * u[1];
    R[3] = t * u[0] * u[1] + s * u[2]; R[4] = c + t * u[1] * u[1];        R[5] = t * u[1] * u[2] - s * u[0];
    R[6] = t * u[0] * u[2] - s * u[1]; R[7] = t * u[1] * u[2] + s * u[0]; R[8] = c + t * u[2] * u[2];
}
static void quat_to_mat(const real *q, real *R) { /* xyzw, body->world */
    real x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}

/* spatial: [ang; lin] */
static void crm(const real *v, const real *m, real *o) { /* v x m (motion) */
    real a[3], b[3], c[3];
    cross3(v, m, a);
    cross3(v, m + 3, b);
    cross3(v + 3, m, c);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2];
    o[3] = b[0] + c[0]; o[4] = b[1] + c[1]; o[5] = b[2] + c[2];
}
static void crf(const real *v, const real *f, real *o) { /* v x* f (force) */
    real a[3], b[3], c[3];
    cross3(v, f, a);
    cross3(v + 3, f + 3, b);
    cross3(v, f + 3, c);
    o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2];
    o[3] = c[0]; o[4] = c[1]; o[5] = c[2];
}
static void mv6(const real *M, const real *v, real *o) {
    real t[6];
    for (int i = 0; i < 6; i++) {
        real s = 0;
        for (int j = 0; j < 6; j++) s += M[6 * i + j] * v[j];
        t[i] = s;
    }
    memcpy(o, t, sizeof(t));
}
/* Pluecker motion transform as 6x6 from (E: parent->child rotation, r: child origin in parent) */
static void xform6(const real *E, const real *r, real *X) {
    real rx[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
    real Erx[9];
    matmul3(E, rx, Erx);
    memset(X, 0, 36 * sizeof(real));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            X[6 * i + j] = E[3 * i + j];
            X[6 * (i + 3) + j + 3] = E[3 * i + j];
            X[6 * (i + 3) + j] = -Erx[3 * i + j];
        }
}
static void rigid_inertia6(real m, const real *c, const real *I6, real *M) {
    real Ic[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]};
    real cx[9] = {0, -c[2], c[1], c[2], 0, -c[0], -c[1], c[0], 0};
    real cxT[9], cc[9];
    transpose3(cx, cxT);
    matmul3(cx, cxT, cc);
    memset(M, 0, 36 * sizeof(real));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            M[6 * i + j] = Ic[3 * i + j] + m * cc[3 * i + j];
            M[6 * i + j + 3] = m * cx[3 * i + j];
            M[6 * (i + 3) + j] = m * cxT[3 * i + j];
        }
    M[6 * 3 + 3] = M[6 * 4 + 4] = M[6 * 5 + 5] = m;
}
/* solve A x = b, A n x n (destroyed), partial pivoting */
static int solve_dense(real *A, real *b, int n) {
    for (int k = 0; k < n; k++) {
        int p = k;
        for (int i = k + 1; i < n; i++)
            if (fabs(A[i * n + k]) > fabs(A[p * n + k])) p = i;
        if (fabs(A[p * n + k]) < 1e-30) return 1;
        if (p != k) {
            for (int j = 0; j < n; j++) { real t = A[k * n + j]; A[k * n + j] = A[p * n + j]; A[p * n + j] = t; }
            real t = b[k]; b[k] = b[p]; b[p] = t;
        }
        for (int i = k + 1; i < n; i++) {
            real f = A[i * n + k] / A[k * n + k];
            for (int j = k; j < n; j++) A[i * n + j] -= f * A[k * n + j];
            b[i] -= f * b[k];
        }
    }
    for (int i = n - 1; i >= 0; i--) {
        real s = b[i];
        for (int j = i + 1; j < n; j++) s -= A[i * n + j] * b[j];
        b[i] = s / A[i * n + i];
    }
    return 0;
}

/* ---------- per-env working set -------------------------------------------------------- */
typedef struct {
    int nb, nd, jpl;             /* jpl: joints per leg = (n_bodies - 1) / n_legs */
    const LgModelDesc *m;
    real mass0, com0[3];         /* base with domain-randomised mass / com shift */
    real arm[ND], jdamp[ND], jfric[ND];
    /* kinematics */
    real Rw[NB][9], Pw[NB][3];   /* body->world rotation, world position of body origin */
    real E[NB][9], X[NB][36];    /* parent->child rotation, 6x6 motion transform */
    real S[NB][6], v[NB][6], c[NB][6];
    real I[NB][36];
    /* ABA factorisation */
    real IA[NB][36], U[NB][6], d[NB];
} Work;

static void kinematics(Work *w, const real *pos, const real *quat, const real *vw, const real *ww,
                       const real *q, const real *qd) {
    const LgModelDesc *m = w->m;
    quat_to_mat(quat, w->Rw[0]);
    for (int k = 0; k < 3; k++) w->Pw[0][k] = pos[k];
    matTvec3(w->Rw[0], ww, w->v[0]);
    matTvec3(w->Rw[0], vw, w->v[0] + 3);
    {
        real i6[6];
        for (int k = 0; k < 6; k++) i6[k] = m->inertia[0][k];
        rigid_inertia6(w->mass0, w->com0, i6, w->I[0]);
    }
    for (int i = 1; i < w->nb; i++) {
        int p = (i - 1) % w->jpl == 0 ? 0 : i - 1;
        real ax[3] = {m->axis[i][0], m->axis[i][1], m->axis[i][2]};
        real jr[9], Rj[9], Rpc[9], jp[3] = {m->jpos[i][0], m->jpos[i][1], m->jpos[i][2]};
        for (int k = 0; k < 9; k++) jr[k] = m->jrot[i][k];
        axis_angle(ax, q[i - 1], Rj);
        matmul3(jr, Rj, Rpc);
        transpose3(Rpc, w->E[i]);
        xform6(w->E[i], jp, w->X[i]);
        matmul3(w->Rw[p], Rpc, w->Rw[i]);
        real t[3];
        matvec3(w->Rw[p], jp, t);
        for (int k = 0; k < 3; k++) w->Pw[i][k] = w->Pw[p][k] + t[k];
        for (int k = 0; k < 3; k++) { w->S[i][k] = ax[k]; w->S[i][k + 3] = 0; }
        real vj[6];
        for (int k = 0; k < 6; k++) vj[k] = w->S[i][k] * qd[i - 1];
        mv6(w->X[i], w->v[p], w->v[i]);
        for (int k = 0; k < 6; k++) w->v[i][k] += vj[k];
        crm(w->v[i], vj, w->c[i]);
        real cm[3] = {m->com[i][0], m->com[i][1], m->com[i][2]};
        real i6[6] = {m->inertia[i][0], m->inertia[i][1], m->inertia[i][2], m->inertia[i][3], m->inertia[i][4], m->inertia[i][5]};
        rigid_inertia6(m->mass[i], cm, i6, w->I[i]);
    }
}
static real clampr(real x, real lim) { return x > lim ? lim : (x < -lim ? -lim : x); }
static int parent_of(const Work *w, int i) { return (i - 1) % w->jpl == 0 ? 0 : i - 1; }

/* articulated inertias; independent of forces */
static void aba_factor(Work *w) {
    for (int i = 0; i < w->nb; i++) memcpy(w->IA[i], w->I[i], sizeof(w->I[i]));
    for (int i = w->nb - 1; i >= 1; i--) {
        mv6(w->IA[i], w->S[i], w->U[i]);
        real d = 0;
        for (int k = 0; k < 6; k++) d += w->S[i][k] * w->U[i][k];
        w->d[i] = d + w->arm[i - 1];
        real Ia[36];
        for (int a = 0; a < 6; a++)
            for (int b = 0; b < 6; b++) Ia[6 * a + b] = w->IA[i][6 * a + b] - w->U[i][a] * w->U[i][b] / w->d[i];
        /* IA[p] += X^T Ia X */
        real T[36];
        for (int a = 0; a < 6; a++)
            for (int b = 0; b < 6; b++) {
                real s = 0;
                for (int k = 0; k < 6; k++) s += Ia[6 * a + k] * w->X[i][6 * k + b];
                T[6 * a + b] = s;
            }
        int p = parent_of(w, i);
        for (int a = 0; a < 6; a++)
            for (int b = 0; b < 6; b++) {
                real s = 0;
                for (int k = 0; k < 6; k++) s += w->X[i][6 * k + a] * T[6 * k + b];
                w->IA[p][6 * a + b] += s;
            }
    }
}

/* bias pass + acceleration pass.
 * with_vel: include velocity-product terms (c, v x* I v); tau may be NULL (zeros);
 * fext[i]: external spatial force on body i in body coordinates (may be NULL).
 * Outputs: a[i] spatial accelerations (body coords, true acceleration: gravity is an explicit
 * force), qdd. */
static void aba_solve(const Work *w, int with_vel, const real *tau, real (*fext)[6], real (*a)[6], real *qdd) {
    real pA[NB][6], u[NB];
    for (int i = 0; i < w->nb; i++) {
        if (with_vel) {
            real Iv[6];
            mv6(w->I[i], w->v[i], Iv);
            crf(w->v[i], Iv, pA[i]);
        } else
            memset(pA[i], 0, sizeof(pA[i]));
        if (fext)
            for (int k = 0; k < 6; k++) pA[i][k] -= fext[i][k];
    }
    for (int i = w->nb - 1; i >= 1; i--) {
        real sp = 0;
        for (int k = 0; k < 6; k++) sp += w->S[i][k] * pA[i][k];
        u[i] = (tau ? tau[i - 1] : 0) - sp;
        real pa[6], Iac[6] = {0, 0, 0, 0, 0, 0};
        if (with_vel) {
            /* Ia c = IA c - U (U^T c)/d */
            mv6(w->IA[i], w->c[i], Iac);
            real uc = 0;
            for (int k = 0; k < 6; k++) uc += w->U[i][k] * w->c[i][k];
            for (int k = 0; k < 6; k++) Iac[k] -= w->U[i][k] * uc / w->d[i];
        }
        for (int k = 0; k < 6; k++) pa[k] = pA[i][k] + Iac[k] + w->U[i][k] * u[i] / w->d[i];
        int p = parent_of(w, i);
        for (int k = 0; k < 6; k++) {
            real s = 0;
            for (int j = 0; j < 6; j++) s += w->X[i][6 * j + k] * pa[j];
            pA[p][k] += s;
        }
    }
    real A[36], b[6];
    memcpy(A, w->IA[0], sizeof(A));
    for (int k = 0; k < 6; k++) b[k] = -pA[0][k];
    solve_dense(A, b, 6);
    memcpy(a[0], b, sizeof(b));
    for (int i = 1; i < w->nb; i++) {
        int p = parent_of(w, i);
        real ap[6];
        mv6(w->X[i], a[p], ap);
        if (with_vel)
            for (int k = 0; k < 6; k++) ap[k] += w->c[i][k];
        real ua = 0;
        for (int k = 0; k < 6; k++) ua += w->U[i][k] * ap[k];
        qdd[i - 1] = (u[i] - ua) / w->d[i];
        for (int k = 0; k < 6; k++) a[i][k] = ap[k] + w->S[i][k] * qdd[i - 1];
    }
}

/* world force f at world point p on body i -> body-coordinate spatial force, accumulated */
static void add_world_force(const Work *w, int i, const real *p, const real *f, real (*fext)[6]) {
    real r[3] = {p[0] - w->Pw[i][0], p[1] - w->Pw[i][1], p[2] - w->Pw[i][2]};
    real n[3], nb[3], fb[3];
    cross3(r, f, n);
    matTvec3(w->Rw[i], n, nb);
    matTvec3(w->Rw[i], f, fb);
    for (int k = 0; k < 3; k++) { fext[i][k] += nb[k]; fext[i][k + 3] += fb[k]; }
}
/* world velocity of the material point at body-frame offset rl of body i */
static void point_velocity(const Work *w, int i, const real *rl, real *vout) {
    real t[3], vb[3];
    cross3(w->v[i], rl, t);
    for (int k = 0; k < 3; k++) vb[k] = w->v[i][k + 3] + t[k];
    matvec3(w->Rw[i], vb, vout);
}
/* world classical acceleration of that point given body spatial acceleration a (body coords) */
static void point_accel(const Work *w, int i, const real *rl, const real *a, int with_vel, real *aout) {
    real t[3], ab[3];
    cross3(a, rl, t);
    for (int k = 0; k < 3; k++) ab[k] = a[k + 3] + t[k];
    if (with_vel) {
        real wr[3], vp[3], cc[3];
        cross3(w->v[i], rl, wr);
        for (int k = 0; k < 3; k++) vp[k] = w->v[i][k + 3] + wr[k];
        cross3(w->v[i], vp, cc);
        for (int k = 0; k < 3; k++) ab[k] += cc[k];
    }
    matvec3(w->Rw[i], ab, aout);
}

/* terrain height + unit normal at world (x, y) */
static void terrain_at(const LgSimOptions *o, const int16_t *hf, real x, real y, real *h, real *n) {
    if (o->terrain_rows <= 0 || !hf) { *h = 0; n[0] = 0; n[1] = 0; n[2] = 1; return; }
    real gx = (x + o->border) / o->hscale, gy = (y + o->border) / o->hscale;
    int ix = (int)floor(gx), iy = (int)floor(gy);
    if (ix < 0) ix = 0; if (ix > o->terrain_rows - 2) ix = o->terrain_rows - 2;
    if (iy < 0) iy = 0; if (iy > o->terrain_cols - 2) iy = o->terrain_cols - 2;
    real fx = gx - ix, fy = gy - iy;
    if (fx < 0) fx = 0; if (fx > 1) fx = 1;
    if (fy < 0) fy = 0; if (fy > 1) fy = 1;
    int C = o->terrain_cols;
    real h00 = hf[ix * C + iy] * o->vscale, h10 = hf[(ix + 1) * C + iy] * o->vscale;
    real h01 = hf[ix * C + iy + 1] * o->vscale, h11 = hf[(ix + 1) * C + iy + 1] * o->vscale;
    *h = (h00 * (1 - fx) + h10 * fx) * (1 - fy) + (h01 * (1 - fx) + h11 * fx) * fy;
    real hx = ((h10 - h00) * (1 - fy) + (h11 - h01) * fy) / o->hscale;
    real hy = ((h01 - h00) * (1 - fx) + (h11 - h10) * fx) / o->hscale;
    real inv = 1 / sqrt(hx * hx + hy * hy + 1);
    n[0] = -hx * inv; n[1] = -hy * inv; n[2] = inv;
}

static void tangent_basis(const real *n, real *t1, real *t2) {
    real ex[3] = {1, 0, 0};
    real d = dot3(ex, n);
    for (int k = 0; k < 3; k++) t1[k] = ex[k] - d * n[k];
    real inv = 1 / sqrt(dot3(t1, t1));
    for (int k = 0; k < 3; k++) t1[k] *= inv;
    cross3(n, t1, t2);
}

/* math_utils.py:64-76 */
static void quat_rotate_inverse(const real *q, const real *v, real *o) {
    real qw = q[3], a[3], b[3], c[3];
    real s = 2 * qw * qw - 1;
    for (int k = 0; k < 3; k++) a[k] = v[k] * s;
    cross3(q, v, b);
    for (int k = 0; k < 3; k++) b[k] *= 2 * qw;
    real d = 2 * dot3(q, v);
    for (int k = 0; k < 3; k++) c[k] = q[k] * d;
    for (int k = 0; k < 3; k++) o[k] = a[k] - b[k] + c[k];
}
/* math_utils.py:90-109 */
static void get_euler_xyz(const real *q, real *e) {
    real qx = q[0], qy = q[1], qz = q[2], qw = q[3];
    real sinr = 2 * (qw * qx + qy * qz), cosr = qw * qw - qx * qx - qy * qy + qz * qz;
    e[0] = atan2(sinr, cosr);
    real sinp = 2 * (qw * qy - qz * qx);
    e[1] = fabs(sinp) >= 1 ? copysign((real)M_PI / 2, sinp) : asin(sinp);
    real siny = 2 * (qw * qz + qx * qy), cosy = qw * qw + qx * qx - qy * qy - qz * qz;
    e[2] = atan2(siny, cosy);
}

/* One env, one control step: genesis_simulator.py:20-60. */
static void env_step(const LgModelDesc *m, const LgSimOptions *o, const int16_t *hf, const LgBuffers *B,
                     const float *actions, int e) {
    Work w;
    memset(&w, 0, sizeof(w));
    const int nb = m->n_bodies, nd = nb - 1, L = m->n_links, F = m->n_legs;
    w.m = m; w.nb = nb; w.nd = nd; w.jpl = nd / F;
    real pos[3], quat[4], vw[3], ww[3], q[ND], qd[ND];
    for (int k = 0; k < 3; k++) { pos[k] = B->base_pos[3 * e + k]; vw[k] = B->base_lin_vel_w[3 * e + k]; ww[k] = B->base_ang_vel_w[3 * e + k]; }
    for (int k = 0; k < 4; k++) quat[k] = B->base_quat[4 * e + k];
    for (int k = 0; k < nd; k++) { q[k] = B->dof_pos[nd * e + k]; qd[k] = B->dof_vel[nd * e + k]; }

    /* "last" snapshots: genesis_simulator.py:21-24 */
    for (int k = 0; k < 3; k++) {
        B->last_base_lin_vel[3 * e + k] = B->base_lin_vel[3 * e + k];
        B->last_base_ang_vel[3 * e + k] = B->base_ang_vel[3 * e + k];
    }
    for (int k = 0; k < 3 * F; k++) B->last_feet_vel[3 * F * e + k] = B->feet_vel[3 * F * e + k];
    for (int k = 0; k < nd; k++) B->last_dof_vel[nd * e + k] = B->dof_vel[nd * e + k];

    /* domain-randomised base (set_mass_shift / set_COM_shift, genesis_simulator.py:677-702) */
    w.mass0 = m->mass[0] + (B->added_base_mass ? B->added_base_mass[e] : 0);
    for (int k = 0; k < 3; k++) w.com0[k] = m->com[0][k] + (B->base_com_bias ? B->base_com_bias[3 * e + k] : 0);
    for (int k = 0; k < nd; k++) {
        w.arm[k] = B->joint_armature ? B->joint_armature[e] : m->armature[k];
        w.jdamp[k] = B->joint_damping ? B->joint_damping[e] : m->damping[k];
        w.jfric[k] = B->joint_friction ? B->joint_friction[e] : m->frictionloss[k];
    }
    const real mu = o->terrain_friction * (B->friction_values ? B->friction_values[e] : 1);
    const real dt = o->dt, kc = o->contact_k, bc = o->contact_b, kappa = kc * dt + bc;
    real torque_raw[ND];
    real link_f[LG_MAX_LINKS][3];
    real Wc[LG_MAX_LEGS][9];
    memset(Wc, 0, sizeof(Wc));

    for (int sub = 0; sub < o->decimation; sub++) {
        kinematics(&w, pos, quat, vw, ww, q, qd);
        aba_factor(&w);
        memset(link_f, 0, sizeof(link_f));
        real fext[NB][6];
        memset(fext, 0, sizeof(fext));
        /* gravity as an explicit force at each body's centre of mass */
        for (int i = 0; i < nb; i++) {
            real cl[3], cwv[3], pw[3], g[3] = {0, 0, 0};
            real mi = i == 0 ? w.mass0 : m->mass[i];
            for (int k = 0; k < 3; k++) cl[k] = i == 0 ? w.com0[k] : m->com[i][k];
            matvec3(w.Rw[i], cl, cwv);
            for (int k = 0; k < 3; k++) pw[k] = w.Pw[i][k] + cwv[k];
            g[2] = mi * o->gravity_z;
            add_world_force(&w, i, pw, g, fext);
        }
        /* body (non-foot) spheres: penalty force made implicit with the conservative point
         * inverse mass sph_w (free-body bound) */
        for (int s = 0; s < m->n_spheres; s++) {
            int is_foot = 0;
            for (int l = 0; l < F; l++) is_foot |= (m->foot_sphere[l] == s);
            if (is_foot) continue;
            int i = m->sph_body[s];
            real rl[3] = {m->sph_pos[s][0], m->sph_pos[s][1], m->sph_pos[s][2]}, rw[3], c[3], h, n[3];
            matvec3(w.Rw[i], rl, rw);
            for (int k = 0; k < 3; k++) c[k] = w.Pw[i][k] + rw[k];
            terrain_at(o, hf, c[0], c[1], &h, n);
            real depth = m->sph_r[s] - (c[2] - h) * n[2];
            const int sole = w.jpl == 4 && i % w.jpl == 0;
            real vref = 0;
            if (sole) {
                /* spheres of a four-joint leg's foot body besides the foot sphere (the sole corners): they hold the foot flat while the
                 * sole centre (foot sphere, solved exactly below) carries the load -- penetration and approach velocity are measured
                 * relative to the centre's while that one is in the ground */
                int sc = m->foot_sphere[(i - 1) / w.jpl];
                real cl[3] = {m->sph_pos[sc][0], m->sph_pos[sc][1], m->sph_pos[sc][2]}, cw[3], cc[3], hc, nc[3], vc[3];
                matvec3(w.Rw[i], cl, cw);
                for (int k = 0; k < 3; k++) cc[k] = w.Pw[i][k] + cw[k];
                terrain_at(o, hf, cc[0], cc[1], &hc, nc);
                real dc = m->sph_r[sc] - (cc[2] - hc) * nc[2];
                if (dc > 0) { point_velocity(&w, i, cl, vc); depth -= dc; vref = dot3(vc, nc); }
            }
            if (depth <= -o->contact_margin) continue;
            real v[3];
            point_velocity(&w, i, rl, v);
            real vnf = dot3(v, n), vn = vnf - vref, wi = m->sph_w[s];
            if (sole) {
                /* inverse mass: twice the ankle joint's own compliance at the contact point (the two corners of an edge share that
                 * rotation), 2 (n . (s x (c - P)))^2 / (S^T I_foot S + armature), plus the model's sph_w */
                real Is[6], dd = 0, sw[3], rc[3], sxr[3];
                mv6(w.I[i], w.S[i], Is);
                for (int k = 0; k < 6; k++) dd += w.S[i][k] * Is[k];
                dd += w.arm[i - 1];
                matvec3(w.Rw[i], w.S[i], sw);
                for (int k = 0; k < 3; k++) rc[k] = c[k] - m->sph_r[s] * n[k] - w.Pw[i][k];
                cross3(sw, rc, sxr);
                real g = dot3(n, sxr);
                wi += 2 * g * g / dd;
            }
            real fn = (kc * depth - kappa * vn) / (1 + kappa * dt * wi);
            if (fn <= 0) continue;
            real vt[3];
            for (int k = 0; k < 3; k++) vt[k] = v[k] - vnf * n[k];
            real vtn = sqrt(dot3(vt, vt));
            real ft = vtn / (dt * wi);
            if (ft > mu * fn) ft = mu * fn;
            real f[3];
            for (int k = 0; k < 3; k++) f[k] = fn * n[k] - (vtn > 1e-9 ? ft * vt[k] / vtn : 0);
            real cp[3];
            for (int k = 0; k < 3; k++) cp[k] = c[k] - m->sph_r[s] * n[k];
            add_world_force(&w, i, cp, f, fext);
            for (int k = 0; k < 3; k++) link_f[m->sph_link[s]][k] += f[k];
        }
        /* actuation: genesis_simulator.py:630-642 (torque reported unclipped), effort clamp,
         * passive joint terms */
        real tau[ND];
        for (int k = 0; k < nd; k++) {
            real kps = B->kp_scale ? B->kp_scale[nd * e + k] : 1, kds = B->kd_scale ? B->kd_scale[nd * e + k] : 1;
            real a_s = (real)actions[nd * e + k] * o->action_scale;
            real t = kps * o->kp[k] * (a_s + o->default_dof_pos[k] - q[k]) - kds * o->kd[k] * qd[k];
            torque_raw[k] = t;
            real lim = m->effort[k];
            if (t > lim) t = lim;
            if (t < -lim) t = -lim;
            real sgn = qd[k] / (real)0.05;
            if (sgn > 1) sgn = 1;
            if (sgn < -1) sgn = -1;
            t -= w.jdamp[k] * qd[k] + w.jfric[k] * sgn;
            tau[k] = t;
        }
        real a[NB][6], qdd[ND];
        aba_solve(&w, 1, tau, fext, a, qdd);

        /* feet: implicit spring-damper normal + stick/slide Coulomb.  Each foot is solved against
         * its own 3x3 operational-space inverse inertia W_ii; the weak coupling between feet
         * (through the base) is resolved by `contact_iters` block-Jacobi sweeps, each followed
         * by an exact joint-space response pass for the current set of foot forces. */
        int active[LG_MAX_LEGS] = {0, 0, 0, 0};
        real cpw[LG_MAX_LEGS][3], cpl[LG_MAX_LEGS][3], nrm[LG_MAX_LEGS][3], t1[LG_MAX_LEGS][3], t2[LG_MAX_LEGS][3];
        real dep[LG_MAX_LEGS], vfree[LG_MAX_LEGS][3], fc[LG_MAX_LEGS][3];
        int any = 0;
        /* W (Wc, kept across sub-steps) is refreshed on sub-steps 0, k, 2k ... for EVERY foot (LgSimOptions.contact_w_every) */
        const int w_every = o->contact_w_every > 1 ? o->contact_w_every : 1;
        const int w_refresh = (sub % w_every) == 0;
        for (int l = 0; l < F; l++) {
            int s = m->foot_sphere[l], i = m->sph_body[s];
            real rl[3] = {m->sph_pos[s][0], m->sph_pos[s][1], m->sph_pos[s][2]}, rw[3], c[3], h;
            matvec3(w.Rw[i], rl, rw);
            for (int k = 0; k < 3; k++) c[k] = w.Pw[i][k] + rw[k];
            terrain_at(o, hf, c[0], c[1], &h, nrm[l]);
            dep[l] = m->sph_r[s] - (c[2] - h) * nrm[l][2];
            fc[l][0] = fc[l][1] = fc[l][2] = 0;
            const int act_l = dep[l] > -o->contact_margin;
            if (!act_l && !w_refresh) continue;
            if (act_l) { active[l] = 1; any = 1; }
            /* contact point = sphere surface point along -n; body-frame offset */
            real rel[3];
            for (int k = 0; k < 3; k++) { cpw[l][k] = c[k] - m->sph_r[s] * nrm[l][k]; rel[k] = cpw[l][k] - w.Pw[i][k]; }
            matTvec3(w.Rw[i], rel, cpl[l]);
            real v[3], af[3];
            point_velocity(&w, i, cpl[l], v);
            point_accel(&w, i, cpl[l], a[i], 1, af);
            tangent_basis(nrm[l], t1[l], t2[l]);
            const real *ax3[3] = {nrm[l], t1[l], t2[l]};
            for (int col = 0; w_refresh && col < 3; col++) {
                real fe[NB][6], ad[NB][6], qd2[ND], resp[3];
                memset(fe, 0, sizeof(fe));
                add_world_force(&w, i, cpw[l], ax3[col], fe);
                aba_solve(&w, 0, NULL, fe, ad, qd2);
                point_accel(&w, i, cpl[l], ad[i], 0, resp);
                for (int row = 0; row < 3; row++) Wc[l][3 * row + col] = dot3(ax3[row], resp);
            }
            for (int r = 0; r < 3; r++) {
                real vv[3];
                for (int k = 0; k < 3; k++) vv[k] = v[k] + dt * af[k];
                vfree[l][r] = dot3(ax3[r], vv);
            }
        }
        /* joint-limit stops (enable_joint_limit=True, genesis_simulator.py:249): one-sided implicit
         * spring-dampers solved in the same sweeps, each against its own joint-space inverse
         * inertia 1/d_i */
        int lim_on[ND]; real lim_e[ND], lim_s[ND], lim_T[ND];
        for (int k = 0; k < nd; k++) {
            lim_on[k] = 0; lim_T[k] = 0; lim_e[k] = 0; lim_s[k] = 0;
            if (q[k] < m->q_lo[k] + o->limit_margin) { lim_on[k] = 1; lim_e[k] = m->q_lo[k] - q[k]; lim_s[k] = 1; any = 1; }
            else if (q[k] > m->q_hi[k] - o->limit_margin) { lim_on[k] = 1; lim_e[k] = q[k] - m->q_hi[k]; lim_s[k] = -1; any = 1; }
        }
        if (any) {
            real ad[NB][6], qd2[ND];
            const real kl = o->limit_k, kapl = kl * dt + o->limit_b;
            memset(ad, 0, sizeof(ad));
            memset(qd2, 0, sizeof(qd2));
            int iters = o->contact_iters > 0 ? o->contact_iters : 1;
            for (int it = 0; it < iters; it++) {
                real fnew[LG_MAX_LEGS][3];
                for (int l = 0; l < F; l++) {
                    fnew[l][0] = fnew[l][1] = fnew[l][2] = 0;
                    if (!active[l]) continue;
                    int i = m->sph_body[m->foot_sphere[l]];
                    const real *ax3[3] = {nrm[l], t1[l], t2[l]};
                    real A[9], vo[3], resp[3];
                    for (int k = 0; k < 9; k++) A[k] = dt * Wc[l][k];
                    /* end-of-step velocity this foot would have without its own force */
                    point_accel(&w, i, cpl[l], ad[i], 0, resp);
                    for (int r = 0; r < 3; r++) {
                        vo[r] = vfree[l][r] + dt * dot3(ax3[r], resp);
                        for (int k = 0; k < 3; k++) vo[r] -= A[3 * r + k] * fc[l][k];
                    }
                    /* stick trial: 3x3 system */
                    real M3[9] = {1 + kappa * A[0], kappa * A[1], kappa * A[2], A[3], A[4], A[5], A[6], A[7], A[8]};
                    real fsol[3] = {kc * dep[l] - kappa * vo[0], -vo[1], -vo[2]};
                    if (solve_dense(M3, fsol, 3)) continue;
                    if (fsol[0] <= 0) continue;
                    real ftn = sqrt(fsol[1] * fsol[1] + fsol[2] * fsol[2]);
                    if (ftn > mu * fsol[0]) {
                        real e1 = fsol[1] / ftn, e2 = fsol[2] / ftn;
                        /* sliding: v_n,end = vo_n + a_eff f_n with a_eff = A_nn + mu (A_n1 e1 + A_n2 e2).  For mu large and a stretched leg
                         * the friction coupling can cancel A_nn (frictional jamming, Painleve): a_eff -> 0 or below and the penalty
                         * law's f_n = rn / (1 + kappa a_eff) grows without bound (observed: 39 kN on a foot, robot thrown 100 m).
                         * The coupling is therefore limited to three quarters of A_nn. */
                        real aeff = A[0] + mu * (A[1] * e1 + A[2] * e2);
                        if (aeff < (real)0.25 * A[0]) aeff = (real)0.25 * A[0];
                        real fn = (kc * dep[l] - kappa * vo[0]) / (1 + kappa * aeff);
                        if (fn <= 0) continue;
                        fsol[0] = fn; fsol[1] = mu * fn * e1; fsol[2] = mu * fn * e2;
                    }
                    for (int k = 0; k < 3; k++) fnew[l][k] = fsol[k];
                }
                real taul[ND];
                for (int k = 0; k < nd; k++) {
                    taul[k] = 0;
                    if (!lim_on[k]) continue;
                    /* vin: end-of-step velocity INTO the stop without the stop's own torque */
                    /* compliance seen by the stop: 1/d_i (parent held) on the first sweep; afterwards the
                     * response measured in the previous sweep (secant), which includes the give of the
                     * floating base -- without it an even number of sweeps ends on an under-estimate */
                    real C = 1 / w.d[k + 1];
                    if (it > 0 && lim_T[k] > 0) {
                        real Cm = lim_s[k] * qd2[k] / lim_T[k];
                        C = Cm > C ? (Cm < 8 * C ? Cm : 8 * C) : C;
                    }
                    real vin = -lim_s[k] * (qd[k] + dt * (qdd[k] + qd2[k])) + dt * lim_T[k] * C;
                    real T = (kl * lim_e[k] + kapl * vin) / (1 + kapl * dt * C);
                    lim_T[k] = T > 0 ? T : 0;
                    taul[k] = lim_s[k] * lim_T[k];
                }
                real ffoot[NB][6];
                memset(ffoot, 0, sizeof(ffoot));
                for (int l = 0; l < F; l++) {
                    for (int k = 0; k < 3; k++) fc[l][k] = fnew[l][k];
                    if (!active[l]) continue;
                    real f[3];
                    for (int k = 0; k < 3; k++) f[k] = fc[l][0] * nrm[l][k] + fc[l][1] * t1[l][k] + fc[l][2] * t2[l][k];
                    add_world_force(&w, m->sph_body[m->foot_sphere[l]], cpw[l], f, ffoot);
                }
                aba_solve(&w, 0, taul, ffoot, ad, qd2);
            }
            for (int l = 0; l < F; l++) {
                if (!active[l]) continue;
                for (int k = 0; k < 3; k++)
                    link_f[m->sph_link[m->foot_sphere[l]]][k] += fc[l][0] * nrm[l][k] + fc[l][1] * t1[l][k] + fc[l][2] * t2[l][k];
            }
            for (int k = 0; k < nd; k++) qdd[k] += qd2[k];
            for (int k = 0; k < 6; k++) a[0][k] += ad[0][k];
        }
        /* semi-implicit Euler: velocities first, then positions with the new velocities */
        real alin_b[3], t[3], acl_w[3], aang_w[3];
        cross3(w.v[0], w.v[0] + 3, t);
        for (int k = 0; k < 3; k++) alin_b[k] = a[0][k + 3] + t[k];
        matvec3(w.Rw[0], alin_b, acl_w);
        matvec3(w.Rw[0], a[0], aang_w);
        for (int k = 0; k < 3; k++) {
            vw[k] = clampr(vw[k] + dt * acl_w[k], o->max_base_lin_vel);
            ww[k] = clampr(ww[k] + dt * aang_w[k], o->max_base_ang_vel);
        }
        for (int k = 0; k < nd; k++) {
            qd[k] = clampr(qd[k] + dt * qdd[k], o->joint_vel_clamp * m->vel_limit[k]);
            q[k] += dt * qd[k];
        }
        for (int k = 0; k < 3; k++) pos[k] += dt * vw[k];
        real wn = sqrt(dot3(ww, ww)), dq[4];
        real half = (real)0.5 * wn * dt;
        real sc = wn > 1e-12 ? sin(half) / wn : (real)0.5 * dt;
        dq[0] = ww[0] * sc; dq[1] = ww[1] * sc; dq[2] = ww[2] * sc; dq[3] = cos(half);
        real nq[4];
        nq[3] = dq[3] * quat[3] - dq[0] * quat[0] - dq[1] * quat[1] - dq[2] * quat[2];
        nq[0] = dq[3] * quat[0] + quat[3] * dq[0] + dq[1] * quat[2] - dq[2] * quat[1];
        nq[1] = dq[3] * quat[1] + quat[3] * dq[1] + dq[2] * quat[0] - dq[0] * quat[2];
        nq[2] = dq[3] * quat[2] + quat[3] * dq[2] + dq[0] * quat[1] - dq[1] * quat[0];
        real inv = 1 / sqrt(nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3]);
        for (int k = 0; k < 4; k++) quat[k] = nq[k] * inv;
    }

    /* ---- read-back: genesis_simulator.py:35-60 ---- */
    {   /* non-finite state (cannot happen with the clamps above short of inf inputs): re-seat the robot */
        real chk = pos[0] + pos[1] + pos[2] + quat[0] + quat[1] + quat[2] + quat[3] + vw[0] + vw[1] + vw[2] + ww[0] + ww[1] + ww[2];
        for (int k = 0; k < nd; k++) chk += q[k] + qd[k];
        if (!isfinite(chk)) {
            for (int k = 0; k < 3; k++) { pos[k] = o->base_init_pos[k] + (B->env_origins ? B->env_origins[3 * e + k] : 0); vw[k] = 0; ww[k] = 0; }
            quat[0] = quat[1] = quat[2] = 0; quat[3] = 1;
            for (int k = 0; k < nd; k++) { q[k] = o->default_dof_pos[k]; qd[k] = 0; torque_raw[k] = 0; }
            memset(link_f, 0, sizeof(link_f));
            /* never silent (lgsim.h LgBuffers.nonfinite_count): counted, and the episode ends at this step's check_termination */
            if (B->nonfinite_count) {
#pragma omp atomic
                (*B->nonfinite_count)++;
            }
            if (B->fail_buf) B->fail_buf[e] = LG_FAIL_NONFINITE;
        }
    }
    if (pos[0] >= o->bound_x[1] || pos[0] <= o->bound_x[0] || pos[1] >= o->bound_y[1] || pos[1] <= o->bound_y[0])
        for (int k = 0; k < 3; k++) pos[k] = o->base_init_pos[k] + (B->env_origins ? B->env_origins[3 * e + k] : 0);
    for (int k = 0; k < 3; k++) { B->base_pos[3 * e + k] = pos[k]; B->base_lin_vel_w[3 * e + k] = vw[k]; B->base_ang_vel_w[3 * e + k] = ww[k]; }
    for (int k = 0; k < 4; k++) B->base_quat[4 * e + k] = quat[k];
    for (int k = 0; k < nd; k++) { B->dof_pos[nd * e + k] = q[k]; B->dof_vel[nd * e + k] = qd[k]; B->torques[nd * e + k] = torque_raw[k]; }
    real eul[3], lv[3], av[3], pg[3], g[3] = {0, 0, -1};
    get_euler_xyz(quat, eul);
    quat_rotate_inverse(quat, vw, lv);
    quat_rotate_inverse(quat, ww, av);
    quat_rotate_inverse(quat, g, pg);
    for (int k = 0; k < 3; k++) {
        B->base_euler[3 * e + k] = eul[k]; B->base_lin_vel[3 * e + k] = lv[k];
        B->base_ang_vel[3 * e + k] = av[k]; B->projected_gravity[3 * e + k] = pg[k];
    }
    for (int l = 0; l < L; l++)
        for (int k = 0; k < 3; k++) B->link_contact_forces[(L * e + l) * 3 + k] = link_f[l][k];
    /* feet frames at the final state; slot order = link order (feet_indices) */
    kinematics(&w, pos, quat, vw, ww, q, qd);
    int slot = 0;
    for (int l = 0; l < L; l++) {
        int is_foot = 0;
        for (int f = 0; f < F; f++) is_foot |= (m->foot_link[f] == l);
        if (!is_foot) continue;
        int i = m->link_body[l];
        real rl[3] = {m->link_pos[l][0], m->link_pos[l][1], m->link_pos[l][2]}, rw[3], v[3];
        matvec3(w.Rw[i], rl, rw);
        point_velocity(&w, i, rl, v);
        for (int k = 0; k < 3; k++) {
            B->feet_pos[(F * e + slot) * 3 + k] = w.Pw[i][k] + rw[k];
            B->feet_vel[(F * e + slot) * 3 + k] = v[k];
        }
        slot++;
    }
    if (B->link_contact_states) {
        int kk = 0, K = 0;
        for (int l = 0; l < L; l++) K += (m->state_link_mask >> l) & 1;
        for (int l = 0; l < L; l++) {
            if (!((m->state_link_mask >> l) & 1)) continue;
            real fn = sqrt(dot3(link_f[l], link_f[l]));
            B->link_contact_states[K * e + kk++] = fn > 1 ? 1.f : 0.f;
        }
    }
}

/* ---------- exported ------------------------------------------------------------------- */
/* simulator.step(actions) + simulator.post_physics_step() for envs [0, n_envs) on host buffers. */
int lgo_sim_step(const LgModelDesc *m, const LgSimOptions *o, const int16_t *hf, const LgBuffers *B,
                 const float *actions, int n_threads) {
    int n = B->n_envs;
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads > 0 ? n_threads : 1) schedule(static)
#endif
    for (int e = 0; e < n; e++) env_step(m, o, hf, B, actions, e);
    (void)n_threads;
    return 0;
}

/* forward dynamics of one env without contacts: qdd, base classical world acceleration.
 * method 0: ABA; method 1: dense solve of [H C] built column-by-column from inverse dynamics
 * (recursive Newton-Euler), an algorithm sharing no recursion with the ABA above. */
static void rnea(const Work *w, const real *a0, const real *qdd, int with_vel, real g_z, real *f0, real *tau) {
    real a[NB][6], f[NB][6];
    memcpy(a[0], a0, sizeof(a[0]));
    for (int i = 0; i < w->nb; i++) {
        if (i > 0) {
            int p = parent_of(w, i);
            mv6(w->X[i], a[p], a[i]);
            for (int k = 0; k < 6; k++) a[i][k] += w->S[i][k] * qdd[i - 1] + (with_vel ? w->c[i][k] : 0);
        }
        mv6(w->I[i], a[i], f[i]);
        if (with_vel) {
            real Iv[6], t[6];
            mv6(w->I[i], w->v[i], Iv);
            crf(w->v[i], Iv, t);
            for (int k = 0; k < 6; k++) f[i][k] += t[k];
        }
        if (g_z != 0) { /* subtract gravity force */
            real mi = i == 0 ? w->mass0 : w->m->mass[i], cl[3], gw[3] = {0, 0, mi * g_z}, gb[3], n[3];
            for (int k = 0; k < 3; k++) cl[k] = i == 0 ? w->com0[k] : w->m->com[i][k];
            matTvec3(w->Rw[i], gw, gb);
            cross3(cl, gb, n);
            for (int k = 0; k < 3; k++) { f[i][k] -= n[k]; f[i][k + 3] -= gb[k]; }
        }
    }
    for (int i = w->nb - 1; i >= 1; i--) {
        real s = 0;
        for (int k = 0; k < 6; k++) s += w->S[i][k] * f[i][k];
        tau[i - 1] = s + w->arm[i - 1] * qdd[i - 1];
        int p = parent_of(w, i);
        for (int k = 0; k < 6; k++) {
            real t = 0;
            for (int j = 0; j < 6; j++) t += w->X[i][6 * j + k] * f[i][j];
            f[p][k] += t;
        }
    }
    memcpy(f0, f[0], sizeof(f[0]));
}

int lgo_forward_dynamics(const LgModelDesc *m, const LgSimOptions *o, const float *pos, const float *quat,
                         const float *vw, const float *ww, const float *q, const float *qd, const float *tau_in,
                         int method, double *qdd_out, double *base_acc_out /* [ang_w(3), lin_w(3)] */) {
    Work w;
    memset(&w, 0, sizeof(w));
    const int nb = m->n_bodies, nd = nb - 1;
    w.m = m; w.nb = nb; w.nd = nd; w.jpl = nd / m->n_legs; w.mass0 = m->mass[0];
    for (int k = 0; k < 3; k++) w.com0[k] = m->com[0][k];
    for (int k = 0; k < nd; k++) w.arm[k] = m->armature[k];
    real p_[3], q_[4], v_[3], w_[3], qq[ND], qqd[ND], tau[ND];
    for (int k = 0; k < 3; k++) { p_[k] = pos[k]; v_[k] = vw[k]; w_[k] = ww[k]; }
    for (int k = 0; k < 4; k++) q_[k] = quat[k];
    for (int k = 0; k < nd; k++) { qq[k] = q[k]; qqd[k] = qd[k]; tau[k] = tau_in[k]; }
    kinematics(&w, p_, q_, v_, w_, qq, qqd);
    real a0[6], qdd[ND];
    if (method == 0) {
        aba_factor(&w);
        real fext[NB][6], a[NB][6];
        memset(fext, 0, sizeof(fext));
        for (int i = 0; i < nb; i++) {
            real cl[3], cwv[3], pw[3], g[3] = {0, 0, 0};
            for (int k = 0; k < 3; k++) cl[k] = i == 0 ? w.com0[k] : m->com[i][k];
            matvec3(w.Rw[i], cl, cwv);
            for (int k = 0; k < 3; k++) pw[k] = w.Pw[i][k] + cwv[k];
            g[2] = (i == 0 ? w.mass0 : m->mass[i]) * o->gravity_z;
            add_world_force(&w, i, pw, g, fext);
        }
        aba_solve(&w, 1, tau, fext, a, qdd);
        memcpy(a0, a[0], sizeof(a0));
    } else {
        int n = 6 + nd;
        real *H = (real *)calloc((size_t)n * n, sizeof(real)), *rhs = (real *)calloc(n, sizeof(real));
        real zero6[6] = {0}, zq[ND] = {0}, f0[6], tt[ND];
        rnea(&w, zero6, zq, 1, o->gravity_z, f0, tt); /* bias */
        for (int k = 0; k < 6; k++) rhs[k] = -f0[k];
        for (int k = 0; k < nd; k++) rhs[6 + k] = tau[k] - tt[k];
        for (int col = 0; col < n; col++) {
            real ea[6] = {0}, eq[ND] = {0};
            if (col < 6) ea[col] = 1; else eq[col - 6] = 1;
            rnea(&w, ea, eq, 0, 0, f0, tt);
            for (int k = 0; k < 6; k++) H[k * n + col] = f0[k];
            for (int k = 0; k < nd; k++) H[(6 + k) * n + col] = tt[k];
        }
        solve_dense(H, rhs, n);
        for (int k = 0; k < 6; k++) a0[k] = rhs[k];
        for (int k = 0; k < nd; k++) qdd[k] = rhs[6 + k];
        free(H); free(rhs);
    }
    real t[3], alin[3], o3[3];
    cross3(w.v[0], w.v[0] + 3, t);
    for (int k = 0; k < 3; k++) alin[k] = a0[k + 3] + t[k];
    matvec3(w.Rw[0], a0, o3);
    for (int k = 0; k < 3; k++) base_acc_out[k] = o3[k];
    matvec3(w.Rw[0], alin, o3);
    for (int k = 0; k < 3; k++) base_acc_out[3 + k] = o3[k];
    for (int k = 0; k < nd; k++) qdd_out[k] = qdd[k];
    return 0;
}

int lgo_real_bytes(void) { return (int)sizeof(real); }
