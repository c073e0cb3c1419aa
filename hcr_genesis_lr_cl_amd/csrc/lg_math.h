// lg_math.h -- small fixed-size linear algebra for the gfx950 env-step kernel.
// Everything is by-value structs of floats so hipcc keeps them in VGPRs (no runtime-indexed
// arrays: cdna_hip_programming.md rule 20).
#pragma once
#include <hip/hip_runtime.h>

#define LG_DEV __device__ __forceinline__

struct V3 { float x, y, z; };
struct M3 { float xx, xy, xz, yx, yy, yz, zx, zy, zz; };   // row-major general 3x3
struct S3 { float xx, yy, zz, xy, xz, yz; };               // symmetric 3x3
struct V6 { V3 a, l; };                                    // spatial vector [angular; linear]

LG_DEV V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
LG_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
LG_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
LG_DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
LG_DEV V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
LG_DEV V3 operator*(float s, V3 a) { return v3(a.x * s, a.y * s, a.z * s); }
LG_DEV V3 &operator+=(V3 &a, V3 b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
LG_DEV V3 &operator-=(V3 &a, V3 b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; return a; }
LG_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
LG_DEV V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
LG_DEV float norm(V3 a) { return sqrtf(dot(a, a)); }

LG_DEV V3 mul(const M3 &m, V3 v) {
    return v3(m.xx * v.x + m.xy * v.y + m.xz * v.z, m.yx * v.x + m.yy * v.y + m.yz * v.z,
              m.zx * v.x + m.zy * v.y + m.zz * v.z);
}
LG_DEV V3 mulT(const M3 &m, V3 v) {
    return v3(m.xx * v.x + m.yx * v.y + m.zx * v.z, m.xy * v.x + m.yy * v.y + m.zy * v.z,
              m.xz * v.x + m.yz * v.y + m.zz * v.z);
}
LG_DEV M3 mul(const M3 &a, const M3 &b) {
    M3 r;
    r.xx = a.xx * b.xx + a.xy * b.yx + a.xz * b.zx; r.xy = a.xx * b.xy + a.xy * b.yy + a.xz * b.zy; r.xz = a.xx * b.xz + a.xy * b.yz + a.xz * b.zz;
    r.yx = a.yx * b.xx + a.yy * b.yx + a.yz * b.zx; r.yy = a.yx * b.xy + a.yy * b.yy + a.yz * b.zy; r.yz = a.yx * b.xz + a.yy * b.yz + a.yz * b.zz;
    r.zx = a.zx * b.xx + a.zy * b.yx + a.zz * b.zx; r.zy = a.zx * b.xy + a.zy * b.yy + a.zz * b.zy; r.zz = a.zx * b.xz + a.zy * b.yz + a.zz * b.zz;
    return r;
}
LG_DEV V3 mul(const S3 &s, V3 v) {
    return v3(s.xx * v.x + s.xy * v.y + s.xz * v.z, s.xy * v.x + s.yy * v.y + s.yz * v.z,
              s.xz * v.x + s.yz * v.y + s.zz * v.z);
}
LG_DEV S3 operator+(const S3 &a, const S3 &b) {
    S3 r = {a.xx + b.xx, a.yy + b.yy, a.zz + b.zz, a.xy + b.xy, a.xz + b.xz, a.yz + b.yz};
    return r;
}
LG_DEV M3 operator+(const M3 &a, const M3 &b) {
    M3 r = {a.xx + b.xx, a.xy + b.xy, a.xz + b.xz, a.yx + b.yx, a.yy + b.yy, a.yz + b.yz, a.zx + b.zx, a.zy + b.zy, a.zz + b.zz};
    return r;
}
// R S R^T for symmetric S
LG_DEV S3 rot_sym(const M3 &R, const S3 &s) {
    // T = R S
    float t00 = R.xx * s.xx + R.xy * s.xy + R.xz * s.xz, t01 = R.xx * s.xy + R.xy * s.yy + R.xz * s.yz, t02 = R.xx * s.xz + R.xy * s.yz + R.xz * s.zz;
    float t10 = R.yx * s.xx + R.yy * s.xy + R.yz * s.xz, t11 = R.yx * s.xy + R.yy * s.yy + R.yz * s.yz, t12 = R.yx * s.xz + R.yy * s.yz + R.yz * s.zz;
    float t20 = R.zx * s.xx + R.zy * s.xy + R.zz * s.xz, t21 = R.zx * s.xy + R.zy * s.yy + R.zz * s.yz, t22 = R.zx * s.xz + R.zy * s.yz + R.zz * s.zz;
    S3 r;
    r.xx = t00 * R.xx + t01 * R.xy + t02 * R.xz;
    r.yy = t10 * R.yx + t11 * R.yy + t12 * R.yz;
    r.zz = t20 * R.zx + t21 * R.zy + t22 * R.zz;
    r.xy = t00 * R.yx + t01 * R.yy + t02 * R.yz;
    r.xz = t00 * R.zx + t01 * R.zy + t02 * R.zz;
    r.yz = t10 * R.zx + t11 * R.zy + t12 * R.zz;
    return r;
}
// m (|c|^2 I - c c^T)
LG_DEV S3 parallel_axis(float m, V3 c) {
    S3 r = {m * (c.y * c.y + c.z * c.z), m * (c.x * c.x + c.z * c.z), m * (c.x * c.x + c.y * c.y),
            -m * c.x * c.y, -m * c.x * c.z, -m * c.y * c.z};
    return r;
}
// m [c]x  (so that mul(skew, v) = m c x v)
LG_DEV M3 skew(V3 h) {
    M3 r = {0.f, -h.z, h.y, h.z, 0.f, -h.x, -h.y, h.x, 0.f};
    return r;
}
LG_DEV M3 quat_to_mat(float x, float y, float z, float w) {
    M3 r = {1.f - 2.f * (y * y + z * z), 2.f * (x * y - z * w), 2.f * (x * z + y * w),
            2.f * (x * y + z * w), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - x * w),
            2.f * (x * z - y * w), 2.f * (y * z + x * w), 1.f - 2.f * (x * x + y * y)};
    return r;
}
// Rodrigues rotation about unit axis u by angle q
LG_DEV M3 axis_angle(V3 u, float q) {
    float s, c;
    __sincosf(q, &s, &c);
    float t = 1.f - c;
    M3 r = {c + t * u.x * u.x, t * u.x * u.y - s * u.z, t * u.x * u.z + s * u.y,
            t * u.x * u.y + s * u.z, c + t * u.y * u.y, t * u.y * u.z - s * u.x,
            t * u.x * u.z - s * u.y, t * u.y * u.z + s * u.x, c + t * u.z * u.z};
    return r;
}

// ---- articulated (general symmetric 6x6) inertia in blocks: [A B; B^T C] ----
struct I6 { S3 A; M3 B; S3 C; };
LG_DEV I6 operator+(const I6 &a, const I6 &b) { I6 r = {a.A + b.A, a.B + b.B, a.C + b.C}; return r; }
// I * [a; l]
LG_DEV V6 mul(const I6 &I, const V6 &v) {
    V6 r;
    r.a = mul(I.A, v.a) + mul(I.B, v.l);
    r.l = mulT(I.B, v.a) + mul(I.C, v.l);
    return r;
}
LG_DEV float dot(const V6 &a, const V6 &b) { return dot(a.a, b.a) + dot(a.l, b.l); }
LG_DEV V6 operator+(const V6 &a, const V6 &b) { V6 r = {a.a + b.a, a.l + b.l}; return r; }
LG_DEV V6 operator-(const V6 &a, const V6 &b) { V6 r = {a.a - b.a, a.l - b.l}; return r; }
LG_DEV V6 operator*(const V6 &a, float s) { V6 r = {a.a * s, a.l * s}; return r; }
// I - U U^T / D
LG_DEV I6 rank1_down(const I6 &I, const V6 &U, float dinv) {
    V3 ua = U.a * dinv, ul = U.l * dinv;
    I6 r;
    r.A.xx = I.A.xx - ua.x * U.a.x; r.A.yy = I.A.yy - ua.y * U.a.y; r.A.zz = I.A.zz - ua.z * U.a.z;
    r.A.xy = I.A.xy - ua.x * U.a.y; r.A.xz = I.A.xz - ua.x * U.a.z; r.A.yz = I.A.yz - ua.y * U.a.z;
    r.B.xx = I.B.xx - ua.x * U.l.x; r.B.xy = I.B.xy - ua.x * U.l.y; r.B.xz = I.B.xz - ua.x * U.l.z;
    r.B.yx = I.B.yx - ua.y * U.l.x; r.B.yy = I.B.yy - ua.y * U.l.y; r.B.yz = I.B.yz - ua.y * U.l.z;
    r.B.zx = I.B.zx - ua.z * U.l.x; r.B.zy = I.B.zy - ua.z * U.l.y; r.B.zz = I.B.zz - ua.z * U.l.z;
    r.C.xx = I.C.xx - ul.x * U.l.x; r.C.yy = I.C.yy - ul.y * U.l.y; r.C.zz = I.C.zz - ul.z * U.l.z;
    r.C.xy = I.C.xy - ul.x * U.l.y; r.C.xz = I.C.xz - ul.x * U.l.z; r.C.yz = I.C.yz - ul.y * U.l.z;
    return r;
}

// ---- 6x6 SPD Cholesky, fully unrolled, in registers --------------------------------------
struct Chol6 {
    float l00, l10, l11, l20, l21, l22, l30, l31, l32, l33, l40, l41, l42, l43, l44, l50, l51, l52, l53, l54, l55;
    float i0, i1, i2, i3, i4, i5;  // reciprocals of the diagonal
};
LG_DEV Chol6 chol6(const I6 &I) {
    // matrix rows: 0..2 angular (A | B), 3..5 linear (B^T | C)
    float a00 = I.A.xx, a10 = I.A.xy, a11 = I.A.yy, a20 = I.A.xz, a21 = I.A.yz, a22 = I.A.zz;
    float a30 = I.B.xx, a31 = I.B.yx, a32 = I.B.zx, a33 = I.C.xx;
    float a40 = I.B.xy, a41 = I.B.yy, a42 = I.B.zy, a43 = I.C.xy, a44 = I.C.yy;
    float a50 = I.B.xz, a51 = I.B.yz, a52 = I.B.zz, a53 = I.C.xz, a54 = I.C.yz, a55 = I.C.zz;
    Chol6 c;
    c.i0 = rsqrtf(a00); c.l00 = a00 * c.i0;
    c.l10 = a10 * c.i0; c.l20 = a20 * c.i0; c.l30 = a30 * c.i0; c.l40 = a40 * c.i0; c.l50 = a50 * c.i0;
    float d1 = a11 - c.l10 * c.l10;
    c.i1 = rsqrtf(d1); c.l11 = d1 * c.i1;
    c.l21 = (a21 - c.l20 * c.l10) * c.i1; c.l31 = (a31 - c.l30 * c.l10) * c.i1;
    c.l41 = (a41 - c.l40 * c.l10) * c.i1; c.l51 = (a51 - c.l50 * c.l10) * c.i1;
    float d2 = a22 - c.l20 * c.l20 - c.l21 * c.l21;
    c.i2 = rsqrtf(d2); c.l22 = d2 * c.i2;
    c.l32 = (a32 - c.l30 * c.l20 - c.l31 * c.l21) * c.i2;
    c.l42 = (a42 - c.l40 * c.l20 - c.l41 * c.l21) * c.i2;
    c.l52 = (a52 - c.l50 * c.l20 - c.l51 * c.l21) * c.i2;
    float d3 = a33 - c.l30 * c.l30 - c.l31 * c.l31 - c.l32 * c.l32;
    c.i3 = rsqrtf(d3); c.l33 = d3 * c.i3;
    c.l43 = (a43 - c.l40 * c.l30 - c.l41 * c.l31 - c.l42 * c.l32) * c.i3;
    c.l53 = (a53 - c.l50 * c.l30 - c.l51 * c.l31 - c.l52 * c.l32) * c.i3;
    float d4 = a44 - c.l40 * c.l40 - c.l41 * c.l41 - c.l42 * c.l42 - c.l43 * c.l43;
    c.i4 = rsqrtf(d4); c.l44 = d4 * c.i4;
    c.l54 = (a54 - c.l50 * c.l40 - c.l51 * c.l41 - c.l52 * c.l42 - c.l53 * c.l43) * c.i4;
    float d5 = a55 - c.l50 * c.l50 - c.l51 * c.l51 - c.l52 * c.l52 - c.l53 * c.l53 - c.l54 * c.l54;
    c.i5 = rsqrtf(d5); c.l55 = d5 * c.i5;
    return c;
}
// solve (L L^T) x = b
LG_DEV V6 chol6_solve(const Chol6 &c, const V6 &b) {
    float y0 = b.a.x * c.i0;
    float y1 = (b.a.y - c.l10 * y0) * c.i1;
    float y2 = (b.a.z - c.l20 * y0 - c.l21 * y1) * c.i2;
    float y3 = (b.l.x - c.l30 * y0 - c.l31 * y1 - c.l32 * y2) * c.i3;
    float y4 = (b.l.y - c.l40 * y0 - c.l41 * y1 - c.l42 * y2 - c.l43 * y3) * c.i4;
    float y5 = (b.l.z - c.l50 * y0 - c.l51 * y1 - c.l52 * y2 - c.l53 * y3 - c.l54 * y4) * c.i5;
    float x5 = y5 * c.i5;
    float x4 = (y4 - c.l54 * x5) * c.i4;
    float x3 = (y3 - c.l43 * x4 - c.l53 * x5) * c.i3;
    float x2 = (y2 - c.l32 * x3 - c.l42 * x4 - c.l52 * x5) * c.i2;
    float x1 = (y1 - c.l21 * x2 - c.l31 * x3 - c.l41 * x4 - c.l51 * x5) * c.i1;
    float x0 = (y0 - c.l10 * x1 - c.l20 * x2 - c.l30 * x3 - c.l40 * x4 - c.l50 * x5) * c.i0;
    V6 r = {v3(x0, x1, x2), v3(x3, x4, x5)};
    return r;
}

// 3x3 general solve by Cramer (well conditioned here: diagonally dominant contact systems)
LG_DEV bool solve3(const M3 &m, V3 b, V3 &x) {
    float c0 = m.yy * m.zz - m.yz * m.zy, c1 = m.yz * m.zx - m.yx * m.zz, c2 = m.yx * m.zy - m.yy * m.zx;
    float det = m.xx * c0 + m.xy * c1 + m.xz * c2;
    if (fabsf(det) < 1e-30f) return false;
    float inv = __builtin_amdgcn_rcpf(det);
    x.x = (b.x * c0 + m.xy * (m.yz * b.z - b.y * m.zz) + m.xz * (b.y * m.zy - m.yy * b.z)) * inv;
    x.y = (m.xx * (b.y * m.zz - m.yz * b.z) + b.x * c1 + m.xz * (m.yx * b.z - b.y * m.zx)) * inv;
    x.z = (m.xx * (m.yy * b.z - b.y * m.zy) + m.xy * (b.y * m.zx - m.yx * b.z) + b.x * c2) * inv;
    return true;
}

// ---- quad (4 consecutive lanes) all-reduce with DPP: no LDS traffic -----------------------
template <int LEGS> LG_DEV float quad_sum(float v) {
    if (LEGS >= 2) {
        int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
        v += __int_as_float(t);
    }
    if (LEGS >= 4) {
        int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
        v += __int_as_float(t);
    }
    return v;
}
template <int LEGS> LG_DEV V3 quad_sum(V3 v) { return v3(quad_sum<LEGS>(v.x), quad_sum<LEGS>(v.y), quad_sum<LEGS>(v.z)); }
template <int LEGS> LG_DEV V6 quad_sum(const V6 &v) { V6 r = {quad_sum<LEGS>(v.a), quad_sum<LEGS>(v.l)}; return r; }
template <int LEGS> LG_DEV I6 quad_sum(const I6 &I) {
    I6 r;
    r.A.xx = quad_sum<LEGS>(I.A.xx); r.A.yy = quad_sum<LEGS>(I.A.yy); r.A.zz = quad_sum<LEGS>(I.A.zz);
    r.A.xy = quad_sum<LEGS>(I.A.xy); r.A.xz = quad_sum<LEGS>(I.A.xz); r.A.yz = quad_sum<LEGS>(I.A.yz);
    r.B.xx = quad_sum<LEGS>(I.B.xx); r.B.xy = quad_sum<LEGS>(I.B.xy); r.B.xz = quad_sum<LEGS>(I.B.xz);
    r.B.yx = quad_sum<LEGS>(I.B.yx); r.B.yy = quad_sum<LEGS>(I.B.yy); r.B.yz = quad_sum<LEGS>(I.B.yz);
    r.B.zx = quad_sum<LEGS>(I.B.zx); r.B.zy = quad_sum<LEGS>(I.B.zy); r.B.zz = quad_sum<LEGS>(I.B.zz);
    r.C.xx = quad_sum<LEGS>(I.C.xx); r.C.yy = quad_sum<LEGS>(I.C.yy); r.C.zz = quad_sum<LEGS>(I.C.zz);
    r.C.xy = quad_sum<LEGS>(I.C.xy); r.C.xz = quad_sum<LEGS>(I.C.xz); r.C.yz = quad_sum<LEGS>(I.C.yz);
    return r;
}
template <int LEGS> LG_DEV int quad_or(int v) {
    if (LEGS >= 2) v |= __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);
    if (LEGS >= 4) v |= __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);
    return v;
}

// ---- Philox4x32-10 (Salmon et al. 2011), counter-based: draws are a pure function of
// (seed, global env id, step counter, slot) so results do not depend on sharding or on which
// envs reset.  Uniform mapping = torch's for float32: 24 random bits * 2^-24.
struct U4 { unsigned x, y, z, w; };
LG_DEV U4 philox4x32_10(U4 c, unsigned k0, unsigned k1) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
        // one v_mad_u64_u32 gives both halves of the 32x32 product (the compiler otherwise emits v_mul_hi_u32 + v_mul_lo_u32,
        // two quarter-rate instructions; Philox is ~40 % of the MDP phases' cycles)
        unsigned long long p0, p1, cy0, cy1;
        asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p0), "=s"(cy0) : "v"(c.x), "v"(0xD2511F53u));
        asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p1), "=s"(cy1) : "v"(c.z), "v"(0xCD9E8D57u));
        const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
        U4 n = {hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        c = n;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c;
}
LG_DEV float u01(unsigned r) { return (float)(r & 0xFFFFFFu) * (1.0f / 16777216.0f); }
