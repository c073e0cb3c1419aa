// lg_quad.h -- physics phase with ONE VECTOR COMPONENT PER LANE (included by lg_kernel.h).
//
// Why a second layout: the headline workload is 4096 envs per GPU.  With one leg per lane that is 256 waves on
// a chip with 1024 SIMDs -- three SIMDs of every CU idle while the fourth walks a ~20 k-instruction serial
// chain.  Here a leg is a QUAD of 4 lanes (x, y, z, w) and a Go2 env a DPP row of 16 lanes:
//   * 3-vectors keep component c in lane c, 3x3 matrices keep ROW c in lane c, the 6x6 articulated inertia keeps
//     rows c and c+3 in lane c.  matrix*vector and matrix*matrix become 3 / 9 fused DPP multiply-adds per lane
//     (operand broadcast inside the quad rides on the VALU instruction), cross products 3, dot products 3;
//   * per-joint scalars (q, qd, tau, limits, PD gains) keep joint j in lane j: the actuator law, the limit stops
//     and the integrator handle the three joints of a leg at once;
//   * the base quaternion keeps (x, y, z, w) in the four lanes;
//   * sums over legs are two DPP row rotations (row_ror:4, row_ror:8).
// The serial chain per wave shrinks ~2.7x and the launch has 4x the waves (one per SIMD at 4096 envs).  Total
// issue slots are ~1.5x those of the leg-per-lane kernel, so that one stays the choice for large batches
// (`LgSimOptions.sim_layout`; default: this layout while it needs at most two waves per SIMD, i.e. up to 8192 Go2 envs).
//
// All control flow is wave-uniform (ballot + scalar branch): a DPP read from a lane switched off by EXEC is
// undefined, so lanes are never masked off; dead lanes (past the last env) shadow the last env and only their
// stores are predicated.  Lane 3 of a quad carries no vector component; its vector values are "don't care" and never
// reach lanes 0-2 (quad permutes used for vector work map lane 3 onto itself).  It does carry the quaternion's w, and with
// four-joint legs (JPL = 4) the fourth joint's scalars; reductions are only formed in lanes 0-2, so what lane 3 needs from one
// is fetched from lane 0.
//
// Algorithm = lg_kernel.h's SIM phase statement for statement (same world-aligned ABA about the base origin,
// same contact / limit laws); tests/test_gpu_physics.py checks both layouts against the same f64 CPU restatement.
#pragma once

namespace q4 {

constexpr int QUAD_AXES[4] = {0, 1, 1, 1};   // hip-x / thigh-y / knee-y (/ ankle-y: four-joint legs)

LG_DEV float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }   // v_sqrt_f32 (1 ulp) without the denormal-range rescue of sqrtf

constexpr int QP(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }
template <int CTRL> LG_DEV float dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int J> LG_DEV float bc(float v) { return dpp<J * 0x55>(v); }    // lane J of the quad, to all four
LG_DEV float bcj(float v, int j) { return j == 0 ? bc<0>(v) : (j == 1 ? bc<1>(v) : (j == 2 ? bc<2>(v) : bc<3>(v))); }   // j: compile time after unrolling
LG_DEV float rot1(float v) { return dpp<QP(1, 2, 0, 3)>(v); }              // lane c <- lane (c+1)%3
LG_DEV float rot2(float v) { return dpp<QP(2, 0, 1, 3)>(v); }              // lane c <- lane (c+2)%3
LG_DEV float sum3(float t) { return t + rot1(t) + rot2(t); }               // x+y+z in lanes 0..2
LG_DEV float sum4(float t) { t += dpp<QP(1, 0, 3, 2)>(t); return t + dpp<QP(2, 3, 0, 1)>(t); }
// lane ^ 4 (the other leg of a biped env: 8 lanes) with two masked DPP moves -- quads 0 / 2 of a row take lane + 4 (row_shl:4, banks 0 and 2),
// quads 1 / 3 lane - 4 (row_shr:4, banks 1 and 3) -- instead of ds_swizzle: no LDS round trip (~100 cycles exposed wherever the result is
// needed at once: the sweeps of the contact solve, every reduction of the MDP tail)
LG_DEV int xor4i(int x) {
    int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);
    return __builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false);
}
LG_DEV float xor4(float v) { return __int_as_float(xor4i(__float_as_int(v))); }
template <int LEGS> LG_DEV float legsum(float v) {                         // sum over the legs of an env, to all
    if (LEGS == 4) { v += dpp<0x124>(v); v += dpp<0x128>(v); }             // row_ror:4, row_ror:8
    else if (LEGS == 2) v += xor4(v);
    return v;
}
LG_DEV int sum4i_or(int v) {
    v |= __builtin_amdgcn_update_dpp(0, v, QP(1, 0, 3, 2), 0xF, 0xF, true);
    return v | __builtin_amdgcn_update_dpp(0, v, QP(2, 3, 0, 1), 0xF, 0xF, true);
}
template <int LEGS> LG_DEV int env_or(int v) {
    v = sum4i_or(v);
    if (LEGS == 4) { v |= __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, true); v |= __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true); }
    else if (LEGS == 2) v |= xor4i(v);
    return v;
}

// 16 bytes of the kernel argument segment per lane (lane i: bytes [16 i, 16 i + 16), clamped to `n16` chunks)
LG_DEV uint4 kernarg_lane(int n16) {
#if defined(__HIP_DEVICE_COMPILE__)
    return reinterpret_cast<const uint4 *>(__builtin_amdgcn_kernarg_segment_ptr())[min((int)(threadIdx.x & 63u), n16 - 1)];
#else
    (void)n16;
    return make_uint4(0u, 0u, 0u, 0u);
#endif
}

struct QM { float c0, c1, c2; };       // 3x3: this lane's row
typedef float f2 __attribute__((ext_vector_type(2)));   // <2 x float>: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on an aligned register pair
struct QV6 { float a, l; };            // spatial vector [angular; linear], one component of each
struct QI6 { QM A, B, Bt, C; };        // [A B; B^T C]: rows c of (A|B) and of (B^T|C)

// lane constants
struct Lane {
    int c; bool is0, is1, is2, is3; float d0, d1, d2;
    LG_DEV float sel(float a0, float a1, float a2) const { return is0 ? a0 : (is1 ? a1 : a2); }
    LG_DEV float sel4(float a0, float a1, float a2, float a3) const { return is0 ? a0 : (is1 ? a1 : (is2 ? a2 : a3)); }
};

// NOTE on cross-lane reads (DPP broadcasts, ds_bpermute) and `?:`: the arms of a conditional expression are evaluated under the
// condition, and these intrinsics are not speculated -- `L.is0 ? bc<3>(a) : bc<3>(b)` becomes a divergent branch whose first arm runs
// with only the is0 lanes in EXEC, so the read from lane 3 returns 0 (bound_ctrl).  Cross-lane values are therefore always formed in
// their own statements (or as arguments of L.sel / L.sel4, which are evaluated before the call) and selected afterwards.
LG_DEV float dot3(float a, float b) { return sum3(a * b); }
// The quad broadcast of an operand rides ON the multiply-add (v_mul_f32_dpp / v_fmac_f32_dpp with quad_perm:[k,k,k,k] on src0).
// LLVM folds a DPP mov into v_mul / v_add but not into v_fmac, leaving one v_mov_b32_dpp per multiply-add (~20 % of the physics
// loop's issue slots); these helpers spell the sequences out.  One `s_nop 1` in front of a group covers the DPP read-after-VALU-
// write hazard of every source in it (2 wait states; the hazard recogniser does not look inside inline asm); inside a group
// the DPP sources are never written.  LG_NO_DPP_ASM selects the plain C++ forms (same arithmetic, same order).
#ifndef LG_NO_DPP_ASM
// The marker lets hcr_genesis_lr_cl_amd/dpp_hazard_pass.py (run by build.py on the compiler's assembly) drop the nop wherever the
// sources of the block turn out to be old enough in the instruction stream the compiler actually produced, and keep or shorten it
// where they are not -- an `s_nop 1` is two issue slots of a lone wave (~8.6 cycles), and most blocks do not need it.  Compiled without
// the pass (tools, plain hipcc) the nops simply stay.
#define LG_SNOP "s_nop 1 ; lg-dpp-hazard\n\t"
#define LG_QPR1 "quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf"
#define LG_QPR2 "quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf"
#define LG_QP0 "quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf"
#define LG_QP1 "quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf"
#define LG_QP2 "quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf"
// r = m0 * v[0] + m1 * v[1] + m2 * v[2]   (v: component layout, broadcast inside the quad)
LG_DEV float dpp_mac3(float m0, float m1, float m2, float v) {
    float r;
    asm(LG_SNOP "v_mul_f32_dpp %0, %4, %1 " LG_QP0 "\n\tv_fmac_f32_dpp %0, %4, %2 " LG_QP1 "\n\tv_fmac_f32_dpp %0, %4, %3 " LG_QP2
        : "=&v"(r) : "v"(m0), "v"(m1), "v"(m2), "v"(v));
    return r;
}
// r += m0 * v[0] + m1 * v[1] + m2 * v[2]
LG_DEV float dpp_mac3_acc(float r, float m0, float m1, float m2, float v) {
    asm(LG_SNOP "v_fmac_f32_dpp %0, %4, %1 " LG_QP0 "\n\tv_fmac_f32_dpp %0, %4, %2 " LG_QP1 "\n\tv_fmac_f32_dpp %0, %4, %3 " LG_QP2
        : "+v"(r) : "v"(m0), "v"(m1), "v"(m2), "v"(v));
    return r;
}
// r = m0 * a[K] + m1 * b[K] + m2 * c[K]   (one fixed lane K of three different vectors: rows of a transposed operand)
template <int K> LG_DEV float dpp_mac3t(float m0, float m1, float m2, float a, float b, float c) {
    float r;
    if (K == 0) asm(LG_SNOP "v_mul_f32_dpp %0, %4, %1 " LG_QP0 "\n\tv_fmac_f32_dpp %0, %5, %2 " LG_QP0 "\n\tv_fmac_f32_dpp %0, %6, %3 " LG_QP0
                    : "=&v"(r) : "v"(m0), "v"(m1), "v"(m2), "v"(a), "v"(b), "v"(c));
    else if (K == 1) asm(LG_SNOP "v_mul_f32_dpp %0, %4, %1 " LG_QP1 "\n\tv_fmac_f32_dpp %0, %5, %2 " LG_QP1 "\n\tv_fmac_f32_dpp %0, %6, %3 " LG_QP1
                         : "=&v"(r) : "v"(m0), "v"(m1), "v"(m2), "v"(a), "v"(b), "v"(c));
    else asm(LG_SNOP "v_mul_f32_dpp %0, %4, %1 " LG_QP2 "\n\tv_fmac_f32_dpp %0, %5, %2 " LG_QP2 "\n\tv_fmac_f32_dpp %0, %6, %3 " LG_QP2
             : "=&v"(r) : "v"(m0), "v"(m1), "v"(m2), "v"(a), "v"(b), "v"(c));
    return r;
}
LG_DEV float mulv(const QM &m, float v) { return dpp_mac3(m.c0, m.c1, m.c2, v); }
// Whole products in ONE asm block: a single `s_nop 1` in front covers the DPP read-after-write hazard of every source (none is written
// inside the block), instead of one per three-instruction group (each s_nop is an issue slot of the lone wave: the physics loop had ~190
// of them per sub-step).  The three accumulators are interleaved, so no instruction waits on the one before it.
LG_DEV QM mulmm(const QM &a, const QM &b) {     // a b: r.ck = a.c0 b.ck[0] + a.c1 b.ck[1] + a.c2 b.ck[2]
    QM r;
    asm(LG_SNOP
        "v_mul_f32_dpp %0, %6, %3 " LG_QP0 "\n\tv_mul_f32_dpp %1, %7, %3 " LG_QP0 "\n\tv_mul_f32_dpp %2, %8, %3 " LG_QP0 "\n\t"
        "v_fmac_f32_dpp %0, %6, %4 " LG_QP1 "\n\tv_fmac_f32_dpp %1, %7, %4 " LG_QP1 "\n\tv_fmac_f32_dpp %2, %8, %4 " LG_QP1 "\n\t"
        "v_fmac_f32_dpp %0, %6, %5 " LG_QP2 "\n\tv_fmac_f32_dpp %1, %7, %5 " LG_QP2 "\n\tv_fmac_f32_dpp %2, %8, %5 " LG_QP2
        : "=&v"(r.c0), "=&v"(r.c1), "=&v"(r.c2) : "v"(a.c0), "v"(a.c1), "v"(a.c2), "v"(b.c0), "v"(b.c1), "v"(b.c2));
    return r;
}
LG_DEV QM mulmmt(const QM &a, const QM &b) {    // a b^T: r.ck = a.c0 b.c0[k] + a.c1 b.c1[k] + a.c2 b.c2[k]
    QM r;
    asm(LG_SNOP
        "v_mul_f32_dpp %0, %6, %3 " LG_QP0 "\n\tv_mul_f32_dpp %1, %6, %3 " LG_QP1 "\n\tv_mul_f32_dpp %2, %6, %3 " LG_QP2 "\n\t"
        "v_fmac_f32_dpp %0, %7, %4 " LG_QP0 "\n\tv_fmac_f32_dpp %1, %7, %4 " LG_QP1 "\n\tv_fmac_f32_dpp %2, %7, %4 " LG_QP2 "\n\t"
        "v_fmac_f32_dpp %0, %8, %5 " LG_QP0 "\n\tv_fmac_f32_dpp %1, %8, %5 " LG_QP1 "\n\tv_fmac_f32_dpp %2, %8, %5 " LG_QP2
        : "=&v"(r.c0), "=&v"(r.c1), "=&v"(r.c2) : "v"(a.c0), "v"(a.c1), "v"(a.c2), "v"(b.c0), "v"(b.c1), "v"(b.c2));
    return r;
}
// two independent products a b and c d in one block (inv6: B C^-1 with C^-1 B^T, S^-1 T with Y S^-1)
LG_DEV void mulmm2(const QM &a, const QM &b, const QM &c, const QM &d, QM &r, QM &q) {
    asm(LG_SNOP
        "v_mul_f32_dpp %0, %9, %6 " LG_QP0 "\n\tv_mul_f32_dpp %1, %10, %6 " LG_QP0 "\n\tv_mul_f32_dpp %2, %11, %6 " LG_QP0 "\n\t"
        "v_mul_f32_dpp %3, %15, %12 " LG_QP0 "\n\tv_mul_f32_dpp %4, %16, %12 " LG_QP0 "\n\tv_mul_f32_dpp %5, %17, %12 " LG_QP0 "\n\t"
        "v_fmac_f32_dpp %0, %9, %7 " LG_QP1 "\n\tv_fmac_f32_dpp %1, %10, %7 " LG_QP1 "\n\tv_fmac_f32_dpp %2, %11, %7 " LG_QP1 "\n\t"
        "v_fmac_f32_dpp %3, %15, %13 " LG_QP1 "\n\tv_fmac_f32_dpp %4, %16, %13 " LG_QP1 "\n\tv_fmac_f32_dpp %5, %17, %13 " LG_QP1 "\n\t"
        "v_fmac_f32_dpp %0, %9, %8 " LG_QP2 "\n\tv_fmac_f32_dpp %1, %10, %8 " LG_QP2 "\n\tv_fmac_f32_dpp %2, %11, %8 " LG_QP2 "\n\t"
        "v_fmac_f32_dpp %3, %15, %14 " LG_QP2 "\n\tv_fmac_f32_dpp %4, %16, %14 " LG_QP2 "\n\tv_fmac_f32_dpp %5, %17, %14 " LG_QP2
        : "=&v"(r.c0), "=&v"(r.c1), "=&v"(r.c2), "=&v"(q.c0), "=&v"(q.c1), "=&v"(q.c2)
        : "v"(a.c0), "v"(a.c1), "v"(a.c2), "v"(b.c0), "v"(b.c1), "v"(b.c2), "v"(c.c0), "v"(c.c1), "v"(c.c2), "v"(d.c0), "v"(d.c1), "v"(d.c2));
}
#else
LG_DEV float mulv(const QM &m, float v) { return m.c0 * bc<0>(v) + m.c1 * bc<1>(v) + m.c2 * bc<2>(v); }
LG_DEV QM mulmm(const QM &a, const QM &b) {     // a b
    QM r;
    r.c0 = a.c0 * bc<0>(b.c0) + a.c1 * bc<1>(b.c0) + a.c2 * bc<2>(b.c0);
    r.c1 = a.c0 * bc<0>(b.c1) + a.c1 * bc<1>(b.c1) + a.c2 * bc<2>(b.c1);
    r.c2 = a.c0 * bc<0>(b.c2) + a.c1 * bc<1>(b.c2) + a.c2 * bc<2>(b.c2);
    return r;
}
LG_DEV QM mulmmt(const QM &a, const QM &b) {    // a b^T
    QM r;
    r.c0 = a.c0 * bc<0>(b.c0) + a.c1 * bc<0>(b.c1) + a.c2 * bc<0>(b.c2);
    r.c1 = a.c0 * bc<1>(b.c0) + a.c1 * bc<1>(b.c1) + a.c2 * bc<1>(b.c2);
    r.c2 = a.c0 * bc<2>(b.c0) + a.c1 * bc<2>(b.c1) + a.c2 * bc<2>(b.c2);
    return r;
}
LG_DEV void mulmm2(const QM &a, const QM &b, const QM &c, const QM &d, QM &r, QM &q) { r = mulmm(a, b); q = mulmm(c, d); }
#endif
// Cross products.  (a x b)_c = a_{c+1} b_{c+2} - a_{c+2} b_{c+1} = rot1(a) rot2(b) - rot2(a) rot1(b): with ONE operand given by its two quad
// rotations (QR, two v_mov_b32_dpp, shared by every product that operand enters) the other rides through DPP on a multiply and a
// multiply-add -- two instructions per product, and no rotation of the RESULT: the older form rot1(a rot1(b) - rot1(a) b) was three
// plus a DPP read of the value just computed, i.e. two wait states in front of whatever consumed it (~50 products per sub-step).
struct QR { float r1, r2; };                                               // rot1(v), rot2(v)
LG_DEV QR rots(float v) { QR r = {rot1(v), rot2(v)}; return r; }
#ifndef LG_NO_DPP_ASM
LG_DEV float cross(float a, const QR &b) {      // a x b, b by its rotations
    float r;
    asm(LG_SNOP "v_mul_f32_dpp %0, %1, %2 " LG_QPR1 "\n\tv_fmac_f32_dpp %0, %1, -%3 " LG_QPR2 : "=&v"(r) : "v"(a), "v"(b.r2), "v"(b.r1));
    return r;
}
LG_DEV float cross(const QR &a, float b) {      // a x b, a by its rotations
    float r;
    asm(LG_SNOP "v_mul_f32_dpp %0, %1, %2 " LG_QPR2 "\n\tv_fmac_f32_dpp %0, %1, -%3 " LG_QPR1 : "=&v"(r) : "v"(b), "v"(a.r1), "v"(a.r2));
    return r;
}
#else
LG_DEV float cross(float a, const QR &b) { return rot1(a) * b.r2 - rot2(a) * b.r1; }
LG_DEV float cross(const QR &a, float b) { return a.r1 * rot2(b) - a.r2 * rot1(b); }
#endif
LG_DEV float cross(float a, float b) { return cross(a, rots(b)); }
LG_DEV float multv(const Lane &L, const QM &m, float v) {   // m^T v
    return L.sel(sum3(m.c0 * v), sum3(m.c1 * v), sum3(m.c2 * v));
}
LG_DEV QM operator+(const QM &a, const QM &b) { QM r = {a.c0 + b.c0, a.c1 + b.c1, a.c2 + b.c2}; return r; }
LG_DEV QM operator-(const QM &a, const QM &b) { QM r = {a.c0 - b.c0, a.c1 - b.c1, a.c2 - b.c2}; return r; }
LG_DEV QM operator-(const QM &a) { QM r = {-a.c0, -a.c1, -a.c2}; return r; }
// rows of [v]x : S_cj = -[c == j-1] v_{j+1} + [c == j+1] v_{j+2}
LG_DEV QM skew(const Lane &L, float v) {
    const float v0 = bc<0>(v), v1 = bc<1>(v), v2 = bc<2>(v);
    QM r;
    r.c0 = L.d1 * v2 - L.d2 * v1;     // column 0: row 1 holds +v2, row 2 holds -v1
    r.c1 = L.d2 * v0 - L.d0 * v2;     // column 1: row 2 holds +v0, row 0 holds -v2
    r.c2 = L.d0 * v1 - L.d1 * v0;     // column 2: row 0 holds +v1, row 1 holds -v0
    return r;
}
// inverse of a symmetric 3x3 (rows): row c of the inverse = (row c+1) x (row c+2) / det
LG_DEV QM inv_sym(const Lane &L, const QM &m) {
    const float u0 = rot1(m.c0), u1 = rot1(m.c1), u2 = rot1(m.c2), w0 = rot2(m.c0), w1 = rot2(m.c1), w2 = rot2(m.c2);
    const float x0 = u1 * w2 - u2 * w1, x1 = u2 * w0 - u0 * w2, x2 = u0 * w1 - u1 * w0;
    float det = m.c0 * x0 + m.c1 * x1 + m.c2 * x2;
    det = L.is3 ? 1.f : det;
    const float inv = __builtin_amdgcn_rcpf(det);
    QM r = {x0 * inv, x1 * inv, x2 * inv};
    return r;
}
// LG_PK_F32 (off; kept for the record): the elementwise spatial-vector / 6x6 arithmetic and the rank-1 downdate written as
// packed FP32 (<2 x float>: v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32).  Measured slower -- the pairs cost v_mov_b32 to form and 64-bit
// tuples to allocate -- and the iterative-ilp build of it crashes clang's register allocator.
#ifdef LG_PK_F32
LG_DEV QV6 operator+(const QV6 &a, const QV6 &b) { const f2 r = f2{a.a, a.l} + f2{b.a, b.l}; QV6 o = {r.x, r.y}; return o; }
LG_DEV QV6 operator-(const QV6 &a, const QV6 &b) { const f2 r = f2{a.a, a.l} - f2{b.a, b.l}; QV6 o = {r.x, r.y}; return o; }
LG_DEV QV6 operator*(const QV6 &a, float s) { const f2 r = f2{a.a, a.l} * f2{s, s}; QV6 o = {r.x, r.y}; return o; }
#else
LG_DEV QV6 operator+(const QV6 &a, const QV6 &b) { QV6 r = {a.a + b.a, a.l + b.l}; return r; }
LG_DEV QV6 operator-(const QV6 &a, const QV6 &b) { QV6 r = {a.a - b.a, a.l - b.l}; return r; }
LG_DEV QV6 operator*(const QV6 &a, float s) { QV6 r = {a.a * s, a.l * s}; return r; }
#endif
LG_DEV float dot6(const QV6 &a, const QV6 &b) { return sum3(a.a * b.a + a.l * b.l); }
LG_DEV QV6 muli6(const QI6 &I, const QV6 &v) {
#ifndef LG_NO_DPP_ASM
    QV6 r;      // rows of (A | B) and of (B^T | C) against [v.a; v.l]: twelve multiply-adds, two interleaved accumulators, one s_nop
    asm(LG_SNOP
        "v_mul_f32_dpp %0, %14, %2 " LG_QP0 "\n\tv_mul_f32_dpp %1, %14, %8 " LG_QP0 "\n\t"
        "v_fmac_f32_dpp %0, %14, %3 " LG_QP1 "\n\tv_fmac_f32_dpp %1, %14, %9 " LG_QP1 "\n\t"
        "v_fmac_f32_dpp %0, %14, %4 " LG_QP2 "\n\tv_fmac_f32_dpp %1, %14, %10 " LG_QP2 "\n\t"
        "v_fmac_f32_dpp %0, %15, %5 " LG_QP0 "\n\tv_fmac_f32_dpp %1, %15, %11 " LG_QP0 "\n\t"
        "v_fmac_f32_dpp %0, %15, %6 " LG_QP1 "\n\tv_fmac_f32_dpp %1, %15, %12 " LG_QP1 "\n\t"
        "v_fmac_f32_dpp %0, %15, %7 " LG_QP2 "\n\tv_fmac_f32_dpp %1, %15, %13 " LG_QP2
        : "=&v"(r.a), "=&v"(r.l)
        : "v"(I.A.c0), "v"(I.A.c1), "v"(I.A.c2), "v"(I.B.c0), "v"(I.B.c1), "v"(I.B.c2),
          "v"(I.Bt.c0), "v"(I.Bt.c1), "v"(I.Bt.c2), "v"(I.C.c0), "v"(I.C.c1), "v"(I.C.c2), "v"(v.a), "v"(v.l));
    return r;
#else
    const float a0 = bc<0>(v.a), a1 = bc<1>(v.a), a2 = bc<2>(v.a), l0 = bc<0>(v.l), l1 = bc<1>(v.l), l2 = bc<2>(v.l);
    QV6 r;
    r.a = I.A.c0 * a0 + I.A.c1 * a1 + I.A.c2 * a2 + I.B.c0 * l0 + I.B.c1 * l1 + I.B.c2 * l2;
    r.l = I.Bt.c0 * a0 + I.Bt.c1 * a1 + I.Bt.c2 * a2 + I.C.c0 * l0 + I.C.c1 * l1 + I.C.c2 * l2;
    return r;
#endif
}
#ifdef LG_PK_F32
LG_DEV QI6 operator+(const QI6 &a, const QI6 &b) {
    QI6 r;
#define LG_PKADD(X, Y) { const f2 t = f2{a.X, a.Y} + f2{b.X, b.Y}; r.X = t.x; r.Y = t.y; }
    LG_PKADD(A.c0, Bt.c0) LG_PKADD(A.c1, Bt.c1) LG_PKADD(A.c2, Bt.c2) LG_PKADD(B.c0, C.c0) LG_PKADD(B.c1, C.c1) LG_PKADD(B.c2, C.c2)
#undef LG_PKADD
    return r;
}
// I - U U^T dinv: rows of (A | B) and (B^T | C) share the broadcast operand -> six v_pk_fma_f32
LG_DEV QI6 rank1_down(const QI6 &I, const QV6 &U, float dinv) {
    const f2 u = f2{U.a, U.l} * f2{-dinv, -dinv};
    const float a0 = bc<0>(U.a), a1 = bc<1>(U.a), a2 = bc<2>(U.a), l0 = bc<0>(U.l), l1 = bc<1>(U.l), l2 = bc<2>(U.l);
    QI6 r;
#define LG_PKFMA(X, Y, S) { const f2 t = __builtin_elementwise_fma(u, f2{S, S}, f2{I.X, I.Y}); r.X = t.x; r.Y = t.y; }
    LG_PKFMA(A.c0, Bt.c0, a0) LG_PKFMA(A.c1, Bt.c1, a1) LG_PKFMA(A.c2, Bt.c2, a2) LG_PKFMA(B.c0, C.c0, l0) LG_PKFMA(B.c1, C.c1, l1) LG_PKFMA(B.c2, C.c2, l2)
#undef LG_PKFMA
    return r;
}
#else
LG_DEV QI6 operator+(const QI6 &a, const QI6 &b) { QI6 r = {a.A + b.A, a.B + b.B, a.Bt + b.Bt, a.C + b.C}; return r; }
// I - U U^T dinv
LG_DEV QI6 rank1_down(const QI6 &I, const QV6 &U, float dinv) {
#ifndef LG_NO_DPP_ASM
    // entry (c, k) of each block -= u_c dinv u_k: the broadcast of u_k rides on v_fmac_f32_dpp (twelve instructions, one s_nop) instead
    // of six v_mov_b32_dpp + twelve v_fma
    const float na = -(U.a * dinv), nl = -(U.l * dinv);
    QI6 r = I;
    asm(LG_SNOP
        "v_fmac_f32_dpp %0, %12, %14 " LG_QP0 "\n\tv_fmac_f32_dpp %1, %12, %14 " LG_QP1 "\n\tv_fmac_f32_dpp %2, %12, %14 " LG_QP2 "\n\t"
        "v_fmac_f32_dpp %3, %13, %14 " LG_QP0 "\n\tv_fmac_f32_dpp %4, %13, %14 " LG_QP1 "\n\tv_fmac_f32_dpp %5, %13, %14 " LG_QP2 "\n\t"
        "v_fmac_f32_dpp %6, %12, %15 " LG_QP0 "\n\tv_fmac_f32_dpp %7, %12, %15 " LG_QP1 "\n\tv_fmac_f32_dpp %8, %12, %15 " LG_QP2 "\n\t"
        "v_fmac_f32_dpp %9, %13, %15 " LG_QP0 "\n\tv_fmac_f32_dpp %10, %13, %15 " LG_QP1 "\n\tv_fmac_f32_dpp %11, %13, %15 " LG_QP2
        : "+v"(r.A.c0), "+v"(r.A.c1), "+v"(r.A.c2), "+v"(r.B.c0), "+v"(r.B.c1), "+v"(r.B.c2),
          "+v"(r.Bt.c0), "+v"(r.Bt.c1), "+v"(r.Bt.c2), "+v"(r.C.c0), "+v"(r.C.c1), "+v"(r.C.c2)
        : "v"(U.a), "v"(U.l), "v"(na), "v"(nl));
    return r;
#else
    const float ua = U.a * dinv, ul = U.l * dinv;
    const float a0 = bc<0>(U.a), a1 = bc<1>(U.a), a2 = bc<2>(U.a), l0 = bc<0>(U.l), l1 = bc<1>(U.l), l2 = bc<2>(U.l);
    QI6 r;
    r.A.c0 = I.A.c0 - ua * a0; r.A.c1 = I.A.c1 - ua * a1; r.A.c2 = I.A.c2 - ua * a2;
    r.B.c0 = I.B.c0 - ua * l0; r.B.c1 = I.B.c1 - ua * l1; r.B.c2 = I.B.c2 - ua * l2;
    r.Bt.c0 = I.Bt.c0 - ul * a0; r.Bt.c1 = I.Bt.c1 - ul * a1; r.Bt.c2 = I.Bt.c2 - ul * a2;
    r.C.c0 = I.C.c0 - ul * l0; r.C.c1 = I.C.c1 - ul * l1; r.C.c2 = I.C.c2 - ul * l2;
    return r;
#endif
}
#endif
// inverse of an SPD 6x6 by Schur complement on the C block
LG_DEV QI6 inv6(const Lane &L, const QI6 &N) {
    const QM Ci = inv_sym(L, N.C);
    QM T, Y, SiT, YSi;
    mulmm2(N.B, Ci, Ci, N.Bt, T, Y);        // B C^-1 and C^-1 B^T = T^T, one block
    const QM S = N.A - mulmm(T, N.Bt);      // A - B C^-1 B^T
    const QM Si = inv_sym(L, S);
    mulmm2(Si, T, Y, Si, SiT, YSi);
    QI6 r;
    r.A = Si;
    r.B = -SiT;
    r.Bt = -YSi;
    r.C = Ci - mulmm(Y, r.B);
    return r;
}
// rigid-body spatial inertia about O (world axes): mass m, centre cw, rotational inertia Icw about the centre
LG_DEV QI6 rigid(const Lane &L, float m, float cw, const QM &Icw) {
    const float cc = dot3(cw, cw), mc = m * cw, mcc = m * cc;
    QI6 r;
    r.A.c0 = Icw.c0 + mcc * L.d0 - mc * bc<0>(cw);
    r.A.c1 = Icw.c1 + mcc * L.d1 - mc * bc<1>(cw);
    r.A.c2 = Icw.c2 + mcc * L.d2 - mc * bc<2>(cw);
    r.B = skew(L, mc);
    r.Bt = -r.B;
    r.C.c0 = m * L.d0; r.C.c1 = m * L.d1; r.C.c2 = m * L.d2;
    return r;
}
LG_DEV QM quat_rows(const Lane &L, float qv) {   // qv: (x, y, z, w) across the quad
    const float w = bc<3>(qv), w2 = 2.f * w, a = w2 * w - 1.f, v2 = 2.f * qv;
    const QM S = skew(L, qv);
    QM r;
    r.c0 = a * L.d0 + w2 * S.c0 + v2 * bc<0>(qv);
    r.c1 = a * L.d1 + w2 * S.c1 + v2 * bc<1>(qv);
    r.c2 = a * L.d2 + w2 * S.c2 + v2 * bc<2>(qv);
    return r;
}

// joint axis in world axes and child rotation R = Rp * Rot(axis, angle).  Axis code AK (compile time): 0/1/2 = the joint axis
// is +-e_x / e_y / e_z of the parent frame -- then the axis is a (signed) column of Rp and the product only mixes the other
// two columns, 5 instructions instead of 30; -1 = general axis (Rodrigues + 3x3 product).  This kernel is built for the
// hip-x / thigh-y / knee-y chains of both robots in scope (QUAD_AXES); lg_step checks the model and falls back to the
// leg-per-lane kernel, which takes any axes and joint frames, otherwise.
// `ax`: the axis, this lane's component; `cq`, `sq`: cos / sin of the joint angle, replicated.
template <int AK> LG_DEV void joint_rot(const Lane &L, const QM &Rp, float ax, float cq, float sq, QM &R, float &s) {
    if (AK == 0) { const float g = bc<0>(ax), sg = sq * g; s = Rp.c0 * g; R.c0 = Rp.c0; R.c1 = cq * Rp.c1 + sg * Rp.c2; R.c2 = cq * Rp.c2 - sg * Rp.c1; }
    else if (AK == 1) { const float g = bc<1>(ax), sg = sq * g; s = Rp.c1 * g; R.c1 = Rp.c1; R.c0 = cq * Rp.c0 - sg * Rp.c2; R.c2 = cq * Rp.c2 + sg * Rp.c0; }
    else if (AK == 2) { const float g = bc<2>(ax), sg = sq * g; s = Rp.c2 * g; R.c2 = Rp.c2; R.c0 = cq * Rp.c0 + sg * Rp.c1; R.c1 = cq * Rp.c1 - sg * Rp.c0; }
    else {
        s = mulv(Rp, ax);
        const QM K1 = skew(L, ax);
        const float tq = 1.f - cq, a0 = tq * ax;
        QM Rl;   // Rodrigues: c I + s [ax]x + (1 - c) ax ax^T
        Rl.c0 = cq * L.d0 + sq * K1.c0 + a0 * bc<0>(ax);
        Rl.c1 = cq * L.d1 + sq * K1.c1 + a0 * bc<1>(ax);
        Rl.c2 = cq * L.d2 + sq * K1.c2 + a0 * bc<2>(ax);
        R = mulmm(Rp, Rl);
    }
}

// store / load at base + a 32-bit BYTE offset (base wave-uniform: the saddr form of global_store / global_load)
template <class T> LG_DEV void stq(T *base, unsigned off, T v) { *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + off) = v; }
template <class T> LG_DEV T ldq(const T *base, unsigned off) { return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + off); }

struct QJoint { QV6 S, U, c; float dinv, u; };
struct QKin { QM R; float P; QV6 V; QR Wr; };   // Wr: the rotations of V.a (cross products)
struct Terr { int rows, cols; float border, ihs, vscale; const int16_t *hf; };

// terrain height and unit normal at world (x, y) -- scalar form, any lane
// HF: the caller knows there is a heightfield (rough task profiles, host-checked): no branch in front of the loads, so the loads of
// several lookups can be scheduled together
template <bool HF = false> LG_DEV void terrain(const Terr &T, float x, float y, float &h, float &nx, float &ny, float &nz) {
    if (!HF && T.rows <= 0) { h = 0.f; nx = ny = 0.f; nz = 1.f; return; }
    const float gx = (x + T.border) * T.ihs, gy = (y + T.border) * T.ihs;
    int ix = (int)floorf(gx), iy = (int)floorf(gy);
    ix = min(max(ix, 0), T.rows - 2);
    iy = min(max(iy, 0), T.cols - 2);
    const float fx = fminf(fmaxf(gx - ix, 0.f), 1.f), fy = fminf(fmaxf(gy - iy, 0.f), 1.f);
    const int16_t *q = T.hf + ix * T.cols + iy;
    const float h00 = q[0] * T.vscale, h10 = q[T.cols] * T.vscale, h01 = q[1] * T.vscale, h11 = q[T.cols + 1] * T.vscale;
    h = (h00 * (1 - fx) + h10 * fx) * (1 - fy) + (h01 * (1 - fx) + h11 * fx) * fy;
    const float hx = ((h10 - h00) * (1 - fy) + (h11 - h01) * fy) * T.ihs;
    const float hy = ((h01 - h00) * (1 - fx) + (h11 - h10) * fx) * T.ihs;
    const float inv = rsqrtf(hx * hx + hy * hy + 1.f);
    nx = -hx * inv; ny = -hy * inv; nz = inv;
}

// TWO lookups at once, one in each half of packed-f32 pairs: the float arithmetic of a lookup (cell coordinates, bilinear height, slope,
// normal: ~35 of its ~55 instructions) is plain per-lane arithmetic, which a lone wave issues in pairs at the price of one (tools/ubench/
// pk_issue.hip).  Same statements as terrain<true>, half by half; the cell index, the four loads and the rsq stay per lookup.
LG_DEV void terrain2(const Terr &T, f2 x, f2 y, f2 &h, f2 &nx, f2 &ny, f2 &nz) {
    const f2 gx = (x + T.border) * T.ihs, gy = (y + T.border) * T.ihs;
    int ix0 = (int)floorf(gx.x), ix1 = (int)floorf(gx.y), iy0 = (int)floorf(gy.x), iy1 = (int)floorf(gy.y);
    ix0 = min(max(ix0, 0), T.rows - 2); ix1 = min(max(ix1, 0), T.rows - 2);
    iy0 = min(max(iy0, 0), T.cols - 2); iy1 = min(max(iy1, 0), T.cols - 2);
    const f2 dx = gx - f2{(float)ix0, (float)ix1}, dy = gy - f2{(float)iy0, (float)iy1};
    const f2 fx = {fminf(fmaxf(dx.x, 0.f), 1.f), fminf(fmaxf(dx.y, 0.f), 1.f)}, fy = {fminf(fmaxf(dy.x, 0.f), 1.f), fminf(fmaxf(dy.y, 0.f), 1.f)};
    const int16_t *q0 = T.hf + ix0 * T.cols + iy0, *q1 = T.hf + ix1 * T.cols + iy1;
    const f2 h00 = f2{(float)q0[0], (float)q1[0]} * T.vscale, h10 = f2{(float)q0[T.cols], (float)q1[T.cols]} * T.vscale;
    const f2 h01 = f2{(float)q0[1], (float)q1[1]} * T.vscale, h11 = f2{(float)q0[T.cols + 1], (float)q1[T.cols + 1]} * T.vscale;
    const f2 omx = 1.f - fx, omy = 1.f - fy;
    h = (h00 * omx + h10 * fx) * omy + (h01 * omx + h11 * fx) * fy;
    const f2 hx = ((h10 - h00) * omy + (h11 - h01) * fy) * T.ihs;
    const f2 hy = ((h01 - h00) * omx + (h11 - h10) * fx) * T.ihs;
    const f2 s = hx * hx + hy * hy + 1.f;
    const f2 inv = {rsqrtf(s.x), rsqrtf(s.y)};
    nx = -hx * inv; ny = -hy * inv; nz = inv;
}

// response sweeps (see resp_up / resp_down in lg_kernel.h); du replicated, tl / dqdd in joint lanes
template <int NJ> LG_DEV QV6 resp_up(const QJoint (&J)[NJ], const QV6 &fspat, float tl, float (&du)[NJ]) {
    QV6 dp = {-fspat.a, -fspat.l};
#pragma unroll
    for (int j = NJ - 1; j >= 0; j--) { du[j] = bcj(tl, j) - dot6(J[j].S, dp); dp = dp + J[j].U * (du[j] * J[j].dinv); }
    return dp;
}
template <int NJ> LG_DEV QV6 resp_down(const Lane &L, const QJoint (&J)[NJ], const QV6 &a0, const float (&du)[NJ], float &dqdd) {
    QV6 a = a0;
    float d[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) { d[j] = (du[j] - dot6(J[j].U, a)) * J[j].dinv; a = a + J[j].S * d[j]; }
    // dot products are only formed in lanes 0-2: the fourth joint's lane takes its value from lane 0
    dqdd = NJ == 4 ? L.sel4(d[0], d[1], d[2], bc<0>(d[NJ - 1])) : L.sel(d[0], d[1], d[2]);
    return a;
}

// Blank the observation histories of the envs flagged in `rm` (a ballot: one lane per env) with ALL 64 lanes of the wave: a reset env
// restarts its stacks from zeros (legged_robot_ee.py / go2_wtw.py:174-178) -- 1.5 k floats for the 10-frame stacks of tron1_pf_ee.  Left
// to the env's own 8 or 16 lanes that is 100-190 stores per lane in a wave that has a reset in it, and a launch ends with its slowest wave.
// `e`: this lane's env; obs / priv: start of this launch's window in row 0 of the set being written; n_o / n_p: floats to zero per row.
// n floats from p (4-byte aligned) to zero with the wave's 64 lanes: the unaligned head (< 4 floats), 16-byte stores, the tail.  `p` and `n`
// are wave-uniform, so every lane takes the same path.
LG_DEV void zero_run(float *p, int n, int wl) {
    const int head = min(n, (int)((16u - ((unsigned)(uintptr_t)p & 15u)) & 15u) >> 2);
    if (wl < head) p[wl] = 0.f;
    float4 *q4p = reinterpret_cast<float4 *>(p + head);
    const int n4 = (n - head) >> 2;
    for (int i = wl; i < n4; i += 64) q4p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int done = head + 4 * n4;
    if (wl < n - done) p[done + wl] = 0.f;
}
LG_DEV void blank_histories(unsigned long long rm, int e, float *obs, size_t orow, int n_o, float *priv, size_t prow, int n_p) {
    const int wl = (int)threadIdx.x & 63;
    while (rm) {
        const int bit = __builtin_ctzll(rm);
        rm &= rm - 1;
        const int er = __builtin_amdgcn_readlane(e, bit);
        zero_run(obs + (size_t)er * orow, n_o, wl);
        zero_run(priv + (size_t)er * prow, n_p, wl);
    }
}

}  // namespace q4

// ---------------------------------------------------------------------------------------------
// MPH: MDP phases (LG_PHASE_POST, LG_PHASE_POST | LG_PHASE_RESET or 0) run in the tail of the same launch by the first
// 16 lanes of each wave, one per leg of the wave's envs, through env_step_body: one launch per control step, no second
// ramp-up, tables already in LDS.  The hand-off goes through the state arrays themselves (written above, L2-hot).
// JPL = 4 (TRON1 sole foot: two legs of four joints, physics only -- its MDP phases are the leg-per-lane launch): the fourth joint's
// scalars live in lane 3 of the quad, the chain arrays have four entries, the foot body's sole corners take the calf's second sphere
// slot with the sole law of lg_kernel.h's sphere_contact.
// INJ (test instantiations, lg_step without LG_PHASE_SIM on sim_layout 2): the golden vectors of tests/golden/*_mdp.npz go through the
// component-layout tails themselves -- the sub-step loop and the read-back are skipped, what they would have left in registers (twists,
// torques, contact forces, feet, terrain samples) is loaded from the bound buffers the test filled with the reference's recorded values,
// and every uniform comes from LgBuffers.rand_in instead of Philox.  The MDP statements are the very ones the product instantiation runs.
// RS (host-checked: the task's reward set IS its profile's default, lg_host.hip rs_mask): the set of active reward terms is a compile-time
// constant.  The terms are evaluated behind one scalar test each -- thirty-two branches that cut the reward section into as many basic
// blocks, each term's reduction over the env's lanes (four dependent DPP adds with their wait states) alone in its own; with the set known
// the unused terms are gone and the others interleave (go2: -0.5 us).  Any other set of terms runs the RS = false instantiation.
template <int LEGS, bool DO_PRE, unsigned MPH, int PROF = 0, int JPL = 3, bool INJ = false, bool RS = false>
#ifdef LG_PK_F32   // one wave per SIMD by design: let the allocator use the accumulation registers instead of spilling the 64-bit tuples
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(1, 1))) void quad_sim_kernel(KParams p) {
#else
__global__ __launch_bounds__(PROF == 6 && MPH == (LG_PHASE_POST | LG_PHASE_RESET) ? 2 * BLOCK : BLOCK) void quad_sim_kernel(KParams p) {
#endif
    using namespace q4;
    const unsigned tl_ = threadIdx.x & 63u;   // lane of the wave (a workgroup is one wave, two for the DUO tail below)
    static_assert(JPL == 3 || (JPL == 4 && LEGS == 2 && MPH == 0 && PROF == 0), "four-joint legs: biped physics only");
    static_assert(!INJ || (DO_PRE && MPH == (LG_PHASE_POST | LG_PHASE_RESET) && (PROF == 1 || PROF == 2 || PROF == 3 || PROF == 4 || PROF == 6)),
                  "injected read-backs: the component-layout tails only");
    constexpr bool FLAT = PROF == 1, PLANE = PROF == 1 || PROF == 2;   // host-checked task profiles (lg_kernel.h flat_profile / wtw_profile)
    constexpr unsigned RS_MASK = lg_default_reward_mask(PROF);
    static_assert(!RS || RS_MASK != 0u, "RS: the profile has a default reward set");
    constexpr int A = JPL * LEGS;
    // The kernel argument block (KParams, ~800 B of pointers) through ONE vector load: lane i holds bytes [16 i, 16 i + 16).
    // Fetched with scalar loads it arrives as a dozen dependent dwordx16 chunks (SGPR pressure), each a device-memory round
    // trip in front of the start-of-kernel burst: ~4 k of the prologue's cycles.  Pointers needed by the burst are rebuilt
    // from the lanes with v_readlane (right here, full EXEC); everything later uses `p` as usual.
    static_assert(sizeof(KParams) <= 64 * 16, "kernel argument block must fit one dwordx4 per lane");
    const uint4 kq = q4::kernarg_lane((int)(sizeof(KParams) + 15) / 16);
    auto kdw = [&](int dw) {   // dword `dw` of the block (compile-time index after inlining)
        const unsigned v = (dw & 3) == 0 ? kq.x : ((dw & 3) == 1 ? kq.y : ((dw & 3) == 2 ? kq.z : kq.w));
        return (unsigned)__builtin_amdgcn_readlane((int)v, dw >> 2);
    };
#define GAS __attribute__((address_space(1)))
#define KPTR(T, off) (reinterpret_cast<T>(((unsigned long long)kdw((int)((off) / 4) + 1) << 32) | (unsigned long long)kdw((int)((off) / 4))))
#define KB(T, f) KPTR(T, offsetof(KParams, B) + offsetof(LgBuffers, f))
#define KINT(member) ((int)kdw((int)(offsetof(KParams, member) / 4)))
#define KFLT(member) (__int_as_float((int)kdw((int)(offsetof(KParams, member) / 4))))
    // Three-joint legs: the lane's model / option constants come from the host-built LANE TABLE (lg_shared.h: sixteen rows of 80 floats, staged
    // by the wave, one row read per lane) instead of being derived from the staged model table on every launch (628 instructions).  The model
    // table itself is staged only where something still walks it: four-joint legs, and the generic tail (env_step_body in the same launch).
    constexpr bool USE_LT = JPL == 3;
    constexpr bool NEED_M = !USE_LT || (MPH != 0 && !(((PROF == 1 || PROF == 2 || PROF == 3 || PROF == 4) && MPH == (LG_PHASE_POST | LG_PHASE_RESET)) ||
                                                     (PROF == 6 && MPH == (LG_PHASE_POST | LG_PHASE_RESET))));
    const LgModelDesc GAS *Mg = KPTR(const LgModelDesc GAS *, offsetof(KParams, M));
    __shared__ __attribute__((aligned(16))) uint4 sMraw[NEED_M ? MODEL_STG * BLOCK : 1];
    __shared__ __attribute__((aligned(16))) uint4 sLT[USE_LT ? LG_LT_STG * BLOCK : 1];
    const LgModelDesc *M = reinterpret_cast<const LgModelDesc *>(sMraw);
    const LgSimOptions *__restrict__ O = p.O;
    __shared__ int sHot[256 + 2 * BLOCK];
    int hv0, hv1, hv2, hv3;
    {
        const int GAS *hp = KPTR(const int GAS *, offsetof(KParams, H)) + (tl_ & 63);
        hv0 = hp[0]; hv1 = hp[64]; hv2 = hp[128]; hv3 = hp[192];
    }
    uint4 stg0 = make_uint4(0u, 0u, 0u, 0u), stg1 = stg0, stg2 = stg0, stg3 = stg0, ltg[LG_LT_STG];
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    if constexpr (NEED_M) {
        const u4v GAS *src = reinterpret_cast<const u4v GAS *>(Mg);
        const u4v a0 = src[tl_], a1 = src[tl_ + BLOCK], a2 = src[tl_ + 2 * BLOCK], a3 = src[tl_ + 3 * BLOCK];
        stg0 = make_uint4(a0.x, a0.y, a0.z, a0.w); stg1 = make_uint4(a1.x, a1.y, a1.z, a1.w);
        stg2 = make_uint4(a2.x, a2.y, a2.z, a2.w); stg3 = make_uint4(a3.x, a3.y, a3.z, a3.w);
    }
#pragma unroll
    for (int k = 0; k < LG_LT_STG; k++) ltg[k] = make_uint4(0u, 0u, 0u, 0u);
    if constexpr (USE_LT) {
        const u4v GAS *src = KPTR(const u4v GAS *, offsetof(KParams, LT));
#pragma unroll
        for (int k = 0; k < LG_LT_STG; k++) { const u4v a = src[tl_ + k * BLOCK]; ltg[k] = make_uint4(a.x, a.y, a.z, a.w); }
    }
    const LgBuffers &B = p.B;
    unsigned long long _stamp0 = 0; (void)_stamp0;
    STAMP(0);
    STAMPB(4096);
    // DUO (the tron1_pf_ee tail, PROF 6): a workgroup of TWO waves per group of 8 envs.  4096 biped envs are 512 waves on 1024 SIMDs; with
    // two, both waves run the same physics on the same envs (role 0 owns its stores) and then SPLIT the MDP tail -- role 0: rewards,
    // episode sums, the noisy actor frames; role 1: the critic frames, labels, pushes, reset / task state -- so the serial tail of the
    // wave that ends the launch is about half as long.  Being one workgroup the two are resident together whatever else runs on the
    // chip; barriers order their accesses to the shared state: every start-of-kernel load of both has returned before either stores
    // (below), and role 0's read-back stores are out before role 1's reset stores to the same arrays.
    constexpr bool DUO = PROF == 6 && MPH == (LG_PHASE_POST | LG_PHASE_RESET);
    const int wg = lg_wg();   // XCD-aware (lg_kernel.h)
    const int role = DUO ? (int)(threadIdx.x >> 6) : 0;
    const int tid = wg * BLOCK + tl_;
    Lane L;
    L.c = tid & 3; L.is0 = L.c == 0; L.is1 = L.c == 1; L.is2 = L.c == 2; L.is3 = L.c == 3;
    L.d0 = L.is0 ? 1.f : 0.f; L.d1 = L.is1 ? 1.f : 0.f; L.d2 = L.is2 ? 1.f : 0.f;
    const int cj = min(L.c, 2);               // vector component of this lane; lane 3 shadows lane 2's addresses and never stores one
    const int cjj = JPL == 4 ? L.c : cj;      // joint of this lane (four-joint legs: lane 3 carries the fourth)
    const int quad = tid >> 2, leg = quad % LEGS;
    int e = quad / LEGS;
    const int N = KINT(B.n_envs);
    const bool alive = e < N;                 // this lane's env exists
    const bool live = alive && role == 0;     // ... and this wave owns its physics / state stores
    if (!alive) e = N - 1;
    const bool st = live && !L.is3;           // this lane stores vector components
    const bool stj = JPL == 4 ? live : st;    // ... joint values
    const bool lead = st && leg == 0;         // ... and the per-env ones
    const int nL = KINT(k.m_n_links), F = LEGS;
    const int b0 = 1 + JPL * leg, d0 = JPL * leg;
    const int foot_link = leg == 0 ? KINT(k.m_foot_link[0]) : (leg == 1 ? KINT(k.m_foot_link[1]) : (leg == 2 ? KINT(k.m_foot_link[2]) : KINT(k.m_foot_link[3])));
    int foot_slot = 0;
#pragma unroll
    for (int k = 0; k < LEGS; k++) foot_slot += (KINT(k.m_foot_link[k]) < foot_link) ? 1 : 0;

    // ---------------- start-of-kernel loads (one burst, one wait) --------------------------------
    // pointers of the burst, rebuilt from the lane table
    const float GAS *const k_actions = KB(const float GAS *, actions);
    const float GAS *const k_last_actions = KB(const float GAS *, last_actions);
    const float GAS *const k_base_pos = KB(const float GAS *, base_pos);
    const float GAS *const k_base_lin_vel_w = KB(const float GAS *, base_lin_vel_w);
    const float GAS *const k_base_ang_vel_w = KB(const float GAS *, base_ang_vel_w);
    const float GAS *const k_base_quat = KB(const float GAS *, base_quat);
    const float GAS *const k_dof_pos = KB(const float GAS *, dof_pos);
    const float GAS *const k_dof_vel = KB(const float GAS *, dof_vel);
    const float GAS *const k_feet_vel = KB(const float GAS *, feet_vel);
    const float GAS *const k_base_lin_vel = KB(const float GAS *, base_lin_vel);
    const float GAS *const k_base_ang_vel = KB(const float GAS *, base_ang_vel);
    const float GAS *const k_added_base_mass = KB(const float GAS *, added_base_mass);
    const float GAS *const k_base_com_bias = KB(const float GAS *, base_com_bias);
    const float GAS *const k_friction_values = KB(const float GAS *, friction_values);
    const float GAS *const k_kp_scale = KB(const float GAS *, kp_scale);
    const float GAS *const k_kd_scale = KB(const float GAS *, kd_scale);
    const float GAS *const k_joint_armature = KB(const float GAS *, joint_armature);
    const float GAS *const k_joint_friction = KB(const float GAS *, joint_friction);
    const float GAS *const k_joint_damping = KB(const float GAS *, joint_damping);
    const float GAS *const k_env_origins = KB(const float GAS *, env_origins);
    const float GAS *const k_command_ranges = KB(const float GAS *, command_ranges);
    const float GAS *const k_episode_sums = KB(const float GAS *, episode_sums);
    const float GAS *const k_commands = KB(const float GAS *, commands);
    const float GAS *const k_feet_air_time = KB(const float GAS *, feet_air_time);
    const int32_t GAS *const k_episode_length_buf = KB(const int32_t GAS *, episode_length_buf);
    const int64_t GAS *const k_fail_buf = KB(const int64_t GAS *, fail_buf);
    const uint8_t GAS *const k_last_contacts = KB(const uint8_t GAS *, last_contacts);
    const uint8_t GAS *const k_obs_dirty = KB(const uint8_t GAS *, obs_dirty);
    const float GAS *const k_actions_in = KPTR(const float GAS *, offsetof(KParams, actions));
    const LgSimOptions GAS *kO = KPTR(const LgSimOptions GAS *, offsetof(KParams, O));
    const LgTaskCfg GAS *kT = KPTR(const LgTaskCfg GAS *, offsetof(KParams, T));
    const int ja = e * A + d0 + cjj;          // this lane's joint
    float act, last_act = 0.f, llast_act = 0.f;
    if (DO_PRE) {
        const float ca = KFLT(k.clip_actions);
        last_act = k_actions[ja]; llast_act = k_last_actions[ja];
        act = clampf(k_actions_in[ja], -ca, ca);
    } else {
        act = k_actions_in ? k_actions_in[ja] : k_actions[ja];
    }
    float pos = k_base_pos[3 * e + cj], vw = k_base_lin_vel_w[3 * e + cj], ww = k_base_ang_vel_w[3 * e + cj];
    float quat = k_base_quat[4 * e + L.c];
    float q = k_dof_pos[ja], qd = k_dof_vel[ja];
    const float snap_fv = k_feet_vel[(e * F + foot_slot) * 3 + cj];
    const float snap_blv = k_base_lin_vel[3 * e + cj], snap_bav = k_base_ang_vel[3 * e + cj];
    const float dr_mass = k_added_base_mass ? k_added_base_mass[e] : 0.f;
    const float dr_com = k_base_com_bias ? k_base_com_bias[3 * e + cj] : 0.f;
    const float dr_fric = k_friction_values ? k_friction_values[e] : 1.f;
    const float dr_kp = k_kp_scale ? k_kp_scale[ja] : 1.f, dr_kd = k_kd_scale ? k_kd_scale[ja] : 1.f;
    float gain_p = 0.f, gain_d = 0.f, q0 = 0.f;      // three-joint legs: from the lane table, below
    if constexpr (!USE_LT) { gain_p = kO->kp[d0 + cjj]; gain_d = kO->kd[d0 + cjj]; q0 = kO->default_dof_pos[d0 + cjj]; }
    float dr_arm = 0.f, dr_jf = 0.f, dr_jd = 0.f;
    if (k_joint_armature) { dr_arm = k_joint_armature[e]; dr_jf = k_joint_friction[e]; dr_jd = k_joint_damping[e]; }
    const float origin = k_env_origins ? k_env_origins[3 * e + cj] : 0.f;
    int crv = 0;
    if (MPH != 0) crv = reinterpret_cast<const int GAS *>(k_command_ranges)[min((int)tl_, LG_CMD_RANGE_FLOATS - 1)];
    int prw = 0;   // observation programs for the MDP tail (lg_kernel.h PRG_I / PRG_F)
    if ((MPH & LG_PHASE_RESET) != 0 && (PROF == 4 || ((PROF == 0 || PROF == 5) && KINT(k.obs_layout) == LG_OBS_PROGRAM))) {
        const int tl = (int)tl_;
        prw = reinterpret_cast<const int GAS *>(tl < 26 ? &kT->priv_prog : &kT->labels_prog)[tl < 26 ? tl : min(tl - 26, 25)];
    }
    // MDP working set of the wave's 16 legs, fetched by lanes 0..15 in this same burst and parked in LDS: the MDP tail
    // (env_step_body<.., FUSED>) reads it back after the physics instead of paying the round trips then.  Layout = the
    // stash of env_step_body (NST values x 16 lanes).
    constexpr int NST = LG_R_COUNT + 36;
    constexpr bool QTAIL = (FLAT || PROF == 2 || PROF == 3 || PROF == 4) && MPH == (LG_PHASE_POST | LG_PHASE_RESET);   // MDP phases in component layout on all 64 lanes (below)
    constexpr bool WQ = PROF == 2 && QTAIL;   // ... of the go2_wtw task: gait clock, behaviour targets, 61 x 5 | 99 x 5 observation stacks
    constexpr bool PQ = PROF == 4 && QTAIL;   // ... of the rough heads with observation programs (go2_ts / go2_cts / go2_dreamwaq): go2_ee's MDP, other packaging
    constexpr bool EQ = (PROF == 3 || PROF == 4) && QTAIL;   // ... of the go2_ee family: heightfield, terrain curriculum, 45 x 20 | critic x 5 stacks, labels
    constexpr bool SQ = WQ || EQ;             // stacked observations, PD-gain randomisation, root twist draws
    // PROF 6 (host-checked, lg_host.hip biped_profile): the tron1_pf_ee task -- point-foot biped on a heightfield -- with its MDP phases in
    // component layout on the env's EIGHT lanes (two quads), the second block below
    constexpr bool BQ = PROF == 6 && MPH == (LG_PHASE_POST | LG_PHASE_RESET);
    static_assert(PROF != 6 || (LEGS == 2 && JPL == 3), "PROF 6 is the three-joint biped's tail");
    constexpr bool CTAIL = QTAIL || BQ;       // some component-layout tail: no leg-per-lane stash, no LDS hand-off
    __shared__ float sStF[(MPH != 0 && !CTAIL) ? NST * 16 : 1];
    float wsv[(MPH != 0 && !CTAIL) ? NST : 1];
    // QTAIL working set, one value per lane: command component c, the two episode sums this lane owns (terms ei and ei + 16 of its
    // env, ei = 4 leg + c), this lane's joint constants; per-env / per-leg scalars replicated
    static_assert(LG_R_COUNT <= 32, "two episode sums per lane");
    const int ei = (leg << 2) | L.c;
    float m_cmd = 0.f, m_air = 0.f, m_es0 = 0.f, m_es1 = 0.f, m_slo = 0.f, m_shi = 0.f, m_rlo = 0.f, m_rsp = 0.f, m_nq = 0.f, m_nqd = 0.f;
    int m_ep = 0, m_fail = 0, m_lc = 0;
    // biped tail: four episode sums per lane (terms el + 8 k of its env, el = 4 leg + c), the action / clock noise scales of tron1_pf_ee's
    // frame, the sit pose and the gait offsets a reset needs
    float b_es[4] = {0.f, 0.f, 0.f, 0.f}, b_nact = 0.f, b_nclk0 = 0.f, b_nclk1 = 0.f, b_sitq = 0.f, b_sitp = 0.f, b_sitr = 0.f, b_th0 = 0.f, b_th1 = 0.f;
    if (BQ) {
        static_assert(LG_R_COUNT <= 32, "four episode sums per lane");
#pragma unroll
        for (int k = 0; k < 4; k++) if (ei + 8 * k < LG_R_COUNT) b_es[k] = k_episode_sums[(size_t)(ei + 8 * k) * N + e];
        b_nact = kT->noise_vec[9 + 2 * A + d0 + cj];
        b_nclk0 = kT->noise_vec[9 + 3 * A + foot_slot]; b_nclk1 = kT->noise_vec[9 + 3 * A + LEGS + foot_slot];
        b_sitq = kT->sit_dof_pos[d0 + cj]; b_sitp = kT->sit_pos[cj]; b_sitr = kT->sit_quat[L.c];
        b_th0 = kT->theta_table[0][0]; b_th1 = kT->theta_table[0][1];
    }
    if (CTAIL) {
        m_cmd = k_commands[4 * e + L.c];
        m_ep = k_episode_length_buf[e];
        m_fail = (int)k_fail_buf[e];
        m_air = k_feet_air_time[e * F + foot_slot];
        m_lc = (int)k_last_contacts[e * F + foot_slot];
        if (QTAIL) m_es0 = k_episode_sums[(size_t)ei * N + e];
        if (QTAIL && ei + 16 < LG_R_COUNT) m_es1 = k_episode_sums[(size_t)(ei + 16) * N + e];
        m_slo = kT->soft_dof_lo[d0 + cj]; m_shi = kT->soft_dof_hi[d0 + cj];
        m_rlo = kT->reset_dof_lo[d0 + cj]; m_rsp = kT->reset_dof_span[d0 + cj];
        m_nq = kT->noise_vec[9 + d0 + cj]; m_nqd = kT->noise_vec[9 + A + d0 + cj];
    }
    // go2_wtw: gait clock and behaviour targets (per-env scalars replicated, per-foot entries by leg), the latest push, the deferred-
    // blanking flag of the observation stacks
    float w_gt = 0.f, w_phi = 0.f, w_gp = 1.f, w_bh = 0.f, w_fc = 0.f, w_pt = 0.f, w_th = 0.f, w_ec = 0.f, w_push = 0.f;
    int w_dirty = 0, w_lvl = 0, w_type = 0;
    if (WQ) {
        const float GAS *ts = KB(const float GAS *, task_state) + (size_t)e * LG_TASK_STATE_WTW;
        w_gt = ts[0]; w_phi = ts[1]; w_gp = ts[2]; w_bh = ts[3]; w_fc = ts[4]; w_pt = ts[5];
        w_th = ts[6 + foot_slot]; w_ec = ts[18 + foot_slot];
    }
    if (BQ) {   // tron1_pf_ee: gait clock (LG_TASK_STATE_BIPED, tron1_pf_ee.py:167-184)
        const float GAS *ts = KB(const float GAS *, task_state) + (size_t)e * LG_TASK_STATE_BIPED;
        w_gt = ts[0]; w_phi = ts[1]; w_th = ts[4 + foot_slot]; w_ec = ts[10 + foot_slot];
    }
    if (SQ || BQ) {
        w_push = KB(const float GAS *, rand_push_vels)[3 * e + cj];
        if (k_obs_dirty) w_dirty = (int)k_obs_dirty[e];
    }
    if (EQ || BQ) {   // go2_ee, tron1_pf_ee: terrain level / type of the env (curriculum at reset)
        const int32_t GAS *tl = KB(const int32_t GAS *, terrain_levels), *tt = KB(const int32_t GAS *, terrain_types);
        if (tl) { w_lvl = tl[e]; w_type = tt[e]; }
    }
    if (MPH != 0 && !CTAIL && tl_ < 16) {
        const LgTaskCfg GAS *T = kT;
        const int lt = wg * 16 + (int)tl_, legL = lt % LEGS, dL = 3 * legL;
        const int eL = min(lt / LEGS, N - 1);
        const int flL = legL == 0 ? KINT(k.m_foot_link[0]) : (legL == 1 ? KINT(k.m_foot_link[1]) : (legL == 2 ? KINT(k.m_foot_link[2]) : KINT(k.m_foot_link[3])));
        int fsL = 0;
#pragma unroll
        for (int k = 0; k < LEGS; k++) fsL += (KINT(k.m_foot_link[k]) < flL) ? 1 : 0;
#pragma unroll
        for (int k = 0; k < LG_R_COUNT; k++) wsv[k] = k_episode_sums[(size_t)k * N + eL];   // unconditional: no branch per term
        int c2 = LG_R_COUNT;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            wsv[c2++] = T->soft_dof_lo[dL + j]; wsv[c2++] = T->soft_dof_hi[dL + j];
            wsv[c2++] = T->reset_dof_lo[dL + j]; wsv[c2++] = T->reset_dof_span[dL + j];
            wsv[c2++] = T->noise_vec[9 + dL + j]; wsv[c2++] = T->noise_vec[9 + A + dL + j]; wsv[c2++] = T->noise_vec[9 + 2 * A + dL + j];
        }
        const bool clk = KINT(k.obs_layout) == LG_OBS_TRON1_EE;
        wsv[c2++] = clk ? T->noise_vec[9 + 3 * A + fsL] : 0.f; wsv[c2++] = clk ? T->noise_vec[9 + 3 * A + LEGS + fsL] : 0.f;
        wsv[c2++] = k_commands[4 * eL]; wsv[c2++] = k_commands[4 * eL + 1]; wsv[c2++] = k_commands[4 * eL + 2]; wsv[c2++] = k_commands[4 * eL + 3];
        wsv[c2++] = __int_as_float(k_episode_length_buf[eL]);
        wsv[c2++] = __int_as_float((int)k_fail_buf[eL]);
        wsv[c2++] = k_feet_air_time[eL * F + fsL];
        wsv[c2++] = __int_as_float((int)k_last_contacts[eL * F + fsL]);
        wsv[c2++] = k_env_origins[3 * eL]; wsv[c2++] = k_env_origins[3 * eL + 1]; wsv[c2++] = k_env_origins[3 * eL + 2];
        wsv[c2++] = __int_as_float(((MPH & LG_PHASE_RESET) && k_obs_dirty) ? (int)k_obs_dirty[eL] : 0);
    }

    asm volatile("" ::: "memory");
    sHot[tl_] = hv0; sHot[tl_ + 64] = hv1; sHot[tl_ + 128] = hv2; sHot[tl_ + 192] = hv3;
    if (MPH != 0) {
        sHot[tl_ + 256] = crv;
        sHot[tl_ + 256 + BLOCK] = prw;
        if (!CTAIL && tl_ < 16) {
            const unsigned rm = p.k.reward_mask;
            const bool leadL = (wg * 16 + tl_) % LEGS == 0;
#pragma unroll
            for (int k = 0; k < LG_R_COUNT; k++) sStF[k * 16 + tl_] = (leadL && ((rm >> k) & 1u)) ? wsv[k] : 0.f;
#pragma unroll
            for (int k = LG_R_COUNT; k < NST - 1; k++) sStF[k * 16 + tl_] = wsv[k];
        }
    }
    if constexpr (NEED_M) { sMraw[tl_] = stg0; sMraw[tl_ + BLOCK] = stg1; sMraw[tl_ + 2 * BLOCK] = stg2; sMraw[tl_ + 3 * BLOCK] = stg3; }
    if constexpr (USE_LT) {
#pragma unroll
        for (int k = 0; k < LG_LT_STG; k++) sLT[tl_ + k * BLOCK] = ltg[k];
    }
    if (DUO) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): both waves hold everything they read before anything of the step is stored
    __syncthreads();
    STAMP(23);

    // ---------------- prologue stores ---------------------------------------------------------------
    if (DO_PRE && stj) {
        stq(B.llast_actions, 4u * (unsigned)ja, llast_act);
        stq(B.last_actions, 4u * (unsigned)ja, last_act);
        stq(B.actions, 4u * (unsigned)ja, act);
    }
    float last_foot_v = snap_fv, qd_start = qd;   // kept for the MDP tail (dof_acc, foot_acc)
    // injected read-backs (INJ): what the physics of this step left behind, as the test put it into the buffers
    float inj_tq = 0.f, inj_fl[4] = {0.f, 0.f, 0.f, 0.f}, inj_fb = 0.f, inj_fp = 0.f, inj_pg = 0.f, inj_eul = 0.f;
    if (INJ) {
        qd_start = B.last_dof_vel[ja]; last_foot_v = B.last_feet_vel[(e * F + foot_slot) * 3 + cj];
        inj_tq = B.torques[ja];
#pragma unroll
        for (int k = 0; k < 4; k++) inj_fl[k] = B.link_contact_forces[(e * nL + foot_link - 3 + k) * 3 + cj];
        inj_fb = B.link_contact_forces[(e * nL) * 3 + cj];
        inj_fp = B.feet_pos[(e * F + foot_slot) * 3 + cj];
        inj_pg = B.projected_gravity[3 * e + cj]; inj_eul = B.base_euler[3 * e + cj];
    } else {
    if (stj) stq(B.last_dof_vel, 4u * (unsigned)ja, qd);   // "last" snapshots (genesis_simulator.py:21-24)
    if (st) {
        stq(B.last_feet_vel, 4u * (unsigned)((e * F + foot_slot) * 3 + cj), snap_fv);
        if (leg == 0) { stq(B.last_base_lin_vel, 4u * (unsigned)(3 * e + cj), snap_blv); stq(B.last_base_ang_vel, 4u * (unsigned)(3 * e + cj), snap_bav); }
    }
    }
    (void)last_foot_v; (void)qd_start;

    // ---------------- constants ---------------------------------------------------------------------
    const float dt = HOT(o_dt), kc = HOT(o_contact_k), kappa = kc * dt + HOT(o_contact_b), margin = HOT(o_contact_margin);
    const float kl = HOT(o_limit_k), kapl = kl * dt + HOT(o_limit_b), lmargin = HOT(o_limit_margin);
    const float gravc = L.d2 * HOT(o_gravity_z);          // gravity vector, this lane's component
    const float mu = HOT(o_terrain_friction) * dr_fric;
    const float ascale = HOT(o_action_scale);
    Terr TR;
    TR.rows = PLANE ? 0 : HOT(o_terrain_rows); TR.cols = HOT(o_terrain_cols); TR.border = HOT(o_border); TR.ihs = 1.f / HOT(o_hscale);
    TR.vscale = HOT(o_vscale); TR.hf = p.hf;
    constexpr bool HFC = PROF == 3 || PROF == 4 || PROF == 5 || PROF == 6;   // a heightfield is bound (host-checked: the rough task profiles of lg_host.hip; 5 = that and nothing else, generic tail)
    const bool hfmode = HFC ? true : (!PLANE && TR.rows > 0);
    float lr[LT_USED];                          // this lane's row of the lane table (three-joint legs)
#pragma unroll
    for (int k = 0; k < LT_USED; k++) lr[k] = 0.f;
    if constexpr (USE_LT) {
        const float4 *row = reinterpret_cast<const float4 *>(sLT) + ei * (LG_LT_ROW / 4);     // ei = 4 leg + c: the lane's type
#pragma unroll
        for (int k = 0; k < LT_USED / 4; k++) { const float4 v = row[k]; lr[4 * k] = v.x; lr[4 * k + 1] = v.y; lr[4 * k + 2] = v.z; lr[4 * k + 3] = v.w; }
        gain_p = lr[LT_KP]; gain_d = lr[LT_KD]; q0 = lr[LT_Q0];
    }
    auto sym_row = [&](const float *s6) {   // rows of a symmetric 3x3 stored (xx, yy, zz, xy, xz, yz)
        QM r = {s6[L.is0 ? 0 : (L.is1 ? 3 : 4)], s6[L.is0 ? 3 : (L.is1 ? 1 : 5)], s6[L.is0 ? 4 : (L.is1 ? 5 : 2)]};
        return r;
    };
    float mass0, com0, Lqlo, Lqhi, Leff, Lvlim, arm, jfric, jdamp, foot_c_loc, foot_r, foot_link_pos;
    QM I0;
    float Lm[JPL], Lcom[JPL], Ljpos[JPL], Lax[JPL];
    QM LIc[JPL];
    // collision spheres (foot excluded): five "slots", in each the four lanes of the quad test four different
    // spheres of the SAME body in scalar form.  slot 0: hip, 1: thigh, 2-3: calf, 4: base (4 per quad); four-joint legs:
    // 0: abad, 1: hip, 2: knee, 3: the sole corners of the foot body, 4: base
    constexpr int NSLOT = 5;
    float sx[NSLOT], sy[NSLOT], sz[NSLOT], srad[NSLOT], sden[NSLOT], sidw[NSLOT];   // sden = 1/(1 + kappa dt w), sidw = 1/(dt w)
    float sole_w = 0.f;                        // sph_w of this lane's sole corner (four-joint legs: its inverse mass is formed per sub-step)
    unsigned m_tmask, m_pmask, m_smask;        // link masks of the MDP tails
    if constexpr (USE_LT) {
        mass0 = lr[LT_MASS0] + dr_mass; com0 = lr[LT_COM0] + dr_com;
        I0 = QM{lr[LT_I0], lr[LT_I0 + 1], lr[LT_I0 + 2]};
#pragma unroll
        for (int j = 0; j < JPL; j++) {
            Lm[j] = lr[LT_M + j]; Lcom[j] = lr[LT_COM + j]; Ljpos[j] = lr[LT_JPOS + j]; Lax[j] = lr[LT_AX + j];
            LIc[j] = QM{lr[LT_IC + 3 * j], lr[LT_IC + 3 * j + 1], lr[LT_IC + 3 * j + 2]};
        }
        Lqlo = lr[LT_QLO]; Lqhi = lr[LT_QHI]; Leff = lr[LT_EFF]; Lvlim = lr[LT_VLIM];
        arm = B.joint_armature ? dr_arm : lr[LT_ARM];
        jfric = B.joint_friction ? dr_jf : lr[LT_JFRIC];
        jdamp = B.joint_damping ? dr_jd : lr[LT_JDAMP];
        foot_c_loc = lr[LT_FOOT_C]; foot_r = lr[LT_FOOT_R]; foot_link_pos = lr[LT_LINKPOS];
#pragma unroll
        for (int k = 0; k < NSLOT; k++) {
            sx[k] = lr[LT_SLOT + 6 * k]; sy[k] = lr[LT_SLOT + 6 * k + 1]; sz[k] = lr[LT_SLOT + 6 * k + 2];
            srad[k] = lr[LT_SLOT + 6 * k + 3]; sden[k] = lr[LT_SLOT + 6 * k + 4]; sidw[k] = lr[LT_SLOT + 6 * k + 5];
        }
        m_tmask = __float_as_uint(lr[LT_TMASK]); m_pmask = __float_as_uint(lr[LT_PMASK]); m_smask = __float_as_uint(lr[LT_SMASK]);
    } else {
    mass0 = M->mass[0] + dr_mass;
    com0 = M->com[0][cj] + dr_com;
    I0 = sym_row(M->inertia[0]);
#pragma unroll
    for (int j = 0; j < JPL; j++) {
        const int b = b0 + j;
        Lm[j] = M->mass[b]; Lcom[j] = M->com[b][cj]; Ljpos[j] = M->jpos[b][cj]; Lax[j] = M->axis[b][cj];
        LIc[j] = sym_row(M->inertia[b]);
    }
    // joint-lane constants
    Lqlo = M->q_lo[d0 + cjj]; Lqhi = M->q_hi[d0 + cjj]; Leff = M->effort[d0 + cjj];
    Lvlim = HOT(o_joint_vel_clamp) * M->vel_limit[d0 + cjj];
    arm = B.joint_armature ? dr_arm : M->armature[d0 + cjj];
    jfric = B.joint_friction ? dr_jf : M->frictionloss[d0 + cjj];
    jdamp = B.joint_damping ? dr_jd : M->damping[d0 + cjj];
    const int fs = leg == 0 ? HOT(m_foot_sphere[0]) : (leg == 1 ? HOT(m_foot_sphere[1]) : (leg == 2 ? HOT(m_foot_sphere[2]) : HOT(m_foot_sphere[3])));
    foot_c_loc = M->sph_pos[fs][cj]; foot_r = M->sph_r[fs]; foot_link_pos = M->link_pos[foot_link][cj];
    m_tmask = M->term_link_mask; m_pmask = M->pen_link_mask; m_smask = M->state_link_mask;
    {
        const int a0 = M->body_sph_start[b0], a1 = M->body_sph_start[b0 + 1], a2 = M->body_sph_start[b0 + 2], a3 = M->body_sph_start[b0 + 3];
        const int e0 = M->body_sph_start[0], e1 = M->body_sph_start[1];
        int idx[NSLOT];
        idx[0] = a0 + L.c < a1 ? a0 + L.c : -1;
        idx[1] = a1 + L.c < a2 ? a1 + L.c : -1;
        if (JPL == 4) {
            const int a4 = M->body_sph_start[b0 + JPL];
            idx[2] = a2 + L.c < a3 ? a2 + L.c : -1;
            int s = a3 + L.c;                  // c-th non-foot sphere of the foot body
            if (s >= fs) s++;
            idx[3] = s < a4 ? s : -1;
            sole_w = M->sph_w[max(idx[3], 0)];
        } else {
#pragma unroll
            for (int k = 0; k < 2; k++) {      // n-th non-foot sphere of the calf, n = 4k + c
                int s = a2 + 4 * k + L.c;
                if (s >= fs) s++;              // the foot sphere is one of [a2, a3): skip over it
                idx[2 + k] = s < a3 ? s : -1;
            }
        }
        idx[4] = e0 + leg * 4 + L.c < e1 ? e0 + leg * 4 + L.c : -1;
#pragma unroll
        for (int k = 0; k < NSLOT; k++) {
            const int s = max(idx[k], 0);
            sx[k] = M->sph_pos[s][0]; sy[k] = M->sph_pos[s][1]; sz[k] = M->sph_pos[s][2];
            srad[k] = idx[k] >= 0 ? M->sph_r[s] : -1e30f;    // an empty slot is infinitely far from any surface
            const float wi = M->sph_w[s];
            sden[k] = 1.f / (1.f + kappa * dt * wi);
            sidw[k] = 1.f / (dt * wi);
        }
    }
    }
    const float kps = dr_kp * gain_p, kds = dr_kd * gain_d;
    (void)m_tmask; (void)m_pmask; (void)m_smask; (void)foot_link_pos;

    STAMP(24);
    float torque = 0.f;
    float f_link[4] = {0.f, 0.f, 0.f, 0.f};   // net contact force on hip, thigh, calf, foot links (component)
    float f_base = 0.f;

    const int decim = INJ ? 0 : HOT(o_decimation), iters = HOT(o_contact_iters), w_every = HOT(o_contact_w_every);
    QM Ac_keep = {0.f, 0.f, 0.f};             // dt * W of this leg's foot as of its latest refresh (LgSimOptions.contact_w_every)
    const float mv = HOT(o_max_base_lin_vel), mw = HOT(o_max_base_ang_vel);

    STAMP(12);
    float nf_poison = 0.f;   // becomes NaN when an acceleration or a joint rate of any sub-step was not finite (read-back guard)
    // SPLIT (the two-wave workgroups of the tron1_pf_ee step, DUO): the two waves used to run the SAME sub-steps side by side and only shared
    // out the tail.  Now they share out the sub-step: both run the chain kinematics, then wave 0 does the body collision spheres (broad phase,
    // heightfield lookups, forces) WHILE wave 1 does everything of the articulated-body pass that does not need a force (articulated
    // inertias, joint factors, base inverse); wave 1 takes the forces through LDS, finishes the pass, the contact solve and the integration,
    // and hands the new state back.  Two barriers per sub-step; the sphere section (a quarter of a sub-step) leaves the critical path.
    // While wave 1 then runs the bias recursion, the base and pass 3, wave 0 -- idle otherwise -- forms the feet's operational-space matrices W from
    // the joint factors and the base inverse wave 1 left in LDS, and hands them over at a third barrier.
    constexpr bool SPLIT = DUO && !INJ;
    const bool doA = !SPLIT || role == 0, doB = !SPLIT || role == 1;
    __shared__ float sXch[SPLIT ? 12 * 64 : 1], sXw[SPLIT ? 21 * 64 : 1], sXa[SPLIT ? 3 * 64 : 1];
    for (int sub = 0; sub < decim; sub++) {
        const QM Rb = quat_rows(L, quat);
        QKin K[JPL];
        QJoint J[JPL];
        // ---- chain kinematics (root -> leaf) --------------------------------------------------------
        {
            float sv, cv;
            __sincosf(q, &sv, &cv);      // the joint angles of the leg at once
            QM Rp = Rb;
            float Pp = 0.f;
            QV6 Vp = {ww, vw};
#pragma unroll
            for (int j = 0; j < JPL; j++) {
                K[j].P = Pp + mulv(Rp, Ljpos[j]);
                const float cq = bcj(cv, j), sq = bcj(sv, j);
                float s;
                if (j == 0) joint_rot<QUAD_AXES[0]>(L, Rp, Lax[j], cq, sq, K[j].R, s);
                else if (j == 1) joint_rot<QUAD_AXES[1]>(L, Rp, Lax[j], cq, sq, K[j].R, s);
                else if (j == 2) joint_rot<QUAD_AXES[2]>(L, Rp, Lax[j], cq, sq, K[j].R, s);
                else joint_rot<QUAD_AXES[3]>(L, Rp, Lax[j], cq, sq, K[j].R, s);
                J[j].S.a = s;
                const QR sr = rots(s);           // the joint axis enters three products, the body's angular velocity six (here and in pass 2)
                J[j].S.l = cross(K[j].P, sr);
                const float qdj = bcj(qd, j);
                K[j].V.a = Vp.a + s * qdj;
                K[j].V.l = Vp.l + J[j].S.l * qdj;
                K[j].Wr = rots(K[j].V.a);
                J[j].c.a = cross(K[j].V.a, sr) * qdj;
                J[j].c.l = (cross(K[j].Wr, J[j].S.l) + cross(K[j].V.l, sr)) * qdj;
                Rp = K[j].R; Pp = K[j].P; Vp = K[j].V;
            }
        }
        if (sub == 0) STAMP(13);
        // ---- body collision spheres -----------------------------------------------------------------
        // ext[b]: spatial force about O on chain body b from its spheres; extb: on the base (this quad's share)
        QV6 ext[JPL], extb = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < JPL; j++) ext[j] = QV6{0.f, 0.f};
        float ft_h = 0.f, ft_nx = 0.f, ft_ny = 0.f, ft_nz = 1.f;   // terrain under the foot sphere (looked up with the broad phase below, used in stage 2)
        if (doA) {
            const float px = bc<0>(pos), py = bc<1>(pos), pz = bc<2>(pos);
            // broad phase of one slot: this lane's sphere of body (R, P); returns the penetration depth (and centre / terrain)
            struct Hit { float rx, ry, rz, h, nx, ny, nz, depth; bool on; };
            auto probe = [&](int k, const QM &R, float P, const float dref = 0.f) {
                Hit t;
                t.rx = t.ry = 0.f; t.h = 0.f; t.nx = t.ny = 0.f; t.nz = 1.f;
                t.rz = bc<2>(P) + bc<2>(R.c0) * sx[k] + bc<2>(R.c1) * sy[k] + bc<2>(R.c2) * sz[k];
                if (hfmode) {
                    t.rx = bc<0>(P) + bc<0>(R.c0) * sx[k] + bc<0>(R.c1) * sy[k] + bc<0>(R.c2) * sz[k];
                    t.ry = bc<1>(P) + bc<1>(R.c0) * sx[k] + bc<1>(R.c1) * sy[k] + bc<1>(R.c2) * sz[k];
                    terrain<HFC>(TR, px + t.rx, py + t.ry, t.h, t.nx, t.ny, t.nz);
                }
                t.depth = srad[k] - (pz + t.rz - t.h) * t.nz - dref;
                t.on = t.depth > -margin;
                return t;
            };
            // force of this lane's sphere (scalar form), accumulated into (m, f) about O; body pose / twist in scalar form
            struct BodyS { float x0, x1, x2, y0, y1, y2, Px, Py, wx, wy, wz, vx, vy, vz; };
            auto gather = [&](const QM &R, float P, const QV6 &V) {
                BodyS g = {bc<0>(R.c0), bc<0>(R.c1), bc<0>(R.c2), bc<1>(R.c0), bc<1>(R.c1), bc<1>(R.c2), bc<0>(P), bc<1>(P),
                           bc<0>(V.a), bc<1>(V.a), bc<2>(V.a), bc<0>(V.l), bc<1>(V.l), bc<2>(V.l)};
                return g;
            };
            // `sole` (four-joint legs, slot 3): the sole law of lg_kernel.h's sphere_contact -- inverse mass = sph_w + twice the
            // ankle joint's own compliance at the contact point, approach velocity relative to the sole centre's (vref)
            struct Sole { float sax, say, saz, slx, sly, slz, dinv, vref; };
            auto force = [&](int k, Hit t, const BodyS &g, float (&m)[3], float (&f)[3], const Sole *sole = nullptr) {
                if (!hfmode) {
                    t.rx = g.Px + g.x0 * sx[k] + g.x1 * sy[k] + g.x2 * sz[k];
                    t.ry = g.Py + g.y0 * sx[k] + g.y1 * sy[k] + g.y2 * sz[k];
                }
                const float vx = g.vx + (g.wy * t.rz - g.wz * t.ry), vy = g.vy + (g.wz * t.rx - g.wx * t.rz), vz = g.vz + (g.wx * t.ry - g.wy * t.rx);
                float vn = vx * t.nx + vy * t.ny + vz * t.nz;
                float den = sden[k], idw = sidw[k];
                if (JPL == 4 && sole) {
                    const float cx_ = t.rx - t.nx * srad[k], cy_ = t.ry - t.ny * srad[k], cz_ = t.rz - t.nz * srad[k];
                    // n . (S.l + S.a x c)
                    const float gq = t.nx * (sole->slx + sole->say * cz_ - sole->saz * cy_) + t.ny * (sole->sly + sole->saz * cx_ - sole->sax * cz_) +
                                     t.nz * (sole->slz + sole->sax * cy_ - sole->say * cx_);
                    const float wi = sole_w + 2.f * gq * gq * sole->dinv;
                    den = rcp(1.f + kappa * dt * wi); idw = rcp(dt * wi);
                    vn -= sole->vref;
                }
                const float fn = (kc * t.depth - kappa * vn) * den;
                float fx = 0.f, fy = 0.f, fz = 0.f;
                if (!hfmode && !(JPL == 4 && sole)) {
                    // the plane: n = e_z, so v_n = v_z, the tangential velocity is (vx, vy, 0) and the contact point sits straight below the centre
                    // -- the general statements with the zeros and ones multiplied out (a compiler may not drop x * 0)
                    float gg = 0.f;
                    if (t.on && fn > 0.f) {
                        const float s2 = vx * vx + vy * vy;
                        gg = s2 > 1e-18f ? fminf(idw, mu * fn * rsqrtf(s2)) : 0.f;
                        fz = fn;
                    }
                    fx = -vx * gg; fy = -vy * gg;
                    const float cz = t.rz - srad[k];
                    m[0] += t.ry * fz - cz * fy; m[1] += cz * fx - t.rx * fz; m[2] += t.rx * fy - t.ry * fx;
                    f[0] += fx; f[1] += fy; f[2] += fz;
                    return;
                }
                if (t.on && fn > 0.f) {
                    const float vnf = vx * t.nx + vy * t.ny + vz * t.nz;
                    const float tx = vx - t.nx * vnf, ty = vy - t.ny * vnf, tz = vz - t.nz * vnf;
                    // |f_t| = min(|v_t| / (dt w), mu f_n) along -v_t: one rsq, no sqrt / division
                    const float s2 = tx * tx + ty * ty + tz * tz;
                    const float gg = s2 > 1e-18f ? fminf(idw, mu * fn * rsqrtf(s2)) : 0.f;
                    fx = t.nx * fn - tx * gg; fy = t.ny * fn - ty * gg; fz = t.nz * fn - tz * gg;
                }
                const float cx = t.rx - t.nx * srad[k], cy = t.ry - t.ny * srad[k], cz = t.rz - t.nz * srad[k];
                m[0] += cy * fz - cz * fy; m[1] += cz * fx - cx * fz; m[2] += cx * fy - cy * fx;
                f[0] += fx; f[1] += fy; f[2] += fz;
            };
            // TWO spheres per lane at once, one in each half of a packed-f32 pair (v_pk_fma_f32 & co.): a lone wave issues a packed
            // instruction at the price of a plain one (tools/ubench/pk_issue.hip: 5.1 vs 5.3 cycles), and this law is plain per-lane
            // arithmetic -- no DPP operand anywhere -- so the pair costs about what one sphere does.  Same statements as `force`, half by half.
            struct BodyS2 { f2 x0, x1, x2, y0, y1, y2, Px, Py, wx, wy, wz, vx, vy, vz; };
            auto pair_bodies = [&](const BodyS a, const BodyS b) {
                BodyS2 g = {f2{a.x0, b.x0}, f2{a.x1, b.x1}, f2{a.x2, b.x2}, f2{a.y0, b.y0}, f2{a.y1, b.y1}, f2{a.y2, b.y2}, f2{a.Px, b.Px}, f2{a.Py, b.Py},
                            f2{a.wx, b.wx}, f2{a.wy, b.wy}, f2{a.wz, b.wz}, f2{a.vx, b.vx}, f2{a.vy, b.vy}, f2{a.vz, b.vz}};
                return g;
            };
            struct Acc2 { f2 mx, my, mz, fx, fy, fz; };
            auto force2 = [&](int k0, int k1, const Hit t0, const Hit t1, const BodyS2 &g, Acc2 &acc) {
                const f2 sxx = {sx[k0], sx[k1]}, syy = {sy[k0], sy[k1]}, szz = {sz[k0], sz[k1]}, rad = {srad[k0], srad[k1]};
                f2 rx = {t0.rx, t1.rx}, ry = {t0.ry, t1.ry};
                const f2 rz = {t0.rz, t1.rz}, nx = {t0.nx, t1.nx}, ny = {t0.ny, t1.ny}, nz = {t0.nz, t1.nz}, depth = {t0.depth, t1.depth};
                if (!hfmode) {
                    rx = g.Px + g.x0 * sxx + g.x1 * syy + g.x2 * szz;
                    ry = g.Py + g.y0 * sxx + g.y1 * syy + g.y2 * szz;
                }
                const f2 vx = g.vx + (g.wy * rz - g.wz * ry), vy = g.vy + (g.wz * rx - g.wx * rz), vz = g.vz + (g.wx * ry - g.wy * rx);
                const f2 den = {sden[k0], sden[k1]}, idw = {sidw[k0], sidw[k1]};
                if (!hfmode) {   // the plane (see `force`)
                    const f2 fn = (kc * depth - kappa * vz) * den;
                    const f2 s2 = vx * vx + vy * vy;
                    const f2 mf = mu * fn;
                    const bool a0 = t0.on && fn.x > 0.f, a1 = t1.on && fn.y > 0.f;
                    const f2 fz = {a0 ? fn.x : 0.f, a1 ? fn.y : 0.f};
                    const f2 gg = {a0 && s2.x > 1e-18f ? fminf(idw.x, mf.x * rsqrtf(s2.x)) : 0.f, a1 && s2.y > 1e-18f ? fminf(idw.y, mf.y * rsqrtf(s2.y)) : 0.f};
                    const f2 fx = -vx * gg, fy = -vy * gg;
                    const f2 cz = rz - rad;
                    acc.mx += ry * fz - cz * fy; acc.my += cz * fx - rx * fz; acc.mz += rx * fy - ry * fx;
                    acc.fx += fx; acc.fy += fy; acc.fz += fz;
                    return;
                }
                const f2 vn = vx * nx + vy * ny + vz * nz;
                const f2 fn = (kc * depth - kappa * vn) * den;
                const f2 tx = vx - nx * vn, ty = vy - ny * vn, tz = vz - nz * vn;
                const f2 s2 = tx * tx + ty * ty + tz * tz;
                const f2 mf = mu * fn;
                const bool a0 = t0.on && fn.x > 0.f, a1 = t1.on && fn.y > 0.f;
                // no force: normal and tangential magnitudes both zero (the products below then are)
                const f2 fa = {a0 ? fn.x : 0.f, a1 ? fn.y : 0.f};
                const f2 gg = {a0 && s2.x > 1e-18f ? fminf(idw.x, mf.x * rsqrtf(s2.x)) : 0.f, a1 && s2.y > 1e-18f ? fminf(idw.y, mf.y * rsqrtf(s2.y)) : 0.f};
                const f2 fx = nx * fa - tx * gg, fy = ny * fa - ty * gg, fz = nz * fa - tz * gg;
                const f2 cx = rx - nx * rad, cy = ry - ny * rad, cz = rz - nz * rad;
                acc.mx += cy * fz - cz * fy; acc.my += cz * fx - cx * fz; acc.mz += cx * fy - cy * fx;
                acc.fx += fx; acc.fy += fy; acc.fz += fz;
            };
            auto reduce6 = [&](float mx, float my, float mz, float fx, float fy, float fz, QV6 &acc) {
                acc.a += L.sel(sum4(mx), sum4(my), sum4(mz));
                acc.l += L.sel(sum4(fx), sum4(fy), sum4(fz));
            };
            auto reduce = [&](const float (&m)[3], const float (&f)[3], QV6 &acc) { reduce6(m[0], m[1], m[2], f[0], f[1], f[2], acc); };
            // one slot of a body; the force branch runs only if some sphere of the slot touches somewhere in the wave
            auto slot = [&](int k, const Hit &t, const QM &R, float P, const QV6 &V, QV6 &acc) {
                if (__builtin_amdgcn_ballot_w64(t.on) != 0ull) {
                    float m[3] = {0.f, 0.f, 0.f}, f[3] = {0.f, 0.f, 0.f};
                    force(k, t, gather(R, P, V), m, f);
                    reduce(m, f, acc);
                }
            };
            const QV6 V0 = {ww, vw};
            // Broad phase of EVERY slot first.  On a heightfield each probe is a dependent round trip (cell index -> four int16 loads ->
            // bilinear height and normal): probe, force, probe, force ... exposed one L2 round trip per slot and sub-step in the serial
            // chain (rough terrain: +14 k cycles per control step over the plane); issued together the loads overlap.
            Hit tb, ta, tc, td, te;
#ifndef LG_NO_PK_TERRAIN
            if constexpr (HFC && JPL == 3) {
                // heightfield bound (compile time): the five slots' sphere centres and the foot's, then the six lookups as three packed pairs
                // (base | hip, thigh | calf, calf's second slot | foot)
                auto centre = [&](int k, const QM &R, float P, float &rx, float &ry, float &rz) {
                    rx = bc<0>(P) + bc<0>(R.c0) * sx[k] + bc<0>(R.c1) * sy[k] + bc<0>(R.c2) * sz[k];
                    ry = bc<1>(P) + bc<1>(R.c0) * sx[k] + bc<1>(R.c1) * sy[k] + bc<1>(R.c2) * sz[k];
                    rz = bc<2>(P) + bc<2>(R.c0) * sx[k] + bc<2>(R.c1) * sy[k] + bc<2>(R.c2) * sz[k];
                };
                float cx[6], cy[6], cz[6];
                centre(4, Rb, 0.f, cx[0], cy[0], cz[0]); centre(0, K[0].R, K[0].P, cx[1], cy[1], cz[1]); centre(1, K[1].R, K[1].P, cx[2], cy[2], cz[2]);
                centre(2, K[2].R, K[2].P, cx[3], cy[3], cz[3]); centre(3, K[2].R, K[2].P, cx[4], cy[4], cz[4]);
                const float rf = K[JPL - 1].P + mulv(K[JPL - 1].R, foot_c_loc);
                cx[5] = bc<0>(rf); cy[5] = bc<1>(rf); cz[5] = 0.f;
                f2 hh[3], nnx[3], nny[3], nnz[3];
#pragma unroll
                for (int k = 0; k < 3; k++) terrain2(TR, f2{px + cx[2 * k], px + cx[2 * k + 1]}, f2{py + cy[2 * k], py + cy[2 * k + 1]}, hh[k], nnx[k], nny[k], nnz[k]);
                auto finish = [&](int k, float rx, float ry, float rz, float h, float nx, float ny, float nz) {
                    Hit t;
                    t.rx = rx; t.ry = ry; t.rz = rz; t.h = h; t.nx = nx; t.ny = ny; t.nz = nz;
                    t.depth = srad[k] - (pz + rz - h) * nz;
                    t.on = t.depth > -margin;
                    return t;
                };
                tb = finish(4, cx[0], cy[0], cz[0], hh[0].x, nnx[0].x, nny[0].x, nnz[0].x);
                ta = finish(0, cx[1], cy[1], cz[1], hh[0].y, nnx[0].y, nny[0].y, nnz[0].y);
                tc = finish(1, cx[2], cy[2], cz[2], hh[1].x, nnx[1].x, nny[1].x, nnz[1].x);
                td = finish(2, cx[3], cy[3], cz[3], hh[1].y, nnx[1].y, nny[1].y, nnz[1].y);
                te = finish(3, cx[4], cy[4], cz[4], hh[2].x, nnx[2].x, nny[2].x, nnz[2].x);
                ft_h = hh[2].y; ft_nx = nnx[2].y; ft_ny = nny[2].y; ft_nz = nnz[2].y;
            } else
#endif
            {
            tb = probe(4, Rb, 0.f); ta = probe(0, K[0].R, K[0].P); tc = probe(1, K[1].R, K[1].P);
            td = probe(2, K[2].R, K[2].P);
            te = td;
            if constexpr (JPL == 3) te = probe(3, K[2].R, K[2].P);   // the calf's second slot
            if (HFC) {   // the foot's lookup rides in the same batch of loads
                const float rf = K[JPL - 1].P + mulv(K[JPL - 1].R, foot_c_loc);
                terrain<HFC>(TR, px + bc<0>(rf), py + bc<1>(rf), ft_h, ft_nx, ft_ny, ft_nz);
            }
            }
            slot(4, tb, Rb, 0.f, V0, extb);
#ifndef LG_NO_PK_SPHERES
            {   // hip and thigh slots: as a packed pair when both are live somewhere in the wave, on their own otherwise
                const bool on0 = __builtin_amdgcn_ballot_w64(ta.on) != 0ull, on1 = __builtin_amdgcn_ballot_w64(tc.on) != 0ull;
                if (on0 && on1) {
                    const f2 z2 = {0.f, 0.f};
                    Acc2 a2 = {z2, z2, z2, z2, z2, z2};
                    force2(0, 1, ta, tc, pair_bodies(gather(K[0].R, K[0].P, K[0].V), gather(K[1].R, K[1].P, K[1].V)), a2);
                    reduce6(a2.mx.x, a2.my.x, a2.mz.x, a2.fx.x, a2.fy.x, a2.fz.x, ext[0]);
                    reduce6(a2.mx.y, a2.my.y, a2.mz.y, a2.fx.y, a2.fy.y, a2.fz.y, ext[1]);
                } else {
                    slot(0, ta, K[0].R, K[0].P, K[0].V, ext[0]);
                    slot(1, tc, K[1].R, K[1].P, K[1].V, ext[1]);
                }
            }
#else
            slot(0, ta, K[0].R, K[0].P, K[0].V, ext[0]);
            slot(1, tc, K[1].R, K[1].P, K[1].V, ext[1]);
#endif
            if constexpr (JPL == 4) {
                slot(2, td, K[2].R, K[2].P, K[2].V, ext[2]);
                // the sole corners: penetration / approach velocity relative to the sole centre's while that one is in the ground
                const BodyS g = gather(K[3].R, K[3].P, K[3].V);
                const float rc = K[3].P + mulv(K[3].R, foot_c_loc);
                const float rcx = bc<0>(rc), rcy = bc<1>(rc), rcz = bc<2>(rc);
                float hc = 0.f, ncx = 0.f, ncy = 0.f, ncz = 1.f;
                if (hfmode) terrain<HFC>(TR, px + rcx, py + rcy, hc, ncx, ncy, ncz);
                const float dc = foot_r - (pz + rcz - hc) * ncz;
                const float vcx = g.vx + (g.wy * rcz - g.wz * rcy), vcy = g.vy + (g.wz * rcx - g.wx * rcz), vcz = g.vz + (g.wx * rcy - g.wy * rcx);
                const float dref = dc > 0.f ? dc : 0.f;
                const Hit t3 = probe(3, K[3].R, K[3].P, dref);
                if (__builtin_amdgcn_ballot_w64(t3.on) != 0ull) {
                    // ankle compliance: 1 / (S^T I_foot S + armature) with the foot body's own rigid inertia
                    const float cw = K[3].P + mulv(K[3].R, Lcom[3]);
                    const QM Icw = mulmmt(mulmm(K[3].R, LIc[3]), K[3].R);
                    const QI6 Ifoot = rigid(L, Lm[3], cw, Icw);
                    Sole so;
                    so.dinv = rcp(bc<0>(dot6(J[3].S, muli6(Ifoot, J[3].S))) + bc<3>(arm));   // replicated over the quad (lane 3 tests a sphere too)
                    so.sax = bc<0>(J[3].S.a); so.say = bc<1>(J[3].S.a); so.saz = bc<2>(J[3].S.a);
                    so.slx = bc<0>(J[3].S.l); so.sly = bc<1>(J[3].S.l); so.slz = bc<2>(J[3].S.l);
                    so.vref = dc > 0.f ? vcx * ncx + vcy * ncy + vcz * ncz : 0.f;
                    float m[3] = {0.f, 0.f, 0.f}, f[3] = {0.f, 0.f, 0.f};
                    force(3, t3, g, m, f, &so);
                    reduce(m, f, ext[3]);
                }
            } else {   // the calf's two slots share one gather and one reduction
                const Hit t2 = td, t3 = te;
                if (__builtin_amdgcn_ballot_w64(t2.on || t3.on) != 0ull) {
                    const BodyS g = gather(K[2].R, K[2].P, K[2].V);
#ifndef LG_NO_PK_SPHERES
                    const f2 z2 = {0.f, 0.f};
                    Acc2 a2 = {z2, z2, z2, z2, z2, z2};
                    force2(2, 3, t2, t3, pair_bodies(g, g), a2);
                    reduce6(a2.mx.x + a2.mx.y, a2.my.x + a2.my.y, a2.mz.x + a2.mz.y, a2.fx.x + a2.fx.y, a2.fy.x + a2.fy.y, a2.fz.x + a2.fz.y, ext[2]);
#else
                    float m[3] = {0.f, 0.f, 0.f}, f[3] = {0.f, 0.f, 0.f};
                    force(2, t2, g, m, f);
                    force(3, t3, g, m, f);
                    reduce(m, f, ext[2]);
#endif
                }
            }
        }
        if constexpr (SPLIT) {
            if (role == 0) {
#pragma unroll
                for (int j = 0; j < JPL; j++) { sXch[(2 * j) * 64 + tl_] = ext[j].a; sXch[(2 * j + 1) * 64 + tl_] = ext[j].l; }
                sXch[6 * 64 + tl_] = extb.a; sXch[7 * 64 + tl_] = extb.l;
                sXch[8 * 64 + tl_] = ft_h; sXch[9 * 64 + tl_] = ft_nx; sXch[10 * 64 + tl_] = ft_ny; sXch[11 * 64 + tl_] = ft_nz;
            }
        }

        if (sub == 0) STAMP(14);
        // ---- actuation (genesis_simulator.py:630-642), three joints at once ---------------------------
        torque = kps * (act * ascale + q0 - q) - kds * qd;
        const float tau = clampf(torque, -Leff, Leff) - jdamp * qd - jfric * clampf(qd * 20.f, -1.f, 1.f);

        // ---- ABA pass 2 (leaf -> root), first what needs no force: articulated inertias, joint factors, the inertia's share of the bias
        //      recursion (IA c, U . c) ------------------------------------------------------------
        QI6 IA, IA0, Inv;
        QV6 pacc, Ic6[JPL];
        float dinvv = 0.f;      // 1/D of the joints in joint lanes
        float cwj[JPL], ucj[JPL], cwb = 0.f;
        QM Icwj[JPL], Icwb = {0.f, 0.f, 0.f};
        if (doB) {
#pragma unroll
            for (int j = JPL - 1; j >= 0; j--) {
                cwj[j] = K[j].P + mulv(K[j].R, Lcom[j]);
                Icwj[j] = mulmmt(mulmm(K[j].R, LIc[j]), K[j].R);
                const QI6 Ib = rigid(L, Lm[j], cwj[j], Icwj[j]);
                if (j == JPL - 1) IA = Ib;
                else IA = IA + Ib;
                J[j].U = muli6(IA, J[j].S);
                const float armj = bcj(arm, j);
                J[j].dinv = rcp(dot6(J[j].S, J[j].U) + armj);
                Ic6[j] = muli6(IA, J[j].c);
                ucj[j] = dot6(J[j].U, J[j].c);
                IA = rank1_down(IA, J[j].U, J[j].dinv);
                { const float dj = j == 3 ? bc<0>(J[j].dinv) : J[j].dinv; if (L.c == j) dinvv = dj; }   // lane 3 holds no dot product of its own
            }
            // ---- base inertia and its inverse ----
            IA0.A.c0 = legsum<LEGS>(IA.A.c0); IA0.A.c1 = legsum<LEGS>(IA.A.c1); IA0.A.c2 = legsum<LEGS>(IA.A.c2);
            IA0.B.c0 = legsum<LEGS>(IA.B.c0); IA0.B.c1 = legsum<LEGS>(IA.B.c1); IA0.B.c2 = legsum<LEGS>(IA.B.c2);
            IA0.Bt.c0 = legsum<LEGS>(IA.Bt.c0); IA0.Bt.c1 = legsum<LEGS>(IA.Bt.c1); IA0.Bt.c2 = legsum<LEGS>(IA.Bt.c2);
            IA0.C.c0 = legsum<LEGS>(IA.C.c0); IA0.C.c1 = legsum<LEGS>(IA.C.c1); IA0.C.c2 = legsum<LEGS>(IA.C.c2);
            cwb = mulv(Rb, com0);
            Icwb = mulmmt(mulmm(Rb, I0), Rb);
            IA0 = IA0 + rigid(L, mass0, cwb, Icwb);
            Inv = inv6(L, IA0);
        }
        if constexpr (SPLIT) {
            if (role == 1) {                  // what W is made of (besides the kinematics both waves hold): U_j, 1 / D_j, IA0^-1
#pragma unroll
                for (int j = 0; j < JPL; j++) { sXw[(3 * j) * 64 + tl_] = J[j].U.a; sXw[(3 * j + 1) * 64 + tl_] = J[j].U.l; sXw[(3 * j + 2) * 64 + tl_] = J[j].dinv; }
                sXw[9 * 64 + tl_] = Inv.A.c0; sXw[10 * 64 + tl_] = Inv.A.c1; sXw[11 * 64 + tl_] = Inv.A.c2;
                sXw[12 * 64 + tl_] = Inv.B.c0; sXw[13 * 64 + tl_] = Inv.B.c1; sXw[14 * 64 + tl_] = Inv.B.c2;
                sXw[15 * 64 + tl_] = Inv.Bt.c0; sXw[16 * 64 + tl_] = Inv.Bt.c1; sXw[17 * 64 + tl_] = Inv.Bt.c2;
                sXw[18 * 64 + tl_] = Inv.C.c0; sXw[19 * 64 + tl_] = Inv.C.c1; sXw[20 * 64 + tl_] = Inv.C.c2;
            }
            __syncthreads();                  // wave 0's forces and wave 1's factors are in LDS
            if (role == 1) {
#pragma unroll
                for (int j = 0; j < JPL; j++) { ext[j].a = sXch[(2 * j) * 64 + tl_]; ext[j].l = sXch[(2 * j + 1) * 64 + tl_]; }
                extb.a = sXch[6 * 64 + tl_]; extb.l = sXch[7 * 64 + tl_];
                ft_h = sXch[8 * 64 + tl_]; ft_nx = sXch[9 * 64 + tl_]; ft_ny = sXch[10 * 64 + tl_]; ft_nz = sXch[11 * 64 + tl_];
            }
        }
#pragma unroll
        for (int j = 0; j < JPL; j++) f_link[j] = ext[j].l;
        if (sub == 0) STAMP(15);
        QV6 a0 = {0.f, 0.f}, a_calf = {0.f, 0.f};
        float qdd = 0.f;
        if (doB) {
        // ---- ... then the bias recursion with the forces (same statements, same order as one interleaved pass) ----
#pragma unroll
        for (int j = JPL - 1; j >= 0; j--) {
            const float m = Lm[j], cw = cwj[j];
            const QR cwr = rots(cw);
            const float vc = K[j].V.l + cross(K[j].Wr, cw);
            const float Pm = vc * m;
            const float Lmo = mulv(Icwj[j], K[j].V.a) + cross(cwr, Pm);
            const float fg = gravc * m;
            QV6 pb;
            pb.a = cross(K[j].Wr, Lmo) + cross(K[j].V.l, Pm) - cross(cwr, fg) - ext[j].a;
            pb.l = cross(K[j].Wr, Pm) - fg - ext[j].l;
            if (j == JPL - 1) pacc = pb;
            else pacc = pacc + pb;
            const float tauj = bcj(tau, j);
            J[j].u = tauj - dot6(J[j].S, pacc);
            const float k = (J[j].u - ucj[j]) * J[j].dinv;
            pacc = pacc + Ic6[j] + J[j].U * k;
        }
        // ---- base ---------------------------------------------------------------------------------
        QV6 p0 = {legsum<LEGS>(pacc.a - extb.a), legsum<LEGS>(pacc.l - extb.l)};
        {
            const float cw = cwb;
            const QM Icw = Icwb;
            const QR wr = rots(ww), cwr = rots(cw);
            const float vc = vw + cross(wr, cw);
            const float Pm = vc * mass0;
            const float Lmo = mulv(Icw, ww) + cross(cwr, Pm);
            const float fg = gravc * mass0;
            p0.a += cross(wr, Lmo) + cross(vw, Pm) - cross(cwr, fg);
            p0.l += cross(wr, Pm) - fg;
        }
        a0 = muli6(Inv, QV6{-p0.a, -p0.l});
        if (sub == 0) STAMP(16);
        // ---- pass 3 (root -> leaf) ----------------------------------------------------------------
        {
            QV6 a = a0;
            float gj[JPL];
#pragma unroll
            for (int j = 0; j < JPL; j++) { a = a + J[j].c; gj[j] = (J[j].u - dot6(J[j].U, a)) * J[j].dinv; a = a + J[j].S * gj[j]; }
            qdd = JPL == 4 ? L.sel4(gj[0], gj[1], gj[2], bc<0>(gj[JPL - 1])) : L.sel(gj[0], gj[1], gj[2]);
            a_calf = a;
        }
        if (sub == 0) STAMP(17);
        }   // doB (bias recursion, base, pass 3)
        // ---- stage 2: foot contact (exact 3x3 W) + joint-limit stops, block-Jacobi ----------------------
        float cn = 0.f, ct1 = L.d0, ct2 = L.d1, cp = 0.f, depth = 0.f;
        bool fact = false;
        auto contact_frame = [&]() {
            const float r = K[JPL - 1].P + mulv(K[JPL - 1].R, foot_c_loc);
            float h = ft_h, nx = ft_nx, ny = ft_ny, nz = ft_nz;
            if (!HFC) terrain<HFC>(TR, bc<0>(pos) + bc<0>(r), bc<1>(pos) + bc<1>(r), h, nx, ny, nz);
            depth = foot_r - (bc<2>(pos) + bc<2>(r) - h) * nz;
            fact = depth > -margin;
            cn = L.sel(nx, ny, nz);
            cp = r - cn * foot_r;
            if (hfmode) {   // tangent frame on a slope (identity on the plane)
                const float t = L.d0 - nx * cn;
                const float t1 = t * rsqrtf(dot3(t, t));
                const bool tilt = nz < 0.999999f;
                ct1 = tilt ? t1 : ct1;
                const float t2 = cross(cn, t1);      // formed outside the conditional (cross-lane reads: see the note at the top)
                ct2 = tilt ? t2 : ct2;
            }
        };
        // dt * W, contact frame, rows.  W_rk = f_r . (acceleration the robot answers f_k with), f_k the unit force along contact axis k at the
        // contact point.  The articulated-body passes are an L D L^T factorisation of the inverse inertia, so with the UPWARD pass of each unit
        // force alone (joint residuals du_j, force left at the base p) W = sum_j du_j du_j^T / D_j + P^T IA0^-1 P: no downward passes, no point
        // accelerations -- the three resp_down chains were the longest dependent stretch of this section.  (Checked against the two-pass form on
        // random chains to 1e-16 in double; the CPU oracle and the leg-per-lane kernel keep the two-pass form.)
        auto w_columns = [&](const QR &cpr) {
            const float axs[3] = {cn, ct1, ct2};
            float du[3][JPL];
            QV6 pk[3], qk[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const QV6 fsp = {cross(cpr, axs[k]), axs[k]};
                pk[k] = resp_up(J, fsp, 0.f, du[k]);
                qk[k] = muli6(Inv, pk[k]);
            }
            const float D00 = dot6(pk[0], qk[0]), D01 = dot6(pk[0], qk[1]), D02 = dot6(pk[0], qk[2]);
            const float D11 = dot6(pk[1], qk[1]), D12 = dot6(pk[1], qk[2]), D22 = dot6(pk[2], qk[2]);
            float col[3] = {L.sel(D00, D01, D02), L.sel(D01, D11, D12), L.sel(D02, D12, D22)};   // row r in lane r; symmetric by construction
#pragma unroll
            for (int j = 0; j < JPL; j++) {
                const float us = L.sel(du[0][j], du[1][j], du[2][j]) * J[j].dinv;
                col[0] += us * du[0][j]; col[1] += us * du[1][j]; col[2] += us * du[2][j];
            }
            QM w = {dt * col[0], dt * col[1], dt * col[2]};
            return w;
        };
        const bool w_refresh = w_every <= 1 || (sub % w_every) == 0;   // wave-uniform: W of every foot is recomputed on these sub-steps
        if constexpr (SPLIT) {
            if (role == 0 && w_refresh) {     // wave 0, otherwise idle until the new state arrives: W from wave 1's factors
#pragma unroll
                for (int j = 0; j < JPL; j++) { J[j].U.a = sXw[(3 * j) * 64 + tl_]; J[j].U.l = sXw[(3 * j + 1) * 64 + tl_]; J[j].dinv = sXw[(3 * j + 2) * 64 + tl_]; }
                Inv.A.c0 = sXw[9 * 64 + tl_]; Inv.A.c1 = sXw[10 * 64 + tl_]; Inv.A.c2 = sXw[11 * 64 + tl_];
                Inv.B.c0 = sXw[12 * 64 + tl_]; Inv.B.c1 = sXw[13 * 64 + tl_]; Inv.B.c2 = sXw[14 * 64 + tl_];
                Inv.Bt.c0 = sXw[15 * 64 + tl_]; Inv.Bt.c1 = sXw[16 * 64 + tl_]; Inv.Bt.c2 = sXw[17 * 64 + tl_];
                Inv.C.c0 = sXw[18 * 64 + tl_]; Inv.C.c1 = sXw[19 * 64 + tl_]; Inv.C.c2 = sXw[20 * 64 + tl_];
                contact_frame();
                const QM w = w_columns(rots(cp));
                sXa[tl_] = w.c0; sXa[64 + tl_] = w.c1; sXa[128 + tl_] = w.c2;
            }
            __syncthreads();                  // W is in LDS
            if (role == 1 && w_refresh) { Ac_keep.c0 = sXa[tl_]; Ac_keep.c1 = sXa[64 + tl_]; Ac_keep.c2 = sXa[128 + tl_]; }
        }
        if (doB) {
        float fc = 0.f;                       // foot force in the contact frame (n, t1, t2), component layout
        contact_frame();
        float lim_e = 0.f, lim_s = 0.f, lim_T = 0.f;
        if (JPL == 4 || !L.is3) {
            if (q < Lqlo + lmargin) { lim_s = 1.f; lim_e = Lqlo - q; }
            else if (q > Lqhi - lmargin) { lim_s = -1.f; lim_e = q - Lqhi; }
        }
        float dqdd = 0.f;
        QV6 da0 = {0.f, 0.f};
        if (w_refresh || __builtin_amdgcn_ballot_w64(fact || lim_s != 0.f) != 0ull) {
            // contact-frame projector: row r of E = axis r (n, t1, t2)
            QM E;
            E.c0 = L.sel(bc<0>(cn), bc<0>(ct1), bc<0>(ct2));
            E.c1 = L.sel(bc<1>(cn), bc<1>(ct1), bc<1>(ct2));
            E.c2 = L.sel(bc<2>(cn), bc<2>(ct1), bc<2>(ct2));
            const QM ET = {cn, ct1, ct2};
            const QR cpr = rots(cp);          // the contact point enters every product below
            if (w_refresh && !SPLIT) Ac_keep = w_columns(cpr);
            const QM Ac = Ac_keep;
            float vfree;
            {
                const float vpt = K[JPL - 1].V.l + cross(K[JPL - 1].V.a, cpr);
                const float apt = a_calf.l + cross(a_calf.a, cpr) + cross(K[JPL - 1].Wr, vpt);
                vfree = mulv(E, vpt + apt * dt);
            }
            if (sub == 0) STAMP(18);
            float resp_c = 0.f;
            for (int it = 0; it < iters; it++) {
                // foot: velocity it would have without its own force, then the local law
                {
                    const float vo = vfree + resp_c * dt - mulv(Ac, fc);
                    const float vo0 = bc<0>(vo);
                    const float rn = kc * depth - kappa * vo0;
                    // stick trial: rows of [1 + kappa A00, kappa A01, kappa A02; A1.; A2.]
                    QM m;
                    m.c0 = L.is0 ? 1.f + kappa * Ac.c0 : Ac.c0;
                    m.c1 = L.is0 ? kappa * Ac.c1 : Ac.c1;
                    m.c2 = L.is0 ? kappa * Ac.c2 : Ac.c2;
                    const float rhs = L.is0 ? rn : -vo;
                    // Cramer: column c of adj = (row c+1) x (row c+2)
                    const float u0 = rot1(m.c0), u1 = rot1(m.c1), u2 = rot1(m.c2), w0 = rot2(m.c0), w1 = rot2(m.c1), w2 = rot2(m.c2);
                    const float x0 = u1 * w2 - u2 * w1, x1 = u2 * w0 - u0 * w2, x2 = u0 * w1 - u1 * w0;
                    const float det = bc<0>(m.c0 * x0 + m.c1 * x1 + m.c2 * x2);
                    const float inv = rcp(det);
                    float f0 = sum3(x0 * rhs) * inv, f1 = sum3(x1 * rhs) * inv, f2 = sum3(x2 * rhs) * inv;
                    bool ok = fact && fabsf(det) >= 1e-30f && f0 > 0.f;
                    const float A00 = bc<0>(Ac.c0), A01 = bc<0>(Ac.c1), A02 = bc<0>(Ac.c2);   // DPP stays outside the branch
                    const float ft2 = f1 * f1 + f2 * f2, iftn = rsqrtf(ft2), ftn = ft2 * iftn;
                    const float e1 = f1 * iftn, e2 = f2 * iftn;
                    // friction coupling limited to 3/4 of A_nn (frictional jamming; see lg_kernel.h / oracle/lg_oracle.c)
                    const float aeff = fmaxf(A00 + mu * (A01 * e1 + A02 * e2), 0.25f * A00);
                    const float fn = rn * rcp(1.f + kappa * aeff);
                    if (ftn > mu * f0) { ok = ok && fn > 0.f; f0 = fn; f1 = mu * fn * e1; f2 = mu * fn * e2; }
                    fc = ok ? L.sel(f0, f1, f2) : 0.f;
                }
                float tl = 0.f;
                {   // limit stops of the three joints (joint lanes)
                    float C = dinvv;
                    if (it > 0 && lim_T > 0.f) C = fminf(fmaxf(lim_s * dqdd * rcp(lim_T), C), 8.f * C);
                    const float vin = -lim_s * (qd + dt * (qdd + dqdd)) + dt * lim_T * C;
                    const float Tn = (kl * lim_e + kapl * vin) * rcp(1.f + kapl * dt * C);
                    lim_T = lim_s != 0.f ? fmaxf(Tn, 0.f) : 0.f;
                    tl = lim_s * lim_T;
                }
                // exact response of the whole robot to the current force set
                const float fw = mulv(ET, fc);
                const QV6 fsp = {cross(cpr, fw), fw};
                float du[JPL];
                QV6 dp = resp_up(J, fsp, tl, du);
                dp.a = legsum<LEGS>(dp.a); dp.l = legsum<LEGS>(dp.l);
                da0 = muli6(Inv, QV6{-dp.a, -dp.l});
                const QV6 ac = resp_down(L, J, da0, du, dqdd);
                const float ra = ac.l + cross(ac.a, cpr);
                resp_c = mulv(E, ra);
            }
            if (JPL == 4) f_link[3] += mulv(ET, fc);   // the foot is the last body's own link: on top of its sole corners
            else f_link[3] = mulv(ET, fc);
        } else if (JPL != 4) {
            f_link[3] = 0.f;
        }
        if (sub == 0) STAMP(19);
        // ---- semi-implicit Euler ------------------------------------------------------------------
        {
            const float alpha = a0.a + da0.a;
            const float alin = a0.l + da0.l + cross(ww, vw);     // spatial -> classical at O
            // poison for the non-finite guard: the clamps below turn NaN into a limit value (fminf / fmaxf drop it), so what goes into
            // them is watched -- 0 * x is NaN for x = NaN or Inf, 0 otherwise
            nf_poison = fmaf(L.is3 ? (JPL == 4 ? qd + qdd + dqdd : 0.f) : alin + alpha + (qd + qdd + dqdd), 0.f, nf_poison);   // lane 3: no vector component
            vw = clampf(vw + dt * alin, -mv, mv);
            ww = clampf(ww + dt * alpha, -mw, mw);
            qd = clampf(qd + dt * (qdd + dqdd), -Lvlim, Lvlim);
            q += dt * qd;
            pos += vw * dt;
            const float wn = fsqrt(bc<0>(dot3(ww, ww))), half = 0.5f * wn * dt;   // lane 3 (quaternion w) needs the true norm too
            float sh, chh;
            __sincosf(half, &sh, &chh);
            const float sc = wn > 1e-12f ? sh * rcp(wn) : 0.5f * dt;
            const float d = L.is3 ? 0.f : ww * sc;              // vector part of the step quaternion; scalar part chh
            const float qw_b = bc<3>(quat);
            const float dv = sum4(d * quat);                    // d . q_v  (lane 3 contributes 0)
            const float dxq = cross(d, quat);                   // outside the conditional (cross-lane reads)
            const float nq = chh * quat + (L.is3 ? -dv : qw_b * d + dxq);
            quat = nq * rsqrtf(sum4(nq * nq));
        }
        }   // doB
        if constexpr (SPLIT) {
            // wave 1 hands the new state (and what the read-back needs of the contact solve) to wave 0; both continue with the same values
            if (role == 1) {
                sXch[0 * 64 + tl_] = q; sXch[1 * 64 + tl_] = qd; sXch[2 * 64 + tl_] = pos; sXch[3 * 64 + tl_] = quat; sXch[4 * 64 + tl_] = vw;
                sXch[5 * 64 + tl_] = ww; sXch[6 * 64 + tl_] = f_link[3]; sXch[7 * 64 + tl_] = nf_poison; sXch[8 * 64 + tl_] = extb.l;
            }
            __syncthreads();
            if (role == 0) {
                q = sXch[0 * 64 + tl_]; qd = sXch[1 * 64 + tl_]; pos = sXch[2 * 64 + tl_]; quat = sXch[3 * 64 + tl_]; vw = sXch[4 * 64 + tl_];
                ww = sXch[5 * 64 + tl_]; f_link[3] = sXch[6 * 64 + tl_]; nf_poison = sXch[7 * 64 + tl_];
            }
            // (no third barrier: the next writes into the buffer are wave 0's own, behind its reads above; wave 1 reads them behind the next barrier)
        }
        if (sub == 0) STAMP(20);
        f_base = legsum<LEGS>(extb.l);
    }  // sub-steps

    STAMP(21);
    // Every load of the start-of-kernel burst landed during the sub-steps; say so before the read-back stores go out.  vmcnt
    // retires loads and stores in issue order, and the compiler keeps a load "pending" until a wait formally covers it: the first
    // use of a prologue value behind the ~60 read-back stores (episode sums, soft limits, noise scales of the MDP tail) otherwise
    // becomes `s_waitcnt vmcnt(k)` with k = the stores issued since on the shortest path, i.e. a drain of most of those stores.
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0); expcnt / lgkmcnt untouched
    if (!PLANE && p.k.cat_enable) {   // go2_cat's job-wide "some joint faster than 4 rad/s" flag (LG_CR_ANY_FAST)
        if (__builtin_amdgcn_ballot_w64(stj && fabsf(qd) > 4.0f) != 0ull && tl_ == 0) B.command_ranges[LG_CR_ANY_FAST + (int)(p.counter & 1)] = 1.0f;
    }
    // ---------------- read-back (genesis_simulator.py:35-60) ---------------------------------------
    int guard_bad = 0;
    if (!INJ) {   // non-finite guard: re-seat the robot
        const float chk = quat + (L.is3 ? (JPL == 4 ? q + qd : 0.f) : pos + vw + ww + q + qd) + nf_poison;
        const int bad = env_or<LEGS>(isfinite(chk) ? 0 : 1);
        const float reseat = HOT(o_base_init_pos[0]) * L.d0 + HOT(o_base_init_pos[1]) * L.d1 + HOT(o_base_init_pos[2]) * L.d2 + origin;
        if (bad) {
            pos = reseat; vw = 0.f; ww = 0.f; quat = L.is3 ? 1.f : 0.f;
            q = q0; qd = 0.f; torque = 0.f;
            f_link[0] = f_link[1] = f_link[2] = f_link[3] = 0.f; f_base = 0.f;
            // never silent: counted, and the env terminates at this step's check_termination (fail_buf over the threshold)
            // the start-of-step snapshots may be what was not finite: this step's rate terms (dof_acc, foot_acc) read zeros instead
            last_foot_v = 0.f; qd_start = 0.f;
            if (stj) B.last_dof_vel[ja] = 0.f;
            if (st) {
                B.last_feet_vel[(e * F + foot_slot) * 3 + cj] = 0.f;
                if (leg == 0) { B.last_base_lin_vel[3 * e + cj] = 0.f; B.last_base_ang_vel[3 * e + cj] = 0.f; }
            }
            if (lead && L.is0) {
                if (B.nonfinite_count) atomicAdd(B.nonfinite_count, 1);
                if (MPH == 0 && B.fail_buf) B.fail_buf[e] = LG_FAIL_NONFINITE;   // the MDP launch that follows reads it
            }
        }
        guard_bad = bad;
        // out-of-terrain teleport (genesis_simulator.py:612-628)
        const float px = bc<0>(pos), py = bc<1>(pos);
        if (px >= HOT(o_bound_x[1]) || px <= HOT(o_bound_x[0]) || py >= HOT(o_bound_y[1]) || py <= HOT(o_bound_y[0])) pos = reseat;
    }
    const float qx = bc<0>(quat), qy = bc<1>(quat), qz = bc<2>(quat), qw = bc<3>(quat);
    float eul;
    {   // lanes 0 / 2: roll / yaw through one atan2f, lane 1: pitch
        const float ay = L.is0 ? 2.f * (qw * qx + qy * qz) : 2.f * (qw * qz + qx * qy);
        const float ax = L.is0 ? qw * qw - qx * qx - qy * qy + qz * qz : qw * qw + qx * qx - qy * qy - qz * qz;
        const float at = atan2f(ay, ax);
        const float sinp = 2.f * (qw * qy - qz * qx);
        const float pit = fabsf(sinp) >= 1.f ? copysignf(1.5707963267948966f, sinp) : asinf(sinp);
        eul = L.is1 ? pit : at;
    }
    const QM Rb = quat_rows(L, quat);
    float blv = multv(L, Rb, vw), bav = multv(L, Rb, ww);
    float pg = -L.sel(bc<2>(Rb.c0), bc<2>(Rb.c1), bc<2>(Rb.c2));
    float foot_p, foot_v;
    {   // foot frame at the final state
        float sv, cv;
        __sincosf(q, &sv, &cv);
        QM Rp = Rb;
        float Pp = 0.f;
        QV6 Vp = {ww, vw};
#pragma unroll
        for (int j = 0; j < JPL; j++) {
            const float P = Pp + mulv(Rp, Ljpos[j]);
            const float cq = bcj(cv, j), sq = bcj(sv, j), qdj = bcj(qd, j);
            float s;
            QM Rn;
            if (j == 0) joint_rot<QUAD_AXES[0]>(L, Rp, Lax[j], cq, sq, Rn, s);
            else if (j == 1) joint_rot<QUAD_AXES[1]>(L, Rp, Lax[j], cq, sq, Rn, s);
            else if (j == 2) joint_rot<QUAD_AXES[2]>(L, Rp, Lax[j], cq, sq, Rn, s);
            else joint_rot<QUAD_AXES[3]>(L, Rp, Lax[j], cq, sq, Rn, s);
            Vp.a = Vp.a + s * qdj;
            Vp.l = Vp.l + cross(P, s) * qdj;
            Rp = Rn;
            Pp = P;
        }
        const float r = Pp + mulv(Rp, foot_link_pos);
        foot_p = pos + r;
        foot_v = Vp.l + cross(Vp.a, r);
    }
    if (INJ) {
        foot_p = inj_fp; foot_v = snap_fv; blv = snap_blv; bav = snap_bav; pg = inj_pg; eul = inj_eul; torque = inj_tq;
        f_link[0] = inj_fl[0]; f_link[1] = inj_fl[1]; f_link[2] = inj_fl[2]; f_link[3] = inj_fl[3]; f_base = inj_fb;
    }
    // Read-back stores: a wave-uniform base (kernel argument) + a 32-bit byte offset per lane, so that the store takes its base from scalar
    // registers (saddr form) -- one offset per index pattern, shared by the arrays that use it -- instead of a 64-bit pointer computed in
    // vector registers for every store (sign extension, 64-bit shift, add with carry: 3-4 VALU instructions each, ~70 stores per lane in the
    // step).  Every array stays below 4 GiB (checked where the launch is chosen, lg_host.hip).
    const unsigned o_ja = 4u * (unsigned)ja, o_e3 = 4u * (unsigned)(3 * e + cj), o_ft = 4u * (unsigned)((e * F + foot_slot) * 3 + cj);
    if (!INJ && stj) { stq(B.dof_pos, o_ja, q); stq(B.dof_vel, o_ja, qd); stq(B.torques, o_ja, torque); }
    if (!INJ && st) {
        const unsigned o_lk = 4u * (unsigned)((e * nL + foot_link - 3) * 3 + cj);
#pragma unroll
        for (int k = 0; k < 4; k++) stq(B.link_contact_forces, o_lk + 12u * (unsigned)k, f_link[k]);
        stq(B.feet_pos, o_ft, foot_p);
        stq(B.feet_vel, o_ft, foot_v);
        if (leg == 0) {
            stq(B.base_pos, o_e3, pos);
            stq(B.base_lin_vel_w, o_e3, vw); stq(B.base_ang_vel_w, o_e3, ww);
            stq(B.base_lin_vel, o_e3, blv); stq(B.base_ang_vel, o_e3, bav);
            stq(B.projected_gravity, o_e3, pg); stq(B.base_euler, o_e3, eul);
            stq(B.link_contact_forces, 4u * (unsigned)((e * nL) * 3 + cj), f_base);
        }
    }
    if (!INJ && live && leg == 0) stq(B.base_quat, 4u * (unsigned)(4 * e + L.c), quat);

    // ---------------- terrain sampling around the base and the feet (genesis_simulator.py:552-610) ------
    const int P = PLANE ? 0 : HOT(o_n_height_points);
    constexpr int HQ = 7;                        // samples per lane held in flight (go2: 81 / 16 lanes, tron1: 49 / 8)
    const int kstride = 4 * LEGS, hk0 = leg * 4 + L.c;   // this lane's terrain samples: k = hk0 + i kstride
    float hq[HQ];                                // this lane's samples (k = k0 + i kstride), kept for the go2_ee tail
    float f_hmean = 0.f, f_hmax = 0.f;           // mean / max of the nine heights around this leg's foot
    float f_h9[9], f_n3[3] = {0.f, 0.f, 0.f};    // ... the nine heights and the terrain normal there (observation programs)
#pragma unroll
    for (int k = 0; k < 9; k++) f_h9[k] = 0.f;
#pragma unroll
    for (int i = 0; i < HQ; i++) hq[i] = 0.f;
    if (INJ && P > 0) {   // the terrain read-backs as injected (genesis_simulator.py:552-610 ran in the reference's simulator)
#pragma unroll
        for (int i = 0; i < HQ; i++) hq[i] = B.measured_heights[(size_t)e * P + min(hk0 + i * kstride, P - 1)];
        if (HOT(o_feet_terrain_info)) {
            float sm = 0.f, mx = -1e30f;
#pragma unroll
            for (int k = 0; k < 9; k++) { const float hv = B.height_around_feet[((size_t)e * F + foot_slot) * 9 + k]; sm += hv; mx = fmaxf(mx, hv); f_h9[k] = hv; }
            f_hmean = sm / 9.f; f_hmax = mx;
#pragma unroll
            for (int k = 0; k < 3; k++) f_n3[k] = B.normal_vector_around_feet[((size_t)e * F + foot_slot) * 3 + k];
        }
    }
    if (!INJ && P > 0) {
        const float yn = rcp(fmaxf(fsqrt(qz * qz + qw * qw), 1e-9f));
        const float yz = qz * yn, yw = qw * yn;
        const float px = bc<0>(pos), py = bc<1>(pos);
        if (P <= HQ * kstride) {
            // all of this lane's samples at once: points, then the 3 x HQ height loads, then the minima -- instead of one
            // exposed load round trip per sample
            const float brd = HOT(o_border), hs = HOT(o_hscale), vs = HOT(o_vscale);
            const int TRr = HOT(o_terrain_rows), TCc = HOT(o_terrain_cols);
            float vx[HQ], vy[HQ];
#pragma unroll
            for (int i = 0; i < HQ; i++) { const int kc = min(hk0 + i * kstride, P - 1); vx[i] = B.height_points[2 * kc]; vy[i] = B.height_points[2 * kc + 1]; }
            int h1[HQ], h2[HQ], h3[HQ];
#pragma unroll
            for (int i = 0; i < HQ; i++) {
                const float tx = -2.f * yz * vy[i], ty = 2.f * yz * vx[i];
                const float rx = vx[i] + yw * tx - yz * ty, ry = vy[i] + yw * ty + yz * tx;
                // genesis_simulator.py:565-575: cell index by truncation, clipped, min over three neighbours
                int cx = (int)((rx + px + brd) / hs), cy = (int)((ry + py + brd) / hs);
                cx = min(max(cx, 0), TRr - 2);
                cy = min(max(cy, 0), TCc - 2);
                const int16_t *c = p.hf + cx * TCc + cy;
                h1[i] = c[0]; h2[i] = c[TCc]; h3[i] = c[1];
            }
#pragma unroll
            for (int i = 0; i < HQ; i++) {
                const int k = hk0 + i * kstride;
                hq[i] = (float)min(min(h1[i], h2[i]), h3[i]) * vs;
                if (live && k < P) B.measured_heights[(size_t)e * P + k] = hq[i];
            }
        } else {
            for (int k = hk0; k < P; k += kstride) {
                const float vx = B.height_points[2 * k], vy = B.height_points[2 * k + 1];
                const float tx = -2.f * yz * vy, ty = 2.f * yz * vx;
                const float rx = vx + yw * tx - yz * ty, ry = vy + yw * ty + yz * tx;
                const float h = sample_min3(O, p.hf, rx + px, ry + py);
                if (live) B.measured_heights[(size_t)e * P + k] = h;
            }
        }
        if (HOT(o_feet_terrain_info)) {
            const float fx = bc<0>(foot_p), fy = bc<1>(foot_p);
            int gx = (int)((fx + HOT(o_border)) / HOT(o_hscale)), gy = (int)((fy + HOT(o_border)) / HOT(o_hscale));
            gx = min(max(gx, 0), HOT(o_terrain_rows) - 2);
            gy = min(max(gy, 0), HOT(o_terrain_cols) - 2);
            const int Cc = HOT(o_terrain_cols), xm = max(gx - 1, 0), ym = max(gy - 1, 0);
            const int16_t *hf = p.hf;
            // order of genesis_simulator.py:591-599
            const int hh[9] = {hf[xm * Cc + gy], hf[(gx + 1) * Cc + gy], hf[gx * Cc + ym], hf[gx * Cc + gy + 1], hf[gx * Cc + gy],
                               hf[xm * Cc + ym], hf[(gx + 1) * Cc + gy + 1], hf[xm * Cc + gy + 1], hf[(gx + 1) * Cc + ym]};
            {
                float sm = 0.f, mx = -1e30f;
#pragma unroll
                for (int k = 0; k < 9; k++) { const float hv = (float)hh[k] * HOT(o_vscale); sm += hv; mx = fmaxf(mx, hv); f_h9[k] = hv; }
                f_hmean = sm / 9.f; f_hmax = mx;
                const float dx_ = (float)(hh[1] - hh[0]) / (HOT(o_hscale) * 2.f), dy_ = (float)(hh[3] - hh[2]) / (HOT(o_hscale) * 2.f);
                const float nn_ = fsqrt(dx_ * dx_ + dy_ * dy_ + 1.f);
                f_n3[0] = dx_ / nn_; f_n3[1] = dy_ / nn_; f_n3[2] = -1.f / nn_;      // genesis_simulator.py:601-606 (raw int16 differences, reproduced)
            }
            if (live && L.is0) {
#pragma unroll
                for (int k = 0; k < 9; k++) B.height_around_feet[((size_t)e * F + foot_slot) * 9 + k] = (float)hh[k] * HOT(o_vscale);
                const float dx = (float)(hh[1] - hh[0]) / (HOT(o_hscale) * 2.f), dy = (float)(hh[3] - hh[2]) / (HOT(o_hscale) * 2.f);
                const float nn = fsqrt(dx * dx + dy * dy + 1.f);
                st3(B.normal_vector_around_feet + ((size_t)e * F + foot_slot) * 3, v3(dx / nn, dy / nn, -1.f / nn));
            }
        }
    }
    if (!INJ && !PLANE && B.link_contact_states && live) {   // genesis_simulator.py:53-55
        const unsigned mask = m_smask;
        const int l0 = foot_link - 3, nst = __popc(mask);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float fn = fsqrt(dot3(f_link[k], f_link[k]));
            const int l = l0 + k;
            if (L.is0 && ((mask >> l) & 1u)) B.link_contact_states[(size_t)e * nst + __popc(mask & ((1u << l) - 1u))] = fn > 1.f ? 1.f : 0.f;
        }
        const float fb = fsqrt(dot3(f_base, f_base));
        if (L.is0 && leg == 0 && (mask & 1u)) B.link_contact_states[(size_t)e * nst] = fb > 1.f ? 1.f : 0.f;
    }

    if (DUO) __syncthreads();   // role 0's read-back stores are issued; role 1's tail may now store into the same arrays (resets)
    STAMP(22);
    STAMPB(8192);
    // ---------------- MDP phases in the same launch, go2-flat and go2_wtw profiles: component layout, all 64 lanes ------------
    // Same statements as env_step_body's POST / RESET phases (legged_robot.py:55-168, 300-348, go2.py:17-134) for the plain go2
    // task, computed on the registers the physics left behind: per-joint values sit in joint lanes, vectors one component per
    // lane, per-env scalars replicated over the env's 16 lanes.  No LDS hand-off, every lane loads / stores its own share of the
    // MDP state (two episode sums, one command component ...), and independent Philox blocks are evaluated side by side in
    // different lanes (a call is ~800 cycles of quarter-rate multiplies whatever the number of lanes using it).  The random
    // stream is the one of env_step_body (same counters), so both instantiations produce the same rollout.
    if constexpr (QTAIL) {
        STAMP(5);
        // hot constants of the MDP phases in ONE burst of LDS reads (a read at the point of use costs an exposed LDS round trip each:
        // one wave per SIMD, every reward term in its own basic block)
        const auto h_about_landing_threshold = HOT(about_landing_threshold);
        const auto h_add_noise = HOT(add_noise);
        const auto h_base_height_target = HOT(base_height_target);
        const auto h_base_init_quat_0 = HOT(base_init_quat[0]);
        const auto h_base_init_quat_1 = HOT(base_init_quat[1]);
        const auto h_base_init_quat_2 = HOT(base_init_quat[2]);
        const auto h_base_init_quat_3 = HOT(base_init_quat[3]);
        const auto h_clip_obs = HOT(clip_obs);
        const auto h_control_dt = HOT(control_dt);
        const auto h_dr_com_lo_0 = HOT(dr_com_lo[0]);
        const auto h_dr_com_lo_1 = HOT(dr_com_lo[1]);
        const auto h_dr_com_lo_2 = HOT(dr_com_lo[2]);
        const auto h_dr_com_on = HOT(dr_com_on);
        const auto h_dr_com_span_0 = HOT(dr_com_span[0]);
        const auto h_dr_com_span_1 = HOT(dr_com_span[1]);
        const auto h_dr_com_span_2 = HOT(dr_com_span[2]);
        const auto h_dr_friction_lo = HOT(dr_friction_lo);
        const auto h_dr_friction_on = HOT(dr_friction_on);
        const auto h_dr_friction_span = HOT(dr_friction_span);
        const auto h_dr_mass_lo = HOT(dr_mass_lo);
        const auto h_dr_mass_on = HOT(dr_mass_on);
        const auto h_dr_mass_span = HOT(dr_mass_span);
        const auto h_env_id_offset = HOT(env_id_offset);
        const auto h_fail_threshold = HOT(fail_threshold);
        const auto h_feet_air_time_threshold = HOT(feet_air_time_threshold);
        const auto h_foot_clearance_sigma = HOT(foot_clearance_sigma);
        const auto h_foot_clearance_target = HOT(foot_clearance_target);
        const auto h_foot_height_offset = HOT(foot_height_offset);
        const auto h_heading_command = HOT(heading_command);
        const auto h_max_episode_length = HOT(max_episode_length);
        const auto h_max_projected_gravity = HOT(max_projected_gravity);
        const auto h_max_push_vel_xy = HOT(max_push_vel_xy);
        const auto h_noise_lead_0 = HOT(noise_lead[0]);
        const auto h_noise_lead_1 = HOT(noise_lead[1]);
        const auto h_noise_lead_2 = HOT(noise_lead[2]);
        const auto h_noise_lead_3 = HOT(noise_lead[3]);
        const auto h_noise_lead_4 = HOT(noise_lead[4]);
        const auto h_noise_lead_5 = HOT(noise_lead[5]);
        const auto h_o_base_init_pos_0 = HOT(o_base_init_pos[0]);
        const auto h_o_base_init_pos_1 = HOT(o_base_init_pos[1]);
        const auto h_o_base_init_pos_2 = HOT(o_base_init_pos[2]);
        const auto h_obs_scale_ang_vel = HOT(obs_scale_ang_vel);
        const auto h_obs_scale_dof_pos = HOT(obs_scale_dof_pos);
        const auto h_obs_scale_dof_vel = HOT(obs_scale_dof_vel);
        const auto h_obs_scale_lin_vel = HOT(obs_scale_lin_vel);
        const auto h_obs_sets = HOT(obs_sets);
        const auto h_only_positive_rewards = HOT(only_positive_rewards);
        const auto h_push_interval = HOT(push_interval);
        const auto h_resample_steps = HOT(resample_steps);
        const auto h_reset_ang_vel_lo = HOT(reset_ang_vel_lo);
        const auto h_reset_lin_vel_lo = HOT(reset_lin_vel_lo);
        const auto h_seed = HOT(seed);
        const auto h_slots_cb_cmd = HOT(slots.cb_cmd);
        const auto h_slots_push = HOT(slots.push);
        const auto h_slots_reset_dof = HOT(slots.reset_dof);
        const auto h_tracking_sigma = HOT(tracking_sigma);
        const auto h_yaw_clip_0 = HOT(yaw_clip[0]);
        const auto h_yaw_clip_1 = HOT(yaw_clip[1]);
        // go2_wtw only (the reads fold away in the go2 instantiation)
        const auto h_b_swing = WQ ? HOT(b_swing) : 0.f;
        const auto h_base_height_sigma = WQ ? HOT(base_height_sigma) : 1.f;
        const auto h_euler_sigma = WQ ? HOT(euler_sigma) : 1.f;
        const auto h_behavior_resample_steps = WQ ? HOT(behavior_resample_steps) : 0;
        const auto h_slots_task_cb = WQ ? HOT(slots.task_cb) : 0;
        const auto h_slots_task_reset = WQ ? HOT(slots.task_reset) : 0;
        const auto h_slots_dr_kp = SQ ? HOT(slots.dr_kp) : 0;
        const auto h_slots_dr_kd = SQ ? HOT(slots.dr_kd) : 0;
        const auto h_dr_pd_on = SQ ? HOT(dr_pd_on) : 0;
        const auto h_dr_kp_lo = SQ ? HOT(dr_kp_lo) : 0.f;
        const auto h_dr_kp_span = SQ ? HOT(dr_kp_span) : 0.f;
        const auto h_dr_kd_lo = SQ ? HOT(dr_kd_lo) : 0.f;
        const auto h_dr_kd_span = SQ ? HOT(dr_kd_span) : 0.f;
        const auto h_reset_lin_vel_span = SQ ? HOT(reset_lin_vel_span) : 0.f;
        const auto h_reset_ang_vel_span = SQ ? HOT(reset_ang_vel_span) : 0.f;
        const auto h_obs_frame = SQ ? HOT(obs_frame) : 0;
        const auto h_priv_frame = SQ ? HOT(priv_frame) : 0;
        const auto h_obs_stack = SQ ? HOT(obs_stack) : 1;
        const auto h_priv_stack = SQ ? HOT(priv_stack) : 1;
        const auto h_obs_slack = SQ ? HOT(obs_slack) : 0;
        const auto h_num_obs = SQ ? HOT(num_obs) : 0;
        const auto h_num_priv_obs = SQ ? HOT(num_priv_obs) : 0;
        // go2_ee only
        const auto h_terrain_curriculum = EQ ? HOT(terrain_curriculum) : 0;
        const auto h_max_terrain_level = EQ ? HOT(max_terrain_level) : 1;
        const auto h_terrain_cols_n = EQ ? HOT(terrain_cols_n) : 1;
        const auto h_terrain_env_length = EQ ? HOT(terrain_env_length) : 0.f;
        const auto h_episode_length_s = EQ ? HOT(episode_length_s) : 0.f;
        const auto h_slots_reset_root_xy = EQ ? HOT(slots.reset_root_xy) : 0;
        const auto h_slots_terrain_level = EQ ? HOT(slots.terrain_level) : 0;
        const auto h_reset_root_xy_lo = EQ ? HOT(reset_root_xy_lo) : 0.f;
        const auto h_reset_root_xy_span = EQ ? HOT(reset_root_xy_span) : 0.f;
        const auto h_custom_origins = EQ ? HOT(custom_origins) : 0;
        const auto h_heights_offset = EQ ? HOT(heights_offset) : 0.f;
        const auto h_heights_clip_scale = EQ ? HOT(heights_clip_scale) : 0;
        const auto h_obs_scale_height = EQ ? HOT(obs_scale_height) : 1.f;
        const auto h_num_labels = EQ ? HOT(num_labels) : 0;
        const auto h_foot_clearance_ref = EQ ? HOT(foot_clearance_ref) : 0;
        const auto h_friction_offset = EQ ? HOT(friction_offset) : 0.f;
        const auto h_kp_offset = EQ ? HOT(kp_offset) : 0.f;
        const auto h_kd_offset = EQ ? HOT(kd_offset) : 0.f;
        asm volatile("" ::: "memory");
        const float cdt = h_control_dt;
        const unsigned rmask = RS ? RS_MASK : (unsigned)p.k.reward_mask;
        const bool heading = h_heading_command != 0;
        unsigned k0, k1, e_lo, e_hi;
        {
            const unsigned long long seed = h_seed, gid = (unsigned long long)(h_env_id_offset + e);
            k0 = (unsigned)(seed & 0xFFFFFFFFu); k1 = (unsigned)(seed >> 32);
            e_lo = (unsigned)(gid & 0xFFFFFFFFu); e_hi = (unsigned)(gid >> 32);
        }
        const unsigned rstep = (unsigned)p.counter;
        const float *const rin = INJ ? B.rand_in + (size_t)e * HOT(slots.n_slots) : nullptr;   // this env's injected uniforms (LgRandSlots)
        auto philox = [&](unsigned c3) { const U4 c = {e_lo, e_hi, rstep, c3}; return philox4x32_10(c, k0, k1); };
        auto pick = [](const U4 &r, int k) { return k == 0 ? r.x : (k == 1 ? r.y : (k == 2 ? r.z : r.w)); };
        auto anyl = [](bool b) { return __builtin_amdgcn_ballot_w64(b) != 0ull; };
        auto lane_of_row = [&](int l16) { return (int)((tl_ & 48u) | (unsigned)l16) << 2; };   // byte address for ds_bpermute
        auto fetch = [&](float v, int l16) { return __int_as_float(__builtin_amdgcn_ds_bpermute(lane_of_row(l16), __float_as_int(v))); };
        auto jsum = [&](float v) { return bc<0>(legsum<LEGS>(sum3(v))); };       // over the env's joints (value in joint lanes), to all lanes
        auto vnorm2 = [&](float v) { return bc<0>(sum3(v * v)); };               // |v|^2 of a component-layout vector, to the quad

        float cmdv = m_cmd, air = m_air, es0 = m_es0, es1 = m_es1;
        float o_fric = dr_fric, o_mass = dr_mass, o_com = dr_com, o_kp = dr_kp, o_kd = dr_kd, o_push = w_push;   // what the critic frame shows (go2_wtw)
        int ep_len = m_ep + 1, failb = m_fail, last_contact = m_lc;              // legged_robot.py:60
        const float CRlo = L.is0 ? CR(0) : (L.is1 ? CR(2) : (L.is2 ? CR(4) : CR(6)));
        const float CRhi = L.is0 ? CR(1) : (L.is1 ? CR(3) : (L.is2 ? CR(5) : CR(7)));
        auto resample = [&](float cv, float u0, float u1, float u2) {            // legged_robot.py:317-334
            const float u = L.is0 ? u0 : (L.is1 ? u1 : u2);
            const bool upd = heading ? !L.is2 : !L.is3;                          // heading mode draws the heading, not the yaw rate
            cv = upd ? (CRhi - CRlo) * u + CRlo : cv;
            const float keep = sqrtf(bc<0>(sum3(cv * cv))) > 0.2f ? 1.f : 0.f;
            return L.is3 ? cv : cv * keep;
        };
        // ---- _post_physics_step_callback (legged_robot.py:300-315) ----
        {
            const bool need = (ep_len % h_resample_steps) == 0;
            if (anyl(need)) {
                float u0, u1, u2;
                if constexpr (INJ) { u0 = rin[h_slots_cb_cmd]; u1 = rin[h_slots_cb_cmd + 1]; u2 = rin[h_slots_cb_cmd + 2]; }
                else { const U4 r = philox(0x40000000u + (unsigned)h_slots_cb_cmd); u0 = u01(r.x); u1 = u01(r.y); u2 = u01(r.z); }
                const float nc = resample(cmdv, u0, u1, u2);
                cmdv = need ? nc : cmdv;
            }
        }
        if (heading) {   // forward = quat_apply(base_quat, [1,0,0]) (math_utils.py:34-40): t = 2 xyz x b = (0, 2 qz, -2 qy)
            const float ty = 2.f * qz, tz = -2.f * qy;
            const float fx = 1.f + (qy * tz - qz * ty), fy = ty * qw + (0.f - qx * tz);
            const float hd = atan2f(fy, fx);
            const float c2 = clampf(0.5f * wrap_to_pi(bc<3>(cmdv) - hd), h_yaw_clip_0, h_yaw_clip_1);
            cmdv = L.is2 ? c2 : cmdv;
        }
        {
            const int pi_ = h_push_interval;
            if (pi_ > 0 && (p.counter % pi_) == 0) {   // genesis_simulator.py:150-158; lanes 0 / 1 evaluate the two draws' blocks side by side
                const int slot = h_slots_push + (L.is1 ? 1 : 0);
                float up;
                if constexpr (INJ) up = rin[slot];
                else { const U4 r = philox((unsigned)(slot >> 2)); up = u01(pick(r, slot & 3)); }
                const float m = h_max_push_vel_xy;
                const float pv = (m + m) * up - m;
                const bool xy = L.c < 2;
                vw = xy ? vw + pv : vw;
                o_push = xy ? pv : o_push;
                if (live && leg == 0 && xy) { B.rand_push_vels[3 * e + L.c] = pv; B.base_lin_vel_w[3 * e + L.c] = vw; }
            }
        }
        // ---- go2_wtw behaviour parameters (go2_wtw.py:180-218): lane c of every quad evaluates the block that holds draw `slot + c` and
        //      picks it -- gait period, base-height, foot-clearance and pitch targets side by side --, one more call with the
        //      env-independent counter picks the gait (one index per call for the whole batch, as in the reference)
        float gait_time = w_gt, phi = w_phi, gait_period = w_gp, bh_tgt = w_bh, fc_tgt = w_fc, pitch_tgt = w_pt, theta = w_th, expC = w_ec;
        // `u`: lane c holds the uniform of parameter c (gait period, base height, clearance, pitch); `ug`: the gait draw, replicated
        auto behavior_apply = [&](const float u, const float ug, const bool who) {
            const float lo = L.is0 ? CR(8) : (L.is1 ? CR(10) : (L.is2 ? CR(12) : CR(14)));
            const float hi = L.is0 ? CR(9) : (L.is1 ? CR(11) : (L.is2 ? CR(13) : CR(15)));
            const float val = (hi - lo) * u + lo;
            const int ng = (int)CR(16);
            const int sel_l = min((int)floorf(ug * (float)ng), ng - 1);
            // Philox: one gait draw per call for the whole job (env-independent counter), hence wave-uniform; injected rows carry it per env
            const int sel = INJ ? sel_l : __builtin_amdgcn_readfirstlane(sel_l);
            const float GAS *tt = &kT->theta_table[sel][0];
            const float nth = tt[foot_slot];
            const float t0 = tt[0], t1 = tt[1], t2 = tt[2], t3 = tt[3];
            float nfc = bc<2>(val);
            // pronk / bound gaits keep the lowest clearance target (go2_wtw.py:212-218)
            if (t0 == 0.f && t1 == 0.f && ((t2 == 0.f && t3 == 0.f) || (t2 == 0.5f && t3 == 0.5f))) nfc = CR(12);
            const float ngp = bc<0>(val), nbh = bc<1>(val), npt = bc<3>(val);
            if (who) { gait_period = ngp; bh_tgt = nbh; fc_tgt = nfc; pitch_tgt = npt; theta = nth; }
        };
        auto philox_e = [&](unsigned elo, unsigned ehi, unsigned c3) { const U4 c = {elo, ehi, rstep, c3}; return philox4x32_10(c, k0, k1); };
        if constexpr (WQ) {
            const int brs = h_behavior_resample_steps;
            const bool needb = brs > 0 && (ep_len % brs) == 0;                  // go2_wtw.py:258-263
            if (anyl(needb)) {   // rare (every resampling_time): two calls, the parameters' blocks side by side in the quad's lanes
                const int s = h_slots_task_cb + L.c;
                if constexpr (INJ) behavior_apply(rin[s], rin[h_slots_task_cb + 4], needb);
                else {
                    const U4 r = philox((unsigned)(s >> 2));
                    const U4 rg = philox_e(0xFFFFFFFFu, 0xFFFFFFFFu, (unsigned)((h_slots_task_cb + 4) >> 2));
                    behavior_apply(u01(pick(r, s & 3)), u01(pick(rg, (h_slots_task_cb + 4) & 3)), needb);
                }
            }
        }
        const float cmd0 = bc<0>(cmdv), cmd1 = bc<1>(cmdv), cmd2 = bc<2>(cmdv);
        STAMP(6);
        // ---- check_termination (legged_robot.py:78-92) ----
        const unsigned tmask = m_tmask, pmask = m_pmask;
        const int l0 = foot_link - 3;
        float n2[4];
#pragma unroll
        for (int k = 0; k < 4; k++) n2[k] = vnorm2(f_link[k]);
        const float nb2 = vnorm2(f_base);
        const float pgz = bc<2>(pg);
        int fail = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) fail |= (((tmask >> (l0 + k)) & 1u) && n2[k] > 100.0f) ? 1 : 0;
        fail = __builtin_amdgcn_update_dpp(0, fail, 0x124, 0xF, 0xF, true) | fail;
        fail = __builtin_amdgcn_update_dpp(0, fail, 0x128, 0xF, 0xF, true) | fail;
        fail |= ((tmask & 1u) && nb2 > 100.0f) ? 1 : 0;
        fail |= pgz > h_max_projected_gravity ? 1 : 0;
        if (guard_bad) failb = LG_FAIL_NONFINITE;   // a re-seated env ends its episode here (lgsim.h)
        failb += fail;
        const bool time_out = (float)ep_len > h_max_episode_length;
        const bool reset = ((float)failb > h_fail_threshold) || time_out;

        // ---- compute_reward (legged_robot.py:150-168): every term replicated over the env's lanes, summed in alphabetical order ----
        float scl[LG_R_COUNT];
#pragma unroll
        for (int k = 0; k < LG_R_COUNT; k++) scl[k] = HOT(reward_scales[k]);
        float total = 0.f;
        auto add = [&](int id, float r) {
            const float rew = r * scl[id];
            total += rew;
            if (id < 16) es0 = ei == id ? es0 + rew : es0;
            else es1 = ei == id - 16 ? es1 + rew : es1;
        };
        const float cmd_xy = sqrtf(cmd0 * cmd0 + cmd1 * cmd1);
        const float cmd_xyz = sqrtf(cmd0 * cmd0 + cmd1 * cmd1 + cmd2 * cmd2);
        const float dq0 = q - q0;
        const float fz = bc<2>(f_link[3]);                          // vertical foot force of this leg
        const float fpz = bc<2>(foot_p), fvx = bc<0>(foot_v), fvy = bc<1>(foot_v), fvz = bc<2>(foot_v);
        if (RON(LG_R_ACTION_RATE)) { const float d = last_act - act; add(LG_R_ACTION_RATE, jsum(d * d)); }                     // :495-497
        if (RON(LG_R_ACTION_SMOOTHNESS)) { const float d = act - 2.f * last_act + llast_act; add(LG_R_ACTION_SMOOTHNESS, jsum(d * d)); }   // :499-503
        if (RON(LG_R_ANG_VEL_XY)) { const float bx = bc<0>(bav), by = bc<1>(bav); add(LG_R_ANG_VEL_XY, bx * bx + by * by); }    // :462-464
        if (RON(LG_R_BASE_HEIGHT)) { const float d = bc<2>(pos) - h_base_height_target; add(LG_R_BASE_HEIGHT, d * d); }     // :470-476 (plane)
        if (RON(LG_R_COLLISION)) {                                                                                               // :505-512
            float sc_ = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) sc_ += (((pmask >> (l0 + k)) & 1u) && n2[k] > 0.1f * 0.1f) ? 1.f : 0.f;
            sc_ = legsum<LEGS>(sc_);
            sc_ += ((pmask & 1u) && nb2 > 0.1f * 0.1f) ? 1.f : 0.f;
            add(LG_R_COLLISION, sc_);
        }
        if (RON(LG_R_DOF_ACC)) { const float d = (qd_start - qd) / cdt; add(LG_R_DOF_ACC, jsum(d * d)); }                       // :490-493
        if (RON(LG_R_DOF_CLOSE_TO_DEFAULT)) add(LG_R_DOF_CLOSE_TO_DEFAULT, jsum(dq0 * dq0));                                     // :571-573
        if (RON(LG_R_DOF_POS_LIMITS)) add(LG_R_DOF_POS_LIMITS, jsum(-fminf(q - m_slo, 0.f) + fmaxf(q - m_shi, 0.f)));            // :518-522
        if (RON(LG_R_DOF_POS_STAND_STILL)) add(LG_R_DOF_POS_STAND_STILL, jsum(dq0 * dq0) * (cmd_xyz < 0.1f ? 1.f : 0.f));        // :561-563
        if (RON(LG_R_DOF_POWER)) add(LG_R_DOF_POWER, jsum(fabsf(torque * qd)));                                                  // :486-488
        if (RON(LG_R_DOF_VEL)) add(LG_R_DOF_VEL, jsum(qd * qd));                                                                 // :482-484
        if (RON(LG_R_DOF_VEL_STAND_STILL)) add(LG_R_DOF_VEL_STAND_STILL, jsum(fabsf(qd)) * (cmd_xyz < 0.1f ? 1.f : 0.f));        // :557-559
        if (RON(LG_R_FEET_AIR_TIME)) {                                                                                           // :545-555 (stateful)
            const int contact = fz > 1.0f ? 1 : 0;
            const int filt = contact | last_contact;
            last_contact = contact;
            const float first = (air > 0.f ? 1.f : 0.f) * (float)filt;
            air += cdt;
            float r = legsum<LEGS>((air - h_feet_air_time_threshold) * first);
            r *= cmd_xy > 0.1f ? 1.f : 0.f;
            air *= filt ? 0.f : 1.f;
            add(LG_R_FEET_AIR_TIME, r);
        }
        if (RON(LG_R_FEET_CONTACT_STAND_STILL)) {                                                                                // :565-569
            const float cnt = legsum<LEGS>(fz > 0.1f ? 1.f : 0.f);
            add(LG_R_FEET_CONTACT_STAND_STILL, (cnt == (float)LEGS ? 1.f : 0.f) * (cmd_xyz < 0.1f ? 1.f : 0.f));
        }
        if (RON(LG_R_FOOT_ACC)) { const float a = (foot_v - last_foot_v) * (1.f / cdt); add(LG_R_FOOT_ACC, legsum<LEGS>(vnorm2(a))); }    // :605-608
        if (RON(LG_R_FOOT_CLEARANCE)) {                                                                                          // :575-588
            const float vxy = sqrtf(fvx * fvx + fvy * fvy);
            const float d = fpz - (h_foot_clearance_ref == 1 ? f_hmean : (h_foot_clearance_ref == 2 ? f_hmax : 0.f)) - h_foot_clearance_target - h_foot_height_offset;
            add(LG_R_FOOT_CLEARANCE, __expf(-legsum<LEGS>(vxy * (d * d)) / h_foot_clearance_sigma));
        }
        if (RON(LG_R_FOOT_LANDING_VEL)) {                                                                                        // :590-599
            const bool land = ((fpz - h_foot_height_offset) < h_about_landing_threshold) && !(fz > 0.1f) && (fvz < 0.f);
            const float vz = land ? fvz : 0.f;
            add(LG_R_FOOT_LANDING_VEL, legsum<LEGS>(vz * vz));
        }
        if (RON(LG_R_HIP_POS)) { const float h = bc<0>(dq0); add(LG_R_HIP_POS, legsum<LEGS>(h * h)); }                           // go2_ee.py:152-159
        if (RON(LG_R_KEEP_BALANCE)) add(LG_R_KEEP_BALANCE, 1.f);                                                                 // :601-603
        if (RON(LG_R_LIN_VEL_Z)) { const float z = bc<2>(blv); add(LG_R_LIN_VEL_Z, z * z); }                                     // :458-460
        if (RON(LG_R_NO_FLY)) add(LG_R_NO_FLY, legsum<LEGS>(fz > 0.1f ? 1.f : 0.f) == 1.f ? 1.f : 0.f);                         // tron1_pf.py:151-154
        if (RON(LG_R_ORIENTATION)) { const float x = bc<0>(pg), y = bc<1>(pg); add(LG_R_ORIENTATION, x * x + y * y); }           // :466-468
        if (WQ && RON(LG_R_QUAD_PERIODIC_GAIT)) {                                                                              // go2_wtw.py:377-484 ("step" indicator)
            const float two_pi = 6.283185307179586f;
            float ph = phi + theta;
            ph = (ph - floorf(ph)) * two_pi;
            const float b_sw = h_b_swing * two_pi;
            const float c_frc = (ph >= 0.f && ph < b_sw) ? -1.f : 0.f;
            const float c_spd = (ph >= b_sw && ph < two_pi) ? -1.f : 0.f;
            expC = c_frc;
            add(LG_R_QUAD_PERIODIC_GAIT, __expf(legsum<LEGS>(c_spd * sqrtf(vnorm2(foot_v)) + c_frc * sqrtf(n2[3]))));
        }
        if (RON(LG_R_TORQUES)) add(LG_R_TORQUES, jsum(torque * torque));                                                         // :478-480
        if (RON(LG_R_TRACKING_ANG_VEL)) { const float d = cmd2 - bc<2>(bav); add(LG_R_TRACKING_ANG_VEL, __expf(-(d * d) / h_tracking_sigma)); }   // :539-543
        if (WQ && RON(LG_R_TRACKING_BASE_HEIGHT)) { const float d = bc<2>(pos) - bh_tgt; add(LG_R_TRACKING_BASE_HEIGHT, __expf(-(d * d) / h_base_height_sigma)); }   // go2_wtw.py:495-500
        if (WQ && RON(LG_R_TRACKING_FOOT_CLEARANCE)) {                                                                           // go2_wtw.py:507-519
            const float vxy = sqrtf(fvx * fvx + fvy * fvy);
            const float d = fpz - fc_tgt - h_foot_height_offset;
            add(LG_R_TRACKING_FOOT_CLEARANCE, __expf(-legsum<LEGS>(vxy * (d * d)) / h_foot_clearance_sigma));
        }
        if (RON(LG_R_TRACKING_LIN_VEL)) {                                                                                        // :533-537
            const float dx = cmd0 - bc<0>(blv), dy = cmd1 - bc<1>(blv);
            add(LG_R_TRACKING_LIN_VEL, __expf(-(dx * dx + dy * dy) / h_tracking_sigma));
        }
        if (WQ && RON(LG_R_TRACKING_ORIENTATION)) {                                                                              // go2_wtw.py:502-505
            const float ex = bc<0>(eul), dp = bc<1>(eul) - pitch_tgt;
            add(LG_R_TRACKING_ORIENTATION, __expf(-(ex * ex + dp * dp) / h_euler_sigma));
        }
        if (h_only_positive_rewards) total = fmaxf(total, 0.f);                                                               // :161-162
        if (RON(LG_R_TERMINATION)) add(LG_R_TERMINATION, (reset && !time_out) ? 1.f : 0.f);                                      // :163-168
        if constexpr (WQ) {   // gait clock (go2_wtw.py:29-36)
            gait_time += cdt;
            if (gait_time >= gait_period - cdt / 2.f) gait_time = 0.f;
            phi = gait_time / gait_period;
        }

        STAMP(7);
        // ONE Philox call per step and lane for everything drawn every step or at a reset: the four lanes of a quad evaluate four
        // different blocks side by side -- lane 0 / 1: the observation-noise blocks 2 leg / 2 leg + 1 of env_step_body, lane 2: the
        // leg's `_reset_dofs` block, lane 3: the env-level reset block 0x200 + leg.  A launch ends with its slowest wave, and that is
        // always one with a reset in it: with the reset draws inside the call every wave makes anyway, a reset costs no Philox call
        // (a call is ~800 cycles of quarter-rate multiplies).
        U4 rall = {0u, 0u, 0u, 0u};      // (issuing this call under the start-of-kernel load burst paid 0.15 us for go2 until the lane table took the
        if constexpr (!INJ)               //  registers it was hiding in: +0.3 us since, taken out)
            rall = philox(L.is0 ? 0x80000000u + (unsigned)(2 * leg) : (L.is1 ? 0x80000000u + (unsigned)(2 * leg) + 1u
                          : (L.is2 ? 0x40000000u + (unsigned)(h_slots_reset_dof + d0) : 0x80000000u + 0x200u + (unsigned)leg)));
        const float rux = u01(rall.x), ruy = u01(rall.y), ruz = u01(rall.z), ruw = u01(rall.w);
        const float ud_p = L.sel(bc<2>(rux), bc<2>(ruy), bc<2>(ruz));                        // element c of lane 2's block
        const float ud = INJ ? rin[h_slots_reset_dof + d0 + cj] : ud_p;
        const float rc = L.sel4(bc<3>(rux), bc<3>(ruy), bc<3>(ruz), bc<3>(ruw));             // element c of lane 3's block
        // ---- reset_idx (legged_robot.py:94-148, go2.py:17-37, 119-134) + simulator.reset_idx (genesis_simulator.py:62-82) ----
        if (anyl(reset)) {
            // env-level uniforms (block 0x200 + leg sits in quad `leg`): the element each lane needs is fetched from the quad that
            // holds it: v0 = (cmd u0 u1 u2 | friction), v1 = (CoM xyz | mass), v2 / v3 = root twist (slots of env_step_body's eu[])
            float v0 = fetch(rc, L.c), v1 = fetch(rc, 4 + L.c);
            float v2 = SQ ? fetch(rc, 8 + L.c) : 0.f, v3 = SQ ? fetch(rc, 12 + L.c) : 0.f;   // root twist draws (slots 8-10, 12-14)
            if constexpr (INJ) {   // the same quantities from their slots of the injected row
                v0 = L.is3 ? rin[HOT(slots.dr_friction)] : rin[HOT(slots.reset_cmd) + cj];
                v1 = L.is3 ? rin[HOT(slots.dr_mass)] : rin[HOT(slots.dr_com) + cj];
                v2 = rin[HOT(slots.reset_lin_vel) + cj]; v3 = rin[HOT(slots.reset_ang_vel) + cj];
            }
            // go2_wtw: ONE more call for what only a reset of this task draws -- lane 0 / 1: the leg's kp / kd blocks
            // (genesis_simulator.py:735-739), lanes 2 / 3 of quads 0 and 1: the four behaviour parameters' draws, lane 2 of quad 2: the
            // gait draw with the env-independent counter (go2_wtw.py:124-142, 180-218)
            float nkp = 1.f, nkd = 1.f;
            if constexpr (WQ) {
                const int sb = h_slots_task_reset + (leg == 2 ? 4 : 2 * leg + (L.c - 2));    // lanes 2 / 3: behaviour draw index
                const bool gl = leg == 2;
                const unsigned c3 = L.is0 ? 0x40000000u + (unsigned)(h_slots_dr_kp + d0)
                                          : (L.is1 ? 0x40000000u + (unsigned)(h_slots_dr_kd + d0) : (unsigned)(sb >> 2));
                const bool envc = L.c < 2 || !gl;      // this lane's block is keyed on the env (the gait draw is not)
                U4 rB = {0u, 0u, 0u, 0u};
                if constexpr (!INJ) rB = philox_e(envc ? e_lo : 0xFFFFFFFFu, envc ? e_hi : 0xFFFFFFFFu, c3);
                const float bx = u01(rB.x), by = u01(rB.y), bz = u01(rB.z);
                const float kpu = L.sel(bc<0>(bx), bc<0>(by), bc<0>(bz)), kdu = L.sel(bc<1>(bx), bc<1>(by), bc<1>(bz));   // element c of lane 0's / lane 1's block
                nkp = h_dr_kp_span * (INJ ? rin[h_slots_dr_kp + d0 + cj] : kpu) + h_dr_kp_lo;
                nkd = h_dr_kd_span * (INJ ? rin[h_slots_dr_kd + d0 + cj] : kdu) + h_dr_kd_lo;
                const float ub = u01(pick(rB, sb & 3));                                            // valid in lanes 2 / 3
                // parameter c was drawn in lane 2 + (c & 1) of quad c >> 1; the gait draw in lane 2 of quad 2
                const float ubp = fetch(ub, (L.c < 2 ? 2 : 4) + L.c), ubg = fetch(ub, 10);
                if constexpr (INJ) behavior_apply(rin[h_slots_task_reset + L.c], rin[h_slots_task_reset + 4], reset);
                else behavior_apply(ubp, ubg, reset);
            }
            // go2_ee: ONE more call likewise -- lane 0 / 1: the leg's kp / kd blocks, lanes 2 / 3 of quad 0: the root xy draws
            // (legged_robot.py:288), lane 2 of quad 1: the terrain-level draw (legged_robot.py:266-268)
            float u_xy = 0.5f, u_tl = 0.f;
            if constexpr (EQ) {
                const int sb = leg == 0 ? h_slots_reset_root_xy + (L.c - 2) : h_slots_terrain_level;
                const unsigned c3 = L.is0 ? 0x40000000u + (unsigned)(h_slots_dr_kp + d0)
                                          : (L.is1 ? 0x40000000u + (unsigned)(h_slots_dr_kd + d0) : (unsigned)(sb >> 2));
                U4 rB = {0u, 0u, 0u, 0u};
                if constexpr (!INJ) rB = philox(c3);
                const float bx = u01(rB.x), by = u01(rB.y), bz = u01(rB.z);
                const float kpu = L.sel(bc<0>(bx), bc<0>(by), bc<0>(bz)), kdu = L.sel(bc<1>(bx), bc<1>(by), bc<1>(bz));
                nkp = h_dr_kp_span * (INJ ? rin[h_slots_dr_kp + d0 + cj] : kpu) + h_dr_kp_lo;
                nkd = h_dr_kd_span * (INJ ? rin[h_slots_dr_kd + d0 + cj] : kdu) + h_dr_kd_lo;
                const float ub = u01(pick(rB, sb & 3));          // valid in lanes 2 / 3
                u_xy = fetch(ub, 2 + (L.c & 1));                 // x / y draw for components 0 / 1 (quad 0, lanes 2 / 3)
                u_tl = fetch(ub, 6);                             // quad 1, lane 2
                if constexpr (INJ) { u_xy = rin[h_slots_reset_root_xy + (L.c & 1)]; u_tl = rin[h_slots_terrain_level]; }
            }
            // terrain curriculum (legged_robot.py:254-272 + genesis_simulator.py:140-148; skipped on the construction-time reset, where
            // the reference returns early because init_done is False)
            float norg = origin;
            int nlvl = w_lvl;
            if (EQ && h_terrain_curriculum && p.counter > 0) {
                const float dd = pos - origin;
                const float dx = bc<0>(dd), dy = bc<1>(dd);
                const float dist = sqrtf(dx * dx + dy * dy);
                const bool up = dist > h_terrain_env_length / 2.f;
                const bool down = (dist < sqrtf(cmd0 * cmd0 + cmd1 * cmd1) * h_episode_length_s * 0.5f) && !up;
                int lvl = w_lvl + (up ? 1 : 0) - (down ? 1 : 0);
                if (lvl >= h_max_terrain_level) lvl = min((int)floorf(u_tl * (float)h_max_terrain_level), h_max_terrain_level - 1);
                else lvl = max(lvl, 0);
                nlvl = lvl;
                norg = B.terrain_origins[((size_t)lvl * h_terrain_cols_n + w_type) * 3 + cj];
            }
            const float ncmd = resample(cmdv, bc<0>(v0), bc<1>(v0), bc<2>(v0));
            float ipos = L.sel(h_o_base_init_pos_0, h_o_base_init_pos_1, h_o_base_init_pos_2) + (EQ ? norg : origin);
            if (EQ && h_custom_origins && L.c < 2) ipos += h_reset_root_xy_span * u_xy + h_reset_root_xy_lo;      // legged_robot.py:288
            const float iq = L.is3 ? h_base_init_quat_3 : L.sel(h_base_init_quat_0, h_base_init_quat_1, h_base_init_quat_2);
            // go2.py:131-133 draws U(0, 0) (flat_profile: zero spans); go2_wtw: legged_robot.py:289-292
            const float nvw = h_reset_lin_vel_span * v2 + h_reset_lin_vel_lo, nww = h_reset_ang_vel_span * v3 + h_reset_ang_vel_lo;
            // quad broadcasts stay outside the divergent branch
            const float nfric = bc<3>(h_dr_friction_span * v0 + h_dr_friction_lo), nmass = bc<3>(h_dr_mass_span * v1 + h_dr_mass_lo);
            const float ncom = L.sel(h_dr_com_span_0, h_dr_com_span_1, h_dr_com_span_2) * v1 + L.sel(h_dr_com_lo_0, h_dr_com_lo_1, h_dr_com_lo_2);
            if (reset) {
                if (WQ) { gait_time = 0.f; phi = 0.f; }
                if (SQ && h_dr_pd_on) { o_kp = nkp; o_kd = nkd; }
                if (h_dr_friction_on) o_fric = nfric;
                if (h_dr_mass_on) o_mass = nmass;
                if (h_dr_com_on) o_com = ncom;
                cmdv = ncmd;
                q = q0 + (m_rsp * ud + m_rlo); qd = 0.f;
                act = 0.f; last_act = 0.f; llast_act = 0.f;
                pos = ipos; quat = iq; vw = nvw; ww = nww;
                air = 0.f; ep_len = 0; failb = 0;
            }
            // the reference stores the commanded reset twist verbatim in the body-frame properties (genesis_simulator.py:128-129)
            // and refreshes projected gravity (:125)
            const QM Rr = quat_rows(L, quat);
            const float npg = -L.sel(bc<2>(Rr.c0), bc<2>(Rr.c1), bc<2>(Rr.c2));
            if (reset) { blv = vw; bav = ww; pg = npg; }
            if (reset && st) {
                B.dof_pos[ja] = q; B.dof_vel[ja] = 0.f; B.last_dof_vel[ja] = 0.f;
                B.actions[ja] = 0.f; B.last_actions[ja] = 0.f; B.llast_actions[ja] = 0.f;
                if (SQ && h_dr_pd_on) { B.kp_scale[ja] = o_kp; B.kd_scale[ja] = o_kd; }
                if (EQ && h_terrain_curriculum && p.counter > 0 && leg == 0) { B.env_origins[3 * e + cj] = norg; if (L.is0) B.terrain_levels[e] = nlvl; }
                B.last_feet_vel[(e * F + foot_slot) * 3 + cj] = 0.f;
                if (leg == 0) {
                    B.base_pos[3 * e + cj] = pos; B.base_lin_vel_w[3 * e + cj] = vw; B.base_ang_vel_w[3 * e + cj] = ww;
                    B.base_lin_vel[3 * e + cj] = blv; B.base_ang_vel[3 * e + cj] = bav; B.projected_gravity[3 * e + cj] = pg;
                    B.last_base_lin_vel[3 * e + cj] = 0.f; B.last_base_ang_vel[3 * e + cj] = 0.f;
                    if (h_dr_com_on) B.base_com_bias[3 * e + cj] = L.sel(h_dr_com_span_0, h_dr_com_span_1, h_dr_com_span_2) * v1 +
                                                                       L.sel(h_dr_com_lo_0, h_dr_com_lo_1, h_dr_com_lo_2);
                }
            }
            if (reset && live && leg == 0) {
                B.base_quat[4 * e + L.c] = quat;
                if (L.is3) {   // the lane holding the fourth element of blocks 0 / 1: friction and mass draws
                    if (h_dr_friction_on) B.friction_values[e] = h_dr_friction_span * v0 + h_dr_friction_lo;
                    if (h_dr_mass_on) B.added_base_mass[e] = h_dr_mass_span * v1 + h_dr_mass_lo;
                    B.episode_done_step[e] = (int)p.counter;
                }
            }
            if (reset && live) {    // extras["episode"] snapshot (legged_robot.py:128-132), then the sums restart
                if ((rmask >> ei) & 1u) { B.episode_done_sums[(size_t)ei * N + e] = es0; es0 = 0.f; }
                if (ei + 16 < LG_R_COUNT && ((rmask >> (ei + 16)) & 1u)) { B.episode_done_sums[(size_t)(ei + 16) * N + e] = es1; es1 = 0.f; }
            }
        }
        STAMP(9);
        if constexpr (EQ) {
            // ---- compute_observations + clip, go2_ee.py:10-75: actor frame = go2's 45, stacked 20 deep; critic frame = the frame without
            //      noise | DR 31 (friction - offset, mass, CoM 3, push 2, kp - offset 12, kd - offset 12) | contact states K | heights P,
            //      stacked 5 deep; labels = v_b 3 | contact states K | foot height above the local terrain mean F.  Same window / copy /
            //      blanking bookkeeping as the go2_wtw block below; the env's 16 lanes write consecutive columns, every lane the terrain
            //      samples it took itself.
            const float co = h_clip_obs;
            const bool nz = h_add_noise != 0;
            const int FR = h_obs_frame, PF = h_priv_frame, ST = h_obs_stack, PST = h_priv_stack, SL = h_obs_slack;
            const size_t orow = (size_t)(h_num_obs + SL * FR), prow = (size_t)(h_num_priv_obs + SL * PF);
            const bool two = h_obs_sets > 1;
            const int cs = two ? p.obs_set : 0, xs = (two && cs < 2) ? 1 - cs : cs;   // the other copy (two sets); with more sets nothing is stacked and xs is unused
            float *oc = B.obs_buf + ((size_t)cs * N + e) * orow + (size_t)p.obs_win * FR;
            float *ox = B.obs_buf + ((size_t)xs * N + e) * orow + (size_t)p.obs_win * FR;
            float *pc = B.priv_obs_buf + ((size_t)cs * N + e) * prow + (size_t)p.obs_win * PF;
            float *px = B.priv_obs_buf + ((size_t)xs * N + e) * prow + (size_t)p.obs_win * PF;
            float *lab = B.labels_buf + ((size_t)cs * N + e) * h_num_labels;
            if (anyl(reset)) {                         // legged_robot_ee.py: the histories of a reset env restart from zeros
                blank_histories(__builtin_amdgcn_ballot_w64(reset && live && ei == 0), e, B.obs_buf + (size_t)cs * N * orow + (size_t)p.obs_win * FR, orow, (ST - 1) * FR,
                                B.priv_obs_buf + (size_t)cs * N * prow + (size_t)p.obs_win * PF, prow, (PST - 1) * PF);
            }
            if (two && anyl(!reset && w_dirty != 0)) {
                blank_histories(__builtin_amdgcn_ballot_w64(!reset && w_dirty != 0 && live && ei == 0), e, B.obs_buf + (size_t)cs * N * orow + (size_t)p.obs_win * FR, orow,
                                (ST - 2) * FR, B.priv_obs_buf + (size_t)cs * N * prow + (size_t)p.obs_win * PF, prow, (PST - 2) * PF);
            }
            float *on = oc + (ST - 1) * FR, *on2 = ox + (ST - 1) * FR, *pn = pc + (PST - 1) * PF, *pn2 = px + (PST - 1) * PF;
            const bool w2o = two && ST > 1, w2p = two && PST > 1;
            float uq = 0.5f, uqd = 0.5f, ug = 0.5f, ua = 0.5f;
            if (nz) {
                const float ux = rux, uy = ruy, uz = ruz, uw = ruw;   // lanes 0 / 1 hold blocks 2 leg / 2 leg + 1 (shared call above)
                uq = L.sel(bc<0>(ux), bc<0>(uy), bc<0>(uz));
                uqd = L.sel(bc<1>(ux), bc<1>(uy), bc<1>(uz));
                ug = fetch(uw, 4 * (cj >> 1) + (cj & 1));
                ua = fetch(uw, 4 * ((3 + cj) >> 1) + ((3 + cj) & 1));
                if constexpr (INJ) {   // the frame's entries of the injected row (slots.noise + index in the frame)
                    const int ns_ = HOT(slots.noise);
                    uq = rin[ns_ + 9 + d0 + cj]; uqd = rin[ns_ + 9 + A + d0 + cj]; ug = rin[ns_ + 3 + cj]; ua = rin[ns_ + 6 + cj];
                }
            }
            // observation programs (PROF 4): where the noise-free actor frame sits in the critic frame (-1: nowhere), where the "next state"
            // copy sits in the labels row (-1: none), whether the critic frame is clipped
            int pfo = 0, nxo = -1;
            float pclip = co;
            if constexpr (PQ) {
                pfo = -1;
                for (int s_ = 0; s_ < PRG_I(0, 0); s_++) if (PRG_I(0, 2 + s_) == LG_SEG_FRAME) pfo = PRG_I(0, 10 + s_);
                for (int s_ = 0; s_ < PRG_I(1, 0); s_++) if (PRG_I(1, 2 + s_) == LG_SEG_NEXT_STATE) nxo = PRG_I(1, 10 + s_);
                pclip = PRG_I(0, 1) ? co : 3.0e38f;
            }
            const float ascl = HOT(o_action_scale);
            auto W = [&](int idx, float v, float u, float ns) {
                const float nv = clampf(nz ? v + (2.f * u - 1.f) * ns : v, -co, co);
                on[idx] = nv; if (w2o) on2[idx] = nv;
                if (pfo >= 0) { const float cl = clampf(v, -pclip, pclip); pn[pfo + idx] = cl; if (w2p) pn2[pfo + idx] = cl; }
                if (PQ && nxo >= 0) lab[nxo + idx] = idx >= 9 + 2 * A ? v * ascl : v;      // go2_dreamwaq.py:66-74, not clipped
            };
            auto WP = [&](int idx, float v) { const float cl = clampf(v, -co, co); pn[idx] = cl; if (w2p) pn2[idx] = cl; };
            // quad broadcasts outside the divergent branches
            const float env4 = L.sel4(o_fric - h_friction_offset, o_mass, bc<0>(o_push), bc<1>(o_push));
            const float pzn = bc<2>(pos);
            const unsigned smask = m_smask;
            const int K = __popc(smask);
            const float csv = (L.sel4(n2[0], n2[1], n2[2], n2[3]) > 1.f) ? 1.f : 0.f;     // contact state of link l0 + c (physics read-back, stale after a reset as in the reference)
            const int cl_ = l0 + L.c;
            const bool chas = ((smask >> cl_) & 1u) != 0;
            const int cidx = __popc(smask & ((1u << cl_) - 1u));
            // the actor frame (both profiles)
            if (st) {
                W(9 + d0 + cj, (q - q0) * h_obs_scale_dof_pos, uq, m_nq);
                W(9 + A + d0 + cj, qd * h_obs_scale_dof_vel, uqd, m_nqd);
                W(9 + 2 * A + d0 + cj, act, 0.5f, 0.f);
                if (leg == 0) {
                    W(cj, cmdv * (L.is2 ? h_obs_scale_ang_vel : h_obs_scale_lin_vel), 0.5f, 0.f);
                    W(3 + cj, pg, ug, L.sel(h_noise_lead_0, h_noise_lead_1, h_noise_lead_2));
                    W(6 + cj, bav * h_obs_scale_ang_vel, ua, L.sel(h_noise_lead_3, h_noise_lead_4, h_noise_lead_5));
                }
            }
            if constexpr (!PQ) {
            if (st) {
                WP(FR + 7 + d0 + cj, o_kp - h_kp_offset);
                WP(FR + 7 + A + d0 + cj, o_kd - h_kd_offset);
                if (leg == 0) lab[cj] = blv * h_obs_scale_lin_vel;
                if (leg == 2) WP(FR + 2 + cj, o_com);
                if (L.is0) lab[3 + K + foot_slot] = clampf(fpz - f_hmean - h_foot_height_offset, -1.f, 1.f);
            }
            if (live) {
                if (leg == 1) WP(FR + (L.c < 2 ? L.c : 3 + L.c), env4);                 // friction, mass | push x, y at FR + 5, 6
                if (chas) { WP(FR + 7 + 2 * A + cidx, csv); lab[3 + cidx] = csv; }
                if (leg == 0 && L.is3 && (smask & 1u)) { const float cb = nb2 > 1.f ? 1.f : 0.f; WP(FR + 7 + 2 * A, cb); lab[3] = cb; }
#pragma unroll
                for (int i = 0; i < HQ; i++) {
                    const int k = hk0 + i * kstride;
                    float hv = pzn - h_heights_offset - hq[i];
                    if (h_heights_clip_scale) hv = clampf(hv, -1.f, 1.f) * h_obs_scale_height;
                    if (k < P) WP(FR + 7 + 2 * A + K + k, hv);
                }
            }
            } else {
                // observation programs (include/lgsim.h LgObsSeg): the critic frame and the auxiliary row are concatenations of blocks; each
                // block is written by the lanes that hold its values -- joint lanes, component lanes, the lane of a link, the lane that
                // took a terrain sample
                const float blvs = blv * h_obs_scale_lin_vel;
                const float clr = clampf(fpz - f_hmean - h_foot_height_offset, -1.f, 1.f);
                for (int which = 0; which < 2; which++) {
                    const bool to_lab = which == 1;
                    if (to_lab ? h_num_labels <= 0 : h_num_priv_obs <= 0) continue;
                    const float cl = PRG_I(which, 1) ? co : 3.0e38f;
                    auto WS = [&](int idx, float v) {
                        v = clampf(v, -cl, cl);
                        if (to_lab) lab[idx] = v;
                        else { pn[idx] = v; if (w2p) pn2[idx] = v; }
                    };
                    const int n_segs = PRG_I(which, 0);
                    for (int s_ = 0; s_ < n_segs; s_++) {
                        const int kind = PRG_I(which, 2 + s_), off = PRG_I(which, 10 + s_);
                        const float sc = PRG_F(which, 18 + s_);
                        if (kind == LG_SEG_DR || kind == LG_SEG_DR_BASE) {
                            if (st && kind == LG_SEG_DR) { WS(off + 7 + d0 + cj, o_kp - h_kp_offset); WS(off + 7 + A + d0 + cj, o_kd - h_kd_offset); }
                            if (st && leg == 2) WS(off + 2 + cj, o_com);
                            if (live && leg == 1) WS(off + (L.c < 2 ? L.c : 3 + L.c), env4);
                        } else if (kind == LG_SEG_KP || kind == LG_SEG_KD) {
                            if (st) WS(off + d0 + cj, kind == LG_SEG_KP ? o_kp - h_kp_offset : o_kd - h_kd_offset);
                        } else if (kind == LG_SEG_BASE_LIN_VEL) {
                            if (st && leg == 0) WS(off + cj, blvs * sc);
                        } else if (kind == LG_SEG_CONTACT_STATES) {
                            if (live && chas) WS(off + cidx, csv);
                            if (live && leg == 0 && L.is3 && (smask & 1u)) WS(off, nb2 > 1.f ? 1.f : 0.f);
                        } else if (kind == LG_SEG_HEIGHTS) {
                            if (live) {
#pragma unroll
                                for (int i = 0; i < HQ; i++) {
                                    const int k = hk0 + i * kstride;
                                    float hv = pzn - h_heights_offset - hq[i];
                                    if (h_heights_clip_scale) hv = clampf(hv, -1.f, 1.f) * h_obs_scale_height;
                                    if (k < P) WS(off + k, hv);
                                }
                            }
                        } else if (kind == LG_SEG_FEET_REL_HEIGHTS || kind == LG_SEG_FEET_HEIGHTS) {
                            if (live) {   // nine per foot: lane c of the leg's quad writes entries c, c + 4 (and 8)
#pragma unroll
                                for (int k = 0; k < 9; k++)
                                    if ((k & 3) == L.c) WS(off + 9 * foot_slot + k, kind == LG_SEG_FEET_HEIGHTS ? f_h9[k] : clampf(fpz - f_h9[k], -1.f, 1.f));
                            }
                        } else if (kind == LG_SEG_FEET_NORMALS) {
                            if (st) WS(off + 3 * foot_slot + cj, L.sel(f_n3[0], f_n3[1], f_n3[2]));
                        } else if (kind == LG_SEG_FOOT_CLEARANCE) {
                            if (live && L.is0) WS(off + foot_slot, clr);
                        } else if (kind == LG_SEG_LAST_ACTIONS) {
                            if (st) WS(off + d0 + cj, last_act);
                        } else if (kind == LG_SEG_FEET_AIR_TIME) {
                            if (live && L.is0) WS(off + foot_slot, air);
                        }   // LG_SEG_FRAME / LG_SEG_NEXT_STATE: written entry by entry in W(); LG_SEG_DR_JOINT: excluded by the host check
                    }
                }
            }
            if (live && leg == 0 && L.is0 && B.obs_dirty) B.obs_dirty[e] = reset ? 1 : 0;
        } else if constexpr (WQ) {
            // ---- compute_observations + clip, go2_wtw.py:53-111: actor frame 61 = go2's 45 | clock sin 4, cos 4 | gait period, base-height,
            //      clearance, pitch targets | theta 4; critic frame 99 = the frame without noise | v_b 3, push 2, mass, friction, CoM 3 |
            //      kp 12 | kd 12 | exp_C_frc 4.  Five-frame stacks as sliding windows over rows with slack, in `obs_sets` copies: the new
            //      frame goes to this launch's window in this set and to the same frame index of the other set; an env that resets
            //      blanks the older frames of its window, one that reset at the previous launch the frames older than that (`dirty`) --
            //      the same bookkeeping as env_step_body's, with the env's 16 lanes writing consecutive columns.
            const float co = h_clip_obs;
            const bool nz = h_add_noise != 0;
            const int FR = h_obs_frame, PF = h_priv_frame, ST = h_obs_stack, PST = h_priv_stack, SL = h_obs_slack;
            const size_t orow = (size_t)(h_num_obs + SL * FR), prow = (size_t)(h_num_priv_obs + SL * PF);
            const bool two = h_obs_sets > 1;
            const int cs = two ? p.obs_set : 0, xs = (two && cs < 2) ? 1 - cs : cs;   // the other copy (two sets); with more sets nothing is stacked and xs is unused
            float *oc = B.obs_buf + ((size_t)cs * N + e) * orow + (size_t)p.obs_win * FR;
            float *ox = B.obs_buf + ((size_t)xs * N + e) * orow + (size_t)p.obs_win * FR;
            float *pc = B.priv_obs_buf + ((size_t)cs * N + e) * prow + (size_t)p.obs_win * PF;
            float *px = B.priv_obs_buf + ((size_t)xs * N + e) * prow + (size_t)p.obs_win * PF;
            if (anyl(reset)) {                         // go2_wtw.py:174-178
                blank_histories(__builtin_amdgcn_ballot_w64(reset && live && ei == 0), e, B.obs_buf + (size_t)cs * N * orow + (size_t)p.obs_win * FR, orow, (ST - 1) * FR,
                                B.priv_obs_buf + (size_t)cs * N * prow + (size_t)p.obs_win * PF, prow, (PST - 1) * PF);
            }
            if (two && anyl(!reset && w_dirty != 0)) {
                blank_histories(__builtin_amdgcn_ballot_w64(!reset && w_dirty != 0 && live && ei == 0), e, B.obs_buf + (size_t)cs * N * orow + (size_t)p.obs_win * FR, orow,
                                (ST - 2) * FR, B.priv_obs_buf + (size_t)cs * N * prow + (size_t)p.obs_win * PF, prow, (PST - 2) * PF);
            }
            float *on = oc + (ST - 1) * FR, *on2 = ox + (ST - 1) * FR, *pn = pc + (PST - 1) * PF, *pn2 = px + (PST - 1) * PF;
            const bool w2o = two && ST > 1, w2p = two && PST > 1;
            float uq = 0.5f, uqd = 0.5f, ug = 0.5f, ua = 0.5f;
            if (nz) {   // same blocks as the go2 frame below
                const float ux = rux, uy = ruy, uz = ruz, uw = ruw;   // lanes 0 / 1 hold blocks 2 leg / 2 leg + 1 (shared call above)
                uq = L.sel(bc<0>(ux), bc<0>(uy), bc<0>(uz));
                uqd = L.sel(bc<1>(ux), bc<1>(uy), bc<1>(uz));
                ug = fetch(uw, 4 * (cj >> 1) + (cj & 1));
                ua = fetch(uw, 4 * ((3 + cj) >> 1) + ((3 + cj) & 1));
                if constexpr (INJ) {   // the frame's entries of the injected row (slots.noise + index in the frame)
                    const int ns_ = HOT(slots.noise);
                    uq = rin[ns_ + 9 + d0 + cj]; uqd = rin[ns_ + 9 + A + d0 + cj]; ug = rin[ns_ + 3 + cj]; ua = rin[ns_ + 6 + cj];
                }
            }
            // W: an actor-frame entry (noisy into the actor windows, noise-free into the critic frame); WP: a critic-only entry
            auto W = [&](int idx, float v, float u, float ns) {
                const float cl = clampf(v, -co, co);
                const float nv = clampf(nz ? v + (2.f * u - 1.f) * ns : v, -co, co);
                on[idx] = nv; if (w2o) on2[idx] = nv;
                pn[idx] = cl; if (w2p) pn2[idx] = cl;
            };
            auto WP = [&](int idx, float v) { const float cl = clampf(v, -co, co); pn[idx] = cl; if (w2p) pn2[idx] = cl; };
            const float ang = 6.283185307179586f * (phi + theta);        // clock inputs (go2_wtw.py:251-256)
            const float sn = sinf(ang), csn = cosf(ang);
            if (st) {
                W(9 + d0 + cj, (q - q0) * h_obs_scale_dof_pos, uq, m_nq);
                W(9 + A + d0 + cj, qd * h_obs_scale_dof_vel, uqd, m_nqd);
                W(9 + 2 * A + d0 + cj, act, 0.5f, 0.f);
                W((L.is0 ? 45 : (L.is1 ? 49 : 57)) + foot_slot, L.sel(sn, csn, theta), 0.5f, 0.f);
                WP(FR + 10 + d0 + cj, o_kp);
                WP(FR + 10 + A + d0 + cj, o_kd);
                if (L.is0) WP(FR + 10 + 2 * A + foot_slot, expC);
                if (leg == 0) {
                    W(cj, cmdv * (L.is2 ? h_obs_scale_ang_vel : h_obs_scale_lin_vel), 0.5f, 0.f);
                    W(3 + cj, pg, ug, L.sel(h_noise_lead_0, h_noise_lead_1, h_noise_lead_2));
                    W(6 + cj, bav * h_obs_scale_ang_vel, ua, L.sel(h_noise_lead_3, h_noise_lead_4, h_noise_lead_5));
                    WP(FR + cj, blv * h_obs_scale_lin_vel);
                }
                if (leg == 2) WP(FR + 7 + cj, o_com);
            }
            const float env4 = L.sel4(bc<0>(o_push), bc<1>(o_push), o_mass, o_fric);   // quad broadcasts outside the divergent branches
            if (live && leg == 0) W(53 + L.c, L.sel4(gait_period, bh_tgt, fc_tgt, pitch_tgt), 0.5f, 0.f);
            if (live && leg == 1) WP(FR + 3 + L.c, env4);
            // task state as the env class exposes it (go2_wtw.py:295-346), the deferred-blanking flag, the second action-history shift
            // (go2_wtw.py:45-46: afterwards last == llast == a_t)
            float *ts = B.task_state + (size_t)e * LG_TASK_STATE_WTW;
            if (st) {
                ts[(L.is0 ? 6 : (L.is1 ? 10 : 14)) + foot_slot] = L.sel(theta, sn, csn);
                if (L.is0) ts[18 + foot_slot] = expC;
                B.llast_actions[ja] = reset ? 0.f : last_act;
                B.last_actions[ja] = act;
            }
            if (live && leg == 0) {
                ts[L.c] = L.sel4(gait_time, phi, gait_period, bh_tgt);
                if (L.c < 2) ts[4 + L.c] = L.is0 ? fc_tgt : pitch_tgt;
                if (L.is0 && B.obs_dirty) B.obs_dirty[e] = reset ? 1 : 0;
            }
        } else
        // ---- compute_observations + clip (go2.py:40-64, legged_robot.py:48-49) ----
        {
            const float co = h_clip_obs;
            const bool nz = h_add_noise != 0;
            float uq = 0.5f, uqd = 0.5f, ug = 0.5f, ua = 0.5f;
            if (nz) {
                // blocks 2 leg and 2 leg + 1 of env_step_body in lanes 0 and 1 of the quad, side by side: (x, y, z) = the three joints'
                // uniforms (positions / velocities), w = one of the base's six
                const float ux = rux, uy = ruy, uz = ruz, uw = ruw;   // lanes 0 / 1 hold blocks 2 leg / 2 leg + 1 (shared call above)
                uq = L.sel(bc<0>(ux), bc<0>(uy), bc<0>(uz));
                uqd = L.sel(bc<1>(ux), bc<1>(uy), bc<1>(uz));
                // base uniform k sits in quad k / 2, lane k % 2: gravity component c takes k = c, angular velocity k = 3 + c
                ug = fetch(uw, 4 * (cj >> 1) + (cj & 1));
                ua = fetch(uw, 4 * ((3 + cj) >> 1) + ((3 + cj) & 1));
                if constexpr (INJ) {   // the frame's entries of the injected row (slots.noise + index in the frame)
                    const int ns_ = HOT(slots.noise);
                    uq = rin[ns_ + 9 + d0 + cj]; uqd = rin[ns_ + 9 + A + d0 + cj]; ug = rin[ns_ + 3 + cj]; ua = rin[ns_ + 6 + cj];
                }
            }
            float *o = B.obs_buf + ((size_t)(h_obs_sets > 1 ? p.obs_set : 0) * N + e) * (size_t)(9 + 3 * A);
            auto noisy = [&](float v, float u, float ns) { if (nz) v += (2.f * u - 1.f) * ns; return clampf(v, -co, co); };
            if (st) {
                o[9 + d0 + cj] = noisy((q - q0) * h_obs_scale_dof_pos, uq, m_nq);
                o[9 + A + d0 + cj] = noisy(qd * h_obs_scale_dof_vel, uqd, m_nqd);
                o[9 + 2 * A + d0 + cj] = clampf(act, -co, co);
                if (leg == 0) {
                    const float cs_ = L.is2 ? h_obs_scale_ang_vel : h_obs_scale_lin_vel;
                    o[cj] = clampf(cmdv * cs_, -co, co);
                    o[3 + cj] = noisy(pg, ug, L.sel(h_noise_lead_0, h_noise_lead_1, h_noise_lead_2));
                    o[6 + cj] = noisy(bav * h_obs_scale_ang_vel, ua, L.sel(h_noise_lead_3, h_noise_lead_4, h_noise_lead_5));
                }
            }
        }
        STAMP(10);
        // ---- persistent MDP state ----
        if (live) {
            // (32-bit byte offsets from wave-uniform bases, as the read-back stores: (term, N) rows stay below 4 GiB, host-checked)
            const unsigned o_es = 4u * ((unsigned)ei * (unsigned)N + (unsigned)e);
            if ((rmask >> ei) & 1u) stq(B.episode_sums, o_es, es0);
            if (ei + 16 < LG_R_COUNT && ((rmask >> (ei + 16)) & 1u)) stq(B.episode_sums, o_es + 64u * (unsigned)N, es1);
            if (L.is0) { stq(B.feet_air_time, 4u * (unsigned)(e * F + foot_slot), air); stq(B.last_contacts, (unsigned)(e * F + foot_slot), (uint8_t)last_contact); }
            if (leg == 0) {
                stq(B.commands, 4u * (unsigned)(4 * e + L.c), cmdv);
                if (L.is0) {
                    const unsigned ue = (unsigned)e;
                    stq(B.episode_length_buf, 4u * ue, (int32_t)ep_len); stq(B.fail_buf, 8u * ue, (int64_t)failb);
                    stq(B.reset_buf, ue, (uint8_t)(reset ? 1 : 0)); stq(B.time_out_buf, ue, (uint8_t)(time_out ? 1 : 0)); stq(B.rew_buf, 4u * ue, total);
                }
            }
        }
        STAMP(11);
    }
    // ---------------- MDP phases in the same launch, tron1_pf_ee profile: component layout on the env's eight lanes ---------------------
    // Same statements as env_step_body's POST / RESET phases for LG_OBS_TRON1_EE (legged_robot.py:55-168, 300-348; tron1_pf_ee.py:12-141,
    // 193-256, 347-463), computed on the registers the physics left behind like the quadruped tails above: per-joint values in joint lanes,
    // vectors one component per lane, per-env scalars replicated over the env's 8 lanes (lane el = 4 leg + c owns episode sums el + 8 k).
    // The random stream is env_step_body's (same counters, same words), three Philox calls per lane at most:
    //   A (every step)  lane c of leg l: observation-noise block 2 l | 2 l + 1 | 2 LEGS + 2 + l (actions) | 3 LEGS + 2 + l (clock)
    //   B (every step)  lane 0: noise block 2 LEGS (base); lanes 1-6: the env-level reset bundle 0x200 + 0 .. 5; lane 7: the job-wide sit coin
    //   C (a reset in the wave)  lane c of leg l: `_reset_dofs` | kp | kd blocks of the leg; lane 3 of leg 0: the root xy block
    if constexpr (BQ) {
        STAMP(5);
        const auto h_about_landing_threshold = HOT(about_landing_threshold);
        const auto h_add_noise = HOT(add_noise);
        const auto h_air_time_cmd_dims = HOT(air_time_cmd_dims);
        const auto h_b_swing = HOT(b_swing);
        const auto h_base_height_sigma = HOT(base_height_sigma);
        const auto h_base_height_target = HOT(base_height_target);
        const auto h_base_init_quat_0 = HOT(base_init_quat[0]);
        const auto h_base_init_quat_1 = HOT(base_init_quat[1]);
        const auto h_base_init_quat_2 = HOT(base_init_quat[2]);
        const auto h_base_init_quat_3 = HOT(base_init_quat[3]);
        const auto h_clip_obs = HOT(clip_obs);
        const auto h_control_dt = HOT(control_dt);
        const auto h_custom_origins = HOT(custom_origins);
        const auto h_dr_com_lo_0 = HOT(dr_com_lo[0]);
        const auto h_dr_com_lo_1 = HOT(dr_com_lo[1]);
        const auto h_dr_com_lo_2 = HOT(dr_com_lo[2]);
        const auto h_dr_com_on = HOT(dr_com_on);
        const auto h_dr_com_span_0 = HOT(dr_com_span[0]);
        const auto h_dr_com_span_1 = HOT(dr_com_span[1]);
        const auto h_dr_com_span_2 = HOT(dr_com_span[2]);
        const auto h_dr_friction_lo = HOT(dr_friction_lo);
        const auto h_dr_friction_on = HOT(dr_friction_on);
        const auto h_dr_friction_span = HOT(dr_friction_span);
        const auto h_dr_joint_lo_0 = HOT(dr_joint_lo[0]);
        const auto h_dr_joint_lo_1 = HOT(dr_joint_lo[1]);
        const auto h_dr_joint_lo_2 = HOT(dr_joint_lo[2]);
        const auto h_dr_joint_on = HOT(dr_joint_on);
        const auto h_dr_joint_span_0 = HOT(dr_joint_span[0]);
        const auto h_dr_joint_span_1 = HOT(dr_joint_span[1]);
        const auto h_dr_joint_span_2 = HOT(dr_joint_span[2]);
        const auto h_dr_kd_lo = HOT(dr_kd_lo);
        const auto h_dr_kd_span = HOT(dr_kd_span);
        const auto h_dr_kp_lo = HOT(dr_kp_lo);
        const auto h_dr_kp_span = HOT(dr_kp_span);
        const auto h_dr_mass_lo = HOT(dr_mass_lo);
        const auto h_dr_mass_on = HOT(dr_mass_on);
        const auto h_dr_mass_span = HOT(dr_mass_span);
        const auto h_dr_pd_on = HOT(dr_pd_on);
        const auto h_env_id_offset = HOT(env_id_offset);
        const auto h_episode_length_s = HOT(episode_length_s);
        const auto h_fail_threshold = HOT(fail_threshold);
        const auto h_feet_air_time_threshold = HOT(feet_air_time_threshold);
        const auto h_foot_clearance_ref = HOT(foot_clearance_ref);
        const auto h_foot_clearance_sigma = HOT(foot_clearance_sigma);
        const auto h_foot_clearance_target = HOT(foot_clearance_target);
        const auto h_foot_distance_threshold = HOT(foot_distance_threshold);
        const auto h_foot_height_offset = HOT(foot_height_offset);
        const auto h_friction_offset = HOT(friction_offset);
        const auto h_gait_period_fixed = HOT(gait_period_fixed);
        const auto h_heading_command = HOT(heading_command);
        const auto h_heights_clip_scale = HOT(heights_clip_scale);
        const auto h_heights_offset = HOT(heights_offset);
        const auto h_kd_offset = HOT(kd_offset);
        const auto h_kp_offset = HOT(kp_offset);
        const auto h_max_episode_length = HOT(max_episode_length);
        const auto h_max_projected_gravity = HOT(max_projected_gravity);
        const auto h_max_push_vel_xy = HOT(max_push_vel_xy);
        const auto h_max_terrain_level = HOT(max_terrain_level);
        const auto h_noise_act0 = HOT(noise_act0);
        const auto h_noise_lead_0 = HOT(noise_lead[0]);
        const auto h_noise_lead_1 = HOT(noise_lead[1]);
        const auto h_noise_lead_2 = HOT(noise_lead[2]);
        const auto h_noise_lead_3 = HOT(noise_lead[3]);
        const auto h_noise_lead_4 = HOT(noise_lead[4]);
        const auto h_noise_lead_5 = HOT(noise_lead[5]);
        const auto h_num_labels = HOT(num_labels);
        const auto h_num_obs = HOT(num_obs);
        const auto h_num_priv_obs = HOT(num_priv_obs);
        const auto h_o_base_init_pos_0 = HOT(o_base_init_pos[0]);
        const auto h_o_base_init_pos_1 = HOT(o_base_init_pos[1]);
        const auto h_o_base_init_pos_2 = HOT(o_base_init_pos[2]);
        const auto h_obs_frame = HOT(obs_frame);
        const auto h_obs_scale_ang_vel = HOT(obs_scale_ang_vel);
        const auto h_obs_scale_dof_pos = HOT(obs_scale_dof_pos);
        const auto h_obs_scale_dof_vel = HOT(obs_scale_dof_vel);
        const auto h_obs_scale_height = HOT(obs_scale_height);
        const auto h_obs_scale_lin_vel = HOT(obs_scale_lin_vel);
        const auto h_obs_sets = HOT(obs_sets);
        const auto h_obs_slack = HOT(obs_slack);
        const auto h_obs_stack = HOT(obs_stack);
        const auto h_only_positive_rewards = HOT(only_positive_rewards);
        const auto h_priv_frame = HOT(priv_frame);
        const auto h_priv_stack = HOT(priv_stack);
        const auto h_push_interval = HOT(push_interval);
        const auto h_resample_steps = HOT(resample_steps);
        const auto h_reset_ang_vel_lo = HOT(reset_ang_vel_lo);
        const auto h_reset_ang_vel_span = HOT(reset_ang_vel_span);
        const auto h_reset_lin_vel_lo = HOT(reset_lin_vel_lo);
        const auto h_reset_lin_vel_span = HOT(reset_lin_vel_span);
        const auto h_reset_root_xy_lo = HOT(reset_root_xy_lo);
        const auto h_reset_root_xy_span = HOT(reset_root_xy_span);
        const auto h_seed = HOT(seed);
        const auto h_sit_percent = HOT(sit_percent);
        const auto h_slots_cb_cmd = HOT(slots.cb_cmd);
        const auto h_slots_dr_kd = HOT(slots.dr_kd);
        const auto h_slots_dr_kp = HOT(slots.dr_kp);
        const auto h_slots_push = HOT(slots.push);
        const auto h_slots_reset_dof = HOT(slots.reset_dof);
        const auto h_slots_reset_root_xy = HOT(slots.reset_root_xy);
        const auto h_slots_task_reset = HOT(slots.task_reset);
        const auto h_terrain_cols_n = HOT(terrain_cols_n);
        const auto h_terrain_curriculum = HOT(terrain_curriculum);
        const auto h_terrain_env_length = HOT(terrain_env_length);
        const auto h_tracking_sigma = HOT(tracking_sigma);
        const auto h_yaw_clip_0 = HOT(yaw_clip[0]);
        const auto h_yaw_clip_1 = HOT(yaw_clip[1]);
        asm volatile("" ::: "memory");
        const float cdt = h_control_dt;
        const unsigned rmask = RS ? RS_MASK : (unsigned)p.k.reward_mask;
        const bool heading = h_heading_command != 0;
        unsigned k0, k1, e_lo, e_hi;
        {
            const unsigned long long seed = h_seed, gid = (unsigned long long)(h_env_id_offset + e);
            k0 = (unsigned)(seed & 0xFFFFFFFFu); k1 = (unsigned)(seed >> 32);
            e_lo = (unsigned)(gid & 0xFFFFFFFFu); e_hi = (unsigned)(gid >> 32);
        }
        const unsigned rstep = (unsigned)p.counter;
        const float *const rin = INJ ? B.rand_in + (size_t)e * HOT(slots.n_slots) : nullptr;   // this env's injected uniforms (LgRandSlots)
        auto philox = [&](unsigned c3) { const U4 c = {e_lo, e_hi, rstep, c3}; return philox4x32_10(c, k0, k1); };
        auto philox_e = [&](unsigned elo, unsigned ehi, unsigned c3) { const U4 c = {elo, ehi, rstep, c3}; return philox4x32_10(c, k0, k1); };
        auto pick = [](const U4 &r, int k) { return k == 0 ? r.x : (k == 1 ? r.y : (k == 2 ? r.z : r.w)); };
        auto anyl = [](bool b) { return __builtin_amdgcn_ballot_w64(b) != 0ull; };
        // lane l8 (0..7) of this env, to every lane of the env
        auto fetch8 = [&](float v, int l8) { return __int_as_float(__builtin_amdgcn_ds_bpermute((int)((tl_ & 56u) | (unsigned)l8) << 2, __float_as_int(v))); };
        auto jsum = [&](float v) { return bc<0>(legsum<LEGS>(sum3(v))); };       // over the env's joints (value in joint lanes), to all lanes
        auto vnorm2 = [&](float v) { return bc<0>(sum3(v * v)); };               // |v|^2 of a component-layout vector, to the quad
        const int el = ei;                                                       // lane of the env: 4 leg + c
        const bool A_ = role == 0, B_ = role == 1;                               // this wave's share of the tail (wave-uniform)
        const bool liveB = alive && B_, stB = liveB && !L.is3, sv = alive && !L.is3;

        float cmdv = m_cmd, air = m_air;
        float es[4] = {b_es[0], b_es[1], b_es[2], b_es[3]};
        float o_fric = dr_fric, o_mass = dr_mass, o_com = dr_com, o_kp = dr_kp, o_kd = dr_kd, o_push = w_push;   // what the critic frame shows
        float o_jnt = L.sel(dr_arm, dr_jf, dr_jd);                               // per-env joint DR (armature, frictionloss, damping), component c
        int ep_len = m_ep + 1, failb = m_fail, last_contact = m_lc;              // legged_robot.py:60
        float gait_time = w_gt, phi = w_phi, theta = w_th, expC = w_ec;
        const float gait_period = h_gait_period_fixed;
        const float CRlo = L.is0 ? CR(0) : (L.is1 ? CR(2) : (L.is2 ? CR(4) : CR(6)));
        const float CRhi = L.is0 ? CR(1) : (L.is1 ? CR(3) : (L.is2 ? CR(5) : CR(7)));
        auto resample = [&](float cv, float u0, float u1, float u2) {            // legged_robot.py:317-334
            const float u = L.is0 ? u0 : (L.is1 ? u1 : u2);
            const bool upd = heading ? !L.is2 : !L.is3;                          // heading mode draws the heading, not the yaw rate
            cv = upd ? (CRhi - CRlo) * u + CRlo : cv;
            const float keep = sqrtf(bc<0>(sum3(cv * cv))) > 0.2f ? 1.f : 0.f;
            return L.is3 ? cv : cv * keep;
        };
        // ---- _post_physics_step_callback (legged_robot.py:300-315) ----
        {
            const bool need = (ep_len % h_resample_steps) == 0;
            if (anyl(need)) {
                float u0, u1, u2;
                if constexpr (INJ) { u0 = rin[h_slots_cb_cmd]; u1 = rin[h_slots_cb_cmd + 1]; u2 = rin[h_slots_cb_cmd + 2]; }
                else { const U4 r = philox(0x40000000u + (unsigned)h_slots_cb_cmd); u0 = u01(r.x); u1 = u01(r.y); u2 = u01(r.z); }
                const float nc = resample(cmdv, u0, u1, u2);
                cmdv = need ? nc : cmdv;
            }
        }
        if (heading) {   // forward = quat_apply(base_quat, [1,0,0]) (math_utils.py:34-40): t = 2 xyz x b = (0, 2 qz, -2 qy)
            const float ty = 2.f * qz, tz = -2.f * qy;
            const float fx = 1.f + (qy * tz - qz * ty), fy = ty * qw + (0.f - qx * tz);
            const float hd = atan2f(fy, fx);
            const float c2 = clampf(0.5f * wrap_to_pi(bc<3>(cmdv) - hd), h_yaw_clip_0, h_yaw_clip_1);
            cmdv = L.is2 ? c2 : cmdv;
        }
        {
            const int pi_ = h_push_interval;
            if (pi_ > 0 && (p.counter % pi_) == 0) {   // genesis_simulator.py:150-158; lanes 0 / 1 evaluate the two draws' blocks side by side
                const int slot = h_slots_push + (L.is1 ? 1 : 0);
                float up;
                if constexpr (INJ) up = rin[slot];
                else { const U4 r = philox((unsigned)(slot >> 2)); up = u01(pick(r, slot & 3)); }
                const float m = h_max_push_vel_xy;
                const float pv = (m + m) * up - m;
                const bool xy = L.c < 2;
                vw = xy ? vw + pv : vw;
                o_push = xy ? pv : o_push;
                if (liveB && leg == 0 && xy) { B.rand_push_vels[3 * e + L.c] = pv; B.base_lin_vel_w[3 * e + L.c] = vw; }   // role 1: in program order with its reset stores
            }
        }
        const float cmd0 = bc<0>(cmdv), cmd1 = bc<1>(cmdv), cmd2 = bc<2>(cmdv);
        STAMP(6);
        // ---- check_termination (legged_robot.py:78-92) ----
        const unsigned tmask = m_tmask, pmask = m_pmask;
        const int l0 = foot_link - 3;
        float n2[4];
#pragma unroll
        for (int k = 0; k < 4; k++) n2[k] = vnorm2(f_link[k]);
        const float nb2 = vnorm2(f_base);
        const float pgz = bc<2>(pg);
        int fail = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) fail |= (((tmask >> (l0 + k)) & 1u) && n2[k] > 100.0f) ? 1 : 0;
        fail |= xor4i(fail);                                                     // the other leg (lane ^ 4)
        fail |= ((tmask & 1u) && nb2 > 100.0f) ? 1 : 0;
        fail |= pgz > h_max_projected_gravity ? 1 : 0;
        if (guard_bad) failb = LG_FAIL_NONFINITE;   // a re-seated env ends its episode here (lgsim.h)
        failb += fail;
        const bool time_out = (float)ep_len > h_max_episode_length;
        const bool reset = ((float)failb > h_fail_threshold) || time_out;
        // terrain curriculum of a resetting env (legged_robot.py:254-272): the level it moves to is known here, except when it has solved
        // the last one (a random level, drawn below).  The origin of that level is loaded NOW, so that the round trip passes under the
        // reward terms instead of sitting in the reset block of the wave that ends the launch.
        const bool curr = h_terrain_curriculum && p.counter > 0;
        int lvl_pre = w_lvl;
        float norg_pre = origin;
        if (curr && anyl(reset)) {
            const float dd = pos - origin;
            const float dx = bc<0>(dd), dy = bc<1>(dd);
            const float dist = sqrtf(dx * dx + dy * dy);
            const bool up = dist > h_terrain_env_length / 2.f;
            const bool down = (dist < sqrtf(cmd0 * cmd0 + cmd1 * cmd1) * h_episode_length_s * 0.5f) && !up;
            lvl_pre = w_lvl + (up ? 1 : 0) - (down ? 1 : 0);
            norg_pre = B.terrain_origins[((size_t)min(max(lvl_pre, 0), h_max_terrain_level - 1) * h_terrain_cols_n + w_type) * 3 + cj];
        }

        // ---- compute_reward (legged_robot.py:150-168): every term replicated over the env's lanes, summed in alphabetical order ----
        float scl[LG_R_COUNT];
#pragma unroll
        for (int k = 0; k < LG_R_COUNT; k++) scl[k] = HOT(reward_scales[k]);
        float total = 0.f;
        auto add = [&](int id, float r) {            // `id` is a compile-time constant at every call
            const float rew = r * scl[id];
            total += rew;
            es[id >> 3] = el == (id & 7) ? es[id >> 3] + rew : es[id >> 3];
        };
        const float cmd_xy = sqrtf(cmd0 * cmd0 + cmd1 * cmd1);
        const float cmd_xyz = sqrtf(cmd0 * cmd0 + cmd1 * cmd1 + cmd2 * cmd2);
        const float dq0 = q - q0;
        const float fz = bc<2>(f_link[3]);                          // vertical foot force of this leg
        const float fpx = bc<0>(foot_p), fpy = bc<1>(foot_p), fpz = bc<2>(foot_p), fvx = bc<0>(foot_v), fvy = bc<1>(foot_v), fvz = bc<2>(foot_v);
        // mean over the terrain samples of (base z - height) (legged_robot.py:470-476, tron1_pf_ee.py:435-440): every lane adds its own samples
        float mean_height = bc<2>(pos);
        if (P > 0) {
            const float pzv = bc<2>(pos);
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < HQ; i++) acc += (hk0 + i * kstride < P) ? pzv - hq[i] : 0.f;
            mean_height = legsum<LEGS>(sum4(acc)) / (float)P;
        }
        if (B_ && RON(LG_R_BIPED_PERIODIC_GAIT)) {   // role 1 writes the critic frame's gait block: the force indicator of the reward term below
            const float two_pi = 6.283185307179586f;
            float ph = phi + theta;
            ph = (ph - floorf(ph)) * two_pi;
            expC = (ph >= 0.f && ph < h_b_swing * two_pi) ? -1.f : 0.f;
        }
        if (A_) {
        if (RON(LG_R_ACTION_RATE)) { const float d = last_act - act; add(LG_R_ACTION_RATE, jsum(d * d)); }                     // :495-497
        if (RON(LG_R_ACTION_SMOOTHNESS)) { const float d = act - 2.f * last_act + llast_act; add(LG_R_ACTION_SMOOTHNESS, jsum(d * d)); }   // :499-503
        if (RON(LG_R_ANG_VEL_XY)) { const float bx = bc<0>(bav), by = bc<1>(bav); add(LG_R_ANG_VEL_XY, bx * bx + by * by); }    // :462-464
        if (RON(LG_R_BASE_HEIGHT)) { const float d = mean_height - h_base_height_target; add(LG_R_BASE_HEIGHT, d * d); }        // :470-476
        if (RON(LG_R_BIPED_PERIODIC_GAIT)) {                                                                                     // tron1_pf_ee.py:347-433 ("step" indicator)
            const float two_pi = 6.283185307179586f;
            float ph = phi + theta;
            ph = (ph - floorf(ph)) * two_pi;
            const float b_sw = h_b_swing * two_pi;
            const float c_frc = (ph >= 0.f && ph < b_sw) ? -1.f : 0.f;
            const float c_spd = (ph >= b_sw && ph < two_pi) ? -1.f : 0.f;
            expC = c_frc;
            add(LG_R_BIPED_PERIODIC_GAIT, __expf(legsum<LEGS>(c_spd * sqrtf(vnorm2(foot_v)) + c_frc * sqrtf(n2[3]))));
        }
        if (RON(LG_R_COLLISION)) {                                                                                               // :505-512
            float sc_ = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) sc_ += (((pmask >> (l0 + k)) & 1u) && n2[k] > 0.1f * 0.1f) ? 1.f : 0.f;
            sc_ = legsum<LEGS>(sc_);
            sc_ += ((pmask & 1u) && nb2 > 0.1f * 0.1f) ? 1.f : 0.f;
            add(LG_R_COLLISION, sc_);
        }
        if (RON(LG_R_DOF_ACC)) { const float d = (qd_start - qd) / cdt; add(LG_R_DOF_ACC, jsum(d * d)); }                       // :490-493
        if (RON(LG_R_DOF_CLOSE_TO_DEFAULT)) add(LG_R_DOF_CLOSE_TO_DEFAULT, jsum(dq0 * dq0));                                     // :571-573
        if (RON(LG_R_DOF_POS_LIMITS)) add(LG_R_DOF_POS_LIMITS, jsum(-fminf(q - m_slo, 0.f) + fmaxf(q - m_shi, 0.f)));            // :518-522
        if (RON(LG_R_DOF_POS_STAND_STILL)) add(LG_R_DOF_POS_STAND_STILL, jsum(dq0 * dq0) * (cmd_xyz < 0.1f ? 1.f : 0.f));        // :561-563
        if (RON(LG_R_DOF_POWER)) add(LG_R_DOF_POWER, jsum(fabsf(torque * qd)));                                                  // :486-488
        if (RON(LG_R_DOF_VEL)) add(LG_R_DOF_VEL, jsum(qd * qd));                                                                 // :482-484
        if (RON(LG_R_DOF_VEL_STAND_STILL)) add(LG_R_DOF_VEL_STAND_STILL, jsum(fabsf(qd)) * (cmd_xyz < 0.1f ? 1.f : 0.f));        // :557-559
        if (RON(LG_R_FEET_AIR_TIME)) {                                                                                           // :545-555 (stateful)
            const int contact = fz > 1.0f ? 1 : 0;
            const int filt = contact | last_contact;
            last_contact = contact;
            const float first = (air > 0.f ? 1.f : 0.f) * (float)filt;
            air += cdt;
            float r = legsum<LEGS>((air - h_feet_air_time_threshold) * first);
            r *= (h_air_time_cmd_dims == 3 ? cmd_xyz : cmd_xy) > 0.1f ? 1.f : 0.f;
            air *= filt ? 0.f : 1.f;
            add(LG_R_FEET_AIR_TIME, r);
        }
        if (RON(LG_R_FEET_CONTACT_STAND_STILL)) {                                                                                // :565-569
            const float cnt = legsum<LEGS>(fz > 0.1f ? 1.f : 0.f);
            add(LG_R_FEET_CONTACT_STAND_STILL, (cnt == (float)LEGS ? 1.f : 0.f) * (cmd_xyz < 0.1f ? 1.f : 0.f));
        }
        if (RON(LG_R_FEET_DISTANCE)) {                                                                                           // tron1_pf_ee.py:458-463
            const float ox = xor4(fpx), oy = xor4(fpy);
            const float dxy = sqrtf((fpx - ox) * (fpx - ox) + (fpy - oy) * (fpy - oy));
            add(LG_R_FEET_DISTANCE, fmaxf(0.f, h_foot_distance_threshold - dxy));
        }
        if (RON(LG_R_FOOT_ACC)) { const float a = (foot_v - last_foot_v) * (1.f / cdt); add(LG_R_FOOT_ACC, legsum<LEGS>(vnorm2(a))); }    // :605-608
        if (RON(LG_R_FOOT_CLEARANCE)) {                                                                                          // :575-588, tron1_pf_ee.py:442-456
            const float vxy = sqrtf(fvx * fvx + fvy * fvy);
            const float d = fpz - (h_foot_clearance_ref == 1 ? f_hmean : (h_foot_clearance_ref == 2 ? f_hmax : 0.f)) - h_foot_clearance_target - h_foot_height_offset;
            add(LG_R_FOOT_CLEARANCE, __expf(-legsum<LEGS>(vxy * (d * d)) / h_foot_clearance_sigma));
        }
        if (RON(LG_R_FOOT_LANDING_VEL)) {                                                                                        // :590-599
            const bool land = ((fpz - h_foot_height_offset) < h_about_landing_threshold) && !(fz > 0.1f) && (fvz < 0.f);
            const float vz = land ? fvz : 0.f;
            add(LG_R_FOOT_LANDING_VEL, legsum<LEGS>(vz * vz));
        }
        if (RON(LG_R_HIP_POS)) { const float h = bc<0>(dq0); add(LG_R_HIP_POS, legsum<LEGS>(h * h)); }                           // go2_ee.py:152-159
        if (RON(LG_R_KEEP_BALANCE)) add(LG_R_KEEP_BALANCE, 1.f);                                                                 // :601-603
        if (RON(LG_R_LIN_VEL_Z)) { const float z = bc<2>(blv); add(LG_R_LIN_VEL_Z, z * z); }                                     // :458-460
        if (RON(LG_R_NO_FLY)) add(LG_R_NO_FLY, legsum<LEGS>(fz > kT->no_fly_contact_threshold ? 1.f : 0.f) == 1.f ? 1.f : 0.f);  // tron1_pf.py:151-154
        if (RON(LG_R_ORIENTATION)) { const float x = bc<0>(pg), y = bc<1>(pg); add(LG_R_ORIENTATION, x * x + y * y); }           // :466-468
        if (RON(LG_R_TORQUES)) add(LG_R_TORQUES, jsum(torque * torque));                                                         // :478-480
        if (RON(LG_R_TRACKING_ANG_VEL)) { const float d = cmd2 - bc<2>(bav); add(LG_R_TRACKING_ANG_VEL, __expf(-(d * d) / h_tracking_sigma)); }   // :539-543
        if (RON(LG_R_TRACKING_BASE_HEIGHT)) { const float d = mean_height - h_base_height_target; add(LG_R_TRACKING_BASE_HEIGHT, __expf(-(d * d) / h_base_height_sigma)); }   // tron1_pf_ee.py:435-440
        if (RON(LG_R_TRACKING_LIN_VEL)) {                                                                                        // :533-537
            const float dx = cmd0 - bc<0>(blv), dy = cmd1 - bc<1>(blv);
            add(LG_R_TRACKING_LIN_VEL, __expf(-(dx * dx + dy * dy) / h_tracking_sigma));
        }
        if (h_only_positive_rewards) total = fmaxf(total, 0.f);                                                               // :161-162
        if (RON(LG_R_TERMINATION)) add(LG_R_TERMINATION, (reset && !time_out) ? 1.f : 0.f);                                      // :163-168
        }
        {   // gait clock (tron1_pf_ee.py:28-35)
            gait_time += cdt;
            if (gait_time >= gait_period - cdt / 2.f) gait_time = 0.f;
            phi = gait_time / gait_period;
        }

        STAMP(7);
        // ---- Philox calls A and B (every step; see the head of this block) ----
        U4 rA = {0u, 0u, 0u, 0u}, rB = rA;
        const bool coin_lane = el == 7;
        if constexpr (!INJ) {
            if (A_) rA = philox(0x80000000u + (L.is0 ? (unsigned)(2 * leg) : (L.is1 ? (unsigned)(2 * leg) + 1u     // observation noise: the actor frames are role 0's
                                               : (L.is2 ? (unsigned)(2 * LEGS + 2 + leg) : (unsigned)(3 * LEGS + 2 + leg)))));
            rB = philox_e(coin_lane ? 0xFFFFFFFFu : e_lo, coin_lane ? 0xFFFFFFFFu : e_hi,
                          el == 0 ? 0x80000000u + (unsigned)(2 * LEGS) : (coin_lane ? (unsigned)(h_slots_task_reset >> 2) : 0x80000000u + 0x200u + (unsigned)(el - 1)));
        }
        const float ax = u01(rA.x), ay = u01(rA.y), az = u01(rA.z), aw = u01(rA.w);
        const float bx = u01(rB.x), by = u01(rB.y), bz = u01(rB.z), bw = u01(rB.w);
        // ---- reset_idx (legged_robot.py:94-148, tron1_pf_ee.py:193-256, 277-310) + simulator.reset_idx (genesis_simulator.py:62-82) ----
        if (anyl(reset)) {
            // element c of the bundle block held by lane `src` of the env (uniform i of the bundle = block i / 4, word i % 4; slots of
            // env_step_body's eu[]: 0-2 commands, 3 friction, 4-6 CoM, 7 mass, 8-10 root lin vel, 12-14 root ang vel, 16-18 joint
            // armature / friction / damping, 19-20 gait phase offsets, 21 terrain level)
            auto bundle_c = [&](int src) { return L.sel4(fetch8(bx, src), fetch8(by, src), fetch8(bz, src), fetch8(bw, src)); };
            float v0 = bundle_c(1), v1 = bundle_c(2), v2 = bundle_c(3), v3_ = bundle_c(4), v4 = bundle_c(5);
            float u_gt = fetch8(bx, 6), u_tl = fetch8(by, 6);                                     // uniforms 20, 21
            float u_coin = fetch8(u01(pick(rB, h_slots_task_reset & 3)), 7);                      // tron1_pf_ee.py:204-210: one coin for the whole job
            if constexpr (INJ) {   // the same quantities from their slots of the injected row (env_step_body's rs.in paths)
                v0 = L.is3 ? rin[HOT(slots.dr_friction)] : rin[HOT(slots.reset_cmd) + cj];
                v1 = L.is3 ? rin[HOT(slots.dr_mass)] : rin[HOT(slots.dr_com) + cj];
                v2 = rin[HOT(slots.reset_lin_vel) + cj]; v3_ = rin[HOT(slots.reset_ang_vel) + cj];
                v4 = L.is3 ? rin[h_slots_task_reset + 1] : rin[HOT(slots.dr_joint) + cj];
                u_gt = rin[h_slots_task_reset + 2]; u_tl = rin[HOT(slots.terrain_level)]; u_coin = rin[h_slots_task_reset];
            }
            // call C: lane 0 / 1 / 2 of the quad: the leg's `_reset_dofs` / kp / kd blocks, lane 3 of leg 0: the root xy block
            const int sxy = h_slots_reset_root_xy;
            U4 rC = {0u, 0u, 0u, 0u};
            if constexpr (!INJ) rC = philox(L.is0 ? 0x40000000u + (unsigned)(h_slots_reset_dof + d0) : (L.is1 ? 0x40000000u + (unsigned)(h_slots_dr_kp + d0)
                                            : (L.is2 ? 0x40000000u + (unsigned)(h_slots_dr_kd + d0) : (unsigned)(sxy >> 2))));
            const float cx_ = u01(rC.x), cy_ = u01(rC.y), cz_ = u01(rC.z);
            const float ud_p = L.sel(bc<0>(cx_), bc<0>(cy_), bc<0>(cz_)), kp_p = L.sel(bc<1>(cx_), bc<1>(cy_), bc<1>(cz_)), kd_p = L.sel(bc<2>(cx_), bc<2>(cy_), bc<2>(cz_));
            const float ud = INJ ? rin[h_slots_reset_dof + d0 + cj] : ud_p;                       // element c of lane 0's block
            const float nkp = h_dr_kp_span * (INJ ? rin[h_slots_dr_kp + d0 + cj] : kp_p) + h_dr_kp_lo;
            const float nkd = h_dr_kd_span * (INJ ? rin[h_slots_dr_kd + d0 + cj] : kd_p) + h_dr_kd_lo;
            const float pxy0 = u01(pick(rC, sxy & 3)), pxy1 = u01(pick(rC, (sxy + 1) & 3));      // valid in lane 3 of leg 0
            const float u_x = fetch8(pxy0, 3), u_y = fetch8(pxy1, 3);
            const float u_xy = INJ ? rin[sxy + (L.c & 1)] : (L.is0 ? u_x : u_y);
            const bool sit = h_sit_percent > 0.f && u_coin < h_sit_percent;
            // terrain curriculum (legged_robot.py:254-272 + genesis_simulator.py:140-148; skipped on the construction-time reset)
            float norg = origin;
            int nlvl = w_lvl;
            if (curr) {
                int lvl = lvl_pre;
                norg = norg_pre;
                if (lvl >= h_max_terrain_level) {      // solved the last level: a random one (rare), its origin loaded here
                    lvl = min((int)floorf(u_tl * (float)h_max_terrain_level), h_max_terrain_level - 1);
                    norg = B.terrain_origins[((size_t)lvl * h_terrain_cols_n + w_type) * 3 + cj];
                } else lvl = max(lvl, 0);
                nlvl = lvl;
            }
            const float ncmd = resample(cmdv, bc<0>(v0), bc<1>(v0), bc<2>(v0));
            float ipos = (sit ? b_sitp : L.sel(h_o_base_init_pos_0, h_o_base_init_pos_1, h_o_base_init_pos_2)) + norg;
            if (h_custom_origins && L.c < 2) ipos += h_reset_root_xy_span * u_xy + h_reset_root_xy_lo;          // legged_robot.py:288
            const float iq = sit ? b_sitr : (L.is3 ? h_base_init_quat_3 : L.sel(h_base_init_quat_0, h_base_init_quat_1, h_base_init_quat_2));
            const float nvw = sit ? 0.f : h_reset_lin_vel_span * v2 + h_reset_lin_vel_lo;                        // legged_robot.py:289-292; tron1_pf_ee.py:304-309
            const float nww = sit ? 0.f : h_reset_ang_vel_span * v3_ + h_reset_ang_vel_lo;
            // quad broadcasts stay outside the divergent branch
            const float nfric = h_dr_friction_span * bc<3>(v0) + h_dr_friction_lo, nmass = h_dr_mass_span * bc<3>(v1) + h_dr_mass_lo;
            const float ncom = L.sel(h_dr_com_span_0, h_dr_com_span_1, h_dr_com_span_2) * v1 + L.sel(h_dr_com_lo_0, h_dr_com_lo_1, h_dr_com_lo_2);
            const float njnt = L.sel(h_dr_joint_span_0, h_dr_joint_span_1, h_dr_joint_span_2) * v4 + L.sel(h_dr_joint_lo_0, h_dr_joint_lo_1, h_dr_joint_lo_2);
            const float th0 = b_th0 + bc<3>(v4);                                                                  // tron1_pf_ee.py:220-226
            const float ntheta = foot_slot == 0 ? th0 : th0 + (b_th1 - b_th0);
            if (reset) {
                theta = ntheta; gait_time = u_gt * gait_period; phi = gait_time / gait_period;
                if (h_dr_pd_on) { o_kp = nkp; o_kd = nkd; }
                if (h_dr_friction_on) o_fric = nfric;
                if (h_dr_mass_on) o_mass = nmass;
                if (h_dr_com_on) o_com = ncom;
                if (h_dr_joint_on && B.joint_armature) o_jnt = njnt;
                cmdv = ncmd;
                q = sit ? b_sitq : q0 + (m_rsp * ud + m_rlo); qd = 0.f;
                act = 0.f; last_act = 0.f; llast_act = 0.f;
                pos = ipos; quat = iq; vw = nvw; ww = nww;
                air = 0.f; ep_len = 0; failb = 0;
            }
            // the reference stores the commanded reset twist verbatim in the body-frame properties (genesis_simulator.py:128-129)
            // and refreshes projected gravity (:125)
            const QM Rr = quat_rows(L, quat);
            const float npg = -L.sel(bc<2>(Rr.c0), bc<2>(Rr.c1), bc<2>(Rr.c2));
            if (reset) { blv = vw; bav = ww; pg = npg; }
            if (reset && stB) {
                B.dof_pos[ja] = q; B.dof_vel[ja] = 0.f; B.last_dof_vel[ja] = 0.f;
                B.actions[ja] = 0.f; B.last_actions[ja] = 0.f; B.llast_actions[ja] = 0.f;
                if (h_dr_pd_on) { B.kp_scale[ja] = o_kp; B.kd_scale[ja] = o_kd; }
                B.last_feet_vel[(e * F + foot_slot) * 3 + cj] = 0.f;
                if (leg == 0) {
                    if (curr) { B.env_origins[3 * e + cj] = norg; if (L.is0) B.terrain_levels[e] = nlvl; }
                    B.base_pos[3 * e + cj] = pos; B.base_lin_vel_w[3 * e + cj] = vw; B.base_ang_vel_w[3 * e + cj] = ww;
                    B.base_lin_vel[3 * e + cj] = blv; B.base_ang_vel[3 * e + cj] = bav; B.projected_gravity[3 * e + cj] = pg;
                    B.last_base_lin_vel[3 * e + cj] = 0.f; B.last_base_ang_vel[3 * e + cj] = 0.f;
                    if (h_dr_com_on) B.base_com_bias[3 * e + cj] = o_com;
                }
                if (leg == 1 && h_dr_joint_on && B.joint_armature) {   // genesis_simulator.py:704-733: one value per env each
                    float *jp = L.is0 ? B.joint_armature : (L.is1 ? B.joint_friction : B.joint_damping);
                    jp[e] = o_jnt;
                }
            }
            if (reset && liveB && leg == 0) {
                B.base_quat[4 * e + L.c] = quat;
                if (L.is3) {
                    if (h_dr_friction_on) B.friction_values[e] = o_fric;
                    if (h_dr_mass_on) B.added_base_mass[e] = o_mass;
                    B.episode_done_step[e] = (int)p.counter;
                }
            }
            if (reset && live) {    // extras["episode"] snapshot (legged_robot.py:128-132), then the sums restart
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (el + 8 * k < LG_R_COUNT && ((rmask >> (el + 8 * k)) & 1u)) { B.episode_done_sums[(size_t)(el + 8 * k) * N + e] = es[k]; es[k] = 0.f; }
            }
        }
        STAMP(9);
        {
            // ---- compute_observations + clip, tron1_pf_ee.py:53-141.  actor frame: 9 + 3 A + clock 2 F, stacked 10 deep; critic frame: the
            //      frame without noise | DR (7 + 2 A + 3) | gait F | contact states K | heights P | normals 3 F | clip(foot_z - h9) 9 F,
            //      stacked 10 deep; labels: v_b 3 | K | foot height above the local terrain max F | normals 3 F.  Same window / copy /
            //      blanking bookkeeping as the quadruped tails; the env's 8 lanes write consecutive columns.
            const float co = h_clip_obs;
            const bool nz = h_add_noise != 0;
            const int FR = h_obs_frame, PF = h_priv_frame, ST = h_obs_stack, PST = h_priv_stack, SL = h_obs_slack;
            const size_t orow = (size_t)(h_num_obs + SL * FR), prow = (size_t)(h_num_priv_obs + SL * PF);
            const bool two = h_obs_sets > 1;
            const int cs = two ? p.obs_set : 0, xs = (two && cs < 2) ? 1 - cs : cs;
            // Every destination = a wave-uniform base (this set's / the other set's allocation) + a 32-bit byte offset per lane: the store takes
            // the base from scalar registers (saddr form) and the offset costs one add, instead of a 64-bit pointer computation per store
            // (~5 VALU instructions each, ~70 stores per lane).  The allocations stay below 4 GiB (lg_host.hip biped_profile).
            float *const ob_c = B.obs_buf + (size_t)cs * N * orow, *const ob_x = B.obs_buf + (size_t)xs * N * orow;
            float *const pb_c = B.priv_obs_buf + (size_t)cs * N * prow, *const pb_x = B.priv_obs_buf + (size_t)xs * N * prow;
            float *const lb_c = B.labels_buf + (size_t)cs * N * h_num_labels;
            const unsigned io4 = ((unsigned)e * (unsigned)orow + (unsigned)(p.obs_win * FR + (ST - 1) * FR)) * 4u;     // newest frame of this launch's window
            const unsigned ip4 = ((unsigned)e * (unsigned)prow + (unsigned)(p.obs_win * PF + (PST - 1) * PF)) * 4u;
            const unsigned il4 = (unsigned)e * (unsigned)h_num_labels * 4u;
            auto SO = [](float *base, unsigned off, float v) { *reinterpret_cast<float *>(reinterpret_cast<char *>(base) + off) = v; };
            // role 0 blanks the actor rows, role 1 the (four times longer) critic rows
            if (anyl(reset)) {                         // legged_robot_ee.py: the histories of a reset env restart from zeros
                blank_histories(__builtin_amdgcn_ballot_w64(reset && alive && el == 0), e, B.obs_buf + (size_t)cs * N * orow + (size_t)p.obs_win * FR, orow,
                                A_ ? (ST - 1) * FR : 0, B.priv_obs_buf + (size_t)cs * N * prow + (size_t)p.obs_win * PF, prow, B_ ? (PST - 1) * PF : 0);
            }
            if (two && anyl(!reset && w_dirty != 0)) {
                blank_histories(__builtin_amdgcn_ballot_w64(!reset && w_dirty != 0 && alive && el == 0), e, B.obs_buf + (size_t)cs * N * orow + (size_t)p.obs_win * FR, orow,
                                A_ ? (ST - 2) * FR : 0, B.priv_obs_buf + (size_t)cs * N * prow + (size_t)p.obs_win * PF, prow, B_ ? (PST - 2) * PF : 0);
            }
            const bool w2o = two && ST > 1, w2p = two && PST > 1;
            float uq = 0.5f, uqd = 0.5f, uact = 0.5f, uclk = 0.5f, ug = 0.5f, ua = 0.5f;
            if (nz) {
                uq = L.sel(bc<0>(ax), bc<0>(ay), bc<0>(az));                 // block 2 leg in lane 0 of the quad: the three joints' uniforms
                uqd = L.sel(bc<1>(ax), bc<1>(ay), bc<1>(az));                // block 2 leg + 1 in lane 1
                // base uniforms (env_step_body's ub[]): 0 / 1 = fourth words of leg 0's two blocks, 2-5 = block 2 LEGS (call B, lane 0)
                ug = L.sel(fetch8(aw, 0), fetch8(aw, 1), fetch8(bx, 0));
                ua = L.sel(fetch8(by, 0), fetch8(bz, 0), fetch8(bw, 0));
                if (h_noise_act0 != 0.f) {                                   // tron1_pf_ee.py:338-342 (quirk 4): actions and clock are noisy too
                    uact = L.sel(bc<2>(ax), bc<2>(ay), bc<2>(az));           // block 2 LEGS + 2 + leg in lane 2
                    const float ck0 = bc<3>(ax), ck1 = bc<3>(ay);            // block 3 LEGS + 2 + leg in lane 3: sin / cos entries
                    uclk = L.is0 ? ck0 : ck1;
                }
                if constexpr (INJ) {   // the frame's entries of the injected row (slots.noise + index in the frame)
                    const int ns_ = HOT(slots.noise);
                    uq = rin[ns_ + 9 + d0 + cj]; uqd = rin[ns_ + 9 + A + d0 + cj]; ug = rin[ns_ + 3 + cj]; ua = rin[ns_ + 6 + cj];
                    if (h_noise_act0 != 0.f) { uact = rin[ns_ + 9 + 2 * A + d0 + cj]; uclk = rin[ns_ + 9 + 3 * A + (L.is0 ? 0 : F) + foot_slot]; }
                }
            }
            auto W = [&](int idx, float v, float u, float ns) {            // an actor-frame entry: noisy into the actor windows, noise-free into the critic frame
                const float cl = clampf(v, -co, co);
                const float nv = clampf(nz ? v + (2.f * u - 1.f) * ns : v, -co, co);
                const unsigned o4 = io4 + 4u * (unsigned)idx, p4 = ip4 + 4u * (unsigned)idx;
                if (A_) { SO(ob_c, o4, nv); if (w2o) SO(ob_x, o4, nv); }
                else { SO(pb_c, p4, cl); if (w2p) SO(pb_x, p4, cl); }
            };
            auto WP = [&](int idx, float v) { const float cl = clampf(v, -co, co); const unsigned p4 = ip4 + 4u * (unsigned)idx; SO(pb_c, p4, cl); if (w2p) SO(pb_x, p4, cl); };
            auto WL = [&](int idx, float v) { SO(lb_c, il4 + 4u * (unsigned)idx, v); };
            const float ang = 6.283185307179586f * (phi + theta);            // clock inputs (tron1_pf_ee.py:186-191)
            const float sn = sinf(ang), csn = cosf(ang);
            // quad broadcasts outside the divergent branches
            const float env4 = L.sel4(o_fric - h_friction_offset, o_mass, bc<0>(o_push), bc<1>(o_push));
            const float pzn = bc<2>(pos);
            const unsigned smask = m_smask;
            const int K = __popc(smask);
            const float csv = (L.sel4(n2[0], n2[1], n2[2], n2[3]) > 1.f) ? 1.f : 0.f;     // contact state of link l0 + c (physics read-back, stale after a reset as in the reference)
            const int cl_ = l0 + L.c;
            const bool chas = ((smask >> cl_) & 1u) != 0;
            const int cidx = __popc(smask & ((1u << cl_) - 1u));
            const int oDR = FR, oG = FR + 7 + 2 * A + 3, oK = oG + F, oH = oK + K, oN = oH + P, oR = oN + 3 * F;
            const float clr = clampf(fpz - f_hmax - h_foot_height_offset, -1.f, 1.f);
            const float nrm = L.sel(f_n3[0], f_n3[1], f_n3[2]);
            if (sv) {
                W(9 + d0 + cj, (q - q0) * h_obs_scale_dof_pos, uq, m_nq);
                W(9 + A + d0 + cj, qd * h_obs_scale_dof_vel, uqd, m_nqd);
                W(9 + 2 * A + d0 + cj, act, uact, b_nact);
                if (L.c < 2) W(9 + 3 * A + (L.is0 ? 0 : F) + foot_slot, L.is0 ? sn : csn, uclk, L.is0 ? b_nclk0 : b_nclk1);
                if (leg == 0) {
                    W(cj, cmdv * (L.is2 ? h_obs_scale_ang_vel : h_obs_scale_lin_vel), 0.5f, 0.f);
                    W(3 + cj, pg, ug, L.sel(h_noise_lead_0, h_noise_lead_1, h_noise_lead_2));
                    W(6 + cj, bav * h_obs_scale_ang_vel, ua, L.sel(h_noise_lead_3, h_noise_lead_4, h_noise_lead_5));
                }
            }
            if (stB) {
                WP(oDR + 7 + d0 + cj, o_kp - h_kp_offset);
                WP(oDR + 7 + A + d0 + cj, o_kd - h_kd_offset);
                WP(oN + 3 * foot_slot + cj, nrm);
                WL(3 + K + F + 3 * foot_slot + cj, nrm);
                if (L.is2) { WP(oG + foot_slot, expC); WL(3 + K + foot_slot, clr); }
                if (leg == 0) {
                    WP(oDR + 2 + cj, o_com);
                    WP(oDR + 7 + 2 * A + cj, o_jnt);
                    WL(cj, blv * h_obs_scale_lin_vel);
                }
            }
            if (liveB) {
                if (leg == 1) WP(oDR + (L.c < 2 ? L.c : 3 + L.c), env4);                 // friction, mass | push x, y at oDR + 5, 6
                if (chas) { WP(oK + cidx, csv); WL(3 + cidx, csv); }
                if (leg == 0 && L.is3 && (smask & 1u)) { const float cb = nb2 > 1.f ? 1.f : 0.f; WP(oK, cb); WL(3, cb); }
#pragma unroll
                for (int i = 0; i < HQ; i++) {
                    const int k = hk0 + i * kstride;
                    float hv = pzn - h_heights_offset - hq[i];
                    if (h_heights_clip_scale) hv = clampf(hv, -1.f, 1.f) * h_obs_scale_height;
                    if (k < P) WP(oH + k, hv);
                }
#pragma unroll
                for (int k = 0; k < 9; k++)
                    if ((k & 3) == L.c) WP(oR + 9 * foot_slot + k, clampf(fpz - f_h9[k], -1.f, 1.f));
                // task state as the env class exposes it (tron1_pf_ee.py:167-184), the second action-history shift
                // (tron1_pf_ee.py:45-46: afterwards last == llast == a_t)
                float *ts = B.task_state + (size_t)e * LG_TASK_STATE_BIPED;
                if (L.is0) { ts[4 + foot_slot] = theta; ts[10 + foot_slot] = expC; }
                if (L.c < 2) ts[6 + (L.is0 ? 0 : F) + foot_slot] = L.is0 ? sn : csn;
                if (leg == 0 && L.c < 3) ts[L.c] = L.sel(gait_time, phi, gait_period);
                if (leg == 0 && L.is0 && B.obs_dirty) B.obs_dirty[e] = reset ? 1 : 0;
            }
            if (stB) {
                B.llast_actions[ja] = reset ? 0.f : last_act;
                B.last_actions[ja] = act;
            }
        }
        STAMP(10);
        // ---- persistent MDP state ----
        if (live) {        // role 0: what the reward terms produced
            const unsigned es4 = ((unsigned)el * (unsigned)N + (unsigned)e) * 4u;      // (term, N) rows: uniform base + 32-bit offset per lane
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (el + 8 * k < LG_R_COUNT && ((rmask >> (el + 8 * k)) & 1u))
                    *reinterpret_cast<float *>(reinterpret_cast<char *>(B.episode_sums + (size_t)8 * k * N) + es4) = es[k];
            if (L.is0) { B.feet_air_time[e * F + foot_slot] = air; B.last_contacts[e * F + foot_slot] = (uint8_t)last_contact; }
            if (leg == 0 && L.is0) B.rew_buf[e] = total;
        }
        if (liveB && leg == 0) {      // role 1: commands, counters, flags
            B.commands[4 * e + L.c] = cmdv;
            if (L.is0) {
                B.episode_length_buf[e] = ep_len; B.fail_buf[e] = (long long)failb;
                B.reset_buf[e] = reset ? 1 : 0; B.time_out_buf[e] = time_out ? 1 : 0;
            }
        }
        STAMP(11);
    }
    // ---------------- MDP phases in the same launch (every other task): leg-per-lane body on the first 16 lanes -----
    if (MPH != 0 && !CTAIL) {
        // hand the results to the MDP phases through LDS (layout: lg_kernel.h, XA .. XFB): leg-lane l of the tail is quad l
        // of this wave; a quad lane writes its own component.  Nothing the tail reads then comes from the arrays stored
        // above, so those stores drain in the background instead of being waited for.
        __shared__ float sX[NX * 16];
        {
            const int qi = (int)tl_ >> 2;
            auto XW = [&](int k, float v) { if (!L.is3) sX[(k + L.c) * 16 + qi] = v; };
            XW(XA, act); XW(XLA, last_act); XW(XLLA, llast_act); XW(XQ, q); XW(XQD, qd); XW(XLQD, qd_start); XW(XTQ, torque);
            XW(XFL, f_link[0]); XW(XFL + 3, f_link[1]); XW(XFL + 6, f_link[2]); XW(XFL + 9, f_link[3]);
            XW(XFP, foot_p); XW(XFV, foot_v); XW(XLFV, last_foot_v);
            XW(XPOS, pos); XW(XVW, vw); XW(XWW, ww); XW(XBLV, blv); XW(XBAV, bav); XW(XPG, pg); XW(XEUL, eul); XW(XFB, f_base);
            sX[(XQUAT + L.c) * 16 + qi] = quat;
            if (L.is0) sX[XBAD * 16 + qi] = guard_bad ? 1.f : 0.f;
        }
        // terrain samples (heightfield only) still travel through measured_heights / height_around_feet: a workgroup-scope
        // fence (the workgroup is this wave; the CU's L1 is coherent with its own stores) makes them visible.  An
        // agent-scope fence here writes back the XCD's L2 from every wave: measured +29 us per launch.
        if (P > 0) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __syncthreads();
        // the wave's 64 lanes run the leg-per-lane body as FOUR replicas of its 16 leg-lanes (lane = 16 replica + leg-lane): every replica
        // computes the same values, replica 0 owns the state stores, and the observation section deals its stores over the replicas
        // (the four destinations of an actor-frame entry, the two of a critic entry; blanking with 64 lanes)
        env_step_body<LEGS, MPH, true, (PROF == 5 ? 0 : PROF)>(p, sMraw, sHot, sStF, sX, wg * 16 + ((int)tl_ & 15), (int)tl_ & 15);
    }
    STAMPB(12288);
}
