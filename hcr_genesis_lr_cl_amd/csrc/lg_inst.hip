// lg_inst.hip -- the kernel instantiations of the library, compiled once per GROUP (hipcc -DLG_GROUP=g ... -c, the groups in parallel;
// hcr_genesis_lr_cl_amd/build.py).  One translation unit holding all of them took over three minutes to build; a group is a
// handful of kernels that share template parameters.  Groups 0-8 and 17-21 hold the component-per-lane kernels (lg_quad.h), groups 9-16 the
// leg-per-lane ones (lg_kernel.h) and do not include lg_quad.h, so an edit there leaves their objects valid.
//
// The host side (lg_host.hip) calls the launchers declared in lg_shared.h; every instantiation it names must appear in exactly
// one group below (a missing one is a link error, not a run-time surprise).
#ifndef LG_GROUP
#error "compile with -DLG_GROUP=<0..21> (hcr_genesis_lr_cl_amd/build.py)"
#endif
#include "lg_kernel.h"
#if LG_GROUP < 9 || LG_GROUP >= 17
#include "lg_quad.h"

template <int LEGS, bool PRE, unsigned MPH, int PROF, int JPL>
void lg_launch_quad(dim3 grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const KParams &p) {
    const dim3 block(PROF == 6 && MPH == (LG_PHASE_POST | LG_PHASE_RESET) ? 2 * BLOCK : BLOCK);   // PROF 6: two waves per group of envs (lg_quad.h DUO)
    if (e0 || e1) hipExtLaunchKernelGGL((quad_sim_kernel<LEGS, PRE, MPH, PROF, JPL>), grid, block, 0, st, e0, e1, 0, p);
    else hipLaunchKernelGGL((quad_sim_kernel<LEGS, PRE, MPH, PROF, JPL>), grid, block, 0, st, p);
}
#define QUAD(...) template void lg_launch_quad<__VA_ARGS__>(dim3, hipStream_t, hipEvent_t, hipEvent_t, const KParams &);
template <int LEGS, int PROF>
void lg_launch_quad_inj(dim3 grid, hipStream_t st, const KParams &p) {
    hipLaunchKernelGGL((quad_sim_kernel<LEGS, true, LG_PHASE_POST | LG_PHASE_RESET, PROF, 3, true>), grid, dim3(PROF == 6 ? 2 * BLOCK : BLOCK), 0, st, p);
}
#define QUAD_INJ(...) template void lg_launch_quad_inj<__VA_ARGS__>(dim3, hipStream_t, const KParams &);
template <int LEGS, int PROF, bool INJ>
void lg_launch_quad_rs(dim3 grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const KParams &p) {
    constexpr unsigned PR = LG_PHASE_POST | LG_PHASE_RESET;
    const dim3 block(PROF == 6 ? 2 * BLOCK : BLOCK);
    if (e0 || e1) hipExtLaunchKernelGGL((quad_sim_kernel<LEGS, true, PR, PROF, 3, INJ, true>), grid, block, 0, st, e0, e1, 0, p);
    else hipLaunchKernelGGL((quad_sim_kernel<LEGS, true, PR, PROF, 3, INJ, true>), grid, block, 0, st, p);
}
#define QUAD_RS(...) template void lg_launch_quad_rs<__VA_ARGS__>(dim3, hipStream_t, hipEvent_t, hipEvent_t, const KParams &);
#else
template <int LEGS, unsigned PH, int PROF, int JPL, bool REPL>
void lg_launch_env(dim3 grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const KParams &p) {
    if (e0 || e1) hipExtLaunchKernelGGL((env_step_kernel<LEGS, PH, PROF, JPL, REPL>), grid, dim3(BLOCK), 0, st, e0, e1, 0, p);
    else hipLaunchKernelGGL((env_step_kernel<LEGS, PH, PROF, JPL, REPL>), grid, dim3(BLOCK), 0, st, p);
}
#define ENV(...) template void lg_launch_env<__VA_ARGS__>(dim3, hipStream_t, hipEvent_t, hipEvent_t, const KParams &);
// the phase combinations lg_step accepts besides ALL (15) and POST | RESET (12), for one robot shape
#define ENV_PHASES(LEGS, JPL) ENV(LEGS, 2u, 0, JPL, false) ENV(LEGS, 3u, 0, JPL, false) ENV(LEGS, 4u, 0, JPL, false) ENV(LEGS, 5u, 0, JPL, false) \
                              ENV(LEGS, 7u, 0, JPL, false) ENV(LEGS, 8u, 0, JPL, false)
#endif

// ---- component per lane: quad_sim_kernel<LEGS, PRE, MDP phases in the tail, PROF, JPL> ----
#if LG_GROUP == 0      // go2 on the plane, whole step in one launch (the headline kernel); PRE | SIM | POST of the generic tail
QUAD(4, true, 12u, 1, 3) QUAD(4, true, 4u, 0, 3)
#elif LG_GROUP == 1    // go2_wtw, whole step
QUAD(4, true, 12u, 2, 3)
#elif LG_GROUP == 2    // go2_ee, whole step
QUAD(4, true, 12u, 3, 3)
#elif LG_GROUP == 3    // go2_ts / go2_cts / go2_dreamwaq (observation programs), whole step
QUAD(4, true, 12u, 4, 3)
#elif LG_GROUP == 4    // any other quadruped task, whole step (leg-per-lane MDP body in the tail)
QUAD(4, true, 12u, 0, 3)
#elif LG_GROUP == 5    // quadruped physics only (go2_cat, split launches of the tests)
QUAD(4, true, 0u, 0, 3) QUAD(4, true, 0u, 3, 3) QUAD(4, false, 0u, 0, 3) QUAD(4, false, 0u, 3, 3)
#elif LG_GROUP == 6    // biped physics (three joints per leg)
QUAD(2, true, 0u, 0, 3) QUAD(2, true, 0u, 3, 3) QUAD(2, false, 0u, 0, 3) QUAD(2, false, 0u, 3, 3)
#elif LG_GROUP == 7    // biped physics (four joints per leg: TRON1 sole foot)
QUAD(2, true, 0u, 0, 4) QUAD(2, false, 0u, 0, 4)
#elif LG_GROUP == 8    // biped, whole step in one launch (TRON1 point foot): leg-per-lane MDP body in the tail (0, 5: heightfield bound); tron1_pf_ee's component-layout tail (6)
QUAD(2, true, 12u, 0, 3) QUAD(2, true, 12u, 5, 3) QUAD(2, true, 12u, 6, 3)
// ---- leg per lane: env_step_kernel<LEGS, PHASES, PROF, JPL, REPL> ----
#elif LG_GROUP == 9    // whole step, quadruped (large batches)
ENV(4, 15u, 0, 3, false) ENV(4, 15u, 1, 3, false)
#elif LG_GROUP == 10    // whole step, biped
ENV(2, 15u, 0, 3, false) ENV(2, 15u, 0, 4, false)
#elif LG_GROUP == 11   // MDP phases behind a physics launch, quadruped
ENV(4, 12u, 0, 3, false) ENV(4, 12u, 0, 3, true) ENV(4, 13u, 0, 3, false)
#elif LG_GROUP == 12   // ... biped
ENV(2, 12u, 0, 3, false) ENV(2, 12u, 0, 3, true) ENV(2, 13u, 0, 3, false)
#elif LG_GROUP == 13   // ... sole-foot biped
ENV(2, 12u, 0, 4, false) ENV(2, 12u, 0, 4, true) ENV(2, 13u, 0, 4, false)
#elif LG_GROUP == 14   // single phases and their other combinations (golden replays, glue tests), quadruped
ENV_PHASES(4, 3)
#elif LG_GROUP == 15   // ... biped
ENV_PHASES(2, 3)
#elif LG_GROUP == 16   // ... sole-foot biped
ENV_PHASES(2, 4)
#elif LG_GROUP == 17   // the component-layout tails on injected read-backs and uniforms (golden replays through the benchmarked tails)
QUAD_INJ(4, 1) QUAD_INJ(4, 2) QUAD_INJ(4, 3) QUAD_INJ(4, 4) QUAD_INJ(2, 6)
// ---- the whole step / the golden-replay form with the profile's default reward set as a compile-time constant (lg_quad.h RS) ----
#elif LG_GROUP == 18   // go2 (the headline kernel of the default task), go2_wtw
QUAD_RS(4, 1, false) QUAD_RS(4, 2, false)
#elif LG_GROUP == 19   // ... on injected read-backs (golden replays)
QUAD_RS(4, 1, true) QUAD_RS(4, 2, true)
#elif LG_GROUP == 20   // go2_ee, the go2 rough heads
QUAD_RS(4, 3, false) QUAD_RS(4, 4, false)
#elif LG_GROUP == 21   // ... on injected read-backs
QUAD_RS(4, 3, true) QUAD_RS(4, 4, true)
#else
#error "LG_GROUP out of range"
#endif
