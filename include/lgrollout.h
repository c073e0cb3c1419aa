/* lgrollout.h -- C ABI of the rollout-side fusion: the per-step record of the reference's
 * RolloutStorage.add_transitions (rsl_rl/storage/rollout_storage.py:89-102) and its compute_returns
 * (rollout_storage.py:124-138) as HIP kernels.  Same conventions as lgsim.h: plain device pointers, sizes, the caller's HIP
 * stream as void*; 0 on success, otherwise non-zero with the message in lgsim.h's last-error call.  All storage tensors are the reference's:
 * (T, N, width) float32 row-major, dones (T, N, 1) uint8.
 */
#ifndef LGROLLOUT_H
#define LGROLLOUT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LG_ROLLOUT_MAX_COPIES 8

/* one strided row copy of lg_rollout_record: dst[e * width + k] = src[e * src_stride + k], k < width */
typedef struct LgRowCopy {
    const float *src;
    float *dst;
    int32_t width;
    int32_t src_stride;   /* floats between consecutive envs in src (history windows are strided views) */
} LgRowCopy;

/* One step's transition written into row `t` of the storage in ONE launch (the reference issues nine copy_ calls,
 * rollout_storage.py:92-100, after the bootstrap of rsl_rl/algorithms/ppo.py:106-113):
 *   rewards_row[e] = rew[e] + gamma * values_row[e] * time_outs[e]      (time_outs may be NULL: no bootstrap)
 *   dones_row[e]   = reset[e]
 *   and up to LG_ROLLOUT_MAX_COPIES strided row copies (observations, critic observations, labels ...).
 * rew: (N) f32, reset / time_outs: (N) uint8, values_row / rewards_row: (N) f32 rows of the (T, N, 1) tensors. */
int lg_rollout_record(int32_t n_envs, const float *rew, const uint8_t *reset, const uint8_t *time_outs, const float *values_row,
                      float gamma, float *rewards_row, uint8_t *dones_row, const LgRowCopy *copies, int32_t n_copies, void *stream);

/* Generalised advantage estimation over a finished rollout plus the advantage normalisation, rollout_storage.py:124-138:
 *   for t = T-1 .. 0:  next_v = t == T-1 ? last_values : values[t+1];  nt = 1 - dones[t]
 *                      delta = rewards[t] + nt * gamma * next_v - values[t];  adv = delta + nt * gamma * lam * adv
 *                      returns[t] = adv + values[t]
 *   advantages = returns - values;  advantages = (advantages - mean) / (std + 1e-8)   (std unbiased, over all T * N entries)
 * One thread per env walks its T steps (coalesced over envs); mean / std by one block reduction pass in f64.
 * scratch: device buffer of at least 2 doubles (sum, sum of squares). */
int lg_rollout_gae(int32_t n_steps, int32_t n_envs, const float *values, const float *rewards, const uint8_t *dones,
                   const float *last_values, float gamma, float lam, float *returns, float *advantages, double *scratch, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LGROLLOUT_H */
