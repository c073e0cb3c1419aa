// lg_rollout.hip -- rollout-side kernels behind include/lgrollout.h.
//
// Reference call sites replaced: rsl_rl/storage/rollout_storage.py:89-102 (add_transitions: nine copy_ launches per step),
// rsl_rl/algorithms/ppo.py:106-113 (time-out bootstrap of the reward) and rollout_storage.py:124-138 (compute_returns: a
// 24-iteration Python loop of ~8 elementwise launches each, then mean / std / normalise).  All of it is HBM-bound streaming
// work: one env per lane, consecutive lanes on consecutive envs, every array touched once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/lgrollout.h"

extern "C" const char *lg_last_error(void);
int lg_fail_msg(const std::string &m);   // lg_host.hip: sets the thread-local message, returns 1

struct RecordArgs {
    int n; const float *rew; const uint8_t *reset, *time_outs; const float *values; float gamma; float *rewards; uint8_t *dones;
    LgRowCopy c[LG_ROLLOUT_MAX_COPIES]; int nc; long long copy_total;
};

__global__ __launch_bounds__(256) void rollout_record_kernel(RecordArgs a) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long long e = tid; e < a.n; e += stride) {
        float r = a.rew[e];
        if (a.time_outs && a.time_outs[e]) r += a.gamma * a.values[e];       // ppo.py:110-111
        a.rewards[e] = r;
        a.dones[e] = a.reset[e];
    }
    // row copies: one flat index space over all segments, consecutive lanes on consecutive floats of a row
    long long base = 0;
    for (int s = 0; s < a.nc; s++) {
        const LgRowCopy c = a.c[s];
        const long long tot = (long long)a.n * c.width;
        for (long long i = tid; i < tot; i += stride) {
            const long long e = i / c.width;
            const int k = (int)(i - e * c.width);
            c.dst[i] = c.src[e * (long long)c.src_stride + k];
        }
        base += tot;
    }
    (void)base;
}

extern "C" int lg_rollout_record(int32_t n_envs, const float *rew, const uint8_t *reset, const uint8_t *time_outs, const float *values_row,
                                 float gamma, float *rewards_row, uint8_t *dones_row, const LgRowCopy *copies, int32_t n_copies, void *stream) {
    if (n_envs < 1 || !rew || !reset || !rewards_row || !dones_row) return lg_fail_msg("lg_rollout_record: null / empty argument");
    if (time_outs && !values_row) return lg_fail_msg("lg_rollout_record: the time-out bootstrap needs the value row");
    if (n_copies < 0 || n_copies > LG_ROLLOUT_MAX_COPIES || (n_copies && !copies)) return lg_fail_msg("lg_rollout_record: bad copy list");
    RecordArgs a;
    a.n = n_envs; a.rew = rew; a.reset = reset; a.time_outs = time_outs; a.values = values_row; a.gamma = gamma;
    a.rewards = rewards_row; a.dones = dones_row; a.nc = n_copies; a.copy_total = 0;
    long long most = n_envs;
    for (int i = 0; i < n_copies; i++) {
        if (!copies[i].src || !copies[i].dst || copies[i].width < 1 || copies[i].src_stride < copies[i].width)
            return lg_fail_msg("lg_rollout_record: bad row copy (null pointer, width < 1 or stride < width)");
        a.c[i] = copies[i];
        const long long tot = (long long)n_envs * copies[i].width;
        a.copy_total += tot;
        if (tot > most) most = tot;
    }
    const int block = 256;
    long long blocks = (most + block - 1) / block;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(rollout_record_kernel, dim3((unsigned)blocks), dim3(block), 0, (hipStream_t)stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lg_fail_msg(std::string("lg_rollout_record: ") + hipGetErrorString(e));
    return 0;
}

// ---- GAE: one env per lane, T steps backwards; per-block partial sums of the raw advantages for the normalisation ----
__global__ __launch_bounds__(256) void gae_kernel(int T, int N, const float *__restrict__ values, const float *__restrict__ rewards,
                                                  const uint8_t *__restrict__ dones, const float *__restrict__ last_values, float gamma,
                                                  float lam, float *__restrict__ returns, float *__restrict__ adv_out, double *scratch) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    if (e < N) {
        float adv = 0.f, next_v = last_values[e];
        for (int t = T - 1; t >= 0; t--) {
            const size_t i = (size_t)t * N + e;
            const float v = values[i];
            const float nt = 1.0f - (float)dones[i];
            const float delta = rewards[i] + nt * gamma * next_v - v;
            adv = delta + nt * gamma * lam * adv;
            const float ret = adv + v;
            returns[i] = ret;
            const float a = ret - v;                       // rollout_storage.py:137: returns - values (not `adv`: same f32 rounding as the reference)
            adv_out[i] = a;
            s1 += (double)a; s2 += (double)a * (double)a;
            next_v = v;
        }
    }
    // wave, then block reduction of (sum, sum of squares); one atomic pair per block
    for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off, 64); s2 += __shfl_down(s2, off, 64); }
    __shared__ double sh[2][4];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) { sh[0][w] = s1; sh[1][w] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < (int)(blockDim.x >> 6); k++) { a += sh[0][k]; b += sh[1][k]; }
        atomicAdd(&scratch[0], a);
        atomicAdd(&scratch[1], b);
    }
}

__global__ __launch_bounds__(256) void adv_normalize_kernel(long long total, float *__restrict__ adv, const double *scratch) {
    const double n = (double)total;
    const double mean = scratch[0] / n;
    double var = (scratch[1] - n * mean * mean) / (n - 1.0);     // torch.std: unbiased
    if (var < 0.0) var = 0.0;
    const float m = (float)mean, inv = 1.0f / ((float)sqrt(var) + 1e-8f);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
        adv[i] = (adv[i] - m) * inv;
}

extern "C" int lg_rollout_gae(int32_t n_steps, int32_t n_envs, const float *values, const float *rewards, const uint8_t *dones,
                              const float *last_values, float gamma, float lam, float *returns, float *advantages, double *scratch,
                              void *stream) {
    if (n_steps < 1 || n_envs < 1 || !values || !rewards || !dones || !last_values || !returns || !advantages || !scratch)
        return lg_fail_msg("lg_rollout_gae: null / empty argument");
    if ((long long)n_steps * n_envs < 2) return lg_fail_msg("lg_rollout_gae: the normalisation needs at least two entries");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(scratch, 0, 2 * sizeof(double), st);
    if (e != hipSuccess) return lg_fail_msg(std::string("lg_rollout_gae: ") + hipGetErrorString(e));
    const int block = 256;
    hipLaunchKernelGGL(gae_kernel, dim3((n_envs + block - 1) / block), dim3(block), 0, st, n_steps, n_envs, values, rewards, dones, last_values,
                       gamma, lam, returns, advantages, scratch);
    const long long total = (long long)n_steps * n_envs;
    long long blocks = (total + block - 1) / block;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adv_normalize_kernel, dim3((unsigned)blocks), dim3(block), 0, st, total, advantages, scratch);
    e = hipGetLastError();
    if (e != hipSuccess) return lg_fail_msg(std::string("lg_rollout_gae: ") + hipGetErrorString(e));
    return 0;
}
