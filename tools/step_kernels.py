"""Kernel-name substrings (as rocprofv3 prints them) of one control step of each BASELINE task at 4096 envs per GPU.
First entry = the launch that happens exactly once per step."""
STEP_KERNELS = {
    "go2": ["quad_sim_kernel<4, true, 12u, true>"],
    "go2_wtw": ["quad_sim_kernel<4, true, 12u, false>", "obs_compact_kernel"],
    "go2_ee": ["quad_sim_kernel<4, true, 12u, false>", "obs_compact_kernel"],
    "tron1_pf_ee": ["quad_sim_kernel<2, true, 0u, false>", "env_step_kernel<2, 12u>", "obs_compact_kernel"],
}


def key_of(task, n=4096):
    return f"{'go2_flat' if task == 'go2' else task}_{n}"
