"""Kernel-name prefixes (as rocprofv3 prints them, trailing template arguments left open) of one control step of each task at
4096 envs per GPU.  First entry = the launch that happens exactly once per step."""
_QUAD_FUSED = ["quad_sim_kernel<4, true, 12u,", "obs_compact_kernel"]
_BIPED = ["quad_sim_kernel<2, true, 12u,", "obs_compact_kernel"]                                   # one launch since round 3
_BIPED_2 = ["quad_sim_kernel<2, true, 0u,", "env_step_kernel<2, 12u", "obs_compact_kernel"]          # four-joint legs: physics, then the MDP launch
STEP_KERNELS = {
    "go2": ["quad_sim_kernel<4, true, 12u,"],
    "go2_wtw": _QUAD_FUSED, "go2_ee": _QUAD_FUSED, "tron1_pf_ee": _BIPED,
    # not BASELINE configs
    "go2_ts": _QUAD_FUSED, "go2_cts": _QUAD_FUSED, "go2_dreamwaq": _QUAD_FUSED, "tron1_pf": _BIPED,
    "go2_cat": ["quad_sim_kernel<4, true, 0u,", "env_step_kernel<4, 12u", "obs_compact_kernel"],
    "tron1_sf": _BIPED_2,
}


def key_of(task, n=4096):
    return f"{'go2_flat' if task == 'go2' else task}_{n}"


def profile_stamp():
    """What the counters were taken on: the profile tag of the pass (LG_PROFILE_TAG, set by tools/gpu_profile.sh) and the source hash of
    the library that ran (csrc/liblgsim.build.json).  bench.py marks a figure stale when the library it loads has another hash."""
    import json, os
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    try:
        h = json.load(open(os.path.join(root, "hcr_genesis_lr_cl_amd", "csrc", "liblgsim.build.json"))).get("source_hash")
    except (OSError, ValueError):
        h = None
    return {"tag": os.environ.get("LG_PROFILE_TAG"), "source_hash": h}
