"""bench.py -- env-steps/s of the fused Go2 flat-terrain step (BASELINE.json metric).

One "step" = one LeggedRobot.step() over the rank's 4096 envs (PD + 4 physics sub-steps +
termination/rewards/resets/observations) with synthetic N(0,1) actions, state resident in HBM.
Weak scaling: 4096 envs per GPU (4096/8192/16384/32768 at 1/2/4/8 GPUs); with N>1 every step
ends with the RCCL all-gather of the per-env output record (obs 45 + reward + done) named by
BASELINE.json's north_star.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` and
`cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
BYTES_PER_ENV_STEP = 988          # go2 flat, f32, state read once + written once
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured copy)


def host_threads():
    """Threads the CPU leg may use: the cores this process is allowed on, capped at the GPU box's per-GPU share."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def hbm_traffic(workload_key):
    """HBM bytes per launch from the committed PMC pass (profiles/hbm_traffic.json, written by tools/pmc_traffic.py
    from separate rocprofv3 --pmc runs with the gfx950 FETCH_SIZE correction applied); None when not collected."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            return json.load(f).get(workload_key, {}).get("bytes_per_launch")
    except (OSError, ValueError):
        return None


def cpu_baseline(n_envs=2048, steps=150):
    """The CPU oracle (C, OpenMP over envs) on a bounded sample of the same workload."""
    import numpy as np
    from hcr_genesis_lr_cl_amd import builders, config as cfgmod
    from hcr_genesis_lr_cl_amd.config import GO2Cfg
    from hcr_genesis_lr_cl_amd.model_compiler import load_model
    from oracle import oracle as orc
    orc.build()
    model, cfg = load_model("go2"), GO2Cfg()
    desc, opts = builders.make_model_desc(model, cfg), builders.make_sim_options(model, cfg)
    st = orc.HostState(model, n_envs, cfgmod.default_dof_pos(cfg), 0.34)
    rng = np.random.default_rng(1)
    cores = host_threads()
    act = rng.normal(size=(n_envs, 12)).astype(np.float32)
    orc.sim_step(desc, opts, st, act, "f32", threads=cores)      # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        act = rng.normal(size=(n_envs, 12)).astype(np.float32)
        orc.sim_step(desc, opts, st, act, "f32", threads=cores)
    dt = time.perf_counter() - t0
    return {"value": n_envs * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n_envs} envs x {steps} control steps of go2 flat (physics phase of the C oracle, f32, "
                      f"OpenMP {cores} threads; the numpy MDP stack is not included)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--no-gather", action="store_true", help="skip the per-step RCCL all-gather at N>1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-timer-stride", type=int, default=8, help="time the physics kernel of every n-th step (0 = off)")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    n_gpus = world
    dev = f"cuda:{local_rank}"
    torch.cuda.set_device(dev)

    from hcr_genesis_lr_cl_amd.envs import make_env
    n_local = args.envs_per_gpu
    env, cfg = make_env("go2", n_local, dev, env_id_offset=rank * n_local, global_num_envs=n_local * world)
    env.reset()
    # on_policy_runner.py:105-106 init_at_random_ep_len
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    env.episode_length_buf[:] = torch.randint(0, int(env.max_episode_length), (n_local,), generator=g, device=dev, dtype=torch.int32)
    # fixed synthetic action stream: a small bank of N(0,1) batches cycled (clipped +-100 in-kernel)
    bank = [torch.randn(n_local, env.num_actions, generator=g, device=dev) for _ in range(16)]
    rec = torch.empty(n_local, 47, device=dev)
    gathered = torch.empty(world * n_local, 47, device=dev) if world > 1 else None

    def one_step(i):
        obs, _, rew, done, _ = env.step(bank[i % len(bank)])
        if gathered is not None and not args.no_gather:
            rec[:, :45] = obs
            rec[:, 45] = rew
            rec[:, 46] = done
            dist.all_gather_into_tensor(gathered, rec)

    for i in range(args.warmup):
        one_step(i)
    # roofline: the dominant kernel (physics) of every 8th step is bracketed by HIP events on the launch stream
    env._engine.profile(args.kernel_timer_stride)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()           # torch's current stream == the stream lg_step launches on (engine.py)
    for i in range(args.steps):
        one_step(i)
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        total_envs = n_local * world
        value = total_envs * args.steps / elapsed
        kern_us, kern_n = env._engine.profile_read()
        launch_s = kern_us * 1e-6 if kern_n else dev_ms / 1e3 / args.steps
        achieved = BYTES_PER_ENV_STEP * n_local / launch_s / 1e9
        layout = "quad_sim_kernel<4,PRE,POST|RESET> (component-per-lane physics, MDP phases in its tail)" if n_local * 16 <= 1024 * 64 \
            else "env_step_kernel<4,ALL> (leg-per-lane)"
        out = {
            "metric": "env-steps/sec, Go2 flat 12-DOF, 4096 envs @1/2/4/8 MI355X",
            "value": value, "unit": "env-steps/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"go2_flat, {n_local} envs per GPU, flat-plane contact, fused LeggedRobot.step "
                                   f"(4 sub-steps dt=0.005) with synthetic N(0,1) actions",
                       "envs_total": total_envs, "parallelism": f"env-shard x{world}" + (" + all-gather(obs,rew,done)" if world > 1 and not args.no_gather else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": hbm_traffic(f"go2_flat_{n_local}"),
                         "kernel": layout, "launch_us": launch_s * 1e6, "samples": kern_n,
                         "step_device_us": dev_ms * 1e3 / args.steps,
                         "algorithmic_bytes_per_launch": BYTES_PER_ENV_STEP * n_local},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
