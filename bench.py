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
# the other configs: SURVEY 8(d) "roofline env-steps/s at 8 TB/s" column (1.1e9 / 5.4e8 / 5.9e8) turned back into bytes; they
# assume the whole observation history is rewritten every step, which the sliding window no longer does
TASK_BYTES = {"go2": BYTES_PER_ENV_STEP, "go2_wtw": 7273, "go2_ee": 14815, "tron1_pf_ee": 13559}
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured copy)


WORKLOADS = {"go2": "go2_flat, flat-plane contact", "go2_wtw": "go2_wtw, periodic-gait rewards + domain rand, flat plane",
             "go2_ee": "go2_rough (go2_ee), heightfield terrain + terrain curriculum",
             "tron1_pf_ee": "tron1_pf_rough (tron1_pf_ee), biped, heightfield terrain + terrain curriculum"}


def ppo_rollout(env, iters, dev):
    """the rollout loop of rsl_rl/runners/on_policy_runner.py:118-124 -- act, step, store -- with the go2
    policy nets (rsl_rl/modules/actor_critic.py:57-82, dims legged_robot_config.py:279-281) as plain torch modules."""
    import torch
    import torch.nn as nn

    def mlp(i, o):
        return nn.Sequential(nn.Linear(i, 512), nn.ELU(), nn.Linear(512, 256), nn.ELU(), nn.Linear(256, 128), nn.ELU(), nn.Linear(128, o)).to(dev)
    torch.manual_seed(1)
    n, A, O = env.num_envs, env.num_actions, int(env.num_obs)
    actor, critic = mlp(O, A), mlp(O, 1)
    obs = env.get_observations()
    obs = obs[0] if isinstance(obs, tuple) else obs
    store = dict(obs=torch.empty(24, n, O, device=dev), act=torch.empty(24, n, A, device=dev), rew=torch.empty(24, n, device=dev),
                 done=torch.empty(24, n, device=dev), val=torch.empty(24, n, device=dev))

    def rollout():
        nonlocal obs
        with torch.inference_mode():
            for t in range(24):
                act = actor(obs) + torch.randn(n, A, device=dev)          # mean + unit-std exploration noise
                val = critic(obs)
                store["obs"][t], store["act"][t], store["val"][t] = obs, act, val[:, 0]
                out = env.step(act)
                obs, store["rew"][t], store["done"][t] = out[0], out[-3], out[-2]
    rollout()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        rollout()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": n * 24 * iters / dt, "unit": "env-steps/s", "iters": iters, "steps_per_iter": 24,
            "ms_per_step": dt / (24 * iters) * 1e3, "policy": "actor/critic MLP 512-256-128 ELU, f32, torch (hipBLASLt GEMMs)"}


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


def cpu_baseline(n_envs=4096, steps=800):
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
    ap.add_argument("--sync-gather", action="store_true",
                    help="wait for each step's all-gather before the next step (default: it overlaps the next step, double-buffered)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-timer-stride", type=int, default=8, help="time the physics kernel of every n-th step (0 = off)")
    ap.add_argument("--task", default="go2", choices=["go2", "go2_wtw", "go2_ee", "tron1_pf_ee"],
                    help="BASELINE config to run (the headline metric is go2; the others are reported under the same keys "
                         "with their own workload string)")
    ap.add_argument("--ppo-rollout", type=int, default=0, metavar="ITERS",
                    help="also time ITERS rollouts of 24 steps with the go2 actor/critic MLPs (45-512-256-128-12 / -1, ELU) "
                         "run between the steps (SURVEY 8d ii); reported under 'ppo_rollout', never as 'value'")
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
    env, cfg = make_env(args.task, n_local, dev, env_id_offset=rank * n_local, global_num_envs=n_local * world)
    n_obs = int(env.num_obs)
    env.reset()
    # on_policy_runner.py:105-106 init_at_random_ep_len
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    env.episode_length_buf[:] = torch.randint(0, int(env.max_episode_length), (n_local,), generator=g, device=dev, dtype=torch.int32)
    # fixed synthetic action stream: a small bank of N(0,1) batches cycled (clipped +-100 in-kernel)
    bank = [torch.randn(n_local, env.num_actions, generator=g, device=dev) for _ in range(16)]
    from hcr_genesis_lr_cl_amd.distributed import StepGather
    gather = StepGather(n_local, n_obs, world, dev, overlap=not args.sync_gather) if world > 1 and not args.no_gather else None

    def one_step(i):
        out = env.step(bank[i % len(bank)])
        if gather is not None:
            gather(out[0], out[-3], out[-2])      # 5-tuple (go2, wtw) or 6-tuple (estimator tasks): same tail (rew, done, extras)

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
    if gather is not None:
        gather.finish()        # every record of the timed region has arrived before the clock stops
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
        bytes_env = TASK_BYTES[args.task]
        achieved = bytes_env * n_local / launch_s / 1e9
        legs = 2 if args.task == "tron1_pf_ee" else 4
        if n_local * legs * 4 > 2048 * 64:
            layout = f"env_step_kernel<{legs},ALL> (leg-per-lane)"
        elif legs == 4:
            layout = "quad_sim_kernel<4,PRE,POST|RESET> (component-per-lane physics, MDP phases in its tail)"
        else:
            layout = "quad_sim_kernel<2,PRE,0> (component-per-lane physics; MDP phases follow in env_step_kernel<2,POST|RESET>)"
        out = {
            "metric": "env-steps/sec, Go2 flat 12-DOF, 4096 envs @1/2/4/8 MI355X" if args.task == "go2" else f"env-steps/sec, {args.task}",
            "value": value, "unit": "env-steps/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{WORKLOADS[args.task]}, {n_local} envs per GPU, fused LeggedRobot.step "
                                   f"(4 sub-steps dt=0.005) with synthetic N(0,1) actions",
                       "envs_total": total_envs, "parallelism": f"env-shard x{world}" + ((" + all-gather(obs,rew,done) per step" + ("" if args.sync_gather else ", overlapped with the next step")) if world > 1 and not args.no_gather else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": hbm_traffic(f"go2_flat_{n_local}") if args.task == "go2" else None,
                         "kernel": layout, "launch_us": launch_s * 1e6, "samples": kern_n,
                         "step_device_us": dev_ms * 1e3 / args.steps,
                         "algorithmic_bytes_per_launch": bytes_env * n_local},
        }
        if world == 1 and args.ppo_rollout > 0:
            out["ppo_rollout"] = ppo_rollout(env, args.ppo_rollout, dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
