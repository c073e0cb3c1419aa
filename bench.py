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
# SURVEY 8(d) persistent-state floats (read once + written once per control step) of the history-stacked tasks; the
# observation part of their traffic is derived from the task constants in algorithmic_bytes() below (DESIGN.md "Roofline
# bookkeeping"): the sliding window writes ONE new frame per stack per step instead of re-materialising the whole history
STATE_FLOATS = {"go2_wtw": 300, "go2_ee": 320, "tron1_pf_ee": 260,
                # not BASELINE configs (no SURVEY figure): the same count for their robot / terrain -- go2-rough heads as go2_ee, the TRON1
                # plane tasks as tron1_pf_ee without its terrain samples (49 + 2 x 12), tron1_sf with 2 more dofs (x 7 per-dof arrays)
                "go2_ts": 320, "go2_cts": 320, "go2_dreamwaq": 320, "go2_cat": 330, "tron1_pf": 190, "tron1_sf": 218}
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured copy)
VALU_PEAK_ONE_WAVE = 256 * 4 * 16 * 2.4e9   # 39.3e12 lane-ops/s: one wave64 per SIMD issues a non-packed f32 VALU instruction every 4 cycles
VALU_PEAK_CHIP = 78.6e12                     # MI355X_MICROARCH.md: vector FP32 157.3 TFLOP/s = 78.6 T lane-FMA/s (needs >= 2 waves per SIMD)


def algorithmic_bytes(task_name, t):
    """Bytes one env-step has to move (f32), DESIGN.md "Roofline bookkeeping".  go2: SURVEY 8(d)'s 988 B.  History tasks:
    SURVEY's state figure + what the sliding-window design writes per step: one new frame per stack into each of the
    `obs_sets` copies, the labels, and the compaction (read + write of stack-1 frames once every `obs_slack` steps)."""
    if task_name == "go2":
        return BYTES_PER_ENV_STEP
    sets = 2 if int(t.obs_sets) > 1 else 1
    fl = STATE_FLOATS[task_name]
    for frame, stack in ((int(t.obs_frame), int(t.obs_stack)), (int(t.priv_frame), int(t.priv_stack))):
        if frame == 0:
            continue
        fl += frame * (sets if stack > 1 else 1)
        if stack > 1 and int(t.obs_slack) > 0:
            fl += 2.0 * (stack - 1) * frame * sets / int(t.obs_slack)
    fl += int(t.num_labels)
    return 4.0 * fl


WORKLOADS = {"go2": "go2_flat, flat-plane contact", "go2_wtw": "go2_wtw, periodic-gait rewards + domain rand, flat plane",
             "go2_ee": "go2_rough (go2_ee), heightfield terrain + terrain curriculum",
             "tron1_pf_ee": "tron1_pf_rough (tron1_pf_ee), biped, heightfield terrain + terrain curriculum",
             "go2_ts": "go2_ts, teacher-student head on go2_rough", "go2_cts": "go2_cts, concurrent teacher-student head on go2_rough",
             "go2_dreamwaq": "go2_dreamwaq head on go2_rough (3000 envs in the reference's config)",
             "go2_cat": "go2_cat, constraints as terminations on go2_rough", "tron1_pf": "tron1_pf, point-foot biped, flat plane",
             "tron1_sf": "tron1_sf, 8-DOF sole-foot biped, flat plane"}


def ppo_rollout(task, n, iters, dev, T=24, gamma=0.99, lam=0.95):
    """the rollout loop of rsl_rl/runners/on_policy_runner.py:118-124 -- act, step, store, then
    compute_returns -- with the task's policy nets as plain torch modules (f32, hipBLASLt GEMMs):
      * go2 and the other 5-tuple tasks: actor obs->512->256->128->A, critic critic_obs->512->256->128->1, ELU
        (rsl_rl/modules/actor_critic.py:57-82, dims legged_robot_config.py:279-281);
      * go2_ee / tron1_pf_ee (6-tuple): estimator F->256->128->labels, actor (F + labels)->512->256->128->A + Hardtanh, critic
        C->1024->256->128->1 (rsl_rl/modules/actor_critic_ee.py:33-123; dims go2_ee_config.py:68-70, tron1_pf_ee_config.py:176-180).
    Two ways:
      * "eager": the reference's data flow -- policy ops dispatched one by one, a copy_ per stored tensor and step, GAE as the
        reference's Python loop;
      * "fused": hcr_genesis_lr_cl_amd.rollout.RolloutStorage -- reward / done / bootstrap in one record launch, GAE in two; the policy
        part of a step (launch-bound: ~25 small kernels) is captured once per row in a HIP graph and replayed on the storage rows.
        With unstacked observations (go2) the env writes each observation straight into its storage row (obs_sets = T + 1); with
        history stacks the window the env returned is copied into the row first (the window slides, a graph needs a fixed address)."""
    import torch
    import torch.nn as nn
    from hcr_genesis_lr_cl_amd.envs import TASKS, set_seed
    from hcr_genesis_lr_cl_amd.rollout import RolloutStorage

    def mlp(i, hidden, o, tail=None):
        layers, d = [], i
        for h in hidden:
            layers += [nn.Linear(d, h), nn.ELU()]
            d = h
        layers.append(nn.Linear(d, o))
        if tail is not None:
            layers.append(tail)
        return nn.Sequential(*layers).to(dev)
    cls, cfg_cls = TASKS[task]
    cfg = cfg_cls()
    cfg.env.num_envs = n
    stacked = task != "go2"
    if not stacked:
        cfg.hip.obs_sets = T + 1          # rollout-resident observation rows (lg_create refuses this with history stacks)
    set_seed(int(cfg.seed))
    env = cls(cfg, None, dev, True)
    torch.manual_seed(1)
    first = env.reset()
    ee = len(first) == 3                  # (features, labels, critic obs)
    A = env.num_actions
    if ee:
        F, NL, CO = (int(x.shape[1]) for x in first)
        est, actor, critic = mlp(F, [256, 128], NL), mlp(F + NL, [512, 256, 128], A, nn.Hardtanh(-100.0, 100.0)), mlp(CO, [1024, 256, 128], 1)
        nets = f"estimator {F}-256-128-{NL}, actor {F + NL}-512-256-128-{A} + Hardtanh, critic {CO}-1024-256-128-1 (actor_critic_ee.py:33-123)"
    else:
        F, CO = int(first[0].shape[1]), (int(first[1].shape[1]) if first[1] is not None else None)
        est, NL = None, 0
        actor, critic = mlp(F, [512, 256, 128], A), mlp(CO or F, [512, 256, 128], 1)
        nets = f"actor {F}-512-256-128-{A}, critic {CO or F}-512-256-128-1 (actor_critic.py:57-82)"
    log_std = torch.zeros(A, device=dev)
    st = RolloutStorage(n, T, [F], [CO], [A], dev, env=env if not stacked else None)
    lab_rows = torch.zeros(T, n, NL, device=dev) if ee else None
    res = {"steps_per_iter": T, "iters": iters, "zero_copy_observations": bool(st.zero_copy),
           "policy": nets + ", ELU, f32, torch (hipBLASLt GEMMs), unit-std Gaussian sampling + log-prob"}

    def current():                        # (actor input, critic input) the env currently offers
        o = env.get_observations()
        if ee:
            return o[0], o[2]
        p = env.get_privileged_observations()
        return o, (p if p is not None else o)

    def policy_row(t):                    # everything the policy contributes to row t, read from / written to the storage rows
        obs = st.observations[t]
        cobs = st.privileged_observations[t] if st.privileged_observations is not None else obs
        if ee:
            lab = est(obs)
            lab_rows[t].copy_(lab)
            mu = actor(torch.cat((obs, lab), dim=-1))
        else:
            mu = actor(obs)
        st.mu[t].copy_(mu)
        std = log_std.exp().expand_as(mu)
        st.sigma[t].copy_(std)
        act = mu + std * torch.randn_like(mu)
        st.actions[t].copy_(act)
        st.actions_log_prob[t, :, 0].copy_((-0.5 * ((act - mu) / std) ** 2 - log_std - 0.9189385332046727).sum(-1))
        st.values[t].copy_(critic(cobs))

    def stage(t):                         # stacked tasks: the sliding windows the env returned -> the fixed storage rows
        if not st.zero_copy:
            o, c = current()
            st.observations[t].copy_(o)
            if st.privileged_observations is not None:
                st.privileged_observations[t].copy_(c)

    graphs = None
    try:
        with torch.inference_mode():
            stage(0)
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3):
                    policy_row(0)
            torch.cuda.current_stream().wait_stream(s)
            graphs = []
            for t in range(T):
                gph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gph):
                    policy_row(t)
                graphs.append(gph)
    except Exception as e:      # noqa: BLE001 -- report, run the same ops eagerly
        res["graph_error"] = repr(e)[:200]
        graphs = None
    res["hip_graph"] = graphs is not None

    def rollout_fused():
        with torch.inference_mode():
            for t in range(T):
                stage(t)
                if graphs is not None:
                    graphs[t].replay()
                else:
                    policy_row(t)
                out = env.step(st.actions[t])
                st.add_step(out[-3], out[-2], out[-1]["time_outs"], gamma)
            st.compute_returns(critic(current()[1]), gamma, lam)
            st.clear()

    # the reference's flow on the same env / nets (rollout_storage.py:89-102, 124-138)
    ref = dict(obs=torch.zeros(T, n, F, device=dev), act=torch.zeros(T, n, A, device=dev), rew=torch.zeros(T, n, 1, device=dev),
               done=torch.zeros(T, n, 1, device=dev, dtype=torch.uint8), val=torch.zeros(T, n, 1, device=dev), logp=torch.zeros(T, n, 1, device=dev),
               mu=torch.zeros(T, n, A, device=dev), sigma=torch.zeros(T, n, A, device=dev), ret=torch.zeros(T, n, 1, device=dev))
    if CO:
        ref["cobs"] = torch.zeros(T, n, CO, device=dev)

    def rollout_eager():
        with torch.inference_mode():
            for t in range(T):
                obs, cobs = current()
                mu = actor(torch.cat((obs, est(obs)), dim=-1)) if ee else actor(obs)
                std = log_std.exp().expand_as(mu)
                act = mu + std * torch.randn_like(mu)
                logp = (-0.5 * ((act - mu) / std) ** 2 - log_std - 0.9189385332046727).sum(-1)
                val = critic(cobs)
                ref["obs"][t].copy_(obs)              # copied before the step: the returned windows are the env's buffers
                if CO:
                    ref["cobs"][t].copy_(cobs)
                out = env.step(act)
                rew, done, extras = out[-3], out[-2], out[-1]
                rew = rew.clone() + gamma * torch.squeeze(val * extras["time_outs"].unsqueeze(1), 1)
                ref["act"][t].copy_(act); ref["rew"][t].copy_(rew.view(-1, 1)); ref["done"][t].copy_(done.view(-1, 1))
                ref["val"][t].copy_(val); ref["logp"][t].copy_(logp.view(-1, 1)); ref["mu"][t].copy_(mu); ref["sigma"][t].copy_(std)
            last = critic(current()[1])
            adv = 0
            for t in reversed(range(T)):
                nv = last if t == T - 1 else ref["val"][t + 1]
                nt = 1.0 - ref["done"][t].float()
                delta = ref["rew"][t] + nt * gamma * nv - ref["val"][t]
                adv = delta + nt * gamma * lam * adv
                ref["ret"][t] = adv + ref["val"][t]
            a = ref["ret"] - ref["val"]
            a = (a - a.mean()) / (a.std() + 1e-8)
            return a

    for name, fn in (("fused", rollout_fused), ("eager", rollout_eager)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[name] = {"value": n * T * iters / dt, "unit": "env-steps/s", "ms_per_step": dt / (T * iters) * 1e3}
    res["value"], res["unit"], res["ms_per_step"] = res["fused"]["value"], "env-steps/s", res["fused"]["ms_per_step"]
    res["speedup_vs_eager"] = res["fused"]["value"] / res["eager"]["value"]
    return res


def host_threads():
    """Threads the CPU leg may use: the cores this process is allowed on, capped at the GPU box's per-GPU share."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def _profile_entry(name, workload_key):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f).get(workload_key) or {}
    except (OSError, ValueError):
        return {}


def _built_source_hash():
    try:
        with open(os.path.join(ROOT, "hcr_genesis_lr_cl_amd", "csrc", "liblgsim.build.json")) as f:
            return json.load(f).get("source_hash")
    except (OSError, ValueError):
        return None


def _stamp(entry):
    """Where a committed counter figure comes from: the profile tag and the source hash of the library the PMC pass ran on
    (tools/pmc_traffic.py, tools/pmc_sq.py); `stale` when the library loaded now was built from other sources (or the pass is
    unstamped): the counters then describe an older kernel and only the live timings of this line are current."""
    built = _built_source_hash()
    return {"from": "committed rocprofv3 --pmc pass (profiles/), not measured in this run", "tag": entry.get("tag"),
            "source_hash": entry.get("source_hash"), "built_source_hash": built,
            "stale": (entry.get("source_hash") is None) or (built is None) or (entry.get("source_hash") != built)}


def hbm_traffic(workload_key):
    """HBM bytes per control step from the committed PMC passes (profiles/hbm_traffic.json, written by tools/pmc_traffic.py
    from separate rocprofv3 --pmc runs with the gfx950 FETCH_SIZE correction applied); (None, None) when not collected."""
    e = _profile_entry("hbm_traffic.json", workload_key)
    if not e.get("bytes_per_launch"):
        return None, None
    return e["bytes_per_launch"], _stamp(e)


def valu_roofline(workload_key, launch_s, n_envs):
    """Second roofline figure (the go2 step sits on the VALU-issue side): SQ_INSTS_VALU per control step from the
    committed SQ counter pass (profiles/sq_counters.json, tools/pmc_sq.py) x 64 lanes / the live kernel time, against BOTH
    ceilings: what one wave per SIMD can issue (a lone wave64 issues one non-packed f32 VALU instruction per 4 cycles:
    MI355X_MICROARCH.md "one wave alone: 4") and what the chip's FP32 vector pipes can retire (157.3 TFLOP/s = 78.6 T lane-FMA/s)."""
    e = _profile_entry("sq_counters.json", workload_key)
    insts = e.get("SQ_INSTS_VALU")
    if not insts or launch_s <= 0:
        return None
    ach = insts * 64.0 / launch_s
    return {"bound": "valu-issue", "achieved": ach / 1e12, "unit": "T lane-ops/s",
            "peak": VALU_PEAK_ONE_WAVE / 1e12, "frac": ach / VALU_PEAK_ONE_WAVE,
            "peak_chip": VALU_PEAK_CHIP / 1e12, "frac_of_chip": ach / VALU_PEAK_CHIP,
            "peak_note": "peak = 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz: the issue ceiling of ONE wave64 per SIMD (this launch shape); "
                         "peak_chip = the FP32 vector rate of the chip (MI355X_MICROARCH.md: 157.3 TFLOP/s = 78.6 T lane-FMA/s), reachable "
                         "only with two or more waves per SIMD",
            "valu_insts_per_step": insts, "lane_ops_per_env_step": insts * 64.0 / n_envs,
            "valu_busy_frac": e.get("valu_active_share"), "wait_frac": e.get("wait_share"), "counters": _stamp(e)}


def stream_copy_gbs(dev, nbytes=1 << 30, iters=20):
    """Measured float4 stream-copy rate of this device (lg_stream_copy, include/lgsim.h): (read + write) GB/s."""
    import ctypes
    import torch
    from hcr_genesis_lr_cl_amd import abi
    lib = abi.load_lib()
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).fill_(1.0)
    b = torch.empty_like(a)
    g = ctypes.c_float()
    abi.check(lib.lg_stream_copy(a.data_ptr(), b.data_ptr(), nbytes, iters, torch.cuda.current_stream(torch.device(dev)).cuda_stream, ctypes.byref(g)), lib)
    return float(g.value)


_REAL_STDOUT = sys.stdout      # main() swaps in a private copy of fd 1 and points fd 1 at stderr (see there)


def spawn_ranks(n):
    """`python bench.py --gpus N` from a plain shell: start N ranks (one per GPU) with torch.distributed.run as a CHILD process
    -- this process has not touched the GPU (torch is not even imported yet) -- and pass its exit code on.  Rank 0 of the
    children prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


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
    t_phys = (time.perf_counter() - t0) / steps
    # the MDP half of the step (termination, rewards, reset + domain randomisation, commands, observation): the numpy oracle
    # on states of the same batch size (numpy: one thread)
    from oracle import mdp_oracle as mo
    from tests.golden_inputs import random_mdp_inputs
    task = builders.make_task_cfg(model, cfg)
    origins = np.zeros((n_envs, 3), np.float32)
    mdp = mo.MdpOracle(model, cfg, task, n_envs, origins)
    mdp.episode_length_buf[:] = rng.integers(0, 1001, n_envs)
    mdp_steps = 12
    sims = [random_mdp_inputs(rng, model, cfg, n_envs, task.slots.n_slots) for _ in range(4)]
    mdp.step(*sims[0], 1)                                         # warm-up
    t0 = time.perf_counter()
    for i in range(mdp_steps):
        mdp.step(*sims[i % 4], 2 + i)
    t_mdp = (time.perf_counter() - t0) / mdp_steps
    return {"value": n_envs / (t_phys + t_mdp), "unit": "env-steps/s", "cores": cores, "kind": "port",
            "physics_ms_per_step": t_phys * 1e3, "mdp_ms_per_step": t_mdp * 1e3,
            "sample": f"{n_envs} envs of go2 flat: {steps} control steps of the C oracle's physics phase (f32, OpenMP {cores} "
                      f"threads, {t_phys * 1e3:.2f} ms/step) + {mdp_steps} steps of the numpy oracle's MDP phases (one thread, "
                      f"{t_mdp * 1e3:.2f} ms/step); value = envs / (physics + MDP time per step)"}


def build_flags():
    """hipcc flags the loaded liblgsim.so was built with (sidecar written by hcr_genesis_lr_cl_amd/build.py)."""
    try:
        with open(os.path.join(ROOT, "hcr_genesis_lr_cl_amd", "csrc", "liblgsim.build.json")) as f:
            return json.load(f).get("flags")
    except (OSError, ValueError):
        return None


def select_gather(args, world, n_local, widths, dev, probe_with, flag_device):
    """Time `probe_with(gather, steps)` (microseconds per step) under each candidate transport and build the one every rank agrees is
    fastest (hcr_genesis_lr_cl_amd.distributed.pick_fastest: the mode whose slowest rank is fastest).  A transport that raises on some
    rank (e.g. peer mappings unavailable) drops out on all.  -> (gather, info dict for the JSON line)."""
    import torch
    import torch.distributed as dist
    from hcr_genesis_lr_cl_amd.distributed import GATHER_MODES, make_gather, pick_fastest
    n_probe = 6 * max(args.gather_batch, 1)
    step_alone = probe_with(None, n_probe)
    forced = "rccl-sync" if args.sync_gather else (None if args.gather_mode == "auto" else args.gather_mode)
    cand, errors = {}, {}
    for mode in ([forced] if forced else GATHER_MODES):
        g = None
        try:
            g = make_gather(mode, n_local, widths, world, dev, batch=args.gather_batch)
            g.prime()           # communicator / peer-mapping set-up is initialisation, not a step
            probe_with(g, n_probe)
            cand[mode] = probe_with(g, n_probe)
        except Exception as e:      # noqa: BLE001
            errors[mode] = repr(e)[:200]
        if g is not None and hasattr(g, "close"):
            g.close()
    flags = torch.tensor([1.0 if m in cand else 0.0 for m in GATHER_MODES], device=flag_device)
    dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    ok = [m for m, f in zip(GATHER_MODES, flags.tolist()) if f > 0.5]
    if not ok:
        raise SystemExit(f"bench.py: no gather transport works on every rank: {errors}")
    mode, cand_max = pick_fastest({m: cand[m] for m in ok})
    g = make_gather(mode, n_local, widths, world, dev, batch=args.gather_batch)
    g.prime()
    return g, {"mode": mode, "selected": "forced" if forced else "fastest of the candidates timed during the warm-up",
               "candidates_us": cand_max, "candidate_errors": errors or None, "probe_steps": n_probe,
               "step_us_alone": step_alone, "steps_per_exchange": int(g.batch)}


def launcher_dry_run(args, world, rank):
    """N > 1 plumbing of this file without a GPU: gloo process group, the same transport selection (select_gather: all three
    transports -- the peer-write one over /dev/shm -- timed, the fastest built), barrier + max-over-ranks clock, ONE JSON line on
    rank 0.  The env step is replaced by writing the rank id into a fake observation."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    n_local, widths = 32, [5, 3]
    obs = [torch.full((n_local, w), float(rank)) for w in widths]
    rew, done = torch.full((n_local,), 0.5 + rank), torch.zeros(n_local, dtype=torch.bool)
    done[rank] = True
    gather, info, out = None, None, None

    def probe_with(g, n_steps):
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            if g is not None:
                g(obs, rew, done)
        if g is not None:
            g.finish()
        if world > 1:
            dist.barrier()
        return (time.perf_counter() - t0) / n_steps * 1e6
    if world > 1:
        gather, info = select_gather(args, world, n_local, widths, "cpu", probe_with, "cpu")
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = gather(obs, rew, done) if gather is not None else None
    if gather is not None:
        gather.finish()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ok = True
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        last = (args.steps - 1) % gather.batch               # slot of the last step inside its batch
        for k in range(world):
            parts, r, d = gather.split(gather.step_view(out, k, last, gather.last_fill))
            ok &= all(bool((p_ == k).all()) for p_ in parts) and bool((r == 0.5 + k).all()) and int(d.sum()) == 1 and bool(d[k])
    if rank == 0:
        print(json.dumps({"metric": "launcher dry run", "value": n_local * world * args.steps / max(elapsed, 1e-9), "unit": "records/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "gather_ok": bool(ok), "gather": info, "data": "synthetic"}),
              file=_REAL_STDOUT, flush=True)
    if world > 1:
        if hasattr(gather, "close"):
            gather.close()
        dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions of --steps steps each after the one warm-up; value / ms_per_step are the median region's, the "
                         "list is reported as repeats_ms_per_step (SURVEY 8d: 5 repeats, median)")
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--no-stream-copy", action="store_true", help="skip the measured stream-copy bandwidth (roofline.peak_measured)")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL all-gather of the step outputs at N>1")
    ap.add_argument("--gather-batch", type=int, default=4, metavar="K",
                    help="control steps per all-gather at N>1 (default 4: one ~24 MB message per 4 steps at 8 ranks instead of four "
                         "latency-bound 6 MB ones, and a short exposed flush at the end of a 20-step run; 1 = one collective per step)")
    ap.add_argument("--sync-gather", action="store_true",
                    help="wait for each step's all-gather before the next step (default: it overlaps the next step, double-buffered)")
    ap.add_argument("--gather-mode", default="auto", choices=["auto", "rccl", "rccl-sync", "copy-engine"],
                    help="transport of the per-step record exchange at N>1: rccl = one all-gather per --gather-batch steps, overlapped; rccl-sync = "
                         "one per step, awaited; copy-engine = peer writes by the SDMA engines (no collective kernel on the CUs); auto (default) "
                         "times a few steps under each during the warm-up and runs the timed regions with the fastest (reported in `gather`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-timer-stride", type=int, default=8, help="time the physics kernel of every n-th step (0 = off)")
    ap.add_argument("--preroll", type=int, default=300,
                    help="untimed env steps before the warm-up that take the freshly reset population (every robot standing) to the state "
                         "mix of a long run (0 = off)")
    ap.add_argument("--task", default="go2", choices=list(WORKLOADS),
                    help="BASELINE config to run (the headline metric is go2; the others are reported under the same keys "
                         "with their own workload string)")
    ap.add_argument("--ppo-rollout", type=int, default=0, metavar="ITERS",
                    help="also time ITERS rollouts of 24 steps with the go2 actor/critic MLPs (45-512-256-128-12 / -1, ELU) "
                         "run between the steps (SURVEY 8d ii); reported under 'ppo_rollout', never as 'value'")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend at N > 1: nccl (= RCCL over xGMI, one GPU per rank; the measurement) or gloo (ranks may "
                         "share a GPU: rehearsal of the N > 1 path on a one-GPU box, never a reported number)")
    ap.add_argument("--launcher-dry-run", action="store_true",
                    help="rank plumbing only (spawn, rendezvous, gather, max-over-ranks timing, one JSON line) on the CPU with the "
                         "gloo backend and a stand-in for the env step: the world-2 test of this file's N > 1 path")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))          # before anything here touches the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); pass --gpus {world} or run "
                         f"`python bench.py --gpus {args.gpus}` from a plain shell (it spawns the ranks itself)")
    # stdout carries ONE JSON line and nothing else: whatever native libraries write to file descriptor 1 while the run lasts (gloo's
    # connection banner, RCCL / ROCm notices) goes to stderr; the line itself is written to the real stdout at the end
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if args.launcher_dry_run:
        return launcher_dry_run(args, world, rank)

    import torch
    dist = None
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if args.backend == "gloo" else local_rank
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{dev_index}"))
        else:
            dist.init_process_group("gloo")
    n_gpus = world
    dev = f"cuda:{dev_index}"
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
    def obs_outputs(out):      # every observation tensor step() returns: 5-tuple (obs, priv) or 6-tuple (features, labels, critic)
        return [o for o in out[:-3] if o is not None]
    widths = [int(o.shape[1]) for o in obs_outputs(env.step(bank[0]))]
    # Population pre-roll (part of building the workload, like the random episode clocks above): every env starts standing, so under N(0,1)
    # actions all n robots fall during the same first ~100 control steps -- a transient with every contact branch live at once, in which a
    # step costs 23.3 us against 21.8 us once falls and resets are spread out (and 20.9 us while the robots still stand).  The default run
    # (2000 steps after 100 of warm-up) is far behind it; a short one (the driver's --steps 20 --warmup 5) would sit right in it:
    # 159.5 M env-steps/s against 171 / 174 M after 300 / 1000 steps.  Both runs should describe the same state mix, so the env is stepped
    # `--preroll` times (default 300, ~7 ms) before the warm-up; the number is in the JSON line.
    for i in range(max(args.preroll, 0)):
        env.step(bank[i % len(bank)])
    gather = None
    gather_info = None

    def one_step(i):
        out = env.step(bank[i % len(bank)])
        if gather is not None:
            gather(obs_outputs(out), out[-3], out[-2])      # same tail in both arities: (rew, done, extras)

    def probe(n_steps):         # microseconds per step of n_steps steps under the current `gather`, bracketed like a timed region
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_steps):
            one_step(i)
        if gather is not None:
            gather.finish()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        return (time.perf_counter() - t0) / n_steps * 1e6

    if world > 1 and not args.no_gather:
        # SURVEY 8e / DESIGN 5: which transport hides best behind the step is a property of the node (RCCL's collective kernel cannot
        # share a SIMD with the step kernel; the copy engines leave the CUs alone but need peer mappings): measured here, not assumed
        def probe_with(g, n_steps):
            nonlocal gather
            gather = g
            for i in range(4):
                one_step(i)
            us = probe(n_steps)
            gather = None
            return us
        gather, gather_info = select_gather(args, world, n_local, widths, dev, probe_with, dev if args.backend == "nccl" else "cpu")

    for i in range(args.warmup):
        one_step(i)
    if gather is not None:
        gather.finish()
        gather.wait_s = gather.finish_s = 0.0
    # roofline: the dominant kernel (physics) of every `--kernel-timer-stride`-th step is bracketed by HIP events on the launch stream.
    # A timed launch is not free -- events attached to the dispatch cost ~10 us of bubble around it: stride 8 made the driver's 20-step
    # regions 6 % slower (27.5 vs 25.9 us per step, same build) -- so with R >= 3 regions only the FIRST carries the timer (it is the slowest
    # anyway: first dispatches after the warm-up): the median region, which is what `value` reports, is then one of at least four without it, and the kernel is still timed live inside timed regions of this
    # very run.  `roofline.timer` says which regions were sampled.
    n_rep = max(args.repeats, 1)
    timer_regions = list(range(n_rep)) if n_rep < 3 else [0]
    if args.kernel_timer_stride <= 0:
        timer_regions = []
    # at least ~10 samples in the sampled region, so that its first launch (right behind a synchronize: up to 10 us slower) does not
    # dominate the mean of a short region -- the driver's 20 steps are timed at every 2nd step, the default 2000 at every 8th
    timer_stride = max(1, min(args.kernel_timer_stride, args.steps // 10)) if args.kernel_timer_stride > 0 else 0
    # R timed regions of `--steps` steps each after the one warm-up, median reported.  A region is bracketed by a
    # barrier + synchronize on both sides and its time is the maximum over ranks, as the contract says for "the" timed region
    region_s, region_dev_ms = [], []
    step_i = args.warmup
    for rep in range(max(args.repeats, 1)):
        env._engine.profile(timer_stride if rep in timer_regions else 0)     # stride 0 = off; samples taken so far are kept
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        # device-side bracket of the region (step_device_us): two more packets on the queue, so -- like the kernel timer -- only in the regions
        # that are sampled anyway (all of them when there are fewer than three)
        dev_bracket = rep in timer_regions or not timer_regions
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if dev_bracket:
            ev0.record()       # torch's current stream == the stream lg_step launches on (engine.py)
        for i in range(args.steps):
            one_step(step_i + i)
        step_i += args.steps
        if gather is not None:
            gather.finish()        # every record of the timed region has arrived before the clock stops
        if dev_bracket:
            ev1.record()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        region_s.append(el)
        region_dev_ms.append(ev0.elapsed_time(ev1) if dev_bracket else None)
    if gather_info is not None:
        n_timed = args.steps * max(args.repeats, 1)
        gather_info.update({"step_us_with_gather": sorted(region_s)[len(region_s) // 2] / args.steps * 1e6,
                            "wait_us_per_step": gather.wait_s / n_timed * 1e6, "exposed_us_per_step": gather.finish_s / n_timed * 1e6,
                            "exposed_us_per_region": gather.finish_s / max(args.repeats, 1) * 1e6,
                            "note": "rank 0's own waits; wait = host time blocked on an exchange inside the steps, exposed = time from the last step's "
                                    "completion to the arrival of every record (gather.finish()), both inside the timed regions"})
    order = sorted(range(len(region_s)), key=lambda k: region_s[k])
    mid = order[len(order) // 2]                     # the median region (upper median for an even count)
    elapsed = region_s[mid]
    dev_all = sorted(x for x in region_dev_ms if x is not None)
    dev_ms = dev_all[len(dev_all) // 2]              # device-side bracket: median of the regions that carry one
    if rank == 0:
        total_envs = n_local * world
        value = total_envs * args.steps / elapsed
        kern_us, kern_n = env._engine.profile_read()
        launch_s = kern_us * 1e-6 if kern_n else dev_ms / 1e3 / args.steps
        if args.task == "go2_cat":      # physics and MDP phases are two lg_step calls there (a collective may sit between them): the
            launch_s = dev_ms / 1e3 / args.steps   # kernel timer sees the first only, so the whole step on the device is what counts
        bytes_env = algorithmic_bytes(args.task, env._engine.task)
        achieved = bytes_env * n_local / launch_s / 1e9
        wkey = f"{'go2_flat' if args.task == 'go2' else args.task}_{n_local}"
        traffic, traffic_stamp = hbm_traffic(wkey)
        peak_meas = None
        if not args.no_stream_copy:
            try:
                peak_meas = stream_copy_gbs(dev)
            except Exception as e:      # noqa: BLE001
                print(f"bench.py: stream copy measurement failed: {e!r}", file=sys.stderr)
        # the kernel instantiation(s) the LAST step really launched (lg_last_kernel: launcher expressions of lg_host.hip):
        # lg_launch_quad<LEGS, PRE, MDP phases in the tail, PROF, joints per leg> = quad_sim_kernel (one vector component per lane; PROF 1 go2
        # flat, 2 go2_wtw, 3 go2_ee, 4 observation programs, 6 tron1_pf_ee: component-layout tails; 0 / 5: generic tail), lg_launch_env<LEGS,
        # phases, PROF, JPL, REPL> = env_step_kernel (one leg per lane); PR = POST | RESET
        layout = env._engine.last_kernel()
        out = {
            "metric": "env-steps/sec, Go2 flat 12-DOF, 4096 envs @1/2/4/8 MI355X" if args.task == "go2" else f"env-steps/sec, {args.task}",
            "value": value, "unit": "env-steps/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "preroll_steps": max(args.preroll, 0),
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{WORKLOADS[args.task]}, {n_local} envs per GPU, fused LeggedRobot.step "
                                   f"(4 sub-steps dt=0.005) with synthetic N(0,1) actions" + (f", population pre-rolled {args.preroll} steps" if args.preroll > 0 else ""),
                       "envs_total": total_envs, "parallelism": f"env-shard x{world}" + (" [gloo rehearsal, ranks may share a GPU]" if world > 1 and args.backend == "gloo" else "") + ((f" + exchange of (obs, rew, done) of every step by {gather_info['mode']}, {gather_info['steps_per_exchange']} step(s) per exchange" + ("" if gather_info["mode"] == "rccl-sync" else ", overlapped with the following steps")) if gather_info else "")},
            "repeats": len(region_s), "repeats_ms_per_step": [x / args.steps * 1e3 for x in region_s],
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_counters": traffic_stamp,
                         "peak_measured": peak_meas, "frac_of_measured": (achieved / peak_meas) if peak_meas else None,
                         "peak_measured_how": "lg_stream_copy: float4 grid-stride copy of 1 GiB x 20 on this device, (read + write) bytes / time",
                         "kernel": layout, "launch_us": launch_s * 1e6, "samples": kern_n,
                         "timer": {"stride": timer_stride, "sampled_regions": timer_regions, "regions": n_rep,
                                   "note": "HIP events attached to the dispatch of every stride-th step, in the listed timed regions only (a timed launch costs ~10 us of bubble; the median region carries none when regions >= 3)"},
                         "step_device_us": dev_ms * 1e3 / args.steps,
                         "algorithmic_bytes_per_launch": bytes_env * n_local,
                         "valu": valu_roofline(wkey, launch_s, n_local)},
            "build_flags": build_flags(), "source_hash": _built_source_hash(),
        }
        if gather_info is not None:
            out["gather"] = gather_info
        if world == 1 and args.ppo_rollout > 0:
            del env
            try:
                out["ppo_rollout"] = ppo_rollout(args.task, n_local, args.ppo_rollout, dev)
            except Exception as e:      # noqa: BLE001 -- the side leg must never cost the headline line
                out["ppo_rollout_error"] = repr(e)[:300]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), file=_REAL_STDOUT, flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
