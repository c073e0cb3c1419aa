"""GPU: the LeggedRobot/VecEnv surface, determinism, env-shard invariance, fused == split launches."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mk(n, **kw):
    from hcr_genesis_lr_cl_amd.envs import make_env
    return make_env("go2", n, "cuda:0", **kw)[0]


def _rollout(env, steps, seed=3, scale=1.0):
    import torch
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    env.reset()
    out = []
    for _ in range(steps):
        a = torch.randn(env.num_envs, 12, generator=g, device="cuda") * scale
        obs, priv, rew, done, extras = env.step(a)
        out.append((obs.clone(), rew.clone(), done.clone()))
    torch.cuda.synchronize()
    return out


def test_vecenv_surface():
    import torch
    env = _mk(128)
    assert (env.num_envs, env.num_obs, env.num_privileged_obs, env.num_actions) == (128, 45, None, 12)
    assert env.max_episode_length == 1000 and abs(env.dt - 0.02) < 1e-12
    obs, priv = env.reset()
    assert obs.shape == (128, 45) and priv is None and obs.data_ptr() == env.get_observations().data_ptr()
    obs, priv, rew, done, extras = env.step(torch.zeros(128, 12, device="cuda"))
    assert rew.shape == (128,) and done.dtype == torch.bool and done.shape == (128,)
    assert extras["time_outs"].dtype == torch.bool and "episode" in extras
    keys = list(extras["episode"])
    assert "rew_tracking_lin_vel" in keys and "max_command_x" in keys and len(keys) == 16
    assert torch.isfinite(obs).all() and obs.abs().max() <= 100.0
    # properties of the Simulator ABC (simulator.py:243-613) are live tensors of the right shape
    s = env.simulator
    assert s.link_contact_forces.shape == (128, 17, 3) and s.feet_pos.shape == (128, 4, 3)
    assert s.dof_pos_limits.shape == (12, 2) and s.default_dof_pos.shape == (1, 12)
    assert s.feet_indices == [4, 8, 12, 16] and s.termination_contact_indices == [0]
    assert len(s.penalized_contact_indices) == 8 and s.measured_heights.shape == (128, 187)


def test_deterministic_and_shard_invariant():
    import torch
    a = _rollout(_mk(256), 30)
    b = _rollout(_mk(256), 30)
    for (o1, r1, d1), (o2, r2, d2) in zip(a, b):
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2)
    # a rank owning envs [128, 256) of a 256-env job reproduces that slice bit for bit
    env = _mk(128, env_id_offset=128, global_num_envs=256)
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    env.reset()
    for t in range(30):
        act = torch.randn(256, 12, generator=g, device="cuda")[128:].contiguous()
        obs, _, rew, done, _ = env.step(act)
        assert torch.equal(obs, a[t][0][128:]) and torch.equal(rew, a[t][1][128:]) and torch.equal(done, a[t][2][128:])


def _go2_all_terms_cfg():
    """go2 on the plane with EVERY reward term the component-layout tail implements switched on (the stock config uses 15 of them),
    the termination term and yaw-rate (non-heading) commands: still inside flat_profile, so the fused launch runs the FLAT tail."""
    from hcr_genesis_lr_cl_amd.config import GO2Cfg
    cfg = GO2Cfg()
    sc = cfg.rewards.scales
    for name, v in (("dof_close_to_default", -0.02), ("dof_pos_stand_still", -0.5), ("dof_power", -2e-4), ("dof_vel_stand_still", -0.1),
                    ("feet_contact_stand_still", 0.5), ("foot_acc", -1e-5), ("foot_landing_vel", -0.1), ("hip_pos", -0.05),
                    ("keep_balance", 0.3), ("no_fly", 0.2), ("termination", -1.0)):
        setattr(sc, name, v)
    cfg.rewards.about_landing_threshold = 0.05
    cfg.rewards.only_positive_rewards = False
    return cfg


@pytest.mark.parametrize("task", ["go2", "go2-push", "go2-allterms", "go2-yawcmd", "go2_wtw", "go2_ee", "go2_ts", "go2_cts", "go2_dreamwaq", "tron1_pf_ee", "tron1_pf", "tron1_sf",
                                  "go2_cat", "go2_wtw-shift", "go2_ee-shift"])
def test_fused_launch_equals_split_launches(task, monkeypatch):
    """Same state in, one control step through (a) the fused launch -- the instantiation bench.py times,
    quad_sim_kernel<4, true, POST|RESET> with the MDP phases in its tail -- and (b) SIM, then PRE|POST|RESET in
    env_step_kernel<4, PRE|POST|RESET>, the very instantiation the golden replays of tests/test_gpu_mdp.py go through:
    identical integers, floats to 1e-5 (the template instantiations may contract FMAs differently; contact dynamics
    would amplify that over many steps, so states are re-synced every step).  Every quadruped task that uses the fused
    tail is covered: go2 (45-wide frame; its fused launch is the FLAT instantiation whose MDP phases run in component
    layout on all 64 lanes, lg_quad.h), go2_wtw (the same kind of tail: gait state, 5-frame stacks, PD-gain DR), go2_ee and the heads
    go2_ts / go2_cts / go2_dreamwaq (the same kind of tail again: terrain curriculum, terrain samples kept in the lanes that took them,
    20 / 5-frame stacks, labels / observation programs interpreted in component layout).  For the bipeds path (a) is what their env.step() issues -- component-per-lane physics launch, then the
    leg-per-lane MDP launch (tron1_sf: four-joint legs in both)."""
    import torch
    from hcr_genesis_lr_cl_amd import abi
    from hcr_genesis_lr_cl_amd.envs import make_env
    N = 256
    start = 480                                                      # crosses the push step (500) for go2_ee
    want = None
    if task.endswith("-shift"):      # in-place history shift: outside every profile, so the fused launch runs the GENERIC tail (PROF 0: the
        task = task[:-6]             # leg-per-lane body as four replicas in the tail of quad_sim_kernel<4, true, POST | RESET, 0>)
        monkeypatch.setenv("LG_OBS_SLACK", "0")
        want = "lg_launch_quad<4, true, PR, 0, 3>"
    if task == "go2_cat":            # two calls in the product too (physics, then the MDP launch: the job-wide CaT flag sits between them)
        want = "lg_launch_env<LEGS, PR, 0, JPL, false>"
    if task == "go2-push":                                           # go2 again, larger, across its push step (750)
        task, N, start = "go2", 1024, 735
    if task in ("go2-allterms", "go2-yawcmd"):
        from hcr_genesis_lr_cl_amd.envs import GO2, set_seed

        def mk():
            cfg = _go2_all_terms_cfg()
            cfg.env.num_envs = N
            if task == "go2-yawcmd":
                cfg.commands.heading_command = False
            set_seed(1)
            return GO2(cfg, None, "cuda:0", True)
        e1, e2 = mk(), mk()
        assert len(e1.reward_scales) >= 26
    else:
        e1, e2 = make_env(task, N, "cuda:0")[0], make_env(task, N, "cuda:0")[0]
    e1.reset(); e2.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    # spread the episode clocks so that time-outs, command and behaviour resampling fall inside the window
    e1.episode_length_buf[:] = torch.randint(0, 1000, (N,), generator=g, device="cuda", dtype=torch.int32)
    e1.common_step_counter = e2.common_step_counter = start
    loose = {"obs_buf": 2e-4, "priv_obs_buf": 2e-4, "labels_buf": 2e-4, "dof_vel": 2e-3, "last_dof_vel": 2e-3, "feet_vel": 2e-3,
             "last_feet_vel": 2e-3, "base_lin_vel": 2e-4, "base_ang_vel": 1e-3, "base_lin_vel_w": 2e-4, "base_ang_vel_w": 1e-3,
             "torques": 2e-3, "link_contact_forces": 0.05, "last_base_lin_vel": 2e-4, "last_base_ang_vel": 1e-3}
    n_reset = 0
    for t in range(40):
        for k in e1._engine.buf.keys():
            e2._engine.buf.raw(k).copy_(e1._engine.buf.raw(k))
        e2.common_step_counter = e1.common_step_counter
        act = torch.randn(N, e1.num_actions, generator=g, device="cuda") * (1.0 if t % 5 else 4.0)
        e1.step(act)
        if want is not None:
            assert want in e1._engine.last_kernel(), e1._engine.last_kernel()
        e2.common_step_counter += 1
        ca = float(e2.cfg.normalization.clip_actions)
        e2._engine.step(abi.PHASE_SIM, torch.clip(act, -ca, ca), e2.common_step_counter)       # Simulator.step: pre-clipped actions
        e2._engine.step(abi.PHASE_PRE | abi.PHASE_POST | abi.PHASE_RESET, act, e2.common_step_counter)
        torch.cuda.synchronize()
        assert e1._engine.obs_window() == e2._engine.obs_window()
        # velocities through a stiff contact solve carry ~1e-4 of instantiation-dependent round-off
        for k in e1._engine.buf.keys():
            a, b = e1._engine.buf[k], e2._engine.buf[k]
            if a.dtype in (torch.float32,):
                if k in ("episode_done_sums",):
                    continue
                np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-5, atol=loose.get(k, 1e-5), err_msg=f"{k} @ {t}")
            elif k != "episode_done_step":
                assert torch.equal(a, b), f"{k} @ {t}"
        n_reset += int(e1.reset_buf.sum())
    assert n_reset > 0


def test_long_random_rollout_is_sane():
    """600 control steps of N(0,1) actions: nothing diverges, failures terminate and reset,
    episode bookkeeping behaves (quirk 7/8 of SURVEY.md: fail_buf accumulates, reset > 5)."""
    import torch
    env = _mk(1024)
    env.reset()
    env.episode_length_buf[:] = torch.randint(0, 1000, (1024,), device="cuda", dtype=torch.int32)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    n_reset = 0
    for t in range(600):
        obs, _, rew, done, extras = env.step(torch.randn(1024, 12, generator=g, device="cuda"))
        n_reset += int(done.sum())
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all() and (rew >= 0).all()   # only_positive_rewards
    assert n_reset > 600          # random policies fall over (and time out) regularly
    assert int(env.episode_length_buf.max()) <= 1001 and int(env.fail_buf.max()) <= 6
    s = env.simulator
    assert float(s.base_pos[:, 2].min()) > -0.05 and float(s.base_pos[:, 2].max()) < 1.5
    assert float(s.dof_vel.abs().max()) <= 2 * 30.1 + 1e-3
    ep = extras["episode"]
    assert torch.isfinite(torch.as_tensor(float(ep["rew_tracking_lin_vel"])))


def test_philox_known_answers():
    """The kernel's generator against the Random123 known-answer vectors for Philox4x32-10 (kat_vectors: zero, all-ones
    and the pi digits cases)."""
    import ctypes as C
    from hcr_genesis_lr_cl_amd import abi
    lib = abi.load_lib()
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        c, k, o = (C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), (C.c_uint32 * 4)()
        abi.check(lib.lg_philox(C.byref(c), C.byref(k), C.byref(o)), lib)
        assert tuple(o) == want, [hex(v) for v in o]


def test_full_size_batch_properties():
    """BASELINE size (4096 envs per GPU): size-independent properties of a 120-step rollout with both physics layouts --
    unit quaternions, finite state, clipped observations, episode counters that count, time-outs exactly at the episode
    length, a shard of the batch reproducing its slice bit for bit, and the two layouts agreeing on one step taken from
    the same state."""
    import torch
    from hcr_genesis_lr_cl_amd import abi
    N = 4096
    env = _mk(N)
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    env.reset()
    env.episode_length_buf[:] = torch.randint(0, 1000, (N,), generator=g, device="cuda", dtype=torch.int32)
    ep0 = env.episode_length_buf.clone()
    ever_reset = torch.zeros(N, dtype=torch.bool, device="cuda")
    acts = [torch.randn(N, 12, generator=g, device="cuda") for _ in range(120)]
    for t, a in enumerate(acts):
        obs, _, rew, done, extras = env.step(a)
        ever_reset |= done
        # time-outs fire exactly when the counter passes the episode length (legged_robot.py:88-91)
        assert torch.equal(extras["time_outs"] & done, extras["time_outs"])
    torch.cuda.synchronize()
    b = env._engine.buf
    assert torch.isfinite(obs).all() and obs.abs().max() <= 100.0 and torch.isfinite(rew).all()
    for k in ("dof_pos", "dof_vel", "base_pos", "base_lin_vel_w", "base_ang_vel_w", "link_contact_forces", "torques"):
        assert torch.isfinite(b[k]).all(), k
    assert (b["base_quat"].norm(dim=1) - 1).abs().max() < 1e-5
    never = ~ever_reset
    assert torch.equal(env.episode_length_buf[never], ep0[never] + 120)          # counters count
    assert ever_reset.sum() >= 1 and bool((env.episode_length_buf[ever_reset] <= 120).all())
    # a rank owning the upper half of the same job reproduces that half bit for bit
    half = _mk(N // 2, env_id_offset=N // 2, global_num_envs=N)
    half.reset()
    half.episode_length_buf[:] = ep0[N // 2:]
    ref = _mk(N)
    ref.reset()
    ref.episode_length_buf[:] = ep0
    for a in acts[:40]:
        o1 = ref.step(a)[0]
        o2 = half.step(a[N // 2:].contiguous())[0]
    assert torch.equal(o1[N // 2:], o2)
    # both physics layouts, one control step from the same state
    import copy
    e1, e2 = ref._engine, None
    from hcr_genesis_lr_cl_amd.engine import Engine
    opts = copy.copy(e1.opts); opts.sim_layout = 1
    e2 = Engine(e1.model, e1.desc, opts, e1.task, N, "cuda:0")
    for k, v in e1.buf.items():
        if k in e2.buf:
            e2.buf[k].copy_(v)
    e1.step(abi.PHASE_SIM, acts[50], 0)
    e2.step(abi.PHASE_SIM, acts[50], 0)
    torch.cuda.synchronize()
    for k, tol, frac in (("dof_pos", 5e-4, 1e-3), ("base_pos", 5e-5, 1e-3), ("base_quat", 1e-4, 1e-3), ("dof_vel", 0.1, 2e-3)):
        bad = (e1.buf[k] - e2.buf[k]).abs() > tol
        assert bad.float().mean() <= frac, (k, int(bad.sum()), float((e1.buf[k] - e2.buf[k]).abs().max()))


@pytest.mark.parametrize("task", ["go2", "go2_wtw", "go2_ee", "tron1_pf_ee", "tron1_sf"])
def test_long_rollout_stays_sane(task):
    """1500 control steps of aggressive random actions (sigma 2) on every BASELINE task at 1024 envs: state and outputs
    stay finite and bounded, resets keep happening (robots fall) and robots never leave the terrain bounds."""
    import torch
    from hcr_genesis_lr_cl_amd.envs import make_env
    env, cfg = make_env(task, 1024, "cuda:0")
    env.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    A = env.num_actions
    resets = 0
    for t in range(1500):
        out = env.step(2.0 * torch.randn(1024, A, generator=g, device="cuda"))
        if t % 100 == 99:
            obs, rew, done = out[0], out[-3], out[-2]
            assert torch.isfinite(obs).all() and torch.isfinite(rew).all(), (task, t)
            resets += int(done.sum())
    b = env._engine.buf
    for k in ("dof_pos", "dof_vel", "base_pos", "base_quat", "base_lin_vel_w", "base_ang_vel_w", "link_contact_forces", "torques"):
        assert torch.isfinite(b[k]).all(), (task, k)
    assert (b["base_quat"].norm(dim=1) - 1).abs().max() < 1e-4
    assert b["base_pos"][:, 2].abs().max() < 20.0 and b["dof_vel"].abs().max() <= 2.0 * 31.0 + 1e-3
    assert b["base_lin_vel_w"].abs().max() <= 50.0 + 1e-3 and b["base_ang_vel_w"].abs().max() <= 40.0 + 1e-3
    assert resets > 0


@pytest.mark.parametrize("layout", [1, 2])
def test_ragged_batch_sizes(layout, monkeypatch):
    """Env counts that do not fill a wave (7 and 10 envs: dead lanes shadow the last env): the first 7 envs of the 10-env
    batch evolve bit for bit like the 7-env batch (draws are keyed on the env id), in both physics layouts."""
    import torch
    monkeypatch.setenv("LG_SIM_LAYOUT", str(layout))
    g = torch.Generator(device="cuda"); g.manual_seed(21)
    acts = [torch.randn(10, 12, generator=g, device="cuda") for _ in range(60)]
    runs = []
    for n in (7, 10):
        env = _mk(n)
        env.reset()
        outs = []
        for a in acts:
            obs, _, rew, done, _ = env.step(a[:n].contiguous())
            outs.append((obs[:7].clone(), rew[:7].clone(), done[:7].clone()))
        assert torch.isfinite(env._engine.buf["dof_pos"]).all()
        runs.append(outs)
    for (o1, r1, d1), (o2, r2, d2) in zip(*runs):
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2)


@pytest.mark.parametrize("task", ["go2_wtw", "go2_ee", "go2_ts", "go2_dreamwaq", "tron1_pf_ee", "tron1_pf", "tron1_sf"])
def test_ragged_batch_sizes_task_tails(task):
    """The same for the task-specific tails (component-layout MDP on all lanes of the wave, observation stores dealt over the
    lanes of an env, the two-launch bipeds): a 7-env shard of a 10-env job (a wave holds 4 quadruped / 8 biped envs, so the last
    wave is partly dead in both runs) steps bit for bit like the first 7 envs of the 10-env job -- every returned observation
    tensor, reward and done flag."""
    import torch
    from hcr_genesis_lr_cl_amd.envs import make_env
    g = torch.Generator(device="cuda"); g.manual_seed(22)
    runs = []
    for n in (7, 10):
        env = make_env(task, n, "cuda:0", global_num_envs=10)[0]
        env.reset()
        A = env.num_actions
        if not runs:
            bank = [torch.randn(10, A, generator=g, device="cuda") for _ in range(40)]
        outs = []
        for a in bank:
            out = env.step(a[:n].contiguous())
            obs = [o[:7].clone() for o in out[:-3] if o is not None]
            outs.append((obs, out[-3][:7].clone(), out[-2][:7].clone()))
        for k in ("dof_pos", "dof_vel", "base_pos", "base_quat"):
            assert torch.isfinite(env._engine.buf[k]).all(), (task, n, k)
        runs.append(outs)
    for t, ((o1, r1, d1), (o2, r2, d2)) in enumerate(zip(*runs)):
        assert len(o1) == len(o2)
        for a, b in zip(o1, o2):
            assert torch.equal(a, b), (task, t)
        assert torch.equal(r1, r2) and torch.equal(d1, d2), (task, t)


@pytest.mark.parametrize("task", ["tron1_pf_ee", "tron1_pf", "tron1_sf", "go2_cat"])
def test_replicated_mdp_launch_is_bit_identical(task, monkeypatch):
    """MDP phases in a launch of their own (bipeds, go2_cat): plain (one leg per lane) and replicated (16 leg-lanes x 4 replicas per
    wave, observation stores dealt over the replicas, one Philox block per replica fetched where needed) give the same rollout bit
    for bit -- every returned tensor and the state behind it, through resets, pushes and window compaction."""
    import torch
    from hcr_genesis_lr_cl_amd.envs import make_env
    monkeypatch.setenv("LG_OBS_SLACK", "16")
    N = 200
    runs = []
    for repl in ("0", "1"):
        monkeypatch.setenv("LG_MDP_REPLICAS", repl)
        env, _ = make_env(task, N, "cuda:0")
        env.reset()
        g = torch.Generator(device="cuda"); g.manual_seed(8)
        env.episode_length_buf = torch.randint(940, 1001, (N,), generator=g, device="cuda", dtype=torch.int32)
        outs = []
        for t in range(70):
            out = env.step(torch.randn(N, env.num_actions, generator=g, device="cuda") * 1.5)
            outs.append([o.clone() for o in out[:-1] if torch.is_tensor(o)])
        b = env._engine.buf
        state = {k: b[k].clone() for k in ("dof_pos", "dof_vel", "base_pos", "base_quat", "commands", "episode_sums", "kp_scale", "kd_scale",
                                          "friction_values", "added_base_mass", "base_com_bias", "feet_air_time", "task_state") if k in b}
        runs.append((outs, state, sum(int(o[-1].sum()) for o in outs)))
    (a, sa, ra), (b_, sb, rb) = runs
    assert ra == rb and ra > N // 4
    for t, (x, y) in enumerate(zip(a, b_)):
        assert len(x) == len(y)
        for u, v in zip(x, y):
            assert torch.equal(u, v), (task, t)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), (task, k)


def test_step_gather_packs_strided_windows_and_bool_done_on_device():
    """distributed.StepGather on the device with what a history task really returns: observation tensors that are strided row
    windows (go2_wtw: 305 of a wider row), a float reward and a BOOL done -- packed by one torch.cat into the float record, several
    steps per batch, read back through step_view / split (world 1: the packing, batching and partial flush without a collective;
    the collective itself is covered on gloo in tests/test_distributed_cpu.py and by the 2-rank rehearsal of bench.py)."""
    import torch
    from hcr_genesis_lr_cl_amd.distributed import StepGather
    from hcr_genesis_lr_cl_amd.envs import make_env
    N = 64
    env, _ = make_env("go2_wtw", N, "cuda:0")
    env.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(2)
    env.episode_length_buf = torch.randint(990, 1001, (N,), generator=g, device="cuda", dtype=torch.int32)      # some time-outs
    out = env.step(torch.randn(N, 12, generator=g, device="cuda"))
    obs_list = [o for o in out[:-3] if o is not None]
    assert any(o.stride(0) != o.shape[1] for o in obs_list) and out[-2].dtype == torch.bool
    sg = StepGather(N, [int(o.shape[1]) for o in obs_list], 1, "cuda:0", overlap=True, batch=3)
    sg.prime()
    def check(buf, kept, fill=None):
        for slot, (obs_k, rew_k, done_k) in enumerate(kept):
            parts, r, d = sg.split(sg.step_view(buf, 0, slot, fill))
            for a, b in zip(parts, obs_k):
                assert torch.equal(a, b), slot
            assert torch.equal(r, rew_k) and d.dtype == torch.bool and torch.equal(d, done_k), slot

    kept, n_done = [], 0
    for t in range(5):      # 5 steps in batches of 3 (one buffer pair at world 1: a batch is read before the next one overwrites it)
        out = env.step(torch.randn(N, 12, generator=g, device="cuda") * 2.0)
        obs_list = [o for o in out[:-3] if o is not None]
        buf = sg(obs_list, out[-3], out[-2])
        kept.append(([o.clone() for o in obs_list], out[-3].clone(), out[-2].clone()))
        n_done += int(out[-2].sum())
        if t == 2:
            check(buf, kept)
            kept = []
    sg.finish()             # the second batch is flushed partly filled
    assert sg.last_fill == 2 and n_done > 0
    check(buf, kept, sg.last_fill)


def _peer_worker(rank, world, port, q):
    import os
    import torch
    import torch.distributed as dist
    from hcr_genesis_lr_cl_amd.distributed import GATHER_MODES, make_gather, shard
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    off, n = shard(256, world, rank)
    ids = torch.arange(off, off + n, dtype=torch.float32, device="cuda")
    wide = torch.zeros(n, 40, device="cuda")
    wide[:, 7:12] = ids[:, None] + torch.arange(5, device="cuda") * 0.1
    obs = wide[:, 7:12]                                   # a strided window, as the history tasks return
    ok = True
    for mode in GATHER_MODES:
        g = make_gather(mode, n, 5, world, "cuda:0", batch=3)
        g.prime()
        outs = [g(obs + k, ids * (2 + k), (ids % 3 == 0)) for k in range(7)]
        g.finish()
        for k in ((6,) if mode == "rccl-sync" else (3, 5, 6)):
            slot, fill = k % g.batch, (g.last_fill if k // g.batch == 6 // g.batch else None)
            for rk in range(world):
                o_k, n_k = shard(256, world, rk)
                (o,), r, d = g.split(g.step_view(outs[k], rk, slot, fill))
                ids_k = torch.arange(o_k, o_k + n_k, dtype=torch.float32, device="cuda")
                ok &= bool(torch.allclose(r, ids_k * (2 + k))) and bool(torch.allclose(o[:, 0], ids_k + k)) and bool(torch.equal(d, ids_k % 3 == 0))
        if hasattr(g, "close"):
            g.close()
        dist.barrier()
    q.put((rank, ok))
    dist.destroy_process_group()


def test_three_gather_transports_between_two_processes_on_one_gpu():
    """The N > 1 record exchange with the records on the DEVICE, two processes sharing cuda:0 (gloo process group as the control plane):
    all three transports of distributed.make_gather deliver the same records -- in particular "copy-engine", where each process maps the
    other's receive buffer through an IPC memory handle and writes its batches straight into it on a side stream (no collective)."""
    import socket
    import torch.multiprocessing as mp
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_peer_worker, args=(r, 2, port, q)) for r in range(2)]
    [p_.start() for p_ in ps]
    res = [q.get(timeout=300) for _ in range(2)]
    [p_.join(timeout=60) for p_ in ps]
    assert sorted(res) == [(0, True), (1, True)], res


def _obs_tensors(out):
    """(actor obs, critic obs / labels ...) of a step()/reset() result: every tensor-valued observation output."""
    import torch
    return [o for o in out[:3] if torch.is_tensor(o) and o.dtype == torch.float32 and o.dim() == 2]


@pytest.mark.parametrize("slack", ["64", "1"], ids=["default-slack", "min-slack"])
@pytest.mark.parametrize("task", ["go2", "go2_wtw", "go2_ee", "tron1_pf_ee", "tron1_sf"])
def test_returned_observations_survive_the_next_step(task, slack, monkeypatch):
    """The ordering of the reference's rollout loop (rsl_rl/algorithms/ppo.py:103-104 keeps `obs`,
    on_policy_runner.py:118-124 calls env.step, rollout_storage.py:92 copies `obs` only afterwards): what a step
    handed out must still be the PRE-step observation after the next step has run -- including for envs that reset in
    that step (their history is blanked) and on steps where the sliding history window is compacted."""
    import torch
    from hcr_genesis_lr_cl_amd.envs import make_env
    monkeypatch.setenv("LG_OBS_SLACK", slack)
    N = 128
    env, cfg = make_env(task, N, "cuda:0")
    held = _obs_tensors(env.reset())
    g = torch.Generator(device="cuda"); g.manual_seed(2)
    env.episode_length_buf = torch.randint(900, 1001, (N,), generator=g, device="cuda", dtype=torch.int32)   # time-outs inside the window
    n_reset = 0
    for t in range(90):
        snap = [h.clone() for h in held]
        out = env.step(torch.randn(N, env.num_actions, generator=g, device="cuda") * 2.0)
        torch.cuda.synchronize()
        for h, s in zip(held, snap):
            assert torch.equal(h, s), f"observation of step {t - 1} changed under step {t}"
        new = _obs_tensors(out)
        assert all(a.data_ptr() != b.data_ptr() for a, b in zip(new, held))
        held = new
        n_reset += int(out[-2].sum())
    assert n_reset > N // 2


@pytest.mark.parametrize("slack", ["64", "1", "0"], ids=["window", "window-min-slack", "shift"])
@pytest.mark.parametrize("task", ["go2_wtw", "tron1_pf_ee"])
def test_two_observation_sets_equal_a_single_one(task, slack, monkeypatch):
    """Same seeded rollout with LgTaskCfg.obs_sets = 2 (default) and = 1: every returned observation bit-identical.  Covers
    the hand-over between the two copies: the frame written into the other copy, the deferred blanking of an env reset at
    the previous launch, compaction of both copies."""
    import torch
    from hcr_genesis_lr_cl_amd.envs import make_env
    monkeypatch.setenv("LG_OBS_SLACK", slack)
    N = 96

    def run(sets):
        monkeypatch.setenv("LG_OBS_SETS", str(sets))
        env, _ = make_env(task, N, "cuda:0")
        assert int(env._engine.task.obs_sets) == sets
        env.reset()
        g = torch.Generator(device="cuda"); g.manual_seed(4)
        env.episode_length_buf = torch.randint(930, 1001, (N,), generator=g, device="cuda", dtype=torch.int32)
        outs = []
        for t in range(75):
            out = env.step(torch.randn(N, env.num_actions, generator=g, device="cuda") * 2.0)
            outs.append([o.clone() for o in _obs_tensors(out)] + [out[-2].clone()])
        return outs
    a, b = run(2), run(1)
    for t, (x, y) in enumerate(zip(a, b)):
        for u, v in zip(x, y):
            assert torch.equal(u, v), f"step {t}"
    assert sum(int(x[-1].sum()) for x in a) > N // 2


def test_runner_statements_on_the_env():
    """The statements the reference's unmodified runner executes on the env object (on_policy_runner.py:105-106, 182-185)."""
    import torch
    env = _mk(64)
    env.reset()
    # init_at_random_ep_len REBINDS the attribute: the kernel's buffer has to follow
    env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
    assert env.episode_length_buf.data_ptr() == env._engine.buf["episode_length_buf"].data_ptr()
    before = env.episode_length_buf.clone()
    assert int(before.max()) > 10
    _, _, _, done, infos = env.step(torch.zeros(64, 12, device="cuda"))
    assert torch.equal(env.episode_length_buf[~done], before[~done] + 1)
    # log(): scalars and 0-dim tensors are rewritten in place
    ep_info = infos["episode"]
    for key in ep_info:
        if not isinstance(ep_info[key], torch.Tensor):
            ep_info[key] = torch.Tensor([ep_info[key]])
        if len(ep_info[key].shape) == 0:
            ep_info[key] = ep_info[key].unsqueeze(0)
        assert ep_info[key].shape == (1,)
    assert len(list(ep_info)) == 16


@pytest.mark.parametrize("task", ["go2", "go2_wtw", "go2_ee", "tron1_pf_ee", "tron1_sf"])
def test_checkpoint_resume_is_bit_exact(task, tmp_path):
    """save after k steps, keep going, restore into a FRESH env, replay the same actions: every output and every
    buffer identical to the uninterrupted run (curriculum levels, command ranges, episode sums, observation histories and the
    counter-keyed random streams all continue)."""
    import torch
    from hcr_genesis_lr_cl_amd.envs import make_env
    N = 96
    env, _ = make_env(task, N, "cuda:0")
    env.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    env.episode_length_buf = torch.randint(940, 1001, (N,), generator=g, device="cuda", dtype=torch.int32)
    env.common_step_counter = 960                      # the command-curriculum gate (step 1000) falls inside the replayed part
    acts = [torch.randn(N, env.num_actions, generator=g, device="cuda") * 1.5 for _ in range(70)]
    for a in acts[:25]:
        env.step(a)
    path = str(tmp_path / "ckpt.pt")
    env.save_checkpoint(path)
    ref = []
    for a in acts[25:]:
        out = env.step(a)
        ref.append([o.clone() for o in out[:-1] if torch.is_tensor(o)])
    final = {k: env._engine.buf.raw(k).clone() for k in env._engine.buf.keys()}
    ranges = dict(env.command_ranges)
    env2, _ = make_env(task, N, "cuda:0")
    env2.reset()
    env2.load_checkpoint(path)
    assert env2.common_step_counter == 985
    for a, want in zip(acts[25:], ref):
        out = env2.step(a)
        got = [o for o in out[:-1] if torch.is_tensor(o)]
        assert len(got) == len(want) and all(torch.equal(x, y) for x, y in zip(got, want))
    for k, v in final.items():
        assert torch.equal(env2._engine.buf.raw(k), v), k
    assert {k: list(v) for k, v in env2.command_ranges.items()} == {k: list(v) for k, v in ranges.items()}
    with pytest.raises(ValueError):
        make_env("go2" if task != "go2" else "go2_wtw", N, "cuda:0")[0].load_checkpoint(path)


# ---- BASELINE.json configs at their full sizes: size-independent properties (determinism, shard invariance) ----
FULL = [("go2", 4096, 4096, 0), ("go2_ee", 4096, 4096, 0), ("go2_wtw", 4096, 8192, 4096), ("tron1_pf_ee", 4096, 32768, 28672)]


@pytest.mark.parametrize("task,n_local,n_global,offset", FULL, ids=[f"{t}-{g}" for t, _, g, _ in FULL])
def test_baseline_config_full_size_properties(task, n_local, n_global, offset):
    """configs[1..4] of BASELINE.json at 4096 envs per GPU: 40 control steps of N(0,1) actions stay finite and inside the physical
    envelope; resets happen; a second construction reproduces the rollout bit for bit.  For the sharded configs (go2_wtw 8192 over 2
    GPUs, tron1_pf_ee 32768 over 8) the rank owning the LAST block of the global env range is run: its terrain columns, initial
    terrain levels and random streams are keyed on the global index."""
    import torch
    from hcr_genesis_lr_cl_amd.envs import make_env

    def run():
        env, cfg = make_env(task, n_local, "cuda:0", env_id_offset=offset, global_num_envs=n_global)
        env.reset()
        g = torch.Generator(device="cuda"); g.manual_seed(11)
        env.episode_length_buf = torch.randint(0, 1000, (n_local,), generator=g, device="cuda", dtype=torch.int32)
        n_reset, outs = 0, None
        for t in range(40):
            out = env.step(torch.randn(n_local, env.num_actions, generator=g, device="cuda"))
            n_reset += int(out[-2].sum())
        torch.cuda.synchronize()
        outs = [o.clone() for o in out[:-1] if torch.is_tensor(o)]
        s = env.simulator
        assert all(torch.isfinite(o.float()).all() for o in outs)
        rel = s.base_pos[:, 2] - s.env_origins[:, 2]              # height above the env's own tile centre (slopes / stairs: a few metres)
        assert float(rel.max()) < 6.0 and float(rel.min()) > -6.0, (float(rel.min()), float(rel.max()))
        assert float(s.dof_vel.abs().max()) < 70.0
        assert n_reset > n_local // 50
        if cfg.terrain.mesh_type == "heightfield":
            tt = s.terrain_types.long()
            gid = torch.arange(n_local, device="cuda") + offset
            assert torch.equal(tt, torch.div(gid, n_global / cfg.terrain.num_cols, rounding_mode="floor").long())
        return outs
    a, b = run(), run()
    assert len(a) == len(b) and all(torch.equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("task", ["go2", "go2_wtw", "go2_ee", "go2_ts"])
def test_constant_reward_set_instantiation_equals_the_general_one(task, monkeypatch):
    """A task whose reward set is its profile's default runs the instantiation that has the set as a compile-time constant (lg_quad.h RS,
    lg_host.hip); LG_REWARD_SET_CONST=0 sends the same task through the general instantiation.  Both step from the same state every step
    (the two are separate compilations of the same statements: floating-point contraction may differ in the last bits, and a rollout
    left alone would amplify that)."""
    import os
    import torch
    from hcr_genesis_lr_cl_amd.envs import make_env
    N, T = 256, 60
    e1, _ = make_env(task, N)
    e2, _ = make_env(task, N)
    e1.reset(); e2.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    e1.episode_length_buf[:] = torch.randint(0, 900, (N,), generator=g, device="cuda", dtype=torch.int32)
    e1.common_step_counter = e2.common_step_counter = 480
    n_reset = 0
    for t in range(T):
        for k in e1._engine.buf.keys():
            e2._engine.buf.raw(k).copy_(e1._engine.buf.raw(k))
        e2.common_step_counter = e1.common_step_counter
        act = torch.randn(N, e1.num_actions, generator=g, device="cuda")
        monkeypatch.delenv("LG_REWARD_SET_CONST", raising=False)
        e1.step(act)
        assert "lg_launch_quad_rs" in e1._engine.last_kernel(), e1._engine.last_kernel()
        monkeypatch.setenv("LG_REWARD_SET_CONST", "0")
        e2.step(act)
        assert "lg_launch_quad_rs" not in e2._engine.last_kernel() and "lg_launch_quad<4" in e2._engine.last_kernel(), e2._engine.last_kernel()
        torch.cuda.synchronize()
        assert torch.equal(e1.reset_buf, e2.reset_buf), t
        torch.testing.assert_close(e1.rew_buf, e2.rew_buf, rtol=2e-5, atol=2e-6)
        for k in ("episode_sums", "obs_buf", "dof_pos", "dof_vel", "base_pos", "commands", "feet_air_time"):
            a, b = e1._engine.buf[k], e2._engine.buf[k]
            torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-4, msg=lambda m, k=k, t=t: f"{k} @ {t}: {m}")
        n_reset += int(e1.reset_buf.sum())
    assert n_reset > 0
