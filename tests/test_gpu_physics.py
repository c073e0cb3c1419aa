"""GPU parity, physics phase: HIP SIM launch (through the C ABI) vs. the f64 CPU oracle on the
same seeded states.  Tolerances are stated per quantity: the kernel computes in f32 in a
different formulation (world-aligned axes about the base origin, leg-per-lane) from the oracle
(body coordinates, dense 6x6), so agreement is to f32 round-off amplified by the stiff contact
(k = 4e4 N/m: 1e-6 m of position noise is 0.04 N)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIM_OUT = ["base_pos", "base_quat", "base_lin_vel_w", "base_ang_vel_w", "dof_pos", "dof_vel", "torques",
           "link_contact_forces", "feet_pos", "feet_vel", "base_lin_vel", "base_ang_vel", "projected_gravity",
           "base_euler", "last_dof_vel", "last_feet_vel", "last_base_lin_vel", "last_base_ang_vel"]
TOL = dict(base_pos=2e-5, base_quat=2e-5, base_lin_vel_w=2e-3, base_ang_vel_w=1e-2, dof_pos=2e-4, dof_vel=3e-2,
           torques=5e-3, link_contact_forces=None, feet_pos=1e-4, feet_vel=2e-2, base_lin_vel=2e-3,
           base_ang_vel=1e-2, projected_gravity=2e-5, base_euler=5e-5, last_dof_vel=0, last_feet_vel=0,
           last_base_lin_vel=0, last_base_ang_vel=0)


LAYOUTS = [pytest.param(1, id="leg-per-lane"), pytest.param(2, id="component-per-lane")]


@pytest.fixture(scope="module", params=LAYOUTS)
def engine(go2, request):
    """Both physics kernels (LgSimOptions.sim_layout): one leg per lane, one vector component per lane."""
    import copy, torch
    from hcr_genesis_lr_cl_amd import builders
    from hcr_genesis_lr_cl_amd.engine import Engine
    task = builders.make_task_cfg(go2["model"], go2["cfg"])
    opts = copy.copy(go2["opts"])
    opts.sim_layout = request.param
    return Engine(go2["model"], go2["desc"], opts, task, 512, "cuda:0")


def _run_both(go2, engine, seed, steps=1, airborne_frac=0.3, z_offset=0.0):
    import torch
    from hcr_genesis_lr_cl_amd import abi
    from oracle import oracle as orc
    from tests.util import random_sim_state, load_state_into_engine, engine_arrays
    st, actions = random_sim_state(go2["model"], go2["cfg"], engine.n, seed, airborne_frac, z_offset)
    load_state_into_engine(engine, st)
    act = torch.from_numpy(actions).cuda()
    for _ in range(steps):
        engine.step(abi.PHASE_SIM, act, 0)
        orc.sim_step(go2["desc"], go2["opts"], st, actions, "f64", threads=8)
    return engine_arrays(engine, SIM_OUT), st


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_one_control_step_matches_oracle(go2, engine, seed):
    got, st = _run_both(go2, engine, seed)
    for k in SIM_OUT:
        ref = st.arr[k].reshape(engine.n, -1)
        assert np.all(np.isfinite(got[k])), k
        if k == "link_contact_forces":
            # forces: 1% + 0.3 N (last sub-step force of a stiff implicit contact)
            err = np.abs(got[k] - ref)
            assert np.all(err <= 0.3 + 0.01 * np.abs(ref)), (k, err.max())
        else:
            np.testing.assert_allclose(got[k], ref, atol=TOL[k], rtol=1e-4, err_msg=k)


def test_free_flight_matches_oracle_tightly(go2, engine):
    """No contacts at all: pure ABA + PD + integration, four sub-steps -> near round-off."""
    got, st = _run_both(go2, engine, 5, airborne_frac=1.1, z_offset=2.0)
    for k, tol in (("dof_pos", 2e-5), ("dof_vel", 2e-3), ("base_pos", 2e-6), ("base_ang_vel_w", 1e-3), ("base_lin_vel_w", 2e-4)):
        np.testing.assert_allclose(got[k], st.arr[k].reshape(engine.n, -1), atol=tol, rtol=2e-5, err_msg=k)
    assert np.abs(got["link_contact_forces"]).max() == 0.0


@pytest.mark.parametrize("layout", LAYOUTS)
def test_static_stance_on_gpu(go2, layout):
    """The kernel alone (no oracle): stiff stance settles with sum Fz = m g on the feet."""
    import copy, torch
    from hcr_genesis_lr_cl_amd import abi, builders, config as cfgmod
    from hcr_genesis_lr_cl_amd.engine import Engine
    opts = copy.copy(go2["opts"])
    opts.sim_layout = layout
    for k in range(12):
        opts.kp[k] = 80.0; opts.kd[k] = 1.0
    task = builders.make_task_cfg(go2["model"], go2["cfg"])
    eng = Engine(go2["model"], go2["desc"], opts, task, 64, "cuda:0")
    eng.buf["dof_pos"][:] = torch.tensor(cfgmod.default_dof_pos(go2["cfg"])).cuda()
    eng.buf["base_pos"][:, 2] = 0.335
    act = torch.zeros(64, 12, device="cuda")
    for _ in range(200):
        eng.step(abi.PHASE_SIM, act, 0)
    torch.cuda.synchronize()
    f = eng.buf["link_contact_forces"].cpu().numpy()
    mg = go2["model"].total_mass * 9.81
    assert np.allclose(f[:, :, 2].sum(1), mg, rtol=0.02)
    assert np.all(f[:, [4, 8, 12, 16], 2].sum(1) > 0.98 * f[:, :, 2].sum(1))
    assert np.abs(eng.buf["dof_vel"].cpu().numpy()).max() < 0.05


def test_extreme_actions_stay_finite(go2, engine):
    import torch
    from hcr_genesis_lr_cl_amd import abi
    from tests.util import random_sim_state, load_state_into_engine
    st, _ = random_sim_state(go2["model"], go2["cfg"], engine.n, 9)
    load_state_into_engine(engine, st)
    act = torch.full((engine.n, 12), 100.0, device="cuda")
    act[::2] *= -1
    for _ in range(50):
        engine.step(abi.PHASE_SIM, act, 0)
    torch.cuda.synchronize()
    for k in ("dof_pos", "dof_vel", "base_pos", "base_quat", "link_contact_forces"):
        assert torch.isfinite(engine.buf[k]).all(), k


# ---- the other robot / terrain combinations of the BASELINE configs -------------------------------
def _setup(robot, rough, layout):
    import torch
    from hcr_genesis_lr_cl_amd import builders
    from hcr_genesis_lr_cl_amd.config import GO2EECfg, TRON1PFEECfg, GO2Cfg, TRON1SFCfg
    from hcr_genesis_lr_cl_amd.engine import Engine
    from hcr_genesis_lr_cl_amd.model_compiler import load_model
    from hcr_genesis_lr_cl_amd.terrain import Terrain
    cfg = {"go2": GO2EECfg if rough else GO2Cfg, "tron1_pf": TRON1PFEECfg, "tron1_sf": TRON1SFCfg}[robot]()
    cfg.hip.sim_layout = layout
    if not rough:
        cfg.terrain.mesh_type, cfg.terrain.measure_heights, cfg.terrain.curriculum = "plane", False, False
        cfg.terrain.obtain_terrain_info_around_feet = False
    model = load_model(robot)
    terrain = None
    if rough:
        np.random.seed(3)
        terrain = Terrain(cfg.terrain)
    desc, opts, task = builders.make_model_desc(model, cfg), builders.make_sim_options(model, cfg, terrain), builders.make_task_cfg(model, cfg)
    eng = Engine(model, desc, opts, task, 512, "cuda:0")
    if rough:
        hx, hy = np.meshgrid(cfg.terrain.measured_points_x, cfg.terrain.measured_points_y, indexing="ij")
        eng.set_terrain(terrain.height_field_raw, terrain.env_origins, np.stack([hx.ravel(), hy.ravel()], 1).astype(np.float32))
        opts = builders.make_sim_options(model, cfg, terrain)
    return model, cfg, desc, opts, eng, terrain


@pytest.mark.parametrize("robot,rough,layout", [("tron1_pf", False, 1), ("tron1_pf", False, 2), ("go2", True, 1), ("go2", True, 2),
                                                ("tron1_pf", True, 1), ("tron1_pf", True, 2), ("tron1_sf", False, 1), ("tron1_sf", False, 2)])
def test_one_control_step_matches_oracle_other_configs(robot, rough, layout):
    """TRON1 exercises the 2-lanes-per-env instantiation, joint_rot/armature/damping tables; `rough` the
    heightfield contact (bilinear height + gradient normal) on stairs / slopes / obstacles; tron1_sf the four-joint chains and
    the sole contact."""
    import torch
    from hcr_genesis_lr_cl_amd import abi
    from oracle import oracle as orc
    from tests.util import random_sim_state, load_state_into_engine, engine_arrays
    model, cfg, desc, opts, eng, terrain = _setup(robot, rough, layout)
    st, actions = random_sim_state(model, cfg, eng.n, 4)
    if rough:   # scatter the robots over the tiles and drop them onto the local ground
        rng = np.random.default_rng(8)
        tiles = terrain.env_origins.reshape(-1, 3)
        pick = tiles[rng.integers(0, len(tiles), eng.n)]
        st.arr["base_pos"][:, :2] = pick[:, :2] + rng.uniform(-3, 3, (eng.n, 2))
        st.arr["env_origins"][:] = pick
        from oracle import mdp_oracle as mo
        h = mo.sample_heights(st.arr["base_pos"], np.tile([0, 0, 0, 1.0], (eng.n, 1)).astype(np.float32), np.zeros((1, 2), np.float32),
                              terrain.height_field_raw, cfg.terrain.border_size, cfg.terrain.horizontal_scale, cfg.terrain.vertical_scale)
        st.arr["base_pos"][:, 2] += h[:, 0]
    if robot in ("tron1_pf", "tron1_sf"):
        st.arr["joint_armature"] = np.random.default_rng(1).uniform(0.11, 0.13, (eng.n, 1)).astype(np.float32)
        st.arr["joint_friction"] = np.random.default_rng(2).uniform(0.0, 0.01, (eng.n, 1)).astype(np.float32)
        st.arr["joint_damping"] = np.random.default_rng(3).uniform(1.4, 1.45, (eng.n, 1)).astype(np.float32)
    load_state_into_engine(eng, st)
    eng.step(abi.PHASE_SIM, torch.from_numpy(actions).cuda(), 0)
    orc.sim_step(desc, opts, st, actions, "f64", threads=8, heightfield=None if terrain is None else terrain.height_field_raw)
    got = engine_arrays(eng, SIM_OUT)
    for k in SIM_OUT:
        ref = st.arr[k].reshape(eng.n, -1)
        assert np.all(np.isfinite(got[k])), k
        if k == "link_contact_forces":
            err = np.abs(got[k] - ref)
            # heightfield: a sphere within round-off of a cell edge can see the neighbouring facet -> allow 0.5 % outliers
            ok = err <= 0.3 + 0.01 * np.abs(ref)
            assert ok.mean() > (0.995 if rough else 1.0 - 1e-9), (k, (~ok).sum(), err.max())
        else:
            bad = ~np.isclose(got[k], ref, atol=TOL[k] * (3 if rough else 1), rtol=1e-4)
            assert bad.mean() <= (5e-3 if rough else 0.0), (k, bad.sum(), np.abs(got[k] - ref).max())


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("iters", [1, 3])
def test_other_sweep_counts_match_oracle(go2, layout, iters):
    """The contact / limit sweep count is an option (cfg.hip.contact_iters): both kernels follow the CPU restatement at
    1 and 3 sweeps as well as at the default 2."""
    import copy, torch
    from hcr_genesis_lr_cl_amd import abi, builders
    from hcr_genesis_lr_cl_amd.engine import Engine
    from oracle import oracle as orc
    from tests.util import random_sim_state, load_state_into_engine, engine_arrays
    opts = copy.copy(go2["opts"])
    opts.sim_layout, opts.contact_iters = layout, iters
    eng = Engine(go2["model"], go2["desc"], opts, builders.make_task_cfg(go2["model"], go2["cfg"]), 256, "cuda:0")
    st, actions = random_sim_state(go2["model"], go2["cfg"], 256, 13)
    load_state_into_engine(eng, st)
    eng.step(abi.PHASE_SIM, torch.from_numpy(actions).cuda(), 0)
    orc.sim_step(go2["desc"], opts, st, actions, "f64", threads=8)
    got = engine_arrays(eng, ["dof_pos", "dof_vel", "base_pos", "base_quat", "link_contact_forces"])
    for k in ("dof_pos", "dof_vel", "base_pos", "base_quat"):
        np.testing.assert_allclose(got[k], st.arr[k].reshape(256, -1), atol=TOL[k], rtol=1e-4, err_msg=k)
    err = np.abs(got["link_contact_forces"] - st.arr["link_contact_forces"].reshape(256, -1))
    assert np.all(err <= 0.3 + 0.01 * np.abs(st.arr["link_contact_forces"].reshape(256, -1))), err.max()


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("every", [2, 4])
def test_reused_contact_w_matches_oracle(go2, layout, every):
    """LgSimOptions.contact_w_every: the feet's operational-space matrices are recomputed on sub-steps 0, k, 2k ... only (every foot,
    in contact or not) and reused in between -- both kernels follow the CPU restatement's statement of that rule, two control steps."""
    import copy, torch
    from hcr_genesis_lr_cl_amd import abi, builders
    from hcr_genesis_lr_cl_amd.engine import Engine
    from oracle import oracle as orc
    from tests.util import random_sim_state, load_state_into_engine, engine_arrays
    opts = copy.copy(go2["opts"])
    opts.sim_layout, opts.contact_w_every = layout, every
    eng = Engine(go2["model"], go2["desc"], opts, builders.make_task_cfg(go2["model"], go2["cfg"]), 256, "cuda:0")
    st, actions = random_sim_state(go2["model"], go2["cfg"], 256, 17)
    load_state_into_engine(eng, st)
    ref1 = st.copy()
    for _ in range(2):
        eng.step(abi.PHASE_SIM, torch.from_numpy(actions).cuda(), 0)
        orc.sim_step(go2["desc"], opts, st, actions, "f64", threads=8)
    got = engine_arrays(eng, ["dof_pos", "dof_vel", "base_pos", "base_quat", "link_contact_forces"])
    bad = 0
    for k in ("dof_pos", "dof_vel", "base_pos", "base_quat"):
        ref = st.arr[k].reshape(256, -1)
        bad = max(bad, int((np.abs(got[k] - ref) > 2 * TOL[k] + 1e-4 * np.abs(ref)).any(axis=1).sum()))
    assert bad <= 2, bad      # two steps through stiff contact: an env on a branch edge of the contact law may differ
    # and the option does something: the default rule gives a (slightly) different trajectory
    o1 = copy.copy(opts); o1.contact_w_every = 1
    for _ in range(2):
        orc.sim_step(go2["desc"], o1, ref1, actions, "f64", threads=8)
    assert np.abs(ref1.arr["dof_vel"] - st.arr["dof_vel"]).max() > 1e-4


def test_nonfinite_state_is_counted_and_ends_the_episode():
    """the non-finite guard re-seats the robot, and it is never silent -- LgBuffers.nonfinite_count goes up and the env
    terminates in the same step (reset_buf = 1, not a time-out), in the fused launch and in the split launches."""
    import torch
    from hcr_genesis_lr_cl_amd.envs import make_env
    for task in ("go2", "tron1_pf_ee", "go2_cat"):
        env, cfg = make_env(task, 64)
        env.reset()
        g = torch.Generator(device="cuda"); g.manual_seed(3)
        for _ in range(3):
            env.step(torch.randn(64, env.num_actions, generator=g, device="cuda"))
        eng = env._engine
        assert eng.nonfinite_count() == 0
        eng.buf["dof_vel"][5, 1] = float("nan")
        eng.buf["base_pos"][9, 2] = float("inf")
        out = env.step(torch.randn(64, env.num_actions, generator=g, device="cuda"))
        done, extras = out[-2], out[-1]
        assert eng.nonfinite_count() == 2, task
        assert bool(done[5]) and bool(done[9]) and not bool(extras["time_outs"][5]) and not bool(extras["time_outs"][9]), task
        for k in ("dof_pos", "dof_vel", "base_pos", "base_quat", "obs_buf", "rew_buf"):
            assert bool(torch.isfinite(eng.buf[k]).all()), (task, k)
        assert int(eng.buf["fail_buf"][5]) == 0 and int(eng.buf["episode_length_buf"][5]) == 0     # the reset went through
        env.step(torch.randn(64, env.num_actions, generator=g, device="cuda"))
        assert eng.nonfinite_count() == 2, task


@pytest.mark.parametrize("task", ["go2_ee", "tron1_pf_ee", "tron1_sf"])
def test_no_robot_is_thrown_by_the_contact_solver(task):
    """Regression (round 2): with friction ratios up to 1.7 the sliding branch of the foot contact could jam -- friction coupling
    cancelling the normal compliance, f_n = rn / (1 + kappa a_eff) with a_eff -> 0 -- and throw a robot (39 kN on one foot, 47 m/s,
    110 m high; 0.025 % of env-steps of a random go2_ee rollout, reproduced by the f64 CPU oracle on the same state).  The coupling
    is now limited to 3/4 of A_nn in both kernels and the oracle.  Same seeds as the rollout that showed it at step 3."""
    import torch
    from hcr_genesis_lr_cl_amd.envs import make_env
    n = 4096
    env, cfg = make_env(task, n)
    env.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    env.episode_length_buf = torch.randint(0, 1000, (n,), generator=g, device="cuda", dtype=torch.int32)
    s = env.simulator
    worst_v = worst_h = worst_f = 0.0
    for t in range(120):
        env.step(torch.randn(n, env.num_actions, generator=g, device="cuda"))
        worst_v = max(worst_v, float(s._base_lin_vel_w[:, 2].abs().max()))
        worst_h = max(worst_h, float((s.base_pos[:, 2] - s.env_origins[:, 2]).max()))
        worst_f = max(worst_f, float(s.link_contact_forces.norm(dim=-1).max()))
    assert worst_v < 6.0 and worst_h < 3.0 and worst_f < 6000.0, (worst_v, worst_h, worst_f)


# ---- first principles on the kernel itself (no oracle in the loop) ------------------------------------------------------------
def _engine_with(go2, layout, n, **opt_overrides):
    import copy
    from hcr_genesis_lr_cl_amd import builders
    from hcr_genesis_lr_cl_amd.engine import Engine
    opts = copy.copy(go2["opts"])
    opts.sim_layout = layout
    for k, v in opt_overrides.items():
        setattr(opts, k, v)
    task = builders.make_task_cfg(go2["model"], go2["cfg"])
    return Engine(go2["model"], go2["desc"], opts, task, n, "cuda:0")


def _read_back(engine, st, names=("base_pos", "base_quat", "base_lin_vel_w", "base_ang_vel_w", "dof_pos", "dof_vel")):
    got = engine_arrays_like(engine, names)
    for k in names:
        st.arr[k][:] = got[k].reshape(st.arr[k].shape)


def engine_arrays_like(engine, names):
    from tests.util import engine_arrays
    return engine_arrays(engine, list(names))


@pytest.mark.parametrize("layout", LAYOUTS)
def test_zero_gravity_momentum_is_conserved_by_the_kernel(go2, layout):
    """Internal (actuator, joint-limit, damping) torques cannot change the total momentum.  64 floating robots in zero gravity,
    random twists / joint rates / actions, 48 ms of simulated time: the linear and angular momentum (computed from the kernel's state
    by an independent numpy FK, tests/test_oracle_physics._momentum) drift only by the O(dt) global error of the first-order
    integrator -- small at dt = 2 ms and at least 2.5x smaller at dt = 0.5 ms (a force that did not belong there would not scale)."""
    import torch
    from hcr_genesis_lr_cl_amd import abi
    from oracle import oracle as orc
    from hcr_genesis_lr_cl_amd import config as cfgmod
    from tests.test_oracle_physics import _momentum
    from tests.util import load_state_into_engine
    n = 64
    drift = []
    for dt, ctrl_steps in ((2e-3, 6), (5e-4, 24)):
        eng = _engine_with(go2, layout, n, gravity_z=0.0, dt=dt)
        st = orc.HostState(go2["model"], n, cfgmod.default_dof_pos(go2["cfg"]), 3.0)
        rng = np.random.default_rng(7)
        st.arr["base_lin_vel_w"][:] = rng.normal(size=(n, 3))
        st.arr["base_ang_vel_w"][:] = rng.normal(size=(n, 3))
        st.arr["dof_vel"][:] = rng.normal(size=(n, 12))
        st.arr["dof_pos"][:] += rng.uniform(-0.2, 0.2, (n, 12)).astype(np.float32)
        st.arr["base_pos"][:, :2] = 0
        load_state_into_engine(eng, st)
        m0 = [_momentum(go2["model"], st, e) for e in range(n)]
        act = torch.from_numpy(rng.normal(size=(n, 12)).astype(np.float32)).cuda()
        for _ in range(ctrl_steps):
            eng.step(abi.PHASE_SIM, act, 0)
        _read_back(eng, st)
        assert np.abs(eng.buf["link_contact_forces"].cpu().numpy()).max() == 0.0
        dl, da = [], []
        for e in range(n):
            l1, a1, _ = _momentum(go2["model"], st, e)
            l0, a0, _ = m0[e]
            dl.append(np.linalg.norm(l1 - l0) / max(np.linalg.norm(l0), 1.0))
            da.append(np.linalg.norm(a1 - a0) / max(np.linalg.norm(a0), 1.0))
        drift.append((float(np.mean(dl)), float(np.mean(da)), float(np.max(dl)), float(np.max(da))))
    print("momentum drift (mean lin, mean ang, max lin, max ang) at dt = 2 ms / 0.5 ms:", drift)
    assert drift[0][2] < 2e-2 and drift[0][3] < 2e-2, drift          # measured: 7e-3 / 5e-3 at worst, means 1.8e-3 / 1.0e-3
    assert drift[1][0] < drift[0][0] / 2.5 + 2e-6 and drift[1][1] < drift[0][1] / 2.5 + 2e-6, drift   # measured: /4.01 and /4.01


@pytest.mark.parametrize("layout", LAYOUTS)
def test_mirror_symmetry_of_the_kernel(go2, layout):
    """Left/right mirrored state + mirrored actions give the mirrored next state (three control steps with contact), in both
    layouts.  The go2 URDF is not perfectly mirror symmetric (calf collision cylinders 0.012 vs 0.013 m), hence 2e-3."""
    import torch
    from hcr_genesis_lr_cl_amd import abi
    from oracle import oracle as orc
    from hcr_genesis_lr_cl_amd import config as cfgmod
    from tests.util import load_state_into_engine
    n = 32
    eng = _engine_with(go2, layout, n)
    st = orc.HostState(go2["model"], n, cfgmod.default_dof_pos(go2["cfg"]), 0.36)
    rng = np.random.default_rng(3)
    perm = np.array([3, 4, 5, 0, 1, 2, 9, 10, 11, 6, 7, 8]); sgn = np.array([-1, 1, 1] * 4, np.float32)
    acts = np.zeros((n, 12), np.float32)
    for e in range(0, n, 2):
        q = cfgmod.default_dof_pos(go2["cfg"]) + rng.uniform(-0.2, 0.2, 12).astype(np.float32)
        qd = rng.normal(size=12).astype(np.float32)
        a = rng.normal(size=12).astype(np.float32)
        v, w = rng.normal(size=3) * 0.3, rng.normal(size=3) * 0.3
        st.arr["dof_pos"][e], st.arr["dof_vel"][e] = q, qd
        st.arr["dof_pos"][e + 1], st.arr["dof_vel"][e + 1] = q[perm] * sgn, qd[perm] * sgn
        st.arr["base_lin_vel_w"][e] = v; st.arr["base_lin_vel_w"][e + 1] = v * [1, -1, 1]
        st.arr["base_ang_vel_w"][e] = w; st.arr["base_ang_vel_w"][e + 1] = w * [-1, 1, -1]
        acts[e], acts[e + 1] = a, a[perm] * sgn
    st.arr["base_pos"][:, :2] = 0
    load_state_into_engine(eng, st)
    act = torch.from_numpy(acts).cuda()
    for _ in range(3):
        eng.step(abi.PHASE_SIM, act, 0)
    got = engine_arrays_like(eng, ("dof_pos", "base_pos", "base_lin_vel_w"))
    np.testing.assert_allclose(got["dof_pos"][1::2], got["dof_pos"][0::2][:, perm] * sgn, atol=2e-3)
    np.testing.assert_allclose(got["base_pos"][1::2] * [1, -1, 1], got["base_pos"][0::2], atol=1e-3)
    np.testing.assert_allclose(got["base_lin_vel_w"][1::2] * [1, -1, 1], got["base_lin_vel_w"][0::2], atol=2e-2)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_free_fall_closed_form_on_the_kernel(go2, layout):
    """No contact, kp = kd = 0: the base follows the semi-implicit Euler closed form z_n = z0 - g dt^2 n (n + 1) / 2 and a
    multibody falling under gravity alone has no relative acceleration (joints and attitude stay put)."""
    import torch
    from hcr_genesis_lr_cl_amd import abi
    from oracle import oracle as orc
    from hcr_genesis_lr_cl_amd import config as cfgmod
    from tests.util import load_state_into_engine
    n = 8
    eng = _engine_with(go2, layout, n)
    eng.buf["kp_scale"].zero_(); eng.buf["kd_scale"].zero_()          # PD gains off through their per-env multipliers
    st = orc.HostState(go2["model"], n, cfgmod.default_dof_pos(go2["cfg"]), 50.0)
    st.arr["dof_vel"][:] = 0
    st.arr["kp_scale"][:] = 0; st.arr["kd_scale"][:] = 0
    load_state_into_engine(eng, st)
    act = torch.zeros(n, 12, device="cuda")
    n_ctrl = 10
    for _ in range(n_ctrl):
        eng.step(abi.PHASE_SIM, act, 0)
    got = engine_arrays_like(eng, ("base_pos", "base_lin_vel_w", "dof_vel", "base_ang_vel_w", "base_quat"))
    k = n_ctrl * 4
    z = 50.0 - 9.81 * 0.005 ** 2 * k * (k + 1) / 2
    np.testing.assert_allclose(got["base_pos"][:, 2], z, atol=5e-5)            # f32 at z = 50: 4e-6 per ulp
    np.testing.assert_allclose(got["base_lin_vel_w"][:, 2], -9.81 * 0.005 * k, atol=2e-5)
    np.testing.assert_allclose(got["dof_vel"], 0, atol=2e-4)
    np.testing.assert_allclose(got["base_ang_vel_w"], 0, atol=2e-4)


@pytest.mark.parametrize("robot,layout", [("tron1_pf", 1), ("tron1_pf", 2), ("tron1_sf", 1), ("tron1_sf", 2)])
def test_zero_gravity_momentum_other_chains(robot, layout):
    """The same conservation check for the biped (two legs per env: eight envs per wave in the component layout) and for the
    four-joint legs of TRON1-SF (fourth joint in the quad's spare lane)."""
    import copy, torch
    from hcr_genesis_lr_cl_amd import abi, builders, config as cfgmod
    from hcr_genesis_lr_cl_amd.engine import Engine
    from hcr_genesis_lr_cl_amd.model_compiler import load_model
    from oracle import oracle as orc
    from tests.test_oracle_physics import _momentum
    from tests.util import load_state_into_engine
    cfg = {"tron1_pf": cfgmod.TRON1PFCfg, "tron1_sf": cfgmod.TRON1SFCfg}[robot]()
    model = load_model(cfg.asset.name)
    A, n = model.n_dof, 48
    desc, task = builders.make_model_desc(model, cfg), builders.make_task_cfg(model, cfg)
    drift = []
    for dt, ctrl_steps in ((2e-3, 6), (5e-4, 24)):
        opts = builders.make_sim_options(model, cfg)
        opts.sim_layout, opts.gravity_z, opts.dt = layout, 0.0, dt
        eng = Engine(model, desc, opts, task, n, "cuda:0")
        st = orc.HostState(model, n, cfgmod.default_dof_pos(cfg), 3.0)
        rng = np.random.default_rng(9)
        st.arr["base_lin_vel_w"][:] = rng.normal(size=(n, 3))
        st.arr["base_ang_vel_w"][:] = rng.normal(size=(n, 3))
        st.arr["dof_vel"][:] = rng.normal(size=(n, A))
        st.arr["dof_pos"][:] += rng.uniform(-0.15, 0.15, (n, A)).astype(np.float32)
        st.arr["base_pos"][:, :2] = 0
        load_state_into_engine(eng, st)
        m0 = [_momentum(model, st, e) for e in range(n)]
        act = torch.from_numpy(rng.normal(size=(n, A)).astype(np.float32)).cuda()
        for _ in range(ctrl_steps):
            eng.step(abi.PHASE_SIM, act, 0)
        _read_back(eng, st)
        assert np.abs(eng.buf["link_contact_forces"].cpu().numpy()).max() == 0.0
        dl, da = [], []
        for e in range(n):
            l1, a1, _ = _momentum(model, st, e)
            l0, a0, _ = m0[e]
            dl.append(np.linalg.norm(l1 - l0) / max(np.linalg.norm(l0), 1.0))
            da.append(np.linalg.norm(a1 - a0) / max(np.linalg.norm(a0), 1.0))
        drift.append((float(np.mean(dl)), float(np.mean(da)), float(np.max(dl)), float(np.max(da))))
    print(robot, layout, "momentum drift (mean lin, mean ang, max lin, max ang) at dt = 2 ms / 0.5 ms:", drift)
    assert drift[0][0] < 6e-3 and drift[0][1] < 6e-3 and drift[0][2] < 8e-2 and drift[0][3] < 8e-2, drift   # measured (SF): 3.0e-3 / 1.2e-3 mean, 3.1e-2 worst
    assert drift[1][0] < drift[0][0] / 2.5 + 2e-6 and drift[1][1] < drift[0][1] / 2.5 + 2e-6, drift
