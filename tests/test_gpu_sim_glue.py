"""GPU parity of the simulator glue: the HIP kernels and every public method of HipSimulator against golden
vectors produced by running the reference's own GenesisSimulator methods unbound (tests/golden/sim_glue_<task>.npz,
generator tests/golden/gen_sim_glue_fixtures.py):

  * the kernel's PD torque (genesis_simulator.py:630-642) -- both physics layouts, every task;
  * the kernel's read-backs (body-frame twist, projected gravity, euler; :35-60) against the math_utils-pinned
    oracle functions applied to the kernel's own final pose and world twist;
  * the out-of-terrain teleport (:612-628);
  * the RESET phase's domain-randomisation draws (:62-82, 665-739) with the reference's own uniforms injected;
  * HipSimulator.step / reset_idx / reset_dofs / reset_root_states / push_robots / update_terrain_curriculum /
    _compute_torques (the drop-in surface of simulator.py:21-103) with the reference's uniforms served to torch.rand.

Tolerances: 2e-6 relative + 2e-5 absolute on torques (f32, FMA contraction); 1e-5 on read-backs; exact on masks, ids
and copied values."""
import os

import numpy as np
import pytest

from oracle import mdp_oracle as mo

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TASKS = ["go2", "go2_wtw", "go2_ee", "tron1_pf_ee", "tron1_sf"]
LAYOUTS = [pytest.param(1, id="leg-per-lane"), pytest.param(2, id="component-per-lane")]


def load(task):
    return np.load(os.path.join(GOLD, f"sim_glue_{task}.npz"))


def make_sim(task, n, layout=0, decimation=None, inject_rand=False):
    from hcr_genesis_lr_cl_amd import config as cfgmod
    from hcr_genesis_lr_cl_amd.envs import TASKS as REG
    from hcr_genesis_lr_cl_amd.simulator import HipSimulator
    cfg = REG[task][1]()
    cfg.env.num_envs = n
    cfg.hip.sim_layout = layout
    if decimation is not None:
        cfg.control.decimation = decimation
    return HipSimulator(cfg, cfgmod.class_to_dict(cfg.sim), "cuda:0", True, inject_rand=inject_rand), cfg


def put(t, arr):
    import torch
    t.copy_(torch.from_numpy(np.ascontiguousarray(arr)).reshape(t.shape).to(t.dtype))


def cpu(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("task", TASKS)
def test_kernel_torque_is_the_reference_pd_law(task, layout):
    """One sub-step per launch (decimation 1): the reported torque is the PD law on the state handed in -- the reference's
    `_compute_torques` on the same numbers, unclipped (|tau| reaches hundreds of N m with actions x 40)."""
    import torch
    fx = load(task)
    T, dec, N, A = fx["torques_sub"].shape
    sim, cfg = make_sim(task, N, layout, decimation=1)
    b = sim._engine.buf
    for t in range(T):
        for k in range(dec):      # every (state, torque) pair the reference evaluated
            q = fx["dof_pos0"][t] if k == 0 else fx["dof_pos_sub"][t, k - 1]
            qd = fx["dof_vel0"][t] if k == 0 else fx["dof_vel_sub"][t, k - 1]
            put(b["dof_pos"], q); put(b["dof_vel"], qd)
            put(b["kp_scale"], fx["kp_scale"][t]); put(b["kd_scale"], fx["kd_scale"][t])
            b["base_pos"][:, 2] = 3.0                     # in the air: nothing but the actuators acts on the joints
            b["base_quat"].zero_(); b["base_quat"][:, 3] = 1.0
            b["base_lin_vel_w"].zero_(); b["base_ang_vel_w"].zero_()
            act = torch.from_numpy(fx["actions"][t]).cuda()
            np.testing.assert_allclose(cpu(sim._compute_torques(act)), fx["torques_sub"][t, k], rtol=2e-6, atol=2e-5)   # host method
            sim.step(act)
            torch.cuda.synchronize()
            np.testing.assert_allclose(cpu(sim.torques), fx["torques_sub"][t, k], rtol=2e-6, atol=2e-5, err_msg=f"step {t} sub {k}")
            np.testing.assert_array_equal(cpu(sim.last_dof_vel), qd)                 # genesis_simulator.py:21-24
    assert np.abs(fx["torques_sub"]).max() > 100.0


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("task", TASKS)
def test_kernel_read_backs_follow_math_utils(task, layout):
    """SIM-phase base_lin_vel / base_ang_vel / projected_gravity / base_euler == the reference's quat_rotate_inverse /
    get_euler_xyz (oracle functions pinned by math_utils_kat.npz and sim_glue_*.npz) on the kernel's own final state."""
    import torch
    from tests.util import random_sim_state, load_state_into_engine
    sim, cfg = make_sim(task, 256, layout)
    st, actions = random_sim_state(sim._model, cfg, 256, seed=11, z_offset=float(cfg.init_state.pos[2]) - 0.42 + 1.0)
    st.arr["base_pos"][:] += sim.env_origins.cpu().numpy()
    rng = np.random.default_rng(3)
    rpy = rng.uniform(-1.5, 1.5, (256, 3))                   # large tilts: all quadrants of roll / pitch / yaw
    from oracle import sim_glue_oracle as sg
    st.arr["base_quat"][:] = sg.quat_from_euler_xyz(rpy[:, 0].astype(np.float32), rpy[:, 1].astype(np.float32), rpy[:, 2].astype(np.float32))
    load_state_into_engine(sim._engine, st)
    sim.step(torch.from_numpy(actions).cuda())
    sim.post_physics_step()
    torch.cuda.synchronize()
    q, vw, ww = cpu(sim.base_quat), cpu(sim._base_lin_vel_w), cpu(sim._base_ang_vel_w)
    np.testing.assert_allclose(np.linalg.norm(q, axis=1), 1.0, atol=2e-6)
    g = np.tile(np.array([0, 0, -1], np.float32), (256, 1))
    np.testing.assert_allclose(cpu(sim.base_lin_vel), mo.quat_rotate_inverse(q, vw), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(cpu(sim.base_ang_vel), mo.quat_rotate_inverse(q, ww), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(cpu(sim.projected_gravity), mo.quat_rotate_inverse(q, g), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(cpu(sim.base_euler), mo.get_euler_xyz(q), rtol=1e-5, atol=2e-5)
    assert np.abs(cpu(sim.base_euler)).max() > 1.2


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("task", TASKS)
def test_out_of_terrain_teleport(task, layout):
    """Envs placed where the reference's fixture has them (a third outside the terrain box, one exactly on the bound): after
    one step the same envs as in the reference sit at base_init_pos + env_origins, the others have not moved in xy, and the
    twist is kept for all (set_pos(zero_velocity=False), genesis_simulator.py:626-627)."""
    import torch
    fx = load(task)
    T, N = fx["oob_ids"].shape
    sim, cfg = make_sim(task, N, layout)
    b = sim._engine.buf
    put(b["env_origins"], fx["env_origins"])
    for t in range(T):
        pos = fx["rb_base_pos_in"][t].copy()
        pos[:, 2] = 8.0                                       # free fall: xy untouched by the physics
        put(b["base_pos"], pos)
        b["base_quat"].zero_(); b["base_quat"][:, 3] = 1.0
        b["base_lin_vel_w"].zero_(); b["base_ang_vel_w"].zero_(); b["dof_vel"].zero_()
        put(b["dof_pos"], np.tile(fx["default_dof_pos"], (N, 1)))
        sim.step(torch.zeros(N, cfg.env.num_actions, device="cuda"))
        torch.cuda.synchronize()
        got = cpu(sim.base_pos)
        m = fx["oob_ids"][t].astype(bool)
        np.testing.assert_allclose(got[m], fx["base_pos"][t][m], rtol=0, atol=1e-6)          # == base_init_pos + env_origins
        np.testing.assert_allclose(got[~m, :2], pos[~m, :2], rtol=0, atol=1e-6)
        assert (got[~m, 2] < 8.0).all() and (got[~m, 2] > 7.99).all()
        vz = cpu(sim._base_lin_vel_w)[:, 2]
        np.testing.assert_allclose(vz, -9.81 * 0.02, rtol=1e-4)                               # velocity kept, teleported or not
    assert fx["oob_ids"].sum() > 10


def _inject_reset_draws(fx, slots, cfg, N):
    """The reference's uniforms of one reset_idx call, in the slot layout of include/lgsim.h (LgRandSlots)."""
    d = cfg.domain_rand
    ids = fx["reset_ids"]
    R = np.full((N, slots.n_slots), 0.5, np.float32)
    draws = [fx[f"reset_draw_{i}"] for i in range(len(fx["reset_draw_tags"]))]
    it = iter(draws)
    if d.randomize_friction:
        R[ids, slots.dr_friction] = next(it)[:, 0]
    if d.randomize_base_mass:
        R[ids, slots.dr_mass] = next(it)[:, 0]
    if d.randomize_com_displacement:
        for k in range(3):
            R[ids, slots.dr_com + k] = next(it)[:, 0]
    for k, flag in enumerate(("randomize_joint_armature", "randomize_joint_friction", "randomize_joint_damping")):
        if getattr(d, flag):
            R[ids, slots.dr_joint + k] = next(it)
    if d.randomize_pd_gain:
        A = cfg.env.num_actions
        R[np.ix_(ids, range(slots.dr_kp, slots.dr_kp + A))] = next(it)
        R[np.ix_(ids, range(slots.dr_kd, slots.dr_kd + A))] = next(it)
    return R


@pytest.mark.parametrize("task", TASKS)
def test_kernel_reset_phase_draws_domain_randomisation_like_the_reference(task):
    """RESET phase with the reference's own uniforms injected: friction / mass / CoM / joint / PD-gain buffers equal what
    GenesisSimulator.reset_idx left behind, for the reset envs and (untouched) for the others."""
    import torch
    from hcr_genesis_lr_cl_amd import abi
    from hcr_genesis_lr_cl_amd.envs import make_env
    fx = load(task)
    N = len(fx["reset_before_friction_values"])
    env, cfg = make_env(task, N, "cuda:0", inject_rand=True)
    eng = env._engine
    b = eng.buf
    names = ("friction_values", "added_base_mass", "base_com_bias", "kp_scale", "kd_scale", "joint_armature", "joint_friction", "joint_damping")
    for k in names:
        if k in b:
            put(b[k], fx["reset_before_" + k])
    put(b["rand_in"], _inject_reset_draws(fx, eng.task.slots, cfg, N))
    for k in ("last_dof_vel", "last_feet_vel", "last_base_lin_vel", "last_base_ang_vel"):
        b[k].fill_(7.0)
    env.reset_buf.fill_(False)
    env.reset_buf[torch.from_numpy(fx["reset_ids"]).cuda()] = True
    eng.step(abi.PHASE_RESET, None, 5)
    torch.cuda.synchronize()
    for k in names:
        if k in b:
            np.testing.assert_allclose(cpu(b[k]), fx["reset_after_" + k].reshape(cpu(b[k]).shape), rtol=2e-6, atol=2e-6, err_msg=k)
    for k in ("last_dof_vel", "last_feet_vel", "last_base_lin_vel", "last_base_ang_vel"):
        np.testing.assert_array_equal(cpu(b[k]).reshape(N, -1), fx["reset_after_" + k].reshape(N, -1), err_msg=k)


class _Served:
    """torch.rand / torch.randint_like stand-ins that hand out the reference's recorded draws, on the device."""

    def __init__(self, draws):
        self.it = iter(draws)

    def rand(self, *shape, **kw):
        import torch
        v = next(self.it)
        shp = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else tuple(shape)
        assert int(np.prod(shp)) == v.size, (shp, v.shape)
        return torch.from_numpy(v.reshape(shp)).to(kw.get("device", "cpu"))


@pytest.mark.parametrize("task", TASKS)
def test_hip_simulator_public_methods_match_the_reference(task, monkeypatch):
    """Every state-writing method of the drop-in surface (simulator.py:34-89) against what the reference's GenesisSimulator
    did with the same arguments and the same uniforms."""
    import torch
    fx = load(task)
    N = len(fx["reset_before_friction_values"])
    sim, cfg = make_sim(task, N)
    ids = torch.from_numpy(fx["reset_ids"]).cuda()
    A = cfg.env.num_actions
    # ---- reset_idx (genesis_simulator.py:62-82) ----
    names = ("friction_values", "added_base_mass", "base_com_bias", "kp_scale", "kd_scale", "joint_armature", "joint_friction", "joint_damping")
    for k in names:
        put(getattr(sim, "_" + k), fx["reset_before_" + k])
    for k in ("last_dof_vel", "last_feet_vel", "last_base_lin_vel", "last_base_ang_vel"):
        getattr(sim, "_" + k).fill_(7.0)
    served = _Served([fx[f"reset_draw_{i}"] for i in range(len(fx["reset_draw_tags"]))])
    monkeypatch.setattr(torch, "rand", served.rand)
    sim.reset_idx(ids)
    monkeypatch.undo()
    assert next(served.it, None) is None                                # same number of draws, same order, same shapes
    for k in names:
        np.testing.assert_allclose(cpu(getattr(sim, "_" + k)), fx["reset_after_" + k], rtol=2e-6, atol=2e-6, err_msg=k)
    for k in ("last_dof_vel", "last_feet_vel", "last_base_lin_vel", "last_base_ang_vel"):
        np.testing.assert_array_equal(cpu(getattr(sim, "_" + k)).reshape(N, -1), fx["reset_after_" + k].reshape(N, -1), err_msg=k)
    # the engine sees the new parameters through the very same tensors (no setter round trip)
    assert sim.dr_friction_values.data_ptr() == sim._engine.buf["friction_values"].data_ptr()
    # ---- reset_dofs (:84-102) ----
    sim._dof_vel.fill_(3.0)
    sim.reset_dofs(ids, torch.from_numpy(fx["rd_dof_pos_in"]).cuda(), torch.zeros(len(ids), A, device="cuda"))
    np.testing.assert_array_equal(cpu(sim.dof_pos)[fx["reset_ids"]], fx["rd_dof_pos_in"])
    np.testing.assert_array_equal(cpu(sim.dof_vel), fx["rd_dof_vel"])
    # ---- reset_root_states (:104-133) ----
    sim._base_quat.zero_(); sim._base_quat[:, 3] = 1.0
    sim._base_pos.zero_(); sim._base_lin_vel.zero_(); sim._base_ang_vel.zero_()
    sim.reset_root_states(ids, *(torch.from_numpy(fx[k]).cuda() for k in ("rr_base_pos_in", "rr_base_quat_in", "rr_lin_vel_in", "rr_ang_vel_in")))
    r = fx["reset_ids"]
    np.testing.assert_array_equal(cpu(sim.base_pos)[r], fx["rr_base_pos"][r])
    np.testing.assert_array_equal(cpu(sim.base_quat), fx["rr_base_quat"])
    np.testing.assert_allclose(cpu(sim.projected_gravity), fx["rr_projected_gravity"], rtol=2e-6, atol=2e-6)
    np.testing.assert_array_equal(cpu(sim.base_lin_vel)[r], fx["rr_base_lin_vel"][r])
    np.testing.assert_array_equal(cpu(sim.base_ang_vel)[r], fx["rr_base_ang_vel"][r])
    # engine twist = what Genesis' dofs 0-5 received: world-frame [lin, ang]
    np.testing.assert_array_equal(np.concatenate([cpu(sim._base_lin_vel_w)[r], cpu(sim._base_ang_vel_w)[r]], 1), fx["rr_engine_velocity"])
    # ---- push_robots (:150-158) ----
    put(sim._base_lin_vel_w, fx["push_base_lin_vel_w_in"])
    served = _Served([fx["push_u"]])
    monkeypatch.setattr(torch, "rand", served.rand)
    sim.push_robots()
    monkeypatch.undo()
    np.testing.assert_allclose(cpu(sim.dr_rand_push_vels), fx["push_rand_push_vels"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(cpu(sim._base_lin_vel_w), fx["push_engine_velocity"][:, :3], rtol=2e-6, atol=2e-6)
    # ---- update_terrain_curriculum (:140-148) ----
    if "tc_levels" in fx.files:
        put(sim._terrain_levels, fx["tc_levels_in"]); put(sim._terrain_types, fx["tc_types"])
        assert sim._terrain_origins.shape == fx["tc_origins"].shape
        sim._terrain_origins.copy_(torch.from_numpy(fx["tc_origins"]))
        monkeypatch.setattr(torch, "randint_like", lambda t, high: torch.from_numpy(fx["tc_randint"]).to(t.device, t.dtype))
        sim.update_terrain_curriculum(ids, torch.from_numpy(fx["tc_up"]).cuda(), torch.from_numpy(fx["tc_down"]).cuda())
        monkeypatch.undo()
        np.testing.assert_array_equal(cpu(sim.terrain_levels), fx["tc_levels"])
        np.testing.assert_array_equal(cpu(sim.env_origins)[r], fx["tc_env_origins"][r])
    # ---- the rest of the surface exists and is callable (simulator.py:28, 67, 90, 96) ----
    assert sim.post_physics_step() is None and sim.update_sensors() is None
    assert sim.draw_debug_vis() is None and sim.set_viewer_camera([0, 0, 1], [0, 0, 0]) is None
