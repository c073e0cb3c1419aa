"""Pins the physics oracle (oracle/lg_oracle.c) with invariants -- SURVEY.md 8(c): the reference
holds no golden vectors for the Genesis boundary ("parity unpinned"), so the restatement is held
to closed forms and cross-algorithm agreement instead."""
import copy

import numpy as np
import pytest

from oracle import oracle as orc
from hcr_genesis_lr_cl_amd import config as cfgmod


def _state(go2, n=1, h=0.42):
    return orc.HostState(go2["model"], n, cfgmod.default_dof_pos(go2["cfg"]), h)


def test_aba_matches_dense_inverse_dynamics_solve(go2):
    """ABA == (H via RNEA columns) \\ (tau - C): two algorithms sharing no recursion."""
    rng = np.random.default_rng(0)
    for _ in range(20):
        quat = rng.normal(size=4); quat /= np.linalg.norm(quat)
        pos, vw, ww = rng.normal(size=3), rng.normal(size=3), rng.normal(size=3) * 2
        q = cfgmod.default_dof_pos(go2["cfg"]) + rng.uniform(-0.5, 0.5, 12)
        qd, tau = rng.normal(size=12) * 3, rng.normal(size=12) * 10
        a = orc.forward_dynamics(go2["desc"], go2["opts"], pos, quat, vw, ww, q, qd, tau, method=0)
        b = orc.forward_dynamics(go2["desc"], go2["opts"], pos, quat, vw, ww, q, qd, tau, method=1)
        np.testing.assert_allclose(a[0], b[0], rtol=1e-6, atol=1e-5)
        np.testing.assert_allclose(a[1], b[1], rtol=1e-6, atol=1e-5)


def test_free_fall_closed_form(go2):
    """No contact, zero torque request impossible (PD) -> use kp=kd=0: base z follows the
    semi-implicit Euler closed form z_n = z0 + g dt^2 n(n+1)/2 and the joints stay put."""
    opts = copy.copy(go2["opts"])
    for k in range(12):
        opts.kp[k] = 0; opts.kd[k] = 0
    st = _state(go2, 2, h=50.0)
    st.arr["dof_vel"][:] = 0
    act = np.zeros((2, 12), np.float32)
    n_ctrl = 10
    for _ in range(n_ctrl):
        orc.sim_step(go2["desc"], opts, st, act, "f64")
    n = n_ctrl * 4
    z = 50.0 - 9.81 * 0.005 ** 2 * n * (n + 1) / 2
    assert abs(st.arr["base_pos"][0, 2] - z) < 2e-5
    assert abs(st.arr["base_lin_vel_w"][0, 2] + 9.81 * 0.005 * n) < 1e-5
    # a free-falling multibody under gravity alone has zero relative acceleration
    np.testing.assert_allclose(st.arr["dof_vel"], 0, atol=1e-5)
    np.testing.assert_allclose(st.arr["base_ang_vel_w"], 0, atol=1e-5)


def _momentum(model, st, e=0):
    """Linear + angular momentum about the world origin from the state (independent FK in numpy)."""
    a = model.arrays
    from hcr_genesis_lr_cl_amd.model_compiler import _sym
    def qmat(q):
        x, y, z, w = q
        return np.array([[1-2*(y*y+z*z), 2*(x*y-z*w), 2*(x*z+y*w)], [2*(x*y+z*w), 1-2*(x*x+z*z), 2*(y*z-x*w)],
                         [2*(x*z-y*w), 2*(y*z+x*w), 1-2*(x*x+y*y)]])
    def rod(u, t):
        K = np.array([[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]])
        return np.eye(3) + np.sin(t) * K + (1 - np.cos(t)) * K @ K
    s = {k: v[e].astype(np.float64) for k, v in st.arr.items()}
    R, P = [qmat(s["base_quat"])], [s["base_pos"]]
    W, V = [s["base_ang_vel_w"]], [s["base_lin_vel_w"]]
    nb = int(a["mass"].shape[0])
    for i in range(1, nb):
        p = int(a["parent"][i])
        Rpc = a["jrot"][i].reshape(3, 3) @ rod(a["axis"][i], s["dof_pos"][i - 1])
        R.append(R[p] @ Rpc)
        Pi = P[p] + R[p] @ a["jpos"][i]
        P.append(Pi)
        V.append(V[p] + np.cross(W[p], Pi - P[p]))
        W.append(W[p] + R[i] @ a["axis"][i] * s["dof_vel"][i - 1])
    lin, ang, mtot, com = np.zeros(3), np.zeros(3), 0.0, np.zeros(3)
    for i in range(nb):
        c = P[i] + R[i] @ a["com"][i]
        vc = V[i] + np.cross(W[i], c - P[i])
        Iw = R[i] @ _sym(a["inertia"][i]) @ R[i].T
        lin += a["mass"][i] * vc
        ang += Iw @ W[i] + a["mass"][i] * np.cross(c, vc)
        mtot += a["mass"][i]; com += a["mass"][i] * c
    return lin, ang, com / mtot


def test_zero_gravity_momentum_conserved_under_internal_torques(go2):
    """Internal torques cannot change total momentum: d/dt(p, h) = 0 instantaneously.  A first-order
    integrator keeps that to O(dt^2) per step, so the one-step drift must fall ~100x when dt
    falls 10x (and be tiny in absolute terms)."""
    rng = np.random.default_rng(1)
    drift = []
    for dt in (1e-3, 1e-4):
        opts = copy.copy(go2["opts"])
        opts.gravity_z, opts.dt, opts.decimation = 0.0, dt, 1
        st = _state(go2, 1, h=3.0)
        r2 = np.random.default_rng(2)
        st.arr["base_lin_vel_w"][:] = r2.normal(size=3)
        st.arr["base_ang_vel_w"][:] = r2.normal(size=3)
        st.arr["dof_vel"][:] = r2.normal(size=12)
        st.arr["base_pos"][:, :2] = 0
        act = r2.normal(size=(1, 12)).astype(np.float32)
        l0, a0, c0 = _momentum(go2["model"], st)
        orc.sim_step(go2["desc"], opts, st, act, "f64")
        l1, a1, c1 = _momentum(go2["model"], st)
        drift.append((np.linalg.norm(l1 - l0), np.linalg.norm(a1 - a0)))
    assert drift[0][0] < 5e-3 and drift[0][1] < 5e-3
    # f32 state I/O floors the smaller-dt drift at ~1e-6
    assert drift[1][0] < drift[0][0] / 30 + 3e-6 and drift[1][1] < drift[0][1] / 30 + 3e-6


def test_static_stance_force_balance_and_zmp(go2):
    """Standing on the plane with zero actions: sum Fz -> m g (16.087 kg), the robot
    neither sinks nor explodes, feet carry the load, and the ZMP lies inside the support polygon."""
    # the task's own gains (kp 20) let the unactuated stance sag onto the rear thighs, so the
    # balance check uses a stiff stance controller; everything else is the go2 configuration
    opts = copy.copy(go2["opts"])
    for k in range(12):
        opts.kp[k] = 80.0; opts.kd[k] = 1.0
    st = _state(go2, 1, h=0.335)
    st.arr["added_base_mass"][:] = 0
    act = np.zeros((1, 12), np.float32)
    fz_hist = []
    for k in range(150):
        orc.sim_step(go2["desc"], opts, st, act, "f32")
        fz_hist.append(st.arr["link_contact_forces"][0].reshape(17, 3)[:, 2].sum())
    mg = go2["model"].total_mass * 9.81
    assert abs(np.mean(fz_hist[-25:]) - mg) < 0.02 * mg
    f = st.arr["link_contact_forces"][0].reshape(17, 3)
    feet = [4, 8, 12, 16]
    assert f[feet, 2].sum() > 0.98 * f[:, 2].sum()
    assert 0.2 < st.arr["base_pos"][0, 2] < 0.4
    assert np.abs(st.arr["dof_vel"]).max() < 0.05 and np.abs(st.arr["base_lin_vel_w"]).max() < 0.02
    fp = st.arr["feet_pos"][0].reshape(4, 3)
    zmp = (fp[:, :2] * f[feet, 2:3]).sum(0) / f[feet, 2].sum()
    assert fp[:, 0].min() < zmp[0] < fp[:, 0].max() and fp[:, 1].min() < zmp[1] < fp[:, 1].max()
    # no creep: sticking feet do not slide (block-Jacobi sweeps resolve the foot-foot coupling)
    p0 = st.arr["base_pos"][0].copy()
    fp0 = st.arr["feet_pos"][0].copy()
    for k in range(50):
        orc.sim_step(go2["desc"], opts, st, act, "f32")
    assert np.abs(st.arr["feet_pos"][0] - fp0).max() < 3e-4
    assert np.linalg.norm(st.arr["base_pos"][0, :2] - p0[:2]) < 3e-3


def test_mirror_symmetry(go2):
    """Left/right mirrored state + mirrored actions give the mirrored next state."""
    rng = np.random.default_rng(3)
    st = _state(go2, 2, h=0.36)
    q = cfgmod.default_dof_pos(go2["cfg"]) + rng.uniform(-0.2, 0.2, 12).astype(np.float32)
    qd = rng.normal(size=12).astype(np.float32)
    act = rng.normal(size=12).astype(np.float32)
    # policy order FR FL RR RL: swap (FR,FL) and (RR,RL); hip (x-axis) angle flips sign
    perm = np.array([3, 4, 5, 0, 1, 2, 9, 10, 11, 6, 7, 8]); sgn = np.array([-1, 1, 1] * 4, np.float32)
    st.arr["dof_pos"][0], st.arr["dof_vel"][0] = q, qd
    st.arr["dof_pos"][1], st.arr["dof_vel"][1] = q[perm] * sgn, qd[perm] * sgn
    st.arr["base_lin_vel_w"][0] = [0.3, 0.2, -0.1]; st.arr["base_lin_vel_w"][1] = [0.3, -0.2, -0.1]
    st.arr["base_ang_vel_w"][0] = [0.1, 0.2, 0.3]; st.arr["base_ang_vel_w"][1] = [-0.1, 0.2, -0.3]
    a2 = np.stack([act, act[perm] * sgn])
    for _ in range(3):
        orc.sim_step(go2["desc"], go2["opts"], st, a2, "f64")
    # the URDF itself is not perfectly mirror symmetric (calf collision cylinders 0.012 vs 0.013 m,
    # FL vs others) -> loose tolerance once contacts are active
    np.testing.assert_allclose(st.arr["dof_pos"][1], st.arr["dof_pos"][0][perm] * sgn, atol=2e-3)
    np.testing.assert_allclose(st.arr["base_pos"][1] * [1, -1, 1], st.arr["base_pos"][0], atol=1e-3)


def test_joint_limit_and_effort_clamp(go2):
    """Huge action: reported torque is the unclipped PD value (genesis_simulator.py:630-642), the
    applied one is clamped to the URDF effort, and the limit stop keeps q near the range."""
    st = _state(go2, 1, h=20.0)
    act = np.full((1, 12), 100.0, np.float32)
    for _ in range(25):
        orc.sim_step(go2["desc"], go2["opts"], st, act, "f32")
    a = go2["model"].arrays
    assert st.arr["torques"].max() > 100.0
    assert np.all(st.arr["dof_pos"][0] < a["q_hi"] + 1.0) and np.all(st.arr["dof_pos"][0] > a["q_lo"] - 1.0)
    assert np.all(np.isfinite(st.arr["dof_pos"]))


# ---- four-joint legs and the sole foot (tron1_sf) -------------------------------------------------
@pytest.fixture(scope="module")
def sf():
    from hcr_genesis_lr_cl_amd.model_compiler import load_model
    from hcr_genesis_lr_cl_amd.config import TRON1SFCfg
    from hcr_genesis_lr_cl_amd import builders
    model, cfg = load_model("tron1_sf"), TRON1SFCfg()
    return dict(model=model, cfg=cfg, desc=builders.make_model_desc(model, cfg), opts=builders.make_sim_options(model, cfg))


def test_sole_foot_model_topology(sf):
    """SF_TRON1A/urdf/robot.urdf through the model compiler: 2 legs x 4 joints, 20.813 kg (sum of the URDF's link masses), the ankle
    bodies are the feet (tron1_sf_config.py:82-84), each with the sole centre as its foot sphere and the four sole corners 5.5 cm
    below the ankle axis (foot_height_offset, tron1_sf_config.py:100)."""
    m, a = sf["model"], sf["model"].arrays
    assert (m.n_legs, m.n_dof, m.joints_per_leg, m.n_links) == (2, 8, 4, 9) and abs(m.total_mass - 20.813) < 1e-6
    assert m.foot_names == ["ankle_L_Link", "ankle_R_Link"] and list(a["foot_link"][:2]) == [4, 8]
    for leg in range(2):
        fs = int(a["foot_sphere"][leg])
        assert a["sph_body"][fs] == 4 + 4 * leg and abs(a["sph_pos"][fs][2] - a["sph_r"][fs] + 0.055) < 1e-9
        corners = [i for i in range(a["n_spheres"]) if a["sph_body"][i] == 4 + 4 * leg and i != fs]
        assert len(corners) == 4 and np.allclose(a["sph_pos"][corners][:, 2], -0.055) and np.all(a["sph_r"][corners] == 0)
        assert np.ptp(a["sph_pos"][corners][:, 0]) == pytest.approx(0.2) and np.ptp(a["sph_pos"][corners][:, 1]) == pytest.approx(0.06)


def test_aba_matches_dense_solve_for_four_joint_legs(sf):
    rng = np.random.default_rng(3)
    for _ in range(12):
        quat = rng.normal(size=4); quat /= np.linalg.norm(quat)
        pos, vw, ww = rng.normal(size=3), rng.normal(size=3), rng.normal(size=3) * 2
        q, qd, tau = rng.uniform(-0.6, 0.6, 8), rng.normal(size=8) * 3, rng.normal(size=8) * 10
        a = orc.forward_dynamics(sf["desc"], sf["opts"], pos, quat, vw, ww, q, qd, tau, method=0)
        b = orc.forward_dynamics(sf["desc"], sf["opts"], pos, quat, vw, ww, q, qd, tau, method=1)
        np.testing.assert_allclose(a[0], b[0], rtol=1e-6, atol=2e-5)
        np.testing.assert_allclose(a[1], b[1], rtol=1e-6, atol=2e-5)


def test_sole_foot_statue_stands_on_its_soles(sf):
    """With stiff joints the robot is a statue whose centre of mass lies over the soles: it has to stay upright for 10 s, the feet
    carrying m g -- the sole corners transmit the ankle torque that a point foot could not (a point-foot statue topples).  Armature as
    the task always randomises it in (tron1_sf_config.py:157-158)."""
    desc, opts = copy.copy(sf["desc"]), copy.copy(sf["opts"])
    for k in range(8):
        desc.armature[k] = 0.12; opts.kp[k] = 400.0; opts.kd[k] = 10.0
    st = orc.HostState(sf["model"], 1, cfgmod.default_dof_pos(sf["cfg"]), 0.84)
    st.arr["added_base_mass"][:] = 0
    act = np.zeros((1, 8), np.float32)
    for k in range(500):
        orc.sim_step(desc, opts, st, act, "f32")
    f = st.arr["link_contact_forces"][0].reshape(9, 3)
    mg = sf["model"].total_mass * 9.81
    assert st.arr["projected_gravity"][0, 2] < -0.99 and 0.80 < st.arr["base_pos"][0, 2] < 0.84
    assert abs(f[[4, 8], 2].sum() - mg) < 0.02 * mg and abs(f[:, 2].sum() - mg) < 0.02 * mg
    assert np.abs(st.arr["dof_vel"]).max() < 0.05 and np.abs(st.arr["base_lin_vel_w"]).max() < 0.02
