"""The numpy restatement of the reference's simulator glue (oracle/sim_glue_oracle.py, oracle/mdp_oracle.py helpers)
against golden vectors produced by running the reference's own GenesisSimulator methods unbound
(tests/golden/gen_sim_glue_fixtures.py -> sim_glue_<task>.npz) and its own math_utils functions (math_utils_kat.npz).
CPU only.  Float tolerance 2e-6 relative / absolute (same f32 formulas, numpy vs torch association); masks and
integers exact."""
import os

import numpy as np
import pytest

from hcr_genesis_lr_cl_amd import config as cfgmod
from oracle import mdp_oracle as mo
from oracle import sim_glue_oracle as sg

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TASKS = {"go2": cfgmod.GO2Cfg, "go2_wtw": cfgmod.GO2WTWCfg, "go2_ee": cfgmod.GO2EECfg, "tron1_pf_ee": cfgmod.TRON1PFEECfg, "tron1_sf": cfgmod.TRON1SFCfg}
TOL = dict(rtol=2e-6, atol=2e-6)


def load(task):
    return np.load(os.path.join(GOLD, f"sim_glue_{task}.npz"))


def reset_draws(fx):
    return [fx[f"reset_draw_{i}"] for i in range(len(fx["reset_draw_tags"]))]


def test_math_utils_known_answers():
    """math_utils.py:34-112 through the reference's TorchScript functions."""
    k = np.load(os.path.join(GOLD, "math_utils_kat.npz"))
    q, v = k["q"], k["v"]
    np.testing.assert_allclose(mo.quat_rotate_inverse(q, v), k["quat_rotate_inverse"], **TOL)
    np.testing.assert_allclose(mo.quat_apply(q, v), k["quat_apply"], **TOL)
    np.testing.assert_allclose(mo.quat_apply_yaw(q, v[:, None, :])[:, 0], k["quat_apply_yaw"], **TOL)
    np.testing.assert_allclose(mo.get_euler_xyz(q), k["get_euler_xyz"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(mo.wrap_to_pi(k["angles"]), k["wrap_to_pi"], **TOL)
    # the TorchScript quirk (C fmod): negative angles stay negative, also below -pi
    assert (k["wrap_to_pi"][k["angles"] < -np.pi] < -np.pi).any()
    np.testing.assert_allclose(sg.quat_from_euler_xyz(k["rpy"][:, 0], k["rpy"][:, 1], k["rpy"][:, 2]), k["quat_from_euler_xyz"], **TOL)
    # round trip of the two: euler of the quaternion built from in-range angles
    rp = np.clip(k["rpy"], -1.4, 1.4)
    np.testing.assert_allclose(mo.get_euler_xyz(sg.quat_from_euler_xyz(rp[:, 0], rp[:, 1], rp[:, 2])), rp, atol=2e-5)


@pytest.mark.parametrize("task", list(TASKS))
def test_torque_law_every_substep(task):
    """genesis_simulator.py:20-33, 630-642: one PD evaluation per sub-step on the state the engine reported after the previous one;
    `_torques` keeps the last."""
    fx = load(task)
    cfg = TASKS[task]()
    kp, kd = cfgmod.pd_gains(cfg)
    np.testing.assert_allclose(kp, fx["p_gains"]); np.testing.assert_allclose(kd, fx["d_gains"])
    np.testing.assert_allclose(cfgmod.default_dof_pos(cfg), fx["default_dof_pos"])
    assert np.float32(cfg.control.action_scale) == fx["action_scale"]
    T, dec = fx["torques_sub"].shape[:2]
    assert dec == cfg.control.decimation
    for t in range(T):
        q = np.concatenate([fx["dof_pos0"][t][None], fx["dof_pos_sub"][t]], 0)
        qd = np.concatenate([fx["dof_vel0"][t][None], fx["dof_vel_sub"][t]], 0)
        for k in range(dec):
            tau = sg.compute_torques(fx["actions"][t], fx["action_scale"], fx["kp_scale"][t], fx["p_gains"], fx["default_dof_pos"], q[k],
                                     fx["kd_scale"][t], fx["d_gains"], qd[k])
            np.testing.assert_allclose(tau, fx["torques_sub"][t, k], rtol=2e-6, atol=2e-5)
        np.testing.assert_array_equal(fx["torques"][t], fx["torques_sub"][t, -1])
        # "last" snapshots are the values at the start of the control step (genesis_simulator.py:21-24)
        np.testing.assert_array_equal(fx["last_dof_vel"][t], fx["dof_vel0"][t])
    assert np.abs(fx["torques_sub"]).max() > 100.0        # unclipped: beyond every URDF effort limit


@pytest.mark.parametrize("task", list(TASKS))
def test_read_back_and_out_of_bound_teleport(task):
    """genesis_simulator.py:35-60, 612-628."""
    fx = load(task)
    cfg = TASKS[task]()
    (x0, x1), (y0, y1) = cfgmod.terrain_bounds(cfg)        # product host logic vs the reference's box
    np.testing.assert_allclose([x0, x1], fx["terrain_x_range"]); np.testing.assert_allclose([y0, y1], fx["terrain_y_range"])
    np.testing.assert_allclose(np.array(cfg.init_state.pos, np.float32), fx["base_init_pos"])
    from hcr_genesis_lr_cl_amd.model_compiler import load_model
    model = load_model(cfg.asset.name)
    feet = model.find_link_indices([n for n in model.link_names if cfg.asset.foot_name in n])
    T = fx["base_pos"].shape[0]
    seen = 0
    for t in range(T):
        pos, m = sg.teleport_out_of_bound(fx["rb_base_pos_in"][t], fx["terrain_x_range"], fx["terrain_y_range"], fx["base_init_pos"], fx["env_origins"])
        np.testing.assert_array_equal(m.astype(np.uint8), fx["oob_ids"][t])
        np.testing.assert_allclose(pos, fx["base_pos"][t], **TOL)
        seen += int(m.sum())
        rb = sg.read_back(fx["rb_base_quat_wxyz"][t], fx["rb_base_lin_vel_w"][t], fx["rb_base_ang_vel_w"][t])
        for k in ("base_quat", "base_lin_vel", "base_ang_vel", "projected_gravity"):
            np.testing.assert_allclose(rb[k], fx[k][t], **TOL, err_msg=k)
        np.testing.assert_allclose(rb["base_euler"], fx["base_euler"][t], rtol=1e-5, atol=1e-5)
        np.testing.assert_array_equal(fx["feet_pos"][t], fx["rb_links_pos"][t][:, feet])
        np.testing.assert_array_equal(fx["feet_vel"][t], fx["rb_links_vel"][t][:, feet])
        if cfg.asset.obtain_link_contact_states:
            ids = model.find_link_indices(cfg.asset.contact_state_link_names)
            np.testing.assert_array_equal(sg.contact_states(fx["rb_link_contact_forces"][t], ids), fx["link_contact_states"][t])
    assert seen > 10
    # the |sinp| >= 1 branch of get_euler_xyz was exercised
    q = fx["base_quat"][0][:2]
    assert (np.abs(2 * (q[:, 3] * q[:, 1] - q[:, 2] * q[:, 0])) >= 1).all()
    np.testing.assert_array_equal(np.abs(fx["base_euler"][0][:2, 1]), np.float32(np.pi / 2))


@pytest.mark.parametrize("task", list(TASKS))
def test_domain_randomisation_draw_order_and_formulas(task):
    """genesis_simulator.py:62-82, 665-739: the buffers after reset_idx, and what reaches the engine's setters."""
    fx = load(task)
    cfg = TASKS[task]()
    d = cfg.domain_rand
    ids = fx["reset_ids"]
    A = cfg.env.num_actions
    out = sg.randomize(d, A, reset_draws(fx))
    names = ("friction_values", "added_base_mass", "base_com_bias", "kp_scale", "kd_scale", "joint_armature", "joint_friction", "joint_damping")
    for k in names:
        after, before = fx["reset_after_" + k], fx["reset_before_" + k]
        exp = before.copy()
        if k in out:
            exp[ids] = out[k].reshape(exp[ids].shape)
        np.testing.assert_allclose(after, exp, **TOL, err_msg=k)
        others = np.setdiff1d(np.arange(len(after)), ids)
        np.testing.assert_array_equal(after[others], before[others])
    for k in ("_last_dof_vel", "_last_feet_vel", "_last_base_lin_vel", "_last_base_ang_vel"):
        a = fx["reset_after" + k]
        assert (a[ids] == 0).all() and (a[np.setdiff1d(np.arange(len(a)), ids)] == 7.0).all()
    # engine-side semantics: one friction ratio per env replicated over all links, mass / CoM shift on the base link only,
    # one armature / frictionloss / damping value per env replicated over the actuated dofs
    setters = [str(s) for s in fx["reset_setter_names"]]
    i = setters.index("set_friction_ratio")
    r = fx[f"reset_setter_{i}_ratios"]
    assert (r == r[:, :1]).all() and np.allclose(r[:, :1], fx["reset_after_friction_values"][ids])
    i = setters.index("set_mass_shift")
    assert int(fx[f"reset_setter_{i}_link"]) == 0 and np.allclose(fx[f"reset_setter_{i}_mass"], fx["reset_after_added_base_mass"][ids])
    i = setters.index("set_COM_shift")
    assert int(fx[f"reset_setter_{i}_link"]) == 0 and np.allclose(fx[f"reset_setter_{i}_com"][:, 0], fx["reset_after_base_com_bias"][ids])
    if d.randomize_joint_armature:
        for nm, key in (("set_dofs_armature", "_joint_armature"), ("set_dofs_frictionloss", "_joint_friction"), ("set_dofs_damping", "_joint_damping")):
            i = setters.index(nm)
            v = fx[f"reset_setter_{i}_v"]
            assert v.shape == (len(ids), A) and (v == v[:, :1]).all() and np.allclose(v[:, :1], fx["reset_after" + key][ids])
            assert list(fx[f"reset_setter_{i}_idx"]) == list(range(6, 6 + A))
    # the product's task constants carry the same ranges
    from hcr_genesis_lr_cl_amd import builders
    from hcr_genesis_lr_cl_amd.model_compiler import load_model
    T = builders.make_task_cfg(load_model(cfg.asset.name), cfg)
    assert bool(T.dr_friction_on) == bool(d.randomize_friction) and bool(T.dr_pd_on) == bool(d.randomize_pd_gain)
    assert bool(T.dr_joint_on) == bool(d.randomize_joint_armature)
    np.testing.assert_allclose([T.dr_friction_lo, T.dr_friction_lo + T.dr_friction_span], d.friction_range, atol=1e-6)
    np.testing.assert_allclose([T.dr_mass_lo, T.dr_mass_lo + T.dr_mass_span], d.added_mass_range, atol=1e-6)


@pytest.mark.parametrize("task", list(TASKS))
def test_state_write_semantics(task):
    """genesis_simulator.py:84-133, 150-158: reset_dofs zeroes every dof velocity, reset_root_states hands the engine a WORLD-frame
    twist on dofs 0-5 and stores the same numbers in the body-frame properties, push adds to the world xy velocity."""
    fx = load(task)
    ids = fx["reset_ids"]
    assert [str(s) for s in fx["rd_setter_names"]] == ["set_dofs_position", "zero_all_dofs_velocity"]
    np.testing.assert_array_equal(fx["rd_dof_pos"][ids], fx["rd_dof_pos_in"])
    assert (fx["rd_dof_vel"][ids] == 0).all()
    assert [str(s) for s in fx["rr_setter_names"]] == ["set_pos", "set_quat", "zero_all_dofs_velocity", "set_dofs_velocity"]
    np.testing.assert_array_equal(fx["rr_base_pos"][ids], fx["rr_base_pos_in"])
    np.testing.assert_array_equal(fx["rr_base_quat"][ids], fx["rr_base_quat_in"])
    np.testing.assert_array_equal(fx["rr_engine_quat_wxyz"], np.concatenate([fx["rr_base_quat_in"][:, 3:], fx["rr_base_quat_in"][:, :3]], 1))
    np.testing.assert_array_equal(fx["rr_engine_velocity"], np.concatenate([fx["rr_lin_vel_in"], fx["rr_ang_vel_in"]], 1))
    np.testing.assert_array_equal(fx["rr_base_lin_vel"][ids], fx["rr_lin_vel_in"])
    g = np.tile(np.array([0, 0, -1], np.float32), (len(fx["rr_base_quat"]), 1))
    np.testing.assert_allclose(mo.quat_rotate_inverse(fx["rr_base_quat"], g), fx["rr_projected_gravity"], **TOL)   # refreshed for ALL envs
    m = fx["push_max"]
    push = (m + m) * fx["push_u"] - m
    np.testing.assert_allclose(fx["push_rand_push_vels"][:, :2], push, **TOL)
    np.testing.assert_allclose(fx["push_engine_velocity"][:, :2], fx["push_base_lin_vel_w_in"][:, :2] + push, **TOL)
    np.testing.assert_array_equal(fx["push_engine_velocity"][:, 2], fx["push_base_lin_vel_w_in"][:, 2])
    if "tc_levels" in fx.files:
        lv, org = sg.terrain_curriculum(fx["tc_levels_in"], fx["tc_types"], fx["tc_origins"], ids, fx["tc_up"], fx["tc_down"],
                                        fx["tc_origins"].shape[0], fx["tc_randint"])
        np.testing.assert_array_equal(lv, fx["tc_levels"])
        np.testing.assert_array_equal(org, fx["tc_env_origins"][ids])
