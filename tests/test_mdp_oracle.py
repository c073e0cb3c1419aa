"""Pins the numpy MDP oracle (oracle/mdp_oracle.py) against golden vectors produced by running
the reference's own GO2 task class (tests/golden/gen_mdp_fixtures.py).  Tolerances: float sums of
12 squares etc. may associate differently than torch -> rtol 2e-6 / atol 2e-6 on floats;
integers, booleans and counters are exact."""
import os

import numpy as np
import pytest

from hcr_genesis_lr_cl_amd import abi, builders
from hcr_genesis_lr_cl_amd.config import GO2Cfg
from hcr_genesis_lr_cl_amd.model_compiler import load_model
from oracle.mdp_oracle import MdpOracle

GOLD = os.path.join(os.path.dirname(__file__), "golden", "go2_mdp.npz")


GOLD_WTW = os.path.join(os.path.dirname(__file__), "golden", "go2_wtw_mdp.npz")


def replay(make_stepper, check, gold=GOLD):
    """Drive an MDP implementation through the fixture; `make_stepper(fx, N)` returns an object
    with .step(t, sim_in, actions, R, counter, override) -> dict of outputs."""
    fx = np.load(gold)
    T, N = fx["obs"].shape[:2]
    st = make_stepper(fx, N)
    for t in range(T):
        sim_in = {k[len("script_"):]: fx[k][t].copy() for k in fx.files if k.startswith("script_")}
        sim_in["last_dof_vel"] = fx["last_dof_vel_in"][t].copy()
        sim_in["last_feet_vel"] = fx["last_feet_vel_in"][t].copy()
        out = st.step(t, sim_in, fx["actions_in"][t], fx["rand"][t], int(fx["counter"][t]), float(fx["esum_override"][t]))
        check(t, fx, out)


class OracleStepper:
    def __init__(self, fx, N):
        model, cfg = load_model("go2"), GO2Cfg()
        task = builders.make_task_cfg(model, cfg)
        self.o = MdpOracle(model, cfg, task, N, fx["init_env_origins"])
        self.o.episode_length_buf[:] = fx["init_episode_length_buf"]
        self.o.commands[:] = fx["init_commands"]
        self.names = [str(n) for n in fx["reward_names"]]

    def step(self, t, sim, actions, R, counter, override):
        o = self.o
        if override:
            o.episode_sums[abi.REWARD_ID["tracking_lin_vel"]][:] = override
            o.episode_length_buf[:4] = 1000
        o.step(sim, actions, R, counter)
        return dict(obs=o.obs_buf, rew=o.rew_buf, reset=o.reset_buf, time_out=o.time_out_buf, commands=o.commands,
                    ep_len=o.episode_length_buf, fail_buf=o.fail_buf, feet_air_time=o.feet_air_time,
                    last_contacts=o.last_contacts,
                    episode_sums=np.stack([o.episode_sums[abi.REWARD_ID[n]] for n in self.names]),
                    act_hist=np.stack([o.actions, o.last_actions, o.llast_actions]),
                    sim_dof_pos=sim["dof_pos"], sim_dof_vel=sim["dof_vel"], sim_base_pos=sim["base_pos"],
                    sim_base_quat=sim["base_quat"], sim_base_lin_vel_w=sim["base_lin_vel_w"],
                    sim_projected_gravity=o.projected_gravity, sim_base_lin_vel=o.base_lin_vel,
                    dr=np.concatenate([o.friction_values, o.added_base_mass, o.base_com_bias, o.rand_push_vels[:, :2]], 1),
                    cmd_range_x=o.command_ranges[:2])


EXACT = ("reset", "time_out", "ep_len", "fail_buf", "last_contacts")
FLOAT = ("obs", "rew", "commands", "feet_air_time", "episode_sums", "act_hist", "sim_dof_pos", "sim_dof_vel",
         "sim_base_pos", "sim_base_quat", "sim_base_lin_vel_w", "sim_projected_gravity", "sim_base_lin_vel", "cmd_range_x")


def check_against_fixture(t, fx, out, rtol=2e-6, atol=2e-6):
    for k in EXACT:
        np.testing.assert_array_equal(np.asarray(out[k]).astype(np.int64), fx[k][t].astype(np.int64), err_msg=f"{k} @ step {t}")
    for k in FLOAT:
        np.testing.assert_allclose(out[k], fx[k][t], rtol=rtol, atol=atol, err_msg=f"{k} @ step {t}")
    # DR values: friction/mass/com from the fake simulator's draws; added mass starts at 1 in the
    # reference's buffer (genesis_simulator.py:648) but 0 here until the first reset of an env
    dr_ref, dr = fx["dr"][t], np.asarray(out["dr"])
    touched = np.abs(dr_ref[:, 1] - 1.0) > 0
    np.testing.assert_allclose(dr[touched], dr_ref[touched], rtol=rtol, atol=atol, err_msg=f"dr @ step {t}")


def test_fixture_exercises_the_branches():
    fx = np.load(GOLD)
    assert fx["reset"].sum() >= 8 and fx["time_out"].sum() >= 3
    assert (fx["reset"].astype(bool) & ~fx["time_out"].astype(bool)).sum() >= 2      # failure terminations
    assert (fx["counter"] % 750 == 0).any() and (fx["counter"] % 1000 == 0).any()   # push + curriculum steps
    assert fx["cmd_range_x"][-1][1] == 1.0 and fx["cmd_range_x"][0][1] == 0.5        # curriculum fired
    assert np.abs(fx["actions_in"]).max() > 100                                        # action clip exercised


def test_mdp_oracle_reproduces_reference_go2():
    replay(OracleStepper, check_against_fixture)


# ------------------------------- go2_wtw ------------------------------------------------------
class WtwOracleStepper:
    def __init__(self, fx, N):
        from hcr_genesis_lr_cl_amd.config import GO2WTWCfg
        model, cfg = load_model("go2"), GO2WTWCfg()
        task = builders.make_task_cfg(model, cfg)
        o = self.o = MdpOracle(model, cfg, task, N, fx["init_env_origins"])
        o.episode_length_buf[:] = fx["init_episode_length_buf"]
        o.commands[:] = fx["init_commands"]
        o.theta[:], o.gait_period[:], o.gait_time[:], o.phi[:] = fx["init_theta"], fx["init_gait_period"], fx["init_gait_time"], fx["init_phi"]
        br = fx["init_behavior_ranges"]
        o.gait_period_range, o.base_height_target_range = list(br[0:2]), list(br[2:4])
        o.foot_clearance_target_range, o.pitch_target_range, o.num_gaits = list(br[4:6]), list(br[6:8]), int(br[8])
        # the fake simulator of the generator starts with friction 0 / added mass 1 (genesis_simulator.py:646-649
        # before the create-time randomisation, which the fake does not perform)
        o.friction_values[:] = 0; o.added_base_mass[:] = 1
        self.names = [str(n) for n in fx["reward_names"]]

    def step(self, t, sim, actions, R, counter, override):
        o = self.o
        o.step(sim, actions, R, counter)
        ts = np.concatenate([o.gait_time, o.phi, o.gait_period, o.base_height_target, o.foot_clearance_target,
                             o.pitch_target, o.theta, o.clock_input, o.exp_C_frc], 1)
        return dict(obs=o.obs_buf, priv=o.priv_obs_buf, rew=o.rew_buf, reset=o.reset_buf, time_out=o.time_out_buf,
                    commands=o.commands, ep_len=o.episode_length_buf, fail_buf=o.fail_buf,
                    episode_sums=np.stack([o.episode_sums[abi.REWARD_ID[n]] for n in self.names]),
                    act_hist=np.stack([o.actions, o.last_actions, o.llast_actions]),
                    sim_dof_pos=sim["dof_pos"], sim_base_pos=sim["base_pos"], sim_base_lin_vel_w=sim["base_lin_vel_w"],
                    dr_pd=np.concatenate([o.kp_scale, o.kd_scale], 1), task_state=ts)


WTW_EXACT = ("reset", "time_out", "ep_len", "fail_buf")
WTW_FLOAT = ("obs", "priv", "rew", "commands", "episode_sums", "act_hist", "sim_dof_pos", "sim_base_pos",
             "sim_base_lin_vel_w", "dr_pd", "task_state")


def check_wtw(t, fx, out, rtol=2e-6, atol=2e-6):
    for k in WTW_EXACT:
        np.testing.assert_array_equal(np.asarray(out[k]).astype(np.int64), fx[k][t].astype(np.int64), err_msg=f"{k} @ step {t}")
    for k in WTW_FLOAT:
        ref, got = fx[k][t], np.asarray(out[k])
        np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol, err_msg=f"{k} @ step {t}")


def test_wtw_fixture_exercises_the_branches():
    fx = np.load(GOLD_WTW)
    assert fx["reset"].sum() >= 6
    th = fx["task_state"][:, :, 6:10]
    assert len({tuple(r) for r in th.reshape(-1, 4)}) >= 4            # several gaits from the table
    assert (fx["task_state"][1:, :, 0] < fx["task_state"][:-1, :, 0]).any()   # gait clock wrapped
    assert (fx["counter"] % 750 == 0).any()


def test_mdp_oracle_reproduces_reference_go2_wtw():
    replay(WtwOracleStepper, check_wtw, GOLD_WTW)


# ------------------------------- go2_ee (rough terrain) ------------------------------------------
GOLD_EE = os.path.join(os.path.dirname(__file__), "golden", "go2_ee_mdp.npz")
# the other Go2-rough heads replay through the same steppers: (config class, critic frame width, contact-state columns of the
# critic frame, fixture)
HEADS = {"go2_ee": ("GO2EECfg", 174, slice(76, 93)), "go2_ts": ("GO2TSCfg", 177, slice(79, 96)), "go2_cts": ("GO2CTSCfg", 177, slice(79, 96)),
         "go2_dreamwaq": ("GO2DreamwaqCfg", 177, slice(79, 96)), "go2_cat": ("GO2CaTCfg", 177, slice(79, 96))}


def head_gold(head):
    return os.path.join(os.path.dirname(__file__), "golden", f"{head}_mdp.npz")


def ee_terrain(fx, head="go2_ee"):
    from hcr_genesis_lr_cl_amd import config as cfgmod
    from hcr_genesis_lr_cl_amd.terrain import Terrain
    cfg = getattr(cfgmod, HEADS[head][0])()
    np.random.seed(int(fx["terrain_seed"]))
    return cfg, Terrain(cfg.terrain)


class EEOracleStepper:
    head = "go2_ee"

    def __init__(self, fx, N):
        import oracle.mdp_oracle as mo
        self.mo = mo
        model = load_model("go2")
        self.cfg, self.terrain = ee_terrain(fx, self.head)
        task = builders.make_task_cfg(model, self.cfg)
        o = self.o = MdpOracle(model, self.cfg, task, N, fx["init_env_origins"])
        o.episode_length_buf[:] = fx["init_episode_length_buf"]
        o.commands[:] = fx["init_commands"]
        o.terrain_levels[:], o.terrain_types[:] = fx["init_terrain_levels"], fx["init_terrain_types"]
        o.terrain_origins = self.terrain.env_origins.astype(np.float32)
        o.friction_values[:] = 0; o.added_base_mass[:] = 1
        self.hp = fx["init_height_points"]
        self.names = [str(n) for n in fx["reward_names"]]

    def step(self, t, sim, actions, R, counter, override):
        o, c, mo = self.o, self.cfg.terrain, self.mo
        hf = self.terrain.height_field_raw
        sim["measured_heights"] = mo.sample_heights(sim["base_pos"], sim["base_quat"], self.hp, hf, c.border_size, c.horizontal_scale, c.vertical_scale)
        sim["height_around_feet"], sim["normals"] = mo.feet_terrain_info(sim["feet_pos"].reshape(len(actions), 4, 3), hf, c.border_size,
                                                                          c.horizontal_scale, c.vertical_scale)
        mh, har, nrm = sim["measured_heights"].copy(), sim["height_around_feet"].copy(), sim["normals"].copy()
        o.step(sim, actions, R, counter)
        W, cols = HEADS[self.head][1], HEADS[self.head][2]
        return dict(feat_new=o.obs_buf[:, -45:], priv_new=o.priv_obs_buf[:, -W:], labels=o.labels_buf, rew=o.rew_buf,
                    reset=o.reset_buf, time_out=o.time_out_buf, commands=o.commands, ep_len=o.episode_length_buf,
                    fail_buf=o.fail_buf, feet_air_time=o.feet_air_time,
                    episode_sums=np.stack([o.episode_sums[abi.REWARD_ID[n]] for n in self.names]),
                    sim_dof_pos=sim["dof_pos"], sim_base_pos=sim["base_pos"], terrain_levels=o.terrain_levels,
                    env_origins=o.env_origins, measured_heights=mh, height_around_feet=har, normals=nrm,
                    contact_states=o.priv_obs_buf[:, -W:][:, cols], feat_full=o.obs_buf, priv_full=o.priv_obs_buf,
                    obs=np.clip(o.obs_buf[:, -45:], -100.0, 100.0), cstr_prob=o.cstr_prob, cstr_sums=o.cstr_sums)


EE_EXACT = ("reset", "time_out", "ep_len", "fail_buf", "terrain_levels")
EE_FLOAT = ("measured_heights", "height_around_feet", "normals", "contact_states", "feat_new", "priv_new", "labels", "rew",
            "commands", "feet_air_time", "episode_sums", "sim_dof_pos", "sim_base_pos", "env_origins")


def check_ee(t, fx, out, rtol=2e-6, atol=2e-6):
    for k in EE_EXACT:
        np.testing.assert_array_equal(np.asarray(out[k]).astype(np.int64), fx[k][t].astype(np.int64), err_msg=f"{k} @ step {t}")
    for k in EE_FLOAT:
        np.testing.assert_allclose(np.asarray(out[k]), fx[k][t].reshape(np.asarray(out[k]).shape), rtol=rtol, atol=atol, err_msg=f"{k} @ step {t}")
    if t == fx["obs"].shape[0] - 1 if "obs" in fx.files else t == fx["rew"].shape[0] - 1:
        np.testing.assert_allclose(out["feat_full"], fx["feat_last"], rtol=rtol, atol=atol, err_msg="stacked estimator features")
        np.testing.assert_allclose(out["priv_full"], fx["priv_last"], rtol=rtol, atol=atol, err_msg="stacked critic obs")


def replay_ee(make_stepper, check, gold=GOLD_EE):
    fx = np.load(gold)
    T, N = fx["rew"].shape
    st = make_stepper(fx, N)
    for t in range(T):
        sim_in = {k[len("script_"):]: fx[k][t].copy() for k in fx.files if k.startswith("script_")}
        sim_in["last_dof_vel"] = fx["last_dof_vel_in"][t].copy()
        sim_in["last_feet_vel"] = fx["last_feet_vel_in"][t].copy()
        out = st.step(t, sim_in, fx["actions_in"][t], fx["rand"][t], int(fx["counter"][t]), 0.0)
        check(t, fx, out)


def test_ee_fixture_exercises_the_branches():
    fx = np.load(GOLD_EE)
    assert fx["reset"].sum() >= 8
    assert (fx["terrain_levels"][-1] != fx["init_terrain_levels"]).sum() >= 4      # terrain curriculum moved envs
    assert np.ptp(fx["measured_heights"]) > 0.2 and np.abs(fx["normals"][..., 0]).max() > 0.5
    assert (fx["counter"] % 500 == 0).any()


def test_mdp_oracle_reproduces_reference_go2_ee():
    replay_ee(EEOracleStepper, check_ee)


# ------------------------------- go2_ts / go2_cts / go2_dreamwaq (same physics and rewards, other packaging) ------------
def check_head(t, fx, out, rtol=2e-6, atol=2e-6):
    check_ee(t, fx, out, rtol, atol)
    np.testing.assert_allclose(out["obs"], fx["obs"][t], rtol=rtol, atol=atol, err_msg=f"clipped actor frame @ step {t}")
    if fx["cstr_sums"].shape[1]:          # go2_cat: termination probability and per-episode violation counters, exact
        np.testing.assert_array_equal(out["cstr_prob"], fx["cstr_prob"][t], err_msg=f"cstr_prob @ step {t}")
        np.testing.assert_array_equal(out["cstr_sums"], fx["cstr_sums"][t], err_msg=f"cstr_sums @ step {t}")


ALL_HEADS = ["go2_ts", "go2_cts", "go2_dreamwaq", "go2_cat"]


def test_cat_fixture_exercises_the_constraints():
    fx = np.load(head_gold("go2_cat"))
    p = fx["cstr_prob"]
    assert (p == 0).any() and (p == 0.25).any() and (p == 1.0).any() and set(np.unique(p)) <= {0.0, 0.25, 1.0}
    seen = fx["cstr_sums"].max(axis=(0, 2)) > 0
    assert seen[[2, 3, 4, 5]].all()                 # action rate, base height, collision, feet stumble all occur
    assert (fx["reset"].sum(1) > 0).any() and (fx["cstr_sums"][-1] < fx["cstr_sums"].max(0)).any()    # counters zeroed at resets


@pytest.mark.parametrize("head", ALL_HEADS)
def test_head_fixture_is_what_the_reference_emits_as_configured(head):
    """17 contact-state links as configured (common_cfgs.py:100-101) although the head's size fields assume 12
    (go2_ts_config.py:8-14): privileged 99 wide / critic frames 177 wide."""
    fx = np.load(head_gold(head))
    assert fx["priv_new"].shape[-1] == 177 and fx["priv_last"].shape[-1] == 5 * 177 and fx["feat_last"].shape[-1] == 20 * 45
    assert fx["labels"].shape[-1] == (69 if head == "go2_dreamwaq" else 99)
    assert fx["reset"].sum() >= 6 and (fx["counter"] % 500 == 0).any()
    np.testing.assert_array_equal(fx["obs"], np.clip(fx["feat_new"], -100, 100))      # obs = newest history frame, clipped


@pytest.mark.parametrize("head", ALL_HEADS)
def test_mdp_oracle_reproduces_reference_head(head):
    stepper = type("Stepper_" + head, (EEOracleStepper,), {"head": head})
    replay_ee(stepper, check_head, head_gold(head))


# ------------------------------- tron1_pf_ee (biped, rough terrain) -------------------------------
GOLD_TRON1 = os.path.join(os.path.dirname(__file__), "golden", "tron1_pf_ee_mdp.npz")


def tron1_terrain(fx):
    from hcr_genesis_lr_cl_amd.config import TRON1PFEECfg
    from hcr_genesis_lr_cl_amd.terrain import Terrain
    cfg = TRON1PFEECfg()
    np.random.seed(int(fx["terrain_seed"]))
    return cfg, Terrain(cfg.terrain)


class Tron1OracleStepper:
    def __init__(self, fx, N):
        import oracle.mdp_oracle as mo
        self.mo = mo
        model = load_model("tron1_pf")
        self.cfg, self.terrain = tron1_terrain(fx)
        task = builders.make_task_cfg(model, self.cfg)
        o = self.o = MdpOracle(model, self.cfg, task, N, fx["init_env_origins"])
        o.episode_length_buf[:] = fx["init_episode_length_buf"]
        o.commands[:] = fx["init_commands"]
        o.terrain_levels[:], o.terrain_types[:] = fx["init_terrain_levels"], fx["init_terrain_types"]
        o.terrain_origins = self.terrain.env_origins.astype(np.float32)
        o.theta[:], o.gait_time[:], o.phi[:] = fx["init_theta"], fx["init_gait_time"], fx["init_phi"]
        o.friction_values[:] = 0; o.added_base_mass[:] = 1
        self.hp = fx["init_height_points"]
        self.names = [str(n) for n in fx["reward_names"]]

    def step(self, t, sim, actions, R, counter, override):
        o, c, mo = self.o, self.cfg.terrain, self.mo
        hf = self.terrain.height_field_raw
        N = len(actions)
        sim["measured_heights"] = mo.sample_heights(sim["base_pos"], sim["base_quat"], self.hp, hf, c.border_size, c.horizontal_scale, c.vertical_scale)
        sim["height_around_feet"], sim["normals"] = mo.feet_terrain_info(sim["feet_pos"].reshape(N, 2, 3), hf, c.border_size,
                                                                          c.horizontal_scale, c.vertical_scale)
        mh, har, nrm = sim["measured_heights"].copy(), sim["height_around_feet"].copy(), sim["normals"].copy()
        o.step(sim, actions, R, counter)
        ts = np.concatenate([o.gait_time, o.phi, o.theta, o.clock_input, o.exp_C_frc], 1)
        return dict(feat_new=o.obs_buf[:, -31:], priv_new=o.priv_obs_buf[:, -134:], labels=o.labels_buf, rew=o.rew_buf,
                    reset=o.reset_buf, time_out=o.time_out_buf, commands=o.commands, ep_len=o.episode_length_buf, fail_buf=o.fail_buf,
                    episode_sums=np.stack([o.episode_sums[abi.REWARD_ID[n]] for n in self.names]),
                    act_hist=np.stack([o.actions, o.last_actions, o.llast_actions]),
                    sim_dof_pos=sim["dof_pos"], sim_base_pos=sim["base_pos"], sim_base_quat=sim["base_quat"],
                    terrain_levels=o.terrain_levels, env_origins=o.env_origins, measured_heights=mh, height_around_feet=har,
                    normals=nrm, dr_joint=np.concatenate([o.joint_armature, o.joint_friction, o.joint_damping], 1), task_state=ts,
                    feat_full=o.obs_buf, priv_full=o.priv_obs_buf)


T1_EXACT = ("reset", "time_out", "ep_len", "fail_buf", "terrain_levels")
T1_FLOAT = ("measured_heights", "height_around_feet", "normals", "feat_new", "priv_new", "labels", "rew", "commands",
            "episode_sums", "act_hist", "sim_dof_pos", "sim_base_pos", "sim_base_quat", "env_origins", "dr_joint", "task_state")


def check_tron1(t, fx, out, rtol=2e-6, atol=2e-6, skip_env0=False):
    sl = slice(1, None) if skip_env0 else slice(None)
    for k in T1_EXACT:
        np.testing.assert_array_equal(np.asarray(out[k]).astype(np.int64)[sl], fx[k][t].astype(np.int64)[sl], err_msg=f"{k} @ step {t}")
    for k in T1_FLOAT:
        got = np.asarray(out[k])
        ref = fx[k][t].reshape(got.shape)
        if k in ("episode_sums", "act_hist"):
            got, ref = got[:, sl], ref[:, sl]
        else:
            got, ref = got[sl], ref[sl]
        np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol, err_msg=f"{k} @ step {t}")
    if t == fx["rew"].shape[0] - 1:
        np.testing.assert_allclose(out["feat_full"][sl], fx["feat_last"][sl], rtol=rtol, atol=atol, err_msg="stacked estimator features")
        np.testing.assert_allclose(out["priv_full"][sl], fx["priv_last"][sl], rtol=rtol, atol=atol, err_msg="stacked critic obs")


def replay_rough(gold, make_stepper, check):
    fx = np.load(gold)
    T, N = fx["rew"].shape
    st = make_stepper(fx, N)
    for t in range(T):
        sim_in = {k[len("script_"):]: fx[k][t].copy() for k in fx.files if k.startswith("script_")}
        sim_in["last_dof_vel"] = fx["last_dof_vel_in"][t].copy()
        sim_in["last_feet_vel"] = fx["last_feet_vel_in"][t].copy()
        out = st.step(t, sim_in, fx["actions_in"][t], fx["rand"][t], int(fx["counter"][t]), 0.0)
        check(t, fx, out)


def test_tron1_fixture_exercises_the_branches():
    fx = np.load(GOLD_TRON1)
    r = fx["reset"].astype(bool)
    assert r.sum() >= 10
    sit = (np.abs(fx["sim_base_quat"][:, :, 1]) > 0.05) & r
    assert sit.sum() >= 2 and (r & ~sit).sum() >= 2                       # both reset branches (tron1_pf_ee.py:204-210)
    assert (fx["reset"].astype(bool) & ~fx["time_out"].astype(bool)).sum() >= 1
    assert fx["dr_joint"][-1][:, 0].max() > 0.11                           # armature randomised


def test_mdp_oracle_reproduces_reference_tron1_pf_ee():
    replay_rough(GOLD_TRON1, Tron1OracleStepper, check_tron1)


# ------------------------------- tron1_pf (biped on the plane) ------------------------------------
GOLD_PF = os.path.join(os.path.dirname(__file__), "golden", "tron1_pf_mdp.npz")


class PFOracleStepper:
    def __init__(self, fx, N):
        from hcr_genesis_lr_cl_amd.config import TRON1PFCfg
        model, cfg = load_model("tron1_pf"), TRON1PFCfg()
        task = builders.make_task_cfg(model, cfg)
        o = self.o = MdpOracle(model, cfg, task, N, fx["init_env_origins"])
        o.episode_length_buf[:] = fx["init_episode_length_buf"]
        o.commands[:] = fx["init_commands"]
        o.friction_values[:] = 0; o.added_base_mass[:] = 1          # the generator's fake simulator starts like genesis_simulator.py:646-649
        self.names = [str(n) for n in fx["reward_names"]]

    def step(self, t, sim, actions, R, counter, override):
        o = self.o
        o.step(sim, actions, R, counter)
        return dict(obs=np.clip(o.obs_buf, -100, 100), priv=o.priv_obs_buf, rew=o.rew_buf, reset=o.reset_buf, time_out=o.time_out_buf,
                    commands=o.commands, ep_len=o.episode_length_buf, fail_buf=o.fail_buf, feet_air_time=o.feet_air_time,
                    episode_sums=np.stack([o.episode_sums[abi.REWARD_ID[n]] for n in self.names]),
                    act_hist=np.stack([o.actions, o.last_actions, o.llast_actions]),
                    sim_dof_pos=sim["dof_pos"], sim_base_pos=sim["base_pos"], sim_base_lin_vel_w=sim["base_lin_vel_w"],
                    dr=np.concatenate([o.friction_values, o.added_base_mass, o.base_com_bias, o.rand_push_vels[:, :2]], 1))


PF_FLOAT = ("obs", "priv", "rew", "commands", "feet_air_time", "episode_sums", "act_hist", "sim_dof_pos", "sim_base_pos", "sim_base_lin_vel_w", "dr")


def check_pf(t, fx, out, rtol=2e-6, atol=2e-6):
    for k in WTW_EXACT:
        np.testing.assert_array_equal(np.asarray(out[k]).astype(np.int64), fx[k][t].astype(np.int64), err_msg=f"{k} @ step {t}")
    for k in PF_FLOAT:
        np.testing.assert_allclose(np.asarray(out[k]), fx[k][t], rtol=rtol, atol=atol, err_msg=f"{k} @ step {t}")


def test_tron1_pf_fixture_exercises_the_branches():
    fx = np.load(GOLD_PF)
    names = [str(n) for n in fx["reward_names"]]
    assert fx["reset"].sum() >= 8 and (fx["counter"] % 500 == 0).any() and "no_fly" in names and len(names) == 19
    assert fx["obs"].shape[-1] == 5 * 27 and fx["priv"].shape[-1] == 5 * 45
    k = names.index("no_fly")
    d = np.diff(fx["episode_sums"][:, k], axis=0)
    assert (d > 0).any() and (d == 0).any()


def test_mdp_oracle_reproduces_reference_tron1_pf():
    replay(PFOracleStepper, check_pf, GOLD_PF)


# ------------------------------- tron1_sf (8-DOF sole-foot biped on the plane) ---------------------
GOLD_SF = os.path.join(os.path.dirname(__file__), "golden", "tron1_sf_mdp.npz")


def sf_cfg():
    from hcr_genesis_lr_cl_amd.config import TRON1SFCfg
    cfg = TRON1SFCfg()
    cfg.rewards.scales.keep_ankle_pitch_zero_in_air = 0.2     # as in the generator: the class defines it, the shipped config leaves it unscaled
    return cfg


class SFOracleStepper:
    def __init__(self, fx, N):
        model, cfg = load_model("tron1_sf"), sf_cfg()
        task = builders.make_task_cfg(model, cfg)
        o = self.o = MdpOracle(model, cfg, task, N, fx["init_env_origins"])
        o.episode_length_buf[:] = fx["init_episode_length_buf"]
        o.commands[:] = fx["init_commands"]
        o.friction_values[:] = 0; o.added_base_mass[:] = 1          # the generator's fake simulator starts like genesis_simulator.py:646-649
        self.names = [str(n) for n in fx["reward_names"]]

    def step(self, t, sim, actions, R, counter, override):
        o = self.o
        sim.pop("foot_quat", None)     # what the reference read from rigid_body_states; here it follows from base_quat and dof_pos
        o.step(sim, actions, R, counter)
        return dict(obs=np.clip(o.obs_buf, -100, 100), priv=o.priv_obs_buf, rew=o.rew_buf, reset=o.reset_buf, time_out=o.time_out_buf,
                    commands=o.commands, ep_len=o.episode_length_buf, fail_buf=o.fail_buf, feet_air_time=o.feet_air_time,
                    episode_sums=np.stack([o.episode_sums[abi.reward_id(n, 4)] for n in self.names]),
                    act_hist=np.stack([o.actions, o.last_actions, o.llast_actions]),
                    sim_dof_pos=sim["dof_pos"], sim_base_pos=sim["base_pos"], sim_base_quat=sim["base_quat"], sim_base_lin_vel_w=sim["base_lin_vel_w"],
                    dr=np.concatenate([o.friction_values, o.added_base_mass, o.base_com_bias, o.rand_push_vels[:, :2]], 1),
                    dr_pd=np.concatenate([o.kp_scale, o.kd_scale], 1),
                    dr_joint=np.concatenate([o.joint_armature, o.joint_friction, o.joint_damping], 1))


SF_FLOAT = PF_FLOAT + ("sim_base_quat", "dr_pd", "dr_joint")


def check_sf(t, fx, out, rtol=2e-6, atol=2e-6):
    for k in WTW_EXACT:
        np.testing.assert_array_equal(np.asarray(out[k]).astype(np.int64), fx[k][t].astype(np.int64), err_msg=f"{k} @ step {t}")
    for k in SF_FLOAT:
        np.testing.assert_allclose(np.asarray(out[k]), fx[k][t], rtol=rtol, atol=atol, err_msg=f"{k} @ step {t}")


def test_tron1_sf_fixture_exercises_the_branches():
    fx = np.load(GOLD_SF)
    names = [str(n) for n in fx["reward_names"]]
    assert fx["reset"].sum() >= 6 and (fx["counter"] % 500 == 0).any() and len(names) == 21
    assert {"foot_flat", "hip_pos_zero_command", "keep_ankle_pitch_zero_in_air", "no_fly"} <= set(names)
    assert fx["obs"].shape[-1] == 10 * 33 and fx["priv"].shape[-1] == 10 * 72
    r = fx["reset"].astype(bool)
    sit = r & (np.abs(fx["sim_dof_pos"][:, :, 2] - 1.35) < 1e-6)            # both reset branches (tron1_sf.py:160-166)
    assert sit.sum() >= 2 and (r & ~sit).sum() >= 2
    for n in ("foot_flat", "hip_pos_zero_command", "keep_ankle_pitch_zero_in_air"):
        d = np.diff(fx["episode_sums"][:, names.index(n)], axis=0)
        assert (np.abs(d) > 0).any(), n
    # tron1_sf.py:224-231: dof 7 is never offset by a reset, dof 0 moves by at most 0.05
    nonsit = r & ~sit
    assert np.all(fx["sim_dof_pos"][:, :, 7][nonsit] == 0) and np.abs(fx["sim_dof_pos"][:, :, 0][nonsit]).max() <= 0.05 + 1e-6


def test_mdp_oracle_reproduces_reference_tron1_sf():
    replay(SFOracleStepper, check_sf, GOLD_SF)
