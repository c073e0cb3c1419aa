"""Generates golden vectors for the MDP part of the hot path by RUNNING THE REFERENCE'S OWN
env classes (legged_gym/envs/...) on CPU against a fake simulator that feeds scripted physics
read-backs.  Output: tests/golden/<task>_mdp.npz (inputs + expected outputs per step).

Run in the build container only:   PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_mdp_fixtures.py

What is pinned: everything LeggedRobot.step does around the physics -- action clip/history
(legged_robot.py:230-239), command resampling + heading command + push schedule (:300-334),
termination (:78-92), every active reward term and the alphabetical sum (:150-168, 458-608),
reset_idx bookkeeping incl. command curriculum (:94-148, 336-348), task reset distributions
and observation layout/noise/clip (go2.py:17-134, legged_robot.py:48-49).
Uniform draws of the reference are intercepted (DrawRecorder) and stored per env in the slot
layout of include/lgsim.h's LgRandSlots, so the kernel can be fed the very same numbers.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_harness as rh  # noqa: E402

rh.load_reference()
import torch  # noqa: E402

from hcr_genesis_lr_cl_amd import builders  # noqa: E402
from hcr_genesis_lr_cl_amd.model_compiler import load_model  # noqa: E402

torch.set_num_threads(2)


def rand_quat(rng, n, tilt):
    rpy = rng.normal(size=(n, 3)) * [tilt, tilt, 1.5]
    cr, sr = np.cos(rpy[:, 0] / 2), np.sin(rpy[:, 0] / 2)
    cp, sp = np.cos(rpy[:, 1] / 2), np.sin(rpy[:, 1] / 2)
    cy, sy = np.cos(rpy[:, 2] / 2), np.sin(rpy[:, 2] / 2)
    q = np.stack([cy * sr * cp - sy * cr * sp, cy * cr * sp + sy * sr * cp, sy * cr * cp - cy * sr * sp,
                  cy * cr * cp + sy * sr * sp], 1)
    return q.astype(np.float32)


class FakeSimulator:
    """Duck-typed stand-in for GenesisSimulator: serves scripted read-backs, records writes."""

    def __init__(self, cfg, sim_params, device, headless):
        from legged_gym.utils.math_utils import quat_rotate_inverse
        self._qri = quat_rotate_inverse
        self._cfg, self._device = cfg, device
        N = self._num_envs = cfg.env.num_envs
        A = self._num_actions = cfg.env.num_actions
        self.model = load_model(cfg.asset.name)
        m = self.model
        L, F = m.n_links, m.n_legs
        z = lambda *s: torch.zeros(*s)
        self._base_pos, self._base_quat, self._base_euler = z(N, 3), z(N, 4), z(N, 3)
        self._base_quat[:, 3] = 1
        self._base_lin_vel, self._base_ang_vel, self._projected_gravity = z(N, 3), z(N, 3), z(N, 3)
        self._base_lin_vel_w, self._base_ang_vel_w = z(N, 3), z(N, 3)
        self._dof_pos, self._dof_vel, self._last_dof_vel, self._torques = z(N, A), z(N, A), z(N, A), z(N, A)
        self._link_contact_forces = z(N, L, 3)
        self._feet_pos, self._feet_vel, self._last_feet_vel = z(N, F, 3), z(N, F, 3), z(N, F, 3)
        self._feet_indices = m.find_link_indices([n for n in m.link_names if cfg.asset.foot_name in n])
        self._termination_contact_indices = m.find_link_indices(cfg.asset.terminate_after_contacts_on)
        self._penalized_contact_indices = m.find_link_indices(cfg.asset.penalize_contacts_on)
        self._default_dof_pos = torch.tensor([cfg.init_state.default_joint_angles[n] for n in cfg.asset.dof_names]).unsqueeze(0)
        lim = torch.tensor(np.stack([m.arrays["q_lo"], m.arrays["q_hi"]], 1), dtype=torch.float)
        for i in range(A):  # genesis_simulator.py:373-382
            mid = (lim[i, 0] + lim[i, 1]) / 2
            r = lim[i, 1] - lim[i, 0]
            lim[i, 0] = mid - 0.5 * r * cfg.rewards.soft_dof_pos_limit
            lim[i, 1] = mid + 0.5 * r * cfg.rewards.soft_dof_pos_limit
        self._dof_pos_limits = lim
        self._base_init_pos = torch.tensor(cfg.init_state.pos)
        self._base_init_quat = torch.tensor(cfg.init_state.rot)
        self._custom_origins = cfg.terrain.mesh_type in ("heightfield", "trimesh")
        self._env_origins = z(N, 3)
        self._env_origins[:, :2] = torch.from_numpy(np.random.default_rng(5).uniform(-5, 5, (N, 2)).astype(np.float32))
        self._measured_heights = z(N, 187)
        self._friction_values, self._added_base_mass = z(N, 1), torch.ones(N, 1)
        self._base_com_bias, self._rand_push_vels = z(N, 3), z(N, 3)
        self._kp_scale, self._kd_scale = torch.ones(N, A), torch.ones(N, A)
        self._joint_armature, self._joint_friction, self._joint_damping = z(N, 1), z(N, 1), z(N, 1)
        self.dof_names = list(cfg.asset.dof_names)   # tron1_pf_ee.py:171 reads simulator.dof_names (absent from the reference ABC)
        self._global_gravity = torch.tensor([0., 0., -1.]).repeat(N, 1)
        self._rigid_body_states = z(N, L, 13)                   # only [:, feet, 3:7] is read (tron1_sf.py:300)
        self._rigid_body_states[:, :, 6] = 1
        self.script, self.t, self.rec = None, -1, None

    # -- scripted physics ------------------------------------------------------------------
    def step(self, actions):
        self._last_dof_vel[:] = self._dof_vel
        self._last_feet_vel[:] = self._feet_vel
        self.t += 1

    def post_physics_step(self):
        s = {k: torch.from_numpy(v[self.t]) for k, v in self.script.items()}
        for k in ("base_pos", "base_quat", "base_lin_vel_w", "base_ang_vel_w", "dof_pos", "dof_vel", "torques",
                  "link_contact_forces", "feet_pos", "feet_vel"):
            getattr(self, "_" + k)[:] = s[k].reshape(getattr(self, "_" + k).shape)
        if "foot_quat" in s:
            self._rigid_body_states[:, self._feet_indices, 3:7] = s["foot_quat"]
        self._base_lin_vel[:] = self._qri(self._base_quat, self._base_lin_vel_w)
        self._base_ang_vel[:] = self._qri(self._base_quat, self._base_ang_vel_w)
        self._projected_gravity = self._qri(self._base_quat, self._global_gravity)
        from legged_gym.utils.math_utils import get_euler_xyz
        self._base_euler[:] = get_euler_xyz(self._base_quat)

    # -- writes (semantics of genesis_simulator.py:62-158) ---------------------------------------
    def reset_idx(self, env_ids):
        """Order and formulas of genesis_simulator.py:62-82, 665-739; draws go through the recorder."""
        d = self._cfg.domain_rand
        n = len(env_ids)
        ids = torch.as_tensor(env_ids)
        if d.randomize_friction:
            lo, hi = d.friction_range
            self._friction_values[env_ids] = self.rec.tagged("dr_friction", (n, 1), ids) * (hi - lo) + lo
        if d.randomize_base_mass:
            lo, hi = d.added_mass_range
            self._added_base_mass[env_ids] = self.rec.tagged("dr_mass", (n, 1), ids) * (hi - lo) + lo
        if d.randomize_com_displacement:
            for k, (lo, hi) in enumerate((d.com_pos_x_range, d.com_pos_y_range, d.com_pos_z_range)):
                self._base_com_bias[env_ids, k] = self.rec.tagged(f"dr_com+{k}", (n, 1), ids).squeeze(1) * (hi - lo) + lo
        for k, (flag, rng_name, buf) in enumerate((("randomize_joint_armature", "joint_armature_range", "_joint_armature"),
                                                   ("randomize_joint_friction", "joint_friction_range", "_joint_friction"),
                                                   ("randomize_joint_damping", "joint_damping_range", "_joint_damping"))):
            if getattr(d, flag):
                lo, hi = getattr(d, rng_name)
                getattr(self, buf)[env_ids, 0] = self.rec.tagged(f"dr_joint+{k}", (n,), ids) * (hi - lo) + lo
        if d.randomize_pd_gain:
            A = self._num_actions
            self._kp_scale[env_ids] = (d.kp_range[1] - d.kp_range[0]) * self.rec.tagged("dr_kp", (n, A), ids) + d.kp_range[0]
            self._kd_scale[env_ids] = (d.kd_range[1] - d.kd_range[0]) * self.rec.tagged("dr_kd", (n, A), ids) + d.kd_range[0]
        self._last_dof_vel[env_ids] = 0.
        self._last_feet_vel[env_ids] = 0.

    def reset_dofs(self, env_ids, dof_pos, dof_vel):
        self._dof_pos[env_ids] = dof_pos[:]
        self._dof_vel[env_ids] = dof_vel[:]

    def reset_root_states(self, env_ids, base_pos, base_quat, base_lin_vel, base_ang_vel):
        self._base_pos[env_ids, :] = base_pos[:]
        self._base_quat[env_ids, :] = base_quat[:]
        self._projected_gravity = self._qri(self._base_quat, self._global_gravity)
        self._base_lin_vel[env_ids] = base_lin_vel[:]
        self._base_ang_vel[env_ids] = base_ang_vel[:]
        self._base_lin_vel_w[env_ids] = base_lin_vel[:]
        self._base_ang_vel_w[env_ids] = base_ang_vel[:]

    def push_robots(self):
        m = self._cfg.domain_rand.max_push_vel_xy
        env_ids = torch.arange(self._num_envs)  # noqa: F841  (picked up by the recorder)
        push = self.rec.rand_float(-m, m, (self._num_envs, 2), "cpu")
        self._rand_push_vels[:, :2] = push.clone()
        self._base_lin_vel_w[:, :2] += push

    def update_sensors(self):
        pass

    def draw_debug_vis(self):
        pass

    def set_viewer_camera(self, eye, target):
        pass


for _p in ("rigid_body_states", "feet_indices", "termination_contact_indices", "penalized_contact_indices", "dof_pos_limits",
           "base_init_pos", "base_init_quat", "base_lin_vel", "base_ang_vel", "projected_gravity", "dof_pos",
           "dof_vel", "last_dof_vel", "feet_pos", "feet_vel", "last_feet_vel", "base_pos", "base_quat",
           "base_euler", "measured_heights", "link_contact_forces", "torques", "default_dof_pos", "custom_origins",
           "env_origins"):
    setattr(FakeSimulator, _p, property(lambda s, _n="_" + _p: getattr(s, _n)))
FakeSimulator.feet_contact_indices = property(lambda s: s._feet_indices)
FakeSimulator.dr_friction_values = property(lambda s: s._friction_values)
FakeSimulator.dr_added_base_mass = property(lambda s: s._added_base_mass)
FakeSimulator.dr_base_com_bias = property(lambda s: s._base_com_bias)
FakeSimulator.dr_rand_push_vels = property(lambda s: s._rand_push_vels)


class RoughFakeSimulator(FakeSimulator):
    """Adds the heightfield side: the terrain comes from the reference's own Terrain class, and the
    height sampling / feet terrain info / contact states / terrain curriculum are the reference's own
    GenesisSimulator methods executed on this object (genesis_simulator.py:53-60, 140-148, 496-610)."""
    TERRAIN_SEED = 3

    def __init__(self, cfg, sim_params, device, headless):
        super().__init__(cfg, sim_params, device, headless)
        from legged_gym.simulator.genesis_simulator import GenesisSimulator as GS
        from legged_gym.utils.terrain import Terrain
        self.GS = GS
        np.random.seed(self.TERRAIN_SEED)
        self._terrain = Terrain(cfg.terrain)
        self._height_samples = torch.tensor(self._terrain.heightsamples).view(self._terrain.tot_rows, self._terrain.tot_cols)
        N = self._num_envs
        self._custom_origins = True
        self._max_terrain_level = cfg.terrain.num_rows
        rng = np.random.default_rng(17)
        self._terrain_levels = torch.from_numpy(rng.integers(0, cfg.terrain.num_rows, N))
        self._terrain_types = torch.div(torch.arange(N), (N / cfg.terrain.num_cols), rounding_mode='floor').to(torch.long)
        self._terrain_origins = torch.from_numpy(self._terrain.env_origins).to(torch.float)
        self._env_origins = self._terrain_origins[self._terrain_levels, self._terrain_types].clone()
        GS._init_height_points(self)
        self._measured_heights = torch.zeros(N, self._num_height_points)
        F = self.model.n_legs
        self._normal_vector_around_feet = torch.zeros(N, F * 3)
        self._height_around_feet = torch.zeros(N, F, 9)
        self._contact_state_link_indices = self.model.find_link_indices(cfg.asset.contact_state_link_names)
        self._link_contact_states = torch.zeros(N, len(self._contact_state_link_indices))

    def post_physics_step(self):
        super().post_physics_step()
        self._link_contact_states = 1. * (torch.norm(self._link_contact_forces[:, self._contact_state_link_indices, :], dim=-1) > 1.)
        self.GS._update_surrounding_heights(self)
        self.GS._calc_terrain_info_around_feet(self)

    def update_terrain_curriculum(self, env_ids, move_up, move_down):
        # the randint of genesis_simulator.py:144-145 goes through the recorder
        orig = torch.randint_like
        torch.randint_like = lambda t, high: self.rec.randint_like(t, high, env_ids)
        try:
            self.GS.update_terrain_curriculum(self, env_ids, move_up, move_down)
        finally:
            torch.randint_like = orig


for _p in ("terrain_levels", "terrain_types", "link_contact_states", "normal_vector_around_feet", "height_around_feet"):
    setattr(RoughFakeSimulator, _p, property(lambda s, _n="_" + _p: getattr(s, _n)))


def make_script(rng, model, cfg, N, T):
    """Scripted read-backs: independent plausible states per step, biased to exercise branches."""
    A, L, F = model.n_dof, model.n_links, model.n_legs
    q0 = np.array([cfg.init_state.default_joint_angles[n] for n in cfg.asset.dof_names], np.float32)
    s = {}
    s["base_pos"] = (rng.normal(size=(T, N, 3)) * [2, 2, 0.03] + [0, 0, 0.32]).astype(np.float32)
    tilt = np.where(rng.random((T, N)) < 0.08, 1.2, 0.15)
    s["base_quat"] = np.stack([rand_quat(rng, N, tilt[t][:, None] * np.ones((N, 1)))[:, :] if False else rand_quat(rng, N, 0.15) for t in range(T)])
    big = rng.random((T, N)) < 0.08
    for t in range(T):
        if big[t].any():
            s["base_quat"][t][big[t]] = rand_quat(rng, int(big[t].sum()), 1.3)
    s["base_lin_vel_w"] = (rng.normal(size=(T, N, 3)) * [0.6, 0.4, 0.15]).astype(np.float32)
    s["base_ang_vel_w"] = (rng.normal(size=(T, N, 3)) * [0.5, 0.5, 0.8]).astype(np.float32)
    s["dof_pos"] = (q0 + rng.normal(size=(T, N, A)) * 0.35).astype(np.float32)
    s["dof_vel"] = (rng.normal(size=(T, N, A)) * 3.0).astype(np.float32)
    s["torques"] = (rng.normal(size=(T, N, A)) * 8.0).astype(np.float32)
    f = np.zeros((T, N, L, 3), np.float32)
    feet = [int(i) for i in model.arrays["foot_link"][:F]]
    for l in range(L):
        if l in feet:
            on = rng.random((T, N)) < 0.6
            f[:, :, l, 2] = on * rng.uniform(0.05, 80, (T, N))
            f[:, :, l, :2] = on[..., None] * rng.normal(size=(T, N, 2)) * 10
        else:
            on = rng.random((T, N)) < (0.06 if l == 0 else 0.1)
            f[:, :, l] = on[..., None] * rng.normal(size=(T, N, 3)) * (12 if l == 0 else 3)
    s["link_contact_forces"] = f
    s["feet_pos"] = (rng.normal(size=(T, N, F, 3)) * [0.2, 0.15, 0.03] + [0, 0, 0.05]).astype(np.float32)
    s["feet_vel"] = (rng.normal(size=(T, N, F, 3)) * [0.8, 0.4, 0.5]).astype(np.float32)
    return s


def slots_from_calls(calls, slots, N, A, policy_dof_groups):
    """Scatter the recorded draws of one step into the (N, n_slots) row layout."""
    R = np.zeros((N, slots.n_slots), np.float32)
    seen = {}
    for c in calls:
        key = (c["caller"], c["parent"])
        k = seen.get(key, 0)
        seen[key] = k + 1
        u = c["u"].numpy()
        ids = None if c["env_ids"] is None else c["env_ids"].numpy()
        if c["caller"] == "_resample_commands":
            base = slots.cb_cmd if c["parent"] == "_post_physics_step_callback" else slots.reset_cmd
            R[ids, base + k] = u[:, 0]
        elif c["caller"] == "push_robots":
            R[:, slots.push:slots.push + 2] = u
        elif c["caller"] == "_reset_dofs":
            cols = policy_dof_groups[k]
            for j, col in enumerate(cols):
                R[ids, slots.reset_dof + col] = u[:, j]
        elif c["caller"] == "_reset_root_states":
            if u.shape[1] == 2:
                base = slots.reset_root_xy
            else:
                k3 = seen.get("root3", 0)
                seen["root3"] = k3 + 1
                base = [slots.reset_lin_vel, slots.reset_ang_vel][k3]
            R[ids, base:base + u.shape[1]] = u
        elif c["caller"].startswith("tag:"):
            name, _, off = c["caller"][4:].partition("+")
            base = getattr(slots, name) + (int(off) if off else 0)
            uu = u.reshape(len(ids), -1)
            R[ids, base:base + uu.shape[1]] = uu
        elif c["caller"] == "torch_rand:reset_idx":          # tron1_pf_ee.py:220-225: theta offset, gait time
            R[ids, slots.task_reset + 1 + k] = u[:, 0]
        elif c["caller"] == "np_random:reset_idx":           # tron1_pf_ee.py:204 sit-pose coin, one per call
            R[ids, slots.task_reset] = float(u[0])
        elif c["caller"] == "_reset_root_states_sit_pose":
            if u.shape[1] == 2:
                R[ids, slots.reset_root_xy:slots.reset_root_xy + 2] = u
        elif c["caller"] == "_resample_behavior_params":
            base = slots.task_cb if c["parent"] == "_post_physics_step_callback" else slots.task_reset
            R[ids, base + k] = u[:, 0]
        elif c["caller"] == "randint:_resample_behavior_params":
            base = slots.task_cb if c["parent"] == "_post_physics_step_callback" else slots.task_reset
            R[ids, base + 4] = float(u[0])
        elif c["caller"] == "randint_like":
            R[ids, slots.terrain_level] = u
        elif c["caller"] == "rand_like":
            R[:, slots.noise:slots.noise + u.shape[1]] = u
        else:
            raise RuntimeError(f"unmapped draw from {c['caller']} <- {c['parent']}")
    return R


def gen_go2(N=24, T=64, seed=7):
    import legged_gym.envs.base.base_task as base_task
    import legged_gym.envs.base.legged_robot as lr_mod
    import legged_gym.envs.go2.go2 as go2_mod
    from legged_gym.envs.go2.go2_config import GO2Cfg
    from legged_gym.utils.helpers import class_to_dict
    from hcr_genesis_lr_cl_amd.config import GO2Cfg as MyCfg

    rec = rh.DrawRecorder(seed)
    base_task.GenesisSimulator = FakeSimulator
    lr_mod.torch_rand_float = rec.rand_float
    go2_mod.torch_rand_float = rec.rand_float
    orig_rand_like = torch.rand_like
    torch.rand_like = rec.rand_like
    try:
        cfg = GO2Cfg()
        cfg.env.num_envs = N
        env = go2_mod.GO2(cfg, class_to_dict(cfg.sim), "cpu", True)
        sim = env.simulator
        sim.rec = rec
        rng = np.random.default_rng(seed + 1)
        model = sim.model
        sim.script = make_script(rng, model, cfg, N, T)
        mycfg = MyCfg()
        task = builders.make_task_cfg(model, mycfg)
        slots = task.slots
        groups = [[0, 3, 6, 9], [1, 4, 7, 10], [2, 5, 8, 11]]      # go2.py:30-35
        # initial MDP state: as after construction, with episode clocks spread so that resampling
        # (ep_len % 500 == 0) and time-outs (ep_len > 1000) occur inside the window
        env.episode_length_buf[:] = torch.from_numpy(rng.choice(
            [3, 120, 470, 480, 495, 498, 499, 960, 985, 995, 998, 999, 1000], N).astype(np.int32))
        env.commands[:] = torch.from_numpy((rng.normal(size=(N, 4)) * [0.4, 0.4, 0.5, 1.5]).astype(np.float32))
        env.common_step_counter = 745                                   # push at 750
        env.reset_buf[:] = 0
        rec.take()
        init = dict(episode_length_buf=env.episode_length_buf.numpy().copy(), commands=env.commands.numpy().copy(),
                    env_origins=sim._env_origins.numpy().copy(), default_dof_pos=sim._default_dof_pos.numpy().copy())
        out = {k: [] for k in ("actions_in", "rand", "counter", "obs", "rew", "reset", "time_out", "commands", "ep_len",
                               "fail_buf", "feet_air_time", "last_contacts", "episode_sums", "act_hist", "sim_dof_pos",
                               "sim_dof_vel", "sim_base_pos", "sim_base_quat", "sim_base_lin_vel_w", "sim_projected_gravity",
                               "sim_base_lin_vel", "dr", "cmd_range_x", "last_dof_vel_in", "last_feet_vel_in", "esum_override")}
        names = env.reward_names
        for t in range(T):
            override = 0.0
            if t == 30:
                env.common_step_counter = 995                            # command curriculum gate at 1000
            if env.common_step_counter + 1 == 1000:
                override = 18.5                                          # > 0.8 * scale * max_len = 16
                env.episode_sums["tracking_lin_vel"][:] = override
                # make sure somebody resets at this step
                env.episode_length_buf[:4] = 1000
            act = torch.from_numpy((rng.normal(size=(N, 12)) * (1.0 if t % 7 else 60.0)).astype(np.float32))
            # what the kernel is given as "last" values for this step
            out["last_dof_vel_in"].append(sim._dof_vel.numpy().copy())
            out["last_feet_vel_in"].append(sim._feet_vel.numpy().copy())
            obs, priv, rew, reset, extras = env.step(act)
            calls = rec.take()
            out["actions_in"].append(act.numpy().copy())
            out["rand"].append(slots_from_calls(calls, slots, N, 12, groups))
            out["counter"].append(env.common_step_counter)
            out["esum_override"].append(override)
            out["obs"].append(obs.numpy().copy()); out["rew"].append(rew.numpy().copy())
            out["reset"].append(reset.numpy().astype(np.uint8)); out["time_out"].append(env.time_out_buf.numpy().astype(np.uint8))
            out["commands"].append(env.commands.numpy().copy()); out["ep_len"].append(env.episode_length_buf.numpy().copy())
            out["fail_buf"].append(env.fail_buf.numpy().copy()); out["feet_air_time"].append(env.feet_air_time.numpy().copy())
            out["last_contacts"].append(env.last_contacts.numpy().astype(np.uint8))
            out["episode_sums"].append(np.stack([env.episode_sums[n].numpy().copy() for n in names]))
            out["act_hist"].append(np.stack([env.actions.numpy(), env.last_actions.numpy(), env.llast_actions.numpy()]).copy())
            out["sim_dof_pos"].append(sim._dof_pos.numpy().copy()); out["sim_dof_vel"].append(sim._dof_vel.numpy().copy())
            out["sim_base_pos"].append(sim._base_pos.numpy().copy()); out["sim_base_quat"].append(sim._base_quat.numpy().copy())
            out["sim_base_lin_vel_w"].append(sim._base_lin_vel_w.numpy().copy())
            out["sim_projected_gravity"].append(sim._projected_gravity.numpy().copy())
            out["sim_base_lin_vel"].append(sim._base_lin_vel.numpy().copy())
            out["dr"].append(np.concatenate([sim._friction_values.numpy(), sim._added_base_mass.numpy(), sim._base_com_bias.numpy(),
                                             sim._rand_push_vels.numpy()[:, :2]], 1).copy())
            out["cmd_range_x"].append(np.array(env.command_ranges["lin_vel_x"], np.float32))
        arrays = {k: np.stack(v) for k, v in out.items()}
        arrays.update({"script_" + k: v for k, v in sim.script.items()})
        arrays.update({"init_" + k: v for k, v in init.items()})
        arrays["reward_names"] = np.array(names)
        path = os.path.join(HERE, "go2_mdp.npz")
        np.savez_compressed(path, **arrays)
        print("wrote", path, {k: v.shape for k, v in arrays.items() if k in ("obs", "rand", "episode_sums")},
              "resets/step", arrays["reset"].sum(1)[:40], "cmd_range_x", arrays["cmd_range_x"][-1])
    finally:
        torch.rand_like = orig_rand_like


def gen_wtw(N=24, T=64, seed=11):
    """GO2WTW (go2_wtw.py): periodic-gait rewards, behaviour parameters, 5-frame histories.
    The command-curriculum gate step is NOT crossed: the reference calls a method that does not
    exist there (`self.update_command_curriculum`, go2_wtw.py:121) and would raise."""
    import legged_gym.envs.base.base_task as base_task
    import legged_gym.envs.base.legged_robot as lr_mod
    import legged_gym.envs.go2.go2_wtw.go2_wtw as wtw_mod
    from legged_gym.envs.go2.go2_wtw.go2_wtw_config import GO2WTWCfg
    from legged_gym.utils.helpers import class_to_dict
    from hcr_genesis_lr_cl_amd.config import GO2WTWCfg as MyCfg

    rec = rh.DrawRecorder(seed)
    base_task.GenesisSimulator = FakeSimulator
    lr_mod.torch_rand_float = rec.rand_float
    wtw_mod.torch_rand_float = rec.rand_float
    orig_rand_like, orig_randint = torch.rand_like, torch.randint
    torch.rand_like, torch.randint = rec.rand_like, rec.randint
    try:
        cfg = GO2WTWCfg()
        cfg.env.num_envs = N
        env = wtw_mod.GO2WTW(cfg, class_to_dict(cfg.sim), "cpu", True)
        sim = env.simulator
        sim.rec = rec
        rng = np.random.default_rng(seed + 1)
        model = sim.model
        sim.script = make_script(rng, model, cfg, N, T)
        task = builders.make_task_cfg(model, MyCfg())
        slots = task.slots
        groups = [list(range(12))]                                        # legged_robot.py:279-280: one (n,12) draw
        env.episode_length_buf[:] = torch.from_numpy(rng.choice(
            [3, 120, 245, 248, 249, 395, 398, 399, 498, 499, 748, 960, 985, 995, 998, 999, 1000], N).astype(np.int32))
        env.commands[:] = torch.from_numpy((rng.normal(size=(N, 4)) * [0.4, 0.4, 0.5, 1.5]).astype(np.float32))
        env.common_step_counter = 745
        env.reset_buf[:] = 0
        # widen the behaviour ranges / gait set as the curriculum would, so every table entry is exercised
        env.num_gaits = 4
        env.gait_period_range = [0.3, 0.6]
        env.base_height_target_range = [0.2, 0.34]
        env.foot_clearance_target_range = [0.04, 0.12]
        env.pitch_target_range = [-0.3, 0.3]
        env.theta[:] = torch.from_numpy(rng.choice([0.0, 0.5], (N, 4)).astype(np.float32))
        env.gait_period[:] = torch.from_numpy(rng.uniform(0.3, 0.6, (N, 1)).astype(np.float32))
        env.gait_time[:] = torch.from_numpy(rng.uniform(0.0, 0.3, (N, 1)).astype(np.float32))
        env.phi[:] = env.gait_time / env.gait_period
        rec.take()
        init = dict(episode_length_buf=env.episode_length_buf.numpy().copy(), commands=env.commands.numpy().copy(),
                    env_origins=sim._env_origins.numpy().copy(), theta=env.theta.numpy().copy(),
                    gait_period=env.gait_period.numpy().copy(), gait_time=env.gait_time.numpy().copy(), phi=env.phi.numpy().copy(),
                    behavior_ranges=np.array(env.gait_period_range + env.base_height_target_range +
                                             env.foot_clearance_target_range + env.pitch_target_range + [env.num_gaits], np.float32))
        keys = ("actions_in", "rand", "counter", "obs", "priv", "rew", "reset", "time_out", "commands", "ep_len", "fail_buf",
                "episode_sums", "act_hist", "sim_dof_pos", "sim_base_pos", "sim_base_lin_vel_w", "dr_pd", "task_state",
                "last_dof_vel_in", "last_feet_vel_in", "esum_override")
        out = {k: [] for k in keys}
        names = env.reward_names
        for t in range(T):
            act = torch.from_numpy((rng.normal(size=(N, 12)) * (1.0 if t % 7 else 60.0)).astype(np.float32))
            out["last_dof_vel_in"].append(sim._dof_vel.numpy().copy())
            out["last_feet_vel_in"].append(sim._feet_vel.numpy().copy())
            obs, priv, rew, reset, extras = env.step(act)
            calls = rec.take()
            out["actions_in"].append(act.numpy().copy())
            out["rand"].append(slots_from_calls(calls, slots, N, 12, groups))
            out["counter"].append(env.common_step_counter); out["esum_override"].append(0.0)
            out["obs"].append(obs.numpy().copy()); out["priv"].append(priv.numpy().copy()); out["rew"].append(rew.numpy().copy())
            out["reset"].append(reset.numpy().astype(np.uint8)); out["time_out"].append(env.time_out_buf.numpy().astype(np.uint8))
            out["commands"].append(env.commands.numpy().copy()); out["ep_len"].append(env.episode_length_buf.numpy().copy())
            out["fail_buf"].append(env.fail_buf.numpy().copy())
            out["episode_sums"].append(np.stack([env.episode_sums[n].numpy().copy() for n in names]))
            out["act_hist"].append(np.stack([env.actions.numpy(), env.last_actions.numpy(), env.llast_actions.numpy()]).copy())
            out["sim_dof_pos"].append(sim._dof_pos.numpy().copy()); out["sim_base_pos"].append(sim._base_pos.numpy().copy())
            out["sim_base_lin_vel_w"].append(sim._base_lin_vel_w.numpy().copy())
            out["dr_pd"].append(np.concatenate([sim._kp_scale.numpy(), sim._kd_scale.numpy()], 1).copy())
            out["task_state"].append(np.concatenate([env.gait_time.numpy(), env.phi.numpy(), env.gait_period.numpy(),
                                                     env.base_height_target.numpy(), env.foot_clearance_target.numpy(),
                                                     env.pitch_target.numpy(), env.theta.numpy(), env.clock_input.numpy(),
                                                     env.exp_C_frc_fl.numpy(), env.exp_C_frc_fr.numpy(), env.exp_C_frc_rl.numpy(),
                                                     env.exp_C_frc_rr.numpy()], 1).copy())
        arrays = {k: np.stack(v) for k, v in out.items()}
        arrays.update({"script_" + k: v for k, v in sim.script.items()})
        arrays.update({"init_" + k: v for k, v in init.items()})
        arrays["reward_names"] = np.array(names)
        path = os.path.join(HERE, "go2_wtw_mdp.npz")
        np.savez_compressed(path, **arrays)
        print("wrote", path, arrays["obs"].shape, arrays["priv"].shape, "resets/step", arrays["reset"].sum(1)[:40])
    finally:
        torch.rand_like, torch.randint = orig_rand_like, orig_randint


def gen_ee(N=24, T=48, seed=21):
    """Go2EE (go2_ee.py, legged_robot_ee.py): heightfield terrain, terrain curriculum, estimator features /
    labels / critic stacks.  Only the newest frame of each stack is stored per step (plus the full
    stacks at the last step) to keep the fixture small; the stacking itself is checked on those."""
    import legged_gym.envs.base.base_task as base_task
    import legged_gym.envs.base.legged_robot as lr_mod
    import legged_gym.envs.go2.go2_ee.go2_ee as ee_mod
    from legged_gym.envs.go2.go2_ee.go2_ee_config import Go2EECfg
    from legged_gym.utils.helpers import class_to_dict
    from hcr_genesis_lr_cl_amd.config import GO2EECfg as MyCfg

    rec = rh.DrawRecorder(seed)
    base_task.GenesisSimulator = RoughFakeSimulator
    lr_mod.torch_rand_float = rec.rand_float
    ee_mod.torch_rand_float = rec.rand_float
    orig_rand_like = torch.rand_like
    torch.rand_like = rec.rand_like
    try:
        cfg = Go2EECfg()
        cfg.env.num_envs = N
        env = ee_mod.Go2EE(cfg, class_to_dict(cfg.sim), "cpu", True)
        sim = env.simulator
        sim.rec = rec
        rng = np.random.default_rng(seed + 1)
        model = sim.model
        script = make_script(rng, model, cfg, N, T)
        # place the robots on their tiles: around the env origin, some far enough to be promoted (> 4 m)
        org = sim._env_origins.numpy()
        off = rng.normal(size=(T, N, 2)) * 1.5 + np.where(rng.random((T, N, 1)) < 0.3, 4.5, 0.0)
        script["base_pos"][:, :, :2] = (org[None, :, :2] + off).astype(np.float32)
        script["base_pos"][:, :, 2] += org[None, :, 2]
        script["feet_pos"][:, :, :, :2] += script["base_pos"][:, :, None, :2]
        script["feet_pos"][:, :, :, 2] += org[None, :, None, 2]
        sim.script = script
        task = builders.make_task_cfg(model, MyCfg())
        slots = task.slots
        groups = [[0, 3, 6, 9], [1, 4, 7, 10], [2, 5, 8, 11]]
        env.episode_length_buf[:] = torch.from_numpy(rng.choice([3, 120, 470, 495, 498, 499, 960, 985, 995, 998, 999, 1000], N).astype(np.int32))
        env.commands[:] = torch.from_numpy((rng.normal(size=(N, 4)) * [0.4, 0.4, 0.5, 1.5]).astype(np.float32))
        env.common_step_counter = 495                                   # push interval 10 s = 500 steps
        env.reset_buf[:] = 0
        rec.take()
        init = dict(episode_length_buf=env.episode_length_buf.numpy().copy(), commands=env.commands.numpy().copy(),
                    env_origins=sim._env_origins.numpy().copy(), terrain_levels=sim._terrain_levels.numpy().copy(),
                    terrain_types=sim._terrain_types.numpy().copy(), height_points=sim._height_points[0, :, :2].numpy().copy())
        keys = ("actions_in", "rand", "counter", "feat_new", "priv_new", "labels", "rew", "reset", "time_out", "commands", "ep_len",
                "fail_buf", "feet_air_time", "episode_sums", "sim_dof_pos", "sim_base_pos", "terrain_levels", "env_origins",
                "measured_heights", "height_around_feet", "normals", "contact_states", "last_dof_vel_in", "last_feet_vel_in", "esum_override")
        out = {k: [] for k in keys}
        names = env.reward_names
        for t in range(T):
            act = torch.from_numpy((rng.normal(size=(N, 12)) * (1.0 if t % 7 else 60.0)).astype(np.float32))
            out["last_dof_vel_in"].append(sim._dof_vel.numpy().copy()); out["last_feet_vel_in"].append(sim._feet_vel.numpy().copy())
            feat, labels, priv, rew, reset, extras = env.step(act)
            calls = rec.take()
            out["actions_in"].append(act.numpy().copy()); out["rand"].append(slots_from_calls(calls, slots, N, 12, groups))
            out["counter"].append(env.common_step_counter); out["esum_override"].append(0.0)
            out["feat_new"].append(feat.numpy()[:, -45:].copy()); out["priv_new"].append(priv.numpy()[:, -174:].copy())
            out["labels"].append(labels.numpy().copy()); out["rew"].append(rew.numpy().copy())
            out["reset"].append(reset.numpy().astype(np.uint8)); out["time_out"].append(env.time_out_buf.numpy().astype(np.uint8))
            out["commands"].append(env.commands.numpy().copy()); out["ep_len"].append(env.episode_length_buf.numpy().copy())
            out["fail_buf"].append(env.fail_buf.numpy().copy()); out["feet_air_time"].append(env.feet_air_time.numpy().copy())
            out["episode_sums"].append(np.stack([env.episode_sums[n].numpy().copy() for n in names]))
            out["sim_dof_pos"].append(sim._dof_pos.numpy().copy()); out["sim_base_pos"].append(sim._base_pos.numpy().copy())
            out["terrain_levels"].append(sim._terrain_levels.numpy().copy()); out["env_origins"].append(sim._env_origins.numpy().copy())
            out["measured_heights"].append(sim._measured_heights.numpy().copy())
            out["height_around_feet"].append(sim._height_around_feet.numpy().copy())
            out["normals"].append(sim._normal_vector_around_feet.numpy().copy())
            out["contact_states"].append(sim._link_contact_states.numpy().copy())
        arrays = {k: np.stack(v) for k, v in out.items()}
        arrays["feat_last"], arrays["priv_last"] = feat.numpy().copy(), priv.numpy().copy()
        arrays.update({"script_" + k: v for k, v in sim.script.items()})
        arrays.update({"init_" + k: v for k, v in init.items()})
        arrays["reward_names"] = np.array(names)
        arrays["terrain_seed"] = RoughFakeSimulator.TERRAIN_SEED
        path = os.path.join(HERE, "go2_ee_mdp.npz")
        np.savez_compressed(path, **arrays)
        print("wrote", path, os.path.getsize(path), "resets/step", arrays["reset"].sum(1), "levels moved",
              int((arrays["terrain_levels"][-1] != init["terrain_levels"]).sum()))
    finally:
        torch.rand_like = orig_rand_like


def gen_tron1(N=24, T=48, seed=31):
    """TRON1PF_EE (tron1_pf_ee.py): 6-DOF biped, heightfield + curriculum, biped periodic gait, sit-pose resets,
    all domain randomisation incl. joint armature / friction / damping.  The command-curriculum gate step is not
    crossed (the reference calls the non-existent `update_command_curriculum`, tron1_pf_ee.py:201)."""
    import legged_gym.envs.base.base_task as base_task
    import legged_gym.envs.base.legged_robot as lr_mod
    import legged_gym.envs.tron1_pf.tron1_pf_ee.tron1_pf_ee as tr_mod
    from legged_gym.envs.tron1_pf.tron1_pf_ee.tron1_pf_ee_config import TRON1PF_EECfg
    from legged_gym.utils.helpers import class_to_dict
    from hcr_genesis_lr_cl_amd.config import TRON1PFEECfg as MyCfg

    rec = rh.DrawRecorder(seed)
    base_task.GenesisSimulator = RoughFakeSimulator
    lr_mod.torch_rand_float = rec.rand_float
    tr_mod.torch_rand_float = rec.rand_float
    orig = (torch.rand_like, torch.rand, np.random.random)
    torch.rand_like, torch.rand, np.random.random = rec.rand_like, rec.torch_rand, rec.np_random
    try:
        cfg = TRON1PF_EECfg()
        cfg.env.num_envs = N
        torch.rand = orig[1]                      # construction uses torch.rand-free paths; keep the original until stepping
        env = tr_mod.TRON1PF_EE(cfg, class_to_dict(cfg.sim), "cpu", True)
        torch.rand = rec.torch_rand
        sim = env.simulator
        sim.rec = rec
        rng = np.random.default_rng(seed + 1)
        model = sim.model
        script = make_script(rng, model, cfg, N, T)
        org = sim._env_origins.numpy()
        off = rng.normal(size=(T, N, 2)) * 1.5 + np.where(rng.random((T, N, 1)) < 0.3, 4.5, 0.0)
        script["base_pos"][:, :, :2] = (org[None, :, :2] + off).astype(np.float32)
        script["base_pos"][:, :, 2] += org[None, :, 2] + 0.4
        script["feet_pos"][:, :, :, :2] = script["feet_pos"][:, :, :, :2] * 0.4 + script["base_pos"][:, :, None, :2]
        script["feet_pos"][:, :, :, 2] += org[None, :, None, 2]
        sim.script = script
        task = builders.make_task_cfg(model, MyCfg())
        slots = task.slots
        groups = [[0, 3], [1, 4], [2, 5]]                                # tron1_pf_ee.py:268-273
        env.episode_length_buf[:] = torch.from_numpy(rng.choice([3, 120, 470, 495, 498, 499, 960, 985, 995, 998, 999, 1000], N).astype(np.int32))
        env.commands[:] = torch.from_numpy((rng.normal(size=(N, 4)) * [0.4, 0.4, 0.5, 1.5]).astype(np.float32))
        env.common_step_counter = 495
        env.reset_buf[:] = 0
        env.theta[:, 0] = torch.from_numpy(rng.uniform(0, 1, N).astype(np.float32)); env.theta[:, 1] = env.theta[:, 0] + 0.5
        env.gait_time[:] = torch.from_numpy(rng.uniform(0.0, 0.45, (N, 1)).astype(np.float32))
        env.phi[:] = env.gait_time / env.gait_period
        rec.take()
        init = dict(episode_length_buf=env.episode_length_buf.numpy().copy(), commands=env.commands.numpy().copy(),
                    env_origins=sim._env_origins.numpy().copy(), terrain_levels=sim._terrain_levels.numpy().copy(),
                    terrain_types=sim._terrain_types.numpy().copy(), height_points=sim._height_points[0, :, :2].numpy().copy(),
                    theta=env.theta.numpy().copy(), gait_time=env.gait_time.numpy().copy(), phi=env.phi.numpy().copy())
        keys = ("actions_in", "rand", "counter", "feat_new", "priv_new", "labels", "rew", "reset", "time_out", "commands", "ep_len",
                "fail_buf", "episode_sums", "act_hist", "sim_dof_pos", "sim_base_pos", "sim_base_quat", "terrain_levels", "env_origins",
                "measured_heights", "height_around_feet", "normals", "dr_joint", "task_state", "last_dof_vel_in", "last_feet_vel_in", "esum_override")
        out = {k: [] for k in keys}
        names = env.reward_names
        for t in range(T):
            act = torch.from_numpy((rng.normal(size=(N, 6)) * (1.0 if t % 7 else 30.0)).astype(np.float32))
            out["last_dof_vel_in"].append(sim._dof_vel.numpy().copy()); out["last_feet_vel_in"].append(sim._feet_vel.numpy().copy())
            feat, labels, priv, rew, reset, extras = env.step(act)
            calls = rec.take()
            out["actions_in"].append(act.numpy().copy()); out["rand"].append(slots_from_calls(calls, slots, N, 6, groups))
            out["counter"].append(env.common_step_counter); out["esum_override"].append(0.0)
            out["feat_new"].append(feat.numpy()[:, -31:].copy()); out["priv_new"].append(priv.numpy()[:, -134:].copy())
            out["labels"].append(labels.numpy().copy()); out["rew"].append(rew.numpy().copy())
            out["reset"].append(reset.numpy().astype(np.uint8)); out["time_out"].append(env.time_out_buf.numpy().astype(np.uint8))
            out["commands"].append(env.commands.numpy().copy()); out["ep_len"].append(env.episode_length_buf.numpy().copy())
            out["fail_buf"].append(env.fail_buf.numpy().copy())
            out["episode_sums"].append(np.stack([env.episode_sums[n].numpy().copy() for n in names]))
            out["act_hist"].append(np.stack([env.actions.numpy(), env.last_actions.numpy(), env.llast_actions.numpy()]).copy())
            out["sim_dof_pos"].append(sim._dof_pos.numpy().copy()); out["sim_base_pos"].append(sim._base_pos.numpy().copy())
            out["sim_base_quat"].append(sim._base_quat.numpy().copy())
            out["terrain_levels"].append(sim._terrain_levels.numpy().copy()); out["env_origins"].append(sim._env_origins.numpy().copy())
            out["measured_heights"].append(sim._measured_heights.numpy().copy())
            out["height_around_feet"].append(sim._height_around_feet.numpy().copy())
            out["normals"].append(sim._normal_vector_around_feet.numpy().copy())
            out["dr_joint"].append(np.concatenate([sim._joint_armature.numpy(), sim._joint_friction.numpy(), sim._joint_damping.numpy()], 1).copy())
            out["task_state"].append(np.concatenate([env.gait_time.numpy(), env.phi.numpy(), env.theta.numpy(), env.clock_input.numpy(),
                                                     env.exp_C_frc_left.numpy(), env.exp_C_frc_right.numpy()], 1).copy())
        arrays = {k: np.stack(v) for k, v in out.items()}
        arrays["feat_last"], arrays["priv_last"] = feat.numpy().copy(), priv.numpy().copy()
        arrays.update({"script_" + k: v for k, v in sim.script.items()})
        arrays.update({"init_" + k: v for k, v in init.items()})
        arrays["reward_names"] = np.array(names)
        arrays["terrain_seed"] = RoughFakeSimulator.TERRAIN_SEED
        path = os.path.join(HERE, "tron1_pf_ee_mdp.npz")
        np.savez_compressed(path, **arrays)
        sit = (np.abs(arrays["sim_base_quat"][:, :, 1]) > 0.05) & arrays["reset"].astype(bool)
        print("wrote", path, os.path.getsize(path), "resets/step", arrays["reset"].sum(1), "sit resets", int(sit.sum()),
              "levels moved", int((arrays["terrain_levels"][-1] != init["terrain_levels"]).sum()))
    finally:
        torch.rand_like, torch.rand, np.random.random = orig


def gen_tron1_pf(N=16, T=40, seed=51):
    """TRON1PF (tron1_pf.py, experiment "tron1_pf"): the point-foot biped on the plane; base-class resets, 5-frame actor /
    critic stacks, `no_fly` reward.  Stored like gen_wtw: full stacked obs / privileged obs per step."""
    import legged_gym.envs.base.base_task as base_task
    import legged_gym.envs.base.legged_robot as lr_mod
    import legged_gym.envs.tron1_pf.tron1_pf as pf_mod
    from legged_gym.envs.tron1_pf.tron1_pf_config import TRON1PFCfg
    from legged_gym.utils.helpers import class_to_dict
    from hcr_genesis_lr_cl_amd.config import TRON1PFCfg as MyCfg

    rec = rh.DrawRecorder(seed)
    base_task.GenesisSimulator = FakeSimulator
    lr_mod.torch_rand_float = rec.rand_float
    orig_rand_like = torch.rand_like
    torch.rand_like = rec.rand_like
    try:
        cfg = TRON1PFCfg()
        cfg.env.num_envs = N
        env = pf_mod.TRON1PF(cfg, class_to_dict(cfg.sim), "cpu", True)
        sim = env.simulator
        sim.rec = rec
        rng = np.random.default_rng(seed + 1)
        model = sim.model
        script = make_script(rng, model, cfg, N, T)
        script["base_pos"][:, :, 2] += 0.36                       # nominal base height 0.68
        script["feet_pos"][:, :, :, :2] = script["feet_pos"][:, :, :, :2] * 0.4 + script["base_pos"][:, :, None, :2]
        sim.script = script
        task = builders.make_task_cfg(model, MyCfg())
        slots = task.slots
        groups = [list(range(6))]                                  # legged_robot.py:279-280: one (n, 6) draw
        env.episode_length_buf[:] = torch.from_numpy(rng.choice([3, 120, 470, 495, 498, 499, 968, 977, 985, 992, 995, 998, 999, 1000], N).astype(np.int32))
        env.commands[:] = torch.from_numpy((rng.normal(size=(N, 4)) * [0.4, 0.4, 0.5, 1.5]).astype(np.float32))
        env.common_step_counter = 495                              # push interval 10 s = 500 steps
        env.reset_buf[:] = 0
        rec.take()
        init = dict(episode_length_buf=env.episode_length_buf.numpy().copy(), commands=env.commands.numpy().copy(),
                    env_origins=sim._env_origins.numpy().copy())
        keys = ("actions_in", "rand", "counter", "obs", "priv", "rew", "reset", "time_out", "commands", "ep_len", "fail_buf", "feet_air_time",
                "episode_sums", "act_hist", "sim_dof_pos", "sim_base_pos", "sim_base_lin_vel_w", "dr", "last_dof_vel_in", "last_feet_vel_in",
                "esum_override")
        out = {k: [] for k in keys}
        names = env.reward_names
        for t in range(T):
            act = torch.from_numpy((rng.normal(size=(N, 6)) * (1.0 if t % 7 else 60.0)).astype(np.float32))
            out["last_dof_vel_in"].append(sim._dof_vel.numpy().copy()); out["last_feet_vel_in"].append(sim._feet_vel.numpy().copy())
            obs, priv, rew, reset, extras = env.step(act)
            calls = rec.take()
            out["actions_in"].append(act.numpy().copy()); out["rand"].append(slots_from_calls(calls, slots, N, 6, groups))
            out["counter"].append(env.common_step_counter); out["esum_override"].append(0.0)
            out["obs"].append(obs.numpy().copy()); out["priv"].append(priv.numpy().copy()); out["rew"].append(rew.numpy().copy())
            out["reset"].append(reset.numpy().astype(np.uint8)); out["time_out"].append(env.time_out_buf.numpy().astype(np.uint8))
            out["commands"].append(env.commands.numpy().copy()); out["ep_len"].append(env.episode_length_buf.numpy().copy())
            out["fail_buf"].append(env.fail_buf.numpy().copy()); out["feet_air_time"].append(env.feet_air_time.numpy().copy())
            out["episode_sums"].append(np.stack([env.episode_sums[n].numpy().copy() for n in names]))
            out["act_hist"].append(np.stack([env.actions.numpy(), env.last_actions.numpy(), env.llast_actions.numpy()]).copy())
            out["sim_dof_pos"].append(sim._dof_pos.numpy().copy()); out["sim_base_pos"].append(sim._base_pos.numpy().copy())
            out["sim_base_lin_vel_w"].append(sim._base_lin_vel_w.numpy().copy())
            out["dr"].append(np.concatenate([sim._friction_values.numpy(), sim._added_base_mass.numpy(), sim._base_com_bias.numpy(),
                                             sim._rand_push_vels.numpy()[:, :2]], 1).copy())
        arrays = {k: np.stack(v) for k, v in out.items()}
        arrays.update({"script_" + k: v for k, v in sim.script.items()})
        arrays.update({"init_" + k: v for k, v in init.items()})
        arrays["reward_names"] = np.array(names)
        path = os.path.join(HERE, "tron1_pf_mdp.npz")
        np.savez_compressed(path, **arrays)
        print("wrote", path, os.path.getsize(path), arrays["obs"].shape, arrays["priv"].shape, "resets/step", arrays["reset"].sum(1), list(names))
    finally:
        torch.rand_like = orig_rand_like


def _mat_to_quat(R):
    """xyzw quaternions of rotation matrices (..., 3, 3), w >= 0 branch-free enough for the tilts scripted here."""
    w = np.sqrt(np.maximum(0.0, 1.0 + R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2])) / 2
    x = np.sqrt(np.maximum(0.0, 1.0 + R[..., 0, 0] - R[..., 1, 1] - R[..., 2, 2])) / 2 * np.sign(R[..., 2, 1] - R[..., 1, 2] + 1e-30)
    y = np.sqrt(np.maximum(0.0, 1.0 - R[..., 0, 0] + R[..., 1, 1] - R[..., 2, 2])) / 2 * np.sign(R[..., 0, 2] - R[..., 2, 0] + 1e-30)
    z = np.sqrt(np.maximum(0.0, 1.0 - R[..., 0, 0] - R[..., 1, 1] + R[..., 2, 2])) / 2 * np.sign(R[..., 1, 0] - R[..., 0, 1] + 1e-30)
    q = np.stack([x, y, z, w], -1)
    return (q / np.linalg.norm(q, axis=-1, keepdims=True)).astype(np.float32)


def gen_tron1_sf(N=16, T=44, seed=61):
    """TRON1SF (tron1_sf.py, experiment "tron1_sf"): the 8-DOF sole-foot biped on the plane: 10-frame actor / critic stacks, the
    6-DOF index pairs of its `_reset_dofs`, the batch-wide sit-pose coin, `hip_pos_zero_command` / `foot_flat` rewards (and
    `keep_ankle_pitch_zero_in_air`, given a scale here so that it is pinned too).  The foot orientation the class reads from
    rigid_body_states is scripted consistently with the scripted base orientation and joint angles (oracle.mdp_oracle.foot_rotations)."""
    import legged_gym.envs.base.base_task as base_task
    import legged_gym.envs.base.legged_robot as lr_mod
    import legged_gym.envs.tron1_sf.tron1_sf as sf_mod
    from legged_gym.envs.tron1_sf.tron1_sf_config import TRON1SFCfg
    from legged_gym.utils.helpers import class_to_dict
    from hcr_genesis_lr_cl_amd.config import TRON1SFCfg as MyCfg
    from oracle.mdp_oracle import foot_rotations

    rec = rh.DrawRecorder(seed)
    base_task.GenesisSimulator = FakeSimulator
    lr_mod.torch_rand_float = rec.rand_float
    sf_mod.torch_rand_float = rec.rand_float
    orig = (torch.rand_like, np.random.random)
    torch.rand_like, np.random.random = rec.rand_like, rec.np_random
    try:
        cfg = TRON1SFCfg()
        cfg.env.num_envs = N
        cfg.rewards.scales.keep_ankle_pitch_zero_in_air = 0.2     # defined by the class, unscaled in the shipped config
        env = sf_mod.TRON1SF(cfg, class_to_dict(cfg.sim), "cpu", True)
        sim = env.simulator
        sim.rec = rec
        rng = np.random.default_rng(seed + 1)
        model = sim.model
        script = make_script(rng, model, cfg, N, T)
        script["base_pos"][:, :, 2] += 0.43                       # nominal base height 0.75
        script["feet_pos"][:, :, :, :2] = script["feet_pos"][:, :, :, :2] * 0.4 + script["base_pos"][:, :, None, :2]
        script["foot_quat"] = np.stack([_mat_to_quat(foot_rotations(model, script["base_quat"][t], script["dof_pos"][t])) for t in range(T)])
        sim.script = script
        mycfg = MyCfg()
        mycfg.rewards.scales.keep_ankle_pitch_zero_in_air = 0.2
        task = builders.make_task_cfg(model, mycfg)
        slots = task.slots
        groups = [[0, 3], [1, 4], [2, 5], [3, 6]]                  # tron1_sf.py:224-231: the 6-DOF index pairs on the 8-DOF vector
        env.episode_length_buf[:] = torch.from_numpy(rng.choice([3, 120, 470, 495, 498, 499, 968, 977, 985, 992, 995, 998, 999, 1000], N).astype(np.int32))
        env.commands[:] = torch.from_numpy((rng.normal(size=(N, 4)) * [0.4, 0.4, 0.5, 1.5]).astype(np.float32))
        env.commands[::3, :3] *= 0.05                              # ~zero commands: hip_pos_zero_command, the air-time gate
        env.common_step_counter = 495                              # push interval 10 s = 500 steps
        env.reset_buf[:] = 0
        rec.take()
        init = dict(episode_length_buf=env.episode_length_buf.numpy().copy(), commands=env.commands.numpy().copy(),
                    env_origins=sim._env_origins.numpy().copy())
        keys = ("actions_in", "rand", "counter", "obs", "priv", "rew", "reset", "time_out", "commands", "ep_len", "fail_buf", "feet_air_time",
                "episode_sums", "act_hist", "sim_dof_pos", "sim_base_pos", "sim_base_quat", "sim_base_lin_vel_w", "dr", "dr_pd", "dr_joint",
                "last_dof_vel_in", "last_feet_vel_in", "esum_override")
        out = {k: [] for k in keys}
        names = env.reward_names
        for t in range(T):
            act = torch.from_numpy((rng.normal(size=(N, 8)) * (1.0 if t % 7 else 60.0)).astype(np.float32))
            out["last_dof_vel_in"].append(sim._dof_vel.numpy().copy()); out["last_feet_vel_in"].append(sim._feet_vel.numpy().copy())
            obs, priv, rew, reset, extras = env.step(act)
            calls = rec.take()
            out["actions_in"].append(act.numpy().copy()); out["rand"].append(slots_from_calls(calls, slots, N, 8, groups))
            out["counter"].append(env.common_step_counter); out["esum_override"].append(0.0)
            out["obs"].append(obs.numpy().copy()); out["priv"].append(priv.numpy().copy()); out["rew"].append(rew.numpy().copy())
            out["reset"].append(reset.numpy().astype(np.uint8)); out["time_out"].append(env.time_out_buf.numpy().astype(np.uint8))
            out["commands"].append(env.commands.numpy().copy()); out["ep_len"].append(env.episode_length_buf.numpy().copy())
            out["fail_buf"].append(env.fail_buf.numpy().copy()); out["feet_air_time"].append(env.feet_air_time.numpy().copy())
            out["episode_sums"].append(np.stack([env.episode_sums[n].numpy().copy() for n in names]))
            out["act_hist"].append(np.stack([env.actions.numpy(), env.last_actions.numpy(), env.llast_actions.numpy()]).copy())
            out["sim_dof_pos"].append(sim._dof_pos.numpy().copy()); out["sim_base_pos"].append(sim._base_pos.numpy().copy())
            out["sim_base_quat"].append(sim._base_quat.numpy().copy()); out["sim_base_lin_vel_w"].append(sim._base_lin_vel_w.numpy().copy())
            out["dr"].append(np.concatenate([sim._friction_values.numpy(), sim._added_base_mass.numpy(), sim._base_com_bias.numpy(),
                                             sim._rand_push_vels.numpy()[:, :2]], 1).copy())
            out["dr_pd"].append(np.concatenate([sim._kp_scale.numpy(), sim._kd_scale.numpy()], 1).copy())
            out["dr_joint"].append(np.concatenate([sim._joint_armature.numpy(), sim._joint_friction.numpy(), sim._joint_damping.numpy()], 1).copy())
        arrays = {k: np.stack(v) for k, v in out.items()}
        arrays.update({"script_" + k: v for k, v in sim.script.items()})
        arrays.update({"init_" + k: v for k, v in init.items()})
        arrays["reward_names"] = np.array(names)
        path = os.path.join(HERE, "tron1_sf_mdp.npz")
        np.savez_compressed(path, **arrays)
        sit = arrays["reset"].astype(bool) & (np.abs(arrays["sim_dof_pos"][:, :, 2] - 1.35) < 1e-6)
        print("wrote", path, os.path.getsize(path), arrays["obs"].shape, arrays["priv"].shape, "resets/step", arrays["reset"].sum(1),
              "sit resets", int(sit.sum()), list(names))
    finally:
        torch.rand_like, np.random.random = orig


def gen_head(name, N=16, T=36, seed=41):
    """The other Go2-rough heads -- Go2TS, Go2CTS, Go2Dreamwaq, Go2CaT (legged_gym/envs/__init__.py:82-86) -- run AS CONFIGURED
    on the rough fake simulator: same recording as gen_ee.  Stored per step: the clipped actor frame, the newest frame of the
    20-deep actor history and of the 5-deep critic stack, and the single-frame auxiliary output (TS / CTS / CaT: privileged
    encoder input; Dreamwaq: explicit labels | next state), plus the full stacks at the last step.  As configured the asset
    selects 17 contact-state links while the size fields assume 12 (go2_ts_config.py:8-14): what the class EMITS is pinned."""
    import importlib
    import legged_gym.envs.base.base_task as base_task
    import legged_gym.envs.base.legged_robot as lr_mod
    from legged_gym.utils.helpers import class_to_dict
    from hcr_genesis_lr_cl_amd import config as mycfg
    mod_name, cls_name, cfg_mod, cfg_name, my_name = {
        "go2_ts": ("legged_gym.envs.go2.go2_ts.go2_ts", "Go2TS", "legged_gym.envs.go2.go2_ts.go2_ts_config", "Go2TSCfg", "GO2TSCfg"),
        "go2_cts": ("legged_gym.envs.go2.go2_cts.go2_cts", "Go2CTS", "legged_gym.envs.go2.go2_cts.go2_cts_config", "Go2CTSCfg", "GO2CTSCfg"),
        "go2_dreamwaq": ("legged_gym.envs.go2.go2_dreamwaq.go2_dreamwaq", "Go2Dreamwaq", "legged_gym.envs.go2.go2_dreamwaq.go2_dreamwaq_config",
                         "Go2DreamwaqCfg", "GO2DreamwaqCfg"),
        "go2_cat": ("legged_gym.envs.go2.go2_cat.go2_cat", "Go2CaT", "legged_gym.envs.go2.go2_cat.go2_cat_config", "Go2CaTCfg", "GO2CaTCfg"),
    }[name]
    env_mod = importlib.import_module(mod_name)
    ref_cfg_cls = getattr(importlib.import_module(cfg_mod), cfg_name)
    rec = rh.DrawRecorder(seed)
    base_task.GenesisSimulator = RoughFakeSimulator
    lr_mod.torch_rand_float = rec.rand_float
    env_mod.torch_rand_float = rec.rand_float
    orig_rand_like = torch.rand_like
    torch.rand_like = rec.rand_like
    try:
        cfg = ref_cfg_cls()
        cfg.env.num_envs = N
        if hasattr(cfg.env, "num_teacher"):
            cfg.env.num_teacher = N // 4 * 3
        env = getattr(env_mod, cls_name)(cfg, class_to_dict(cfg.sim), "cpu", True)
        sim = env.simulator
        sim.rec = rec
        if name == "go2_cat":          # properties of the Simulator ABC the constraints read (simulator.py: torque_limits, dof_vel_limits)
            type(sim).torque_limits = property(lambda s_: torch.tensor(s_.model.arrays["effort"], dtype=torch.float))
            type(sim).dof_vel_limits = property(lambda s_: torch.tensor(s_._cfg.asset.dof_vel_limits, dtype=torch.float).unsqueeze(0))
        rng = np.random.default_rng(seed + 1)
        model = sim.model
        script = make_script(rng, model, cfg, N, T)
        org = sim._env_origins.numpy()
        off = rng.normal(size=(T, N, 2)) * 1.5 + np.where(rng.random((T, N, 1)) < 0.3, 4.5, 0.0)
        script["base_pos"][:, :, :2] = (org[None, :, :2] + off).astype(np.float32)
        script["base_pos"][:, :, 2] += org[None, :, 2]
        script["feet_pos"][:, :, :, :2] += script["base_pos"][:, :, None, :2]
        script["feet_pos"][:, :, :, 2] += org[None, :, None, 2]
        sim.script = script
        task = builders.make_task_cfg(model, getattr(mycfg, my_name)())
        slots = task.slots
        groups = [[0, 3, 6, 9], [1, 4, 7, 10], [2, 5, 8, 11]]
        env.episode_length_buf[:] = torch.from_numpy(rng.choice([3, 470, 495, 498, 499, 968, 972, 977, 981, 985, 989, 992, 995, 998, 999, 1000], N).astype(np.int32))
        env.commands[:] = torch.from_numpy((rng.normal(size=(N, 4)) * [0.4, 0.4, 0.5, 1.5]).astype(np.float32))
        env.commands[:3] = 0.0                                   # standing-still envs (feet_contact_stand_still, CaT's style constraint)
        env.common_step_counter = 495
        env.reset_buf[:] = 0
        env.extras.setdefault("episode", {})      # exists after the runner's env.reset(); go2_cts.py:96 / go2_cat.py:101 index it on every step
        rec.take()
        init = dict(episode_length_buf=env.episode_length_buf.numpy().copy(), commands=env.commands.numpy().copy(),
                    env_origins=sim._env_origins.numpy().copy(), terrain_levels=sim._terrain_levels.numpy().copy(),
                    terrain_types=sim._terrain_types.numpy().copy(), height_points=sim._height_points[0, :, :2].numpy().copy())
        keys = ("actions_in", "rand", "counter", "obs", "feat_new", "priv_new", "labels", "rew", "reset", "time_out", "commands", "ep_len",
                "fail_buf", "feet_air_time", "episode_sums", "sim_dof_pos", "sim_base_pos", "terrain_levels", "env_origins",
                "measured_heights", "height_around_feet", "normals", "contact_states", "last_dof_vel_in", "last_feet_vel_in", "esum_override",
                "cstr_prob", "cstr_sums")
        out = {k: [] for k in keys}
        names = env.reward_names
        cstr_names = ["torque", "dof_vel", "action_rate", "base_height", "collision", "feet_stumble", "dof_pos", "base_orientation", "stand_still"]
        for t in range(T):
            act = torch.from_numpy((rng.normal(size=(N, 12)) * (1.0 if t % 7 else 60.0)).astype(np.float32))
            out["last_dof_vel_in"].append(sim._dof_vel.numpy().copy()); out["last_feet_vel_in"].append(sim._feet_vel.numpy().copy())
            res = env.step(act)
            if name == "go2_dreamwaq":
                obs, critic, hist, explicit, nxt, rew, reset, extras = res
                aux = torch.cat([explicit, nxt], dim=-1)
            else:
                obs, aux, hist, critic, rew, reset, extras = res
            calls = rec.take()
            FW = critic.shape[1] - 4 * cfg.env.single_critic_obs_len if t < 4 else critic.shape[1] // 5   # newest frame width
            out["actions_in"].append(act.numpy().copy()); out["rand"].append(slots_from_calls(calls, slots, N, 12, groups))
            out["counter"].append(env.common_step_counter); out["esum_override"].append(0.0)
            out["obs"].append(obs.numpy().copy())
            out["feat_new"].append(hist.numpy()[:, -45:].copy()); out["priv_new"].append(critic.numpy()[:, -177:].copy())
            assert critic.shape[1] in (4 * cfg.env.single_critic_obs_len + 177, 3 * cfg.env.single_critic_obs_len + 2 * 177,
                                       2 * cfg.env.single_critic_obs_len + 3 * 177, cfg.env.single_critic_obs_len + 4 * 177, 5 * 177), critic.shape
            out["labels"].append(aux.numpy().copy()); out["rew"].append(rew.numpy().copy())
            out["reset"].append(reset.numpy().astype(np.uint8)); out["time_out"].append(env.time_out_buf.numpy().astype(np.uint8))
            out["commands"].append(env.commands.numpy().copy()); out["ep_len"].append(env.episode_length_buf.numpy().copy())
            out["fail_buf"].append(env.fail_buf.numpy().copy()); out["feet_air_time"].append(env.feet_air_time.numpy().copy())
            out["episode_sums"].append(np.stack([env.episode_sums[n].numpy().copy() for n in names]))
            out["sim_dof_pos"].append(sim._dof_pos.numpy().copy()); out["sim_base_pos"].append(sim._base_pos.numpy().copy())
            out["terrain_levels"].append(sim._terrain_levels.numpy().copy()); out["env_origins"].append(sim._env_origins.numpy().copy())
            out["measured_heights"].append(sim._measured_heights.numpy().copy())
            out["height_around_feet"].append(sim._height_around_feet.numpy().copy())
            out["normals"].append(sim._normal_vector_around_feet.numpy().copy())
            out["contact_states"].append(sim._link_contact_states.numpy().copy())
            if name == "go2_cat":
                out["cstr_prob"].append(env.cstr_prob.numpy().copy())
                out["cstr_sums"].append(np.stack([env.episode_sums["cstr_" + n].numpy().copy() for n in cstr_names]))
            else:
                out["cstr_prob"].append(np.zeros(N, np.float32)); out["cstr_sums"].append(np.zeros((0, N), np.float32))
        arrays = {k: np.stack(v) for k, v in out.items()}
        arrays["feat_last"], arrays["priv_last"] = hist.numpy().copy(), critic.numpy().copy()
        arrays.update({"script_" + k: v for k, v in sim.script.items()})
        arrays.update({"init_" + k: v for k, v in init.items()})
        arrays["reward_names"] = np.array(names)
        arrays["terrain_seed"] = RoughFakeSimulator.TERRAIN_SEED
        path = os.path.join(HERE, f"{name}_mdp.npz")
        np.savez_compressed(path, **arrays)
        print("wrote", path, os.path.getsize(path), "aux width", arrays["labels"].shape[-1], "critic stack", arrays["priv_last"].shape,
              "resets/step", arrays["reset"].sum(1), "levels moved", int((arrays["terrain_levels"][-1] != init["terrain_levels"]).sum()))
    finally:
        torch.rand_like = orig_rand_like


if __name__ == "__main__":
    which = sys.argv[1:] or ["go2", "wtw", "ee", "tron1", "go2_ts", "go2_cts", "go2_dreamwaq", "go2_cat", "tron1_pf", "tron1_sf"]
    if "tron1_sf" in which:
        gen_tron1_sf()
    if "tron1_pf" in which:
        gen_tron1_pf()
    for h in ("go2_ts", "go2_cts", "go2_dreamwaq", "go2_cat"):
        if h in which:
            gen_head(h)
    if "tron1" in which:
        gen_tron1()
    if "ee" in which:
        gen_ee()
    if "go2" in which:
        gen_go2()
    if "wtw" in which:
        gen_wtw()
