"""Thin host wrapper around the C ABI (include/lgsim.h): owns the torch device buffers, binds
them, and enqueues launches on torch's current HIP stream.

PyTorch is plumbing here (device memory + streams); every number is produced by the HIP
kernels in csrc/.  There is no CPU path: constructing an Engine without the built library or
without a GPU raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import abi


def buffer_specs(A, L, F, K, P, num_obs, num_priv, num_labels, n_slots, hist, priv_hist, task_state):
    """name -> (trailing shape, dtype).  Shapes follow genesis_simulator.py:407-494,
    legged_robot.py:380-409 and base_task.py:29-37."""
    f, i32, i64, u8 = torch.float32, torch.int32, torch.int64, torch.uint8
    s = dict(
        base_pos=((3,), f), base_quat=((4,), f), base_lin_vel_w=((3,), f), base_ang_vel_w=((3,), f),
        dof_pos=((A,), f), dof_vel=((A,), f),
        friction_values=((1,), f), added_base_mass=((1,), f), base_com_bias=((3,), f),
        kp_scale=((A,), f), kd_scale=((A,), f),
        joint_armature=((1,), f), joint_friction=((1,), f), joint_damping=((1,), f),
        rand_push_vels=((3,), f), env_origins=((3,), f),
        base_lin_vel=((3,), f), base_ang_vel=((3,), f), projected_gravity=((3,), f), base_euler=((3,), f),
        last_base_lin_vel=((3,), f), last_base_ang_vel=((3,), f), last_dof_vel=((A,), f), last_feet_vel=((F, 3), f),
        torques=((A,), f), link_contact_forces=((L, 3), f), feet_pos=((F, 3), f), feet_vel=((F, 3), f),
        actions=((A,), f), last_actions=((A,), f), llast_actions=((A,), f), commands=((4,), f),
        feet_air_time=((F,), f), last_contacts=((F,), u8), episode_length_buf=((), i32), fail_buf=((), i64),
        reset_buf=((), u8), time_out_buf=((), u8), rew_buf=((), f), obs_buf=((num_obs,), f),
    )
    if K:
        s["link_contact_states"] = ((K,), f)
    if P:
        s["measured_heights"] = ((P,), f)
        s["height_around_feet"] = ((F, 9), f)
        s["normal_vector_around_feet"] = ((3 * F,), f)
        s["terrain_levels"] = ((), i32)
        s["terrain_types"] = ((), i32)
    if num_priv:
        s["priv_obs_buf"] = ((num_priv,), f)
    if num_labels:
        s["labels_buf"] = ((num_labels,), f)
    if hist:
        s["obs_hist"] = (hist, f)
    if priv_hist:
        s["priv_hist"] = (priv_hist, f)
    if task_state:
        s["task_state"] = ((task_state,), f)
    return s


class _Buffers(dict):
    """name -> device tensor.  The observation outputs (obs_buf, priv_obs_buf, labels_buf) exist in
    `LgTaskCfg.obs_sets` copies written alternately, and history-stacked ones live in rows with slack frames
    (LgTaskCfg.obs_slack): `buf["obs_buf"]` is the (N, stack*frame) view of the copy and window the most recent
    observation launch wrote (lg_obs_set, lg_obs_window), `buf.raw("obs_buf")` the whole (sets, N, row) allocation
    the C ABI is bound to."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.windows = {}     # name -> (frame width, stack)
        self.sets = set()     # names allocated as (obs_sets, N, ...)
        self.first_frame = lambda: 0
        self.current_set = lambda: 0

    def raw(self, k):
        return dict.__getitem__(self, k)

    def __getitem__(self, k):
        t = dict.__getitem__(self, k)
        if k in self.sets:
            t = t[self.current_set()]
        w = self.windows.get(k)
        if w is None:
            return t
        off = self.first_frame() * w[0]
        return t[:, off:off + w[0] * w[1]]

    def get(self, k, default=None):
        return self[k] if k in self else default

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]


class Engine:
    def __init__(self, model, desc, opts, task, n_envs, device="cuda:0", inject_rand=False):
        if not torch.cuda.is_available():
            raise RuntimeError("hcr_genesis_lr_cl_amd needs a HIP device (no CPU fallback)")
        self.lib = abi.load_lib()
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.model, self.desc, self.opts, self.task = model, desc, opts, task
        self.n = int(n_envs)
        self._obs_slot = (0, 0)
        A, L, F = model.n_dof, model.n_links, model.n_legs
        K = bin(desc.state_link_mask).count("1")
        P = int(opts.n_height_points)
        hist = (task.obs_stack, task.obs_frame) if task.obs_stack > 1 else None
        phist = (task.priv_stack, task.priv_frame) if task.priv_stack > 1 else None
        specs = buffer_specs(A, L, F, K, P, task.num_obs, task.num_priv_obs, int(task.num_labels), task.slots.n_slots,
                             None, None, int(task.task_state_width))
        SL = int(task.obs_slack)
        if SL:   # rows of (stack + slack) frames; the observation is a sliding window over them
            specs["obs_buf"] = (((task.obs_stack + SL) * task.obs_frame,), torch.float32)
            if task.num_priv_obs:
                specs["priv_obs_buf"] = (((task.priv_stack + SL) * task.priv_frame,), torch.float32)
        sets = int(task.obs_sets) if int(task.obs_sets) > 1 else 1
        outs = [k for k in ("obs_buf", "priv_obs_buf", "labels_buf") if k in specs]
        if sets == 2 and SL:
            specs["obs_dirty"] = ((), torch.uint8)
        self.buf = _Buffers({k: torch.zeros(((sets,) if k in outs else ()) + (self.n,) + tuple(shape), dtype=dt, device=self.device)
                             for k, (shape, dt) in specs.items()})
        self.buf.sets = set(outs)
        self.buf.current_set = self.obs_set
        if SL:
            self.buf.windows["obs_buf"] = (int(task.obs_frame), int(task.obs_stack))
            if task.num_priv_obs:
                self.buf.windows["priv_obs_buf"] = (int(task.priv_frame), int(task.priv_stack))
            self.buf.first_frame = self.obs_window
        self.buf["episode_sums"] = torch.zeros((abi.R_COUNT, self.n), device=self.device)
        self.buf["episode_done_sums"] = torch.zeros((abi.R_COUNT, self.n), device=self.device)
        self.buf["episode_done_step"] = torch.full((self.n,), -1, dtype=torch.int32, device=self.device)
        self.buf["command_ranges"] = torch.zeros(abi.CMD_RANGE_FLOATS, device=self.device)
        self.buf["nonfinite_count"] = torch.zeros(1, dtype=torch.int32, device=self.device)   # LgBuffers.nonfinite_count
        if int(task.cat_enable):
            self.buf["cstr_prob"] = torch.zeros(self.n, device=self.device)
            self.buf["cstr_sums"] = torch.zeros((abi.NUM_CSTR, self.n), device=self.device)
            self.buf["cstr_done_sums"] = torch.zeros((abi.NUM_CSTR, self.n), device=self.device)
        if inject_rand:
            self.buf["rand_in"] = torch.zeros((self.n, task.slots.n_slots), device=self.device)
        b = self.buf
        b["base_quat"][:, 3] = 1.0
        b["friction_values"].fill_(1.0)
        b["kp_scale"].fill_(1.0)
        b["kd_scale"].fill_(1.0)
        b["added_base_mass"].fill_(0.0)
        self.handle = C.c_void_p()
        abi.check(self.lib.lg_create(C.byref(desc), C.byref(opts), C.byref(task), C.byref(self.handle)), self.lib)
        self.bind()

    def bind(self):
        lb = abi.LgBuffers()
        lb.n_envs = self.n
        for name in abi.BUFFER_NAMES:
            t = self.buf.raw(name) if name in self.buf else None
            if t is not None:
                assert t.is_contiguous()
                setattr(lb, name, t.data_ptr())
        self._lb = lb
        abi.check(self.lib.lg_bind(self.handle, C.byref(lb)), self.lib)

    def set_task(self, task):
        self.task = task
        abi.check(self.lib.lg_set_task(self.handle, C.byref(task)), self.lib)

    def set_terrain(self, height_samples, terrain_origins=None, height_points=None):
        """Upload the int16 heightfield (+ per-tile origins, + body-frame height sample offsets)."""
        hs = torch.as_tensor(height_samples).to(self.device, torch.int16).contiguous()
        self.height_samples = hs
        abi.check(self.lib.lg_set_terrain(self.handle, hs.data_ptr(), hs.shape[0], hs.shape[1]), self.lib)
        if terrain_origins is not None:
            self.buf["terrain_origins"] = torch.as_tensor(terrain_origins).to(self.device, torch.float32).contiguous()
        if height_points is not None:
            self.buf["height_points"] = torch.as_tensor(height_points).to(self.device, torch.float32).contiguous()
        self.bind()

    def step(self, phases, actions, counter):
        a = 0
        if actions is not None:
            if actions.dtype != torch.float32 or not actions.is_contiguous() or actions.device != self.device:
                actions = actions.to(self.device, torch.float32).contiguous()
            if tuple(actions.shape) != (self.n, self.model.n_dof):
                raise ValueError(f"actions must be ({self.n}, {self.model.n_dof}), got {tuple(actions.shape)}")
            a = actions.data_ptr()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        abi.check(self.lib.lg_step(self.handle, phases, a, int(counter), stream), self.lib)
        if phases & abi.PHASE_RESET:
            self._refresh_obs_slot()

    def time_steps(self, actions, first_counter, count):
        ms = C.c_float()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        abi.check(self.lib.lg_time_steps(self.handle, actions.data_ptr(), int(first_counter), int(count), stream,
                                         C.byref(ms)), self.lib)
        self._refresh_obs_slot()
        return ms.value

    def _refresh_obs_slot(self):
        """Which copy / window of the observation buffers the latest observation launch wrote (cached: the views are
        looked up on every step)."""
        w, s_ = C.c_int32(), C.c_int32()
        abi.check(self.lib.lg_obs_window(self.handle, C.byref(w)), self.lib)
        abi.check(self.lib.lg_obs_set(self.handle, C.byref(s_)), self.lib)
        self._obs_slot = (s_.value, w.value)

    def obs_window(self):
        """First frame of the window holding the latest stacked observation (0 without history slack)."""
        return self._obs_slot[1]

    def obs_set(self):
        """Copy of the observation buffers the latest observation launch wrote (0 with a single set)."""
        return self._obs_slot[0]

    # ---- checkpointing: the reference never saves simulator or curriculum state; here a run
    #      can be resumed exactly, because every piece of state is a caller-owned buffer plus two integers in the handle ----
    def state_dict(self):
        """All device buffers (the whole allocations, both observation copies and the history slack) and the position of the
        observation window / copy.  Random draws need no state: Philox is keyed on (seed, global env id, step counter, slot)."""
        sd = {"buffers": {k: self.buf.raw(k).detach().clone() for k in self.buf.keys()},
              "obs_slot": tuple(self._obs_slot), "n_envs": self.n}
        if getattr(self, "height_samples", None) is not None:
            sd["height_samples"] = self.height_samples.detach().clone()     # the int16 heightfield the state was produced on
        return sd

    def load_state_dict(self, sd):
        if int(sd["n_envs"]) != self.n:
            raise ValueError("checkpoint was taken with a different number of envs")
        hs = sd.get("height_samples")
        if (hs is None) != (getattr(self, "height_samples", None) is None) or (hs is not None and not torch.equal(hs.to(self.device), self.height_samples)):
            raise ValueError("checkpoint was taken on a different terrain (heightfield differs): build the env with the same terrain seed")
        for k, v in sd["buffers"].items():
            if k not in self.buf:
                raise KeyError(f"checkpoint buffer {k!r} does not exist in this engine")
            dst = self.buf.raw(k)
            if dst.shape != v.shape or dst.dtype != v.dtype:
                raise ValueError(f"checkpoint buffer {k!r}: {tuple(v.shape)} {v.dtype} != {tuple(dst.shape)} {dst.dtype}")
            dst.copy_(v)
        s_, w = sd["obs_slot"]
        abi.check(self.lib.lg_obs_set_select(self.handle, int(s_)), self.lib)
        abi.check(self.lib.lg_obs_window_select(self.handle, int(w)), self.lib)
        self._refresh_obs_slot()

    def nonfinite_count(self):
        """Envs re-seated by the non-finite guard of the physics phase so far (include/lgsim.h LgBuffers.nonfinite_count): 0 in a healthy
        run.  Synchronises."""
        return int(self.buf["nonfinite_count"].item())

    def last_kernel(self):
        """Launcher instantiation(s) of the latest step (include/lgsim.h lg_last_kernel)."""
        return self.lib.lg_last_kernel(self.handle).decode()

    def profile(self, stride):
        """Time the physics kernel of every `stride`-th step with HIP events (0 = off)."""
        abi.check(self.lib.lg_profile(self.handle, int(stride)), self.lib)

    def profile_read(self):
        us, n = C.c_float(), C.c_int32()
        abi.check(self.lib.lg_profile_read(self.handle, C.byref(us), C.byref(n)), self.lib)
        return us.value, n.value

    def close(self):
        if getattr(self, "handle", None):
            self.lib.lg_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
