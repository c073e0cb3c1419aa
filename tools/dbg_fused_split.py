"""Diagnostic: one control step through the production launch(es) and through SIM + PRE|POST|RESET from the same state; prints which
columns of which outputs differ.  usage: python tools/dbg_fused_split.py task"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from hcr_genesis_lr_cl_amd import abi
from hcr_genesis_lr_cl_amd.envs import make_env
task = sys.argv[1] if len(sys.argv) > 1 else "go2_ee"
N = 64
e1, e2 = make_env(task, N, "cuda:0")[0], make_env(task, N, "cuda:0")[0]
e1.reset(); e2.reset()
g = torch.Generator(device="cuda"); g.manual_seed(5)
e1.episode_length_buf[:] = torch.randint(0, 1000, (N,), generator=g, device="cuda", dtype=torch.int32)
e1.common_step_counter = e2.common_step_counter = 480
for t in range(6):
    for k in e1._engine.buf.keys():
        e2._engine.buf.raw(k).copy_(e1._engine.buf.raw(k))
    e2.common_step_counter = e1.common_step_counter
    act = torch.randn(N, e1.num_actions, generator=g, device="cuda")
    e1.step(act)
    e2.common_step_counter += 1
    ca = float(e2.cfg.normalization.clip_actions)
    e2._engine.step(abi.PHASE_SIM, torch.clip(act, -ca, ca), e2.common_step_counter)
    e2._engine.step(abi.PHASE_PRE | abi.PHASE_POST | abi.PHASE_RESET, act, e2.common_step_counter)
    torch.cuda.synchronize()
    for k in e1._engine.buf.keys():
        a, b = e1._engine.buf[k], e2._engine.buf[k]
        if a.dtype != torch.float32 or k == "episode_done_sums":
            continue
        d = (a - b).abs().reshape(a.shape[0], -1) if a.dim() > 1 else (a - b).abs().reshape(-1, 1)
        if float(d.max()) > 2e-3:
            cols = torch.nonzero(d.max(0).values > 2e-3).flatten().tolist()
            fr = {"priv_obs_buf": int(e1._engine.task.priv_frame), "obs_buf": int(e1._engine.task.obs_frame)}.get(k)
            print(f"t={t} {k}: max {float(d.max()):.4f} cols {cols[:12]}{'...' if len(cols) > 12 else ''} n={len(cols)}" + (f" (mod frame: {sorted(set(c % fr for c in cols))[:20]})" if fr else ""), flush=True)
    if t == 0 and "priv_obs_buf" in e1._engine.buf:
        fr = int(e1._engine.task.priv_frame); st_ = int(e1._engine.task.priv_stack)
        a, b = e1._engine.buf["priv_obs_buf"], e2._engine.buf["priv_obs_buf"]
        o = (st_ - 1) * fr
        print("prod ", a[0, o + 90:o + 110].cpu().numpy().round(3))
        print("split", b[0, o + 90:o + 110].cpu().numpy().round(3))
        nzp = torch.nonzero((a[0] - b[0]).abs() > 1e-3).flatten().tolist(); print("diff cols", nzp[:10], len(nzp))
        full = a[0].cpu().numpy(); import numpy as _np; idx = _np.nonzero(_np.abs(full + 0.443) < 2e-3)[0]; print("prod cols with -0.443:", idx.tolist()[:100])
        raw = e1._engine.buf.raw("priv_obs_buf"); print("raw shape", tuple(raw.shape)); r0 = raw.reshape(raw.shape[0], -1)[0] if raw.dim() == 2 else raw[e1._engine.obs_set(), 0]; ii = torch.nonzero((r0 + 0.443).abs() < 2e-3).flatten().tolist(); print("raw idx", ii[:100])
        print("base z", float(e1._engine.buf["base_pos"][0, 2]), "mh", e1._engine.buf["measured_heights"][0, :8].cpu().numpy().round(3))
