"""Diagnostic (LG_LIB=dbg/lg_STAMPS.so): cycle stamps of workgroup 0 through the fused biped step kernel and the per-workgroup durations
(start -> physics done -> end) of one launch, split by whether the workgroup had a reset in it."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hcr_genesis_lr_cl_amd.envs import make_env
NE = int(os.environ.get("NE", "4096"))
TASK = os.environ.get("TASK", "tron1_pf_ee")
env, cfg = make_env(TASK, NE)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
env.episode_length_buf[:] = torch.randint(0, 1000, (NE,), generator=g, device="cuda", dtype=torch.int32)
bank = [torch.randn(NE, env.num_actions, generator=g, device="cuda") for _ in range(8)]
acc = torch.zeros(32); n = 0
phys, tail, rs_, roles = [], [], [], []
for i in range(400):
    env.step(bank[i % 8])
    if i >= 300:
        torch.cuda.synchronize()
        d = env._engine.buf["episode_done_sums"].flatten().cpu()
        acc += d[:32]; n += 1
        ngrp = NE * 2 * 4 // 64                 # groups of 8 envs = workgroups (two waves each: lg_quad.h DUO; the stamps are role 0's)
        nwg = ngrp
        t0, t1, t2 = d[4096:4096 + nwg], d[8192:8192 + nwg], d[12288:12288 + nwg]
        m = float(1 << 24)
        phys.append(((t1 - t0) % m)); tail.append(((t2 - t1) % m))
        # stamp slot = blockIdx; the block's envs are those of workgroup lg_wg(blockIdx) (XCD-aware index, lg_kernel.h)
        b = torch.arange(nwg); wg_raw = (b & 7) * (nwg // 8) + (b >> 3)
        had = (env.reset_buf.view(ngrp, -1).any(dim=1) | env._engine.buf["obs_dirty"].view(ngrp, -1).bool().any(dim=1)).cpu()
        rs_.append(had[wg_raw])
        roles.append(torch.zeros(nwg, dtype=torch.bool))
acc /= n
order = [(23, "quad: load burst + LDS staging"), (24, "quad: lane constants"), (12, "quad: rest of prologue"), (21, "4 sub-steps (to read-back)"), (22, "read-back + stores"),
         (5, "mdp: start"), (6, "mdp: callback"), (7, "mdp: termination + rewards"), (9, "mdp: draws + reset blk"), (10, "mdp: observations"), (11, "mdp: state stores")]
prev = 0.0
for k, name in order:
    print(f"{name:30s} +{acc[k]-prev:8.0f} cycles (cum {acc[k]:8.0f})")
    prev = acc[k]
phys, tail, rs_ = torch.stack(phys), torch.stack(tail), torch.stack(rs_)
ok = (phys < 4e5) & (tail < 4e5)     # the stamp slots share episode_done_sums with the episode snapshots of envs 0..511: drop overwritten ones
q = lambda x, p: float(torch.quantile(x, p))
print(f"per-workgroup cycles ({ok.float().mean() * 100:.0f} % of stamps usable): physics median {q(phys[ok], .5):.0f} p99 {q(phys[ok], .99):.0f} max {phys[ok].max():.0f}; "
      f"tail median {q(tail[ok], .5):.0f} p99 {q(tail[ok], .99):.0f} max {tail[ok].max():.0f}")
roles = torch.stack(roles)
for r in (0,):      # the stamps are written by wave 0 of a workgroup = role 0
    m = ok & (roles == bool(r))
    print(f"  role {r}: tail median {q(tail[m], .5):.0f} p99 {q(tail[m], .99):.0f}; with a reset: median {q(tail[m & rs_], .5):.0f} p99 {q(tail[m & rs_], .99):.0f}")
a, b = ok & rs_, ok & ~rs_
print(f"  tail of workgroups with a reset in this or the previous step ({rs_.float().mean() * 100:.1f} %): median {q(tail[a], .5):.0f} p99 {q(tail[a], .99):.0f}; without: median {q(tail[b], .5):.0f} p99 {q(tail[b], .99):.0f}")
tot = (phys + tail)
print(f"  whole kernel per workgroup: median {q(tot[ok], .5):.0f} p99 {q(tot[ok], .99):.0f} max {tot[ok].max():.0f}; per launch max: median over launches {float(torch.where(ok, tot, torch.zeros_like(tot)).max(dim=1).values.median()):.0f}")
