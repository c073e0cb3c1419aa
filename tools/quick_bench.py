"""Developer micro-benchmark: fused step time vs. env count (not the contract bench)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hcr_genesis_lr_cl_amd.envs import make_env

for n in [int(x) for x in (sys.argv[1:] or ["4096", "16384", "65536", "262144"])]:
    env, cfg = make_env("go2", n)
    env.reset()
    env.episode_length_buf[:] = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
    act = torch.randn(n, 12, device="cuda").clamp(-100, 100)
    for _ in range(50):
        env.step(act)
    torch.cuda.synchronize()
    ms = env._engine.time_steps(act, env.common_step_counter + 1, 200)
    env.common_step_counter += 200
    t0 = time.perf_counter()
    for _ in range(200):
        env.step(act)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 200 * 1e3
    print(f"N={n:7d}  kernel {ms*1e3:8.1f} us/step  {n/ms/1e3:8.2f} M env-steps/s | python loop {wall*1e3:8.1f} us/step  {n/wall/1e3:8.2f} M/s"
          f" | rew mean {env.rew_buf.mean().item():.4f} resets {int(env.reset_buf.sum())}", flush=True)
    del env
