"""BUILD-CONTAINER ONLY (needs /root/reference): how long the reference's own PyTorch env stack takes per control step on the CPU.

SURVEY 8(d) asks for it as a side number next to the GPU measurement: the reference's LeggedRobot.step() *without* any physics
(legged_gym/envs/base/legged_robot.py:37-168 + go2.py, i.e. rows a2, a9-a16 of SURVEY 8a) on torch-CPU, driven by the scripted
fake simulator of tests/golden/gen_mdp_fixtures.py.  It quantifies the Python / torch dispatch cost that the fused kernel
removes; the physics (Genesis) is not installed and is not part of this number.

usage: PYTHONDONTWRITEBYTECODE=1 python tools/ref_env_stack_timing.py [n_envs] [steps] [threads]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)
import gen_mdp_fixtures as gf  # noqa: E402  (loads the reference through ref_harness)
import torch  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    torch.set_num_threads(threads)
    import legged_gym.envs.base.base_task as base_task
    import legged_gym.envs.go2.go2 as go2_mod
    from legged_gym.envs.go2.go2_config import GO2Cfg
    from legged_gym.utils.helpers import class_to_dict
    base_task.GenesisSimulator = gf.FakeSimulator
    cfg = GO2Cfg()
    cfg.env.num_envs = N
    env = go2_mod.GO2(cfg, class_to_dict(cfg.sim), "cpu", True)
    sim = env.simulator
    sim.rec = gf.rh.DrawRecorder(1)
    rng = np.random.default_rng(2)
    sim.script = gf.make_script(rng, sim.model, cfg, N, T + 5)
    env.episode_length_buf[:] = torch.randint(0, 1000, (N,), dtype=env.episode_length_buf.dtype)
    acts = torch.randn(T + 5, N, 12)
    with torch.inference_mode():
        for t in range(5):
            env.step(acts[t])
        t0 = time.perf_counter()
        for t in range(5, T + 5):
            env.step(acts[t])
        dt = time.perf_counter() - t0
    print(f"reference env stack (no physics), torch-CPU {threads} threads, {N} envs: {dt / T * 1e3:.2f} ms per control step = "
          f"{N * T / dt / 1e6:.3f} M env-steps/s")


if __name__ == "__main__":
    main()
