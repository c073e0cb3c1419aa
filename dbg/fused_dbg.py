import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch
from hcr_genesis_lr_cl_amd import abi
from test_gpu_env import _mk
e1, e2 = _mk(256), _mk(256)
e1.reset(); e2.reset()
g = torch.Generator(device="cuda"); g.manual_seed(5)
for t in range(40):
    for k, v in e1._engine.buf.items():
        e2._engine.buf[k].copy_(v)
    e2.common_step_counter = e1.common_step_counter
    act = torch.randn(256, 12, generator=g, device="cuda")
    e1.step(act)
    e2.common_step_counter += 1
    e2._engine.step(abi.PHASE_PRE | abi.PHASE_SIM | abi.PHASE_POST, act, e2.common_step_counter)
    e2._engine.step(abi.PHASE_RESET, None, e2.common_step_counter)
    torch.cuda.synchronize()
    a, b = e1._engine.buf["base_pos"].cpu().numpy(), e2._engine.buf["base_pos"].cpu().numpy()
    bad = np.argwhere(np.abs(a - b) > 1e-4)
    if len(bad):
        e = bad[0][0]
        print("t", t, "env", e, "fused", a[e], "split", b[e])
        for k in ("reset_buf", "episode_length_buf", "dof_pos", "base_quat", "env_origins", "commands", "fail_buf", "time_out_buf"):
            print(k, e1._engine.buf[k][e].cpu().numpy(), e2._engine.buf[k][e].cpu().numpy())
        break
