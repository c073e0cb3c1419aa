"""What the build's DPP hazard pass (hcr_genesis_lr_cl_amd/dpp_hazard_pass.py) takes for granted about the chip, checked on the chip: a DPP
instruction needs its two wait states behind a VALU write of its DPP-ROUTED source only; a plain operand or the accumulator written in the
slot before is forwarded (include/lgsim.h lg_dpp_kat)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rot1(v):      # quad_perm:[1,2,0,3] inside every quad of the wave
    q = v.reshape(-1, 4)
    return q[:, [1, 2, 0, 3]].reshape(-1)


def test_only_the_dpp_routed_operand_needs_wait_states():
    from hcr_genesis_lr_cl_amd import abi
    lib = abi.load_lib()
    rng = np.random.default_rng(7)
    for rep in range(8):
        x = rng.uniform(-4, 4, 64).astype(np.float32)
        y = rng.uniform(-4, 4, 64).astype(np.float32)
        inp = np.concatenate([x, y]).astype(np.float32)
        out = np.zeros(320, np.float32)
        abi.check(lib.lg_dpp_kat(inp.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float))), lib)
        r0, r1, r2, r3, a = out.reshape(5, 64)
        xy = x * y
        np.testing.assert_array_equal(a, xy)
        np.testing.assert_array_equal(r0, rot1(x) + xy)                                   # fresh plain operand: no wait state needed
        fused = (rot1(x).astype(np.float64) * y.astype(np.float64) + xy.astype(np.float64)).astype(np.float32)
        assert np.max(np.abs(r1 - fused) / np.maximum(np.abs(fused), 1e-6)) < 2e-7            # fresh accumulator: likewise (fused multiply-add)
        np.testing.assert_array_equal(r2, rot1(xy) + x)                                   # fresh DPP source behind `s_nop 1`
    # the negative control (fresh DPP source, no wait state) is reported only: on gfx950 it reads the register's previous content
    stale = int(np.sum(~(r3 == rot1(xy) + x)))
    print(f"negative control: {stale} of 64 lanes differ from the waited result")
