"""The C ABI: ctypes mirrors (abi.py) agree with include/lgsim.h byte for byte, and the built
library exports every declared entry point.  No GPU needed (no compute calls)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from hcr_genesis_lr_cl_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lgsim.h")
HEADER_ROLLOUT = os.path.join(ROOT, "include", "lgrollout.h")


def _probe(structs):
    """Compile a C probe printing sizeof + offsetof of every field."""
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', f'#include "{HEADER_ROLLOUT}"', "int main(void){"]
    for cname, cls in structs:
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append("return 0;}")
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "p.c"), os.path.join(d, "p")
        open(src, "w").write("\n".join(lines))
        subprocess.run(["gcc", "-o", exe, src], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    return dict(l.split() for l in out.strip().splitlines())


def test_struct_layouts_match_header():
    structs = [("LgModelDesc", abi.LgModelDesc), ("LgSimOptions", abi.LgSimOptions), ("LgRandSlots", abi.LgRandSlots),
               ("LgObsProgram", abi.LgObsProgram), ("LgTaskCfg", abi.LgTaskCfg), ("LgBuffers", abi.LgBuffers), ("LgRowCopy", abi.LgRowCopy)]
    got = _probe(structs)
    for cname, cls in structs:
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_constants_match_header():
    h = open(HEADER).read()
    d = dict(re.findall(r"#define\s+(LG_\w+)\s+(\d+)", h))
    assert int(d["LG_MAX_DOF"]) == abi.MAX_DOF and int(d["LG_MAX_BODIES"]) == abi.MAX_BODIES
    assert int(d["LG_MAX_LINKS"]) == abi.MAX_LINKS and int(d["LG_MAX_SPHERES"]) == abi.MAX_SPHERES
    assert int(d["LG_MAX_OBS"]) == abi.MAX_OBS and int(d["LG_NUM_REWARDS"]) == abi.NUM_REWARDS
    enum = re.search(r"enum LgReward \{(.*?)\};", h, re.S).group(1)
    names = [n.strip().split("=")[0].strip() for n in re.sub(r"/\*.*?\*/", "", enum, flags=re.S).split(",") if n.strip()]
    assert names[-1] == "LG_R_COUNT"
    assert [n[len("LG_R_"):].lower() for n in names[:-1]] == abi.REWARD_NAMES


def test_library_exports_every_declared_symbol():
    if not os.path.exists(abi.lib_path()):
        from hcr_genesis_lr_cl_amd import build
        build.build()
    h = open(HEADER).read()
    declared = set(re.findall(r"\b(lg_\w+)\s*\(", h))
    assert declared == set(abi.EXPORTS)
    declared_r = set(re.findall(r"\b(lg_\w+)\s*\(", open(HEADER_ROLLOUT).read()))
    assert declared_r == set(abi.ROLLOUT_EXPORTS)
    lib = C.CDLL(abi.lib_path())        # loads without a GPU; nothing is called
    for sym in declared | declared_r:
        assert hasattr(lib, sym), sym


def test_missing_extension_fails_loudly(monkeypatch):
    monkeypatch.setattr(abi, "_LIB", None)
    monkeypatch.setattr(abi, "lib_path", lambda: "/nonexistent/liblgsim.so")
    with pytest.raises(abi.HipExtensionMissing):
        abi.load_lib()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "hcr_genesis_lr_cl_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("oracle/lg_oracle.c", "").replace("CPU oracle", "").replace("(see oracle", "").replace("The CPU oracle", ""), f


def test_build_used_primary_flags():
    """The in-tree library was compiled with the scheduler flags the measurements are quoted for: a silent fallback after a
    compiler crash (build.py) would show up here and in bench.py's `build_flags`."""
    import json
    from hcr_genesis_lr_cl_amd import build
    if not os.path.exists(build.SIDECAR):
        build.build(force=True)
    info = json.load(open(build.SIDECAR))
    assert info["fallback"] is False, info.get("primary_error", "")[-500:]
    assert all(f in info["flags"] for f in build.EXTRA_FLAGS)
    st = info.get("dpp_hazard_pass")
    assert st and not st.get("pipeline_failed"), f"a kernel group fell back to plain hipcc -c (no hazard pass): {st}"
