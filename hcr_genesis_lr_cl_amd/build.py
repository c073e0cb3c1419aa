"""Builds csrc/liblgsim.so with hipcc for gfx950 (in-tree, so it travels to the GPU box)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SRC = ["lg_kernel.hip", "lg_rollout.hip"]
OUT = os.path.join(CSRC, "liblgsim.so")
# -fno-slp-vectorize: packing scalars into v_pk_* costs more v_mov / AGPR shuffles than it saves here.
# iterative-ilp scheduling: the kernels run one wave per SIMD, so occupancy is irrelevant and the scheduler should fill DPP /
# VALU->SGPR hazard slots with independent work (measured -3 % on the physics launch, neutral elsewhere).
EXTRA_FLAGS = os.environ.get("LG_HIPCC_FLAGS", "-fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp").split()


def needs_build():
    if not os.path.exists(OUT):
        return True
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    deps.append(os.path.join(HERE, "..", "include", "lgsim.h"))
    deps.append(os.path.join(HERE, "..", "include", "lgrollout.h"))
    return any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps)


SIDECAR = os.path.join(CSRC, "liblgsim.build.json")
FALLBACK_FLAGS = [["-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp"], ["-fno-slp-vectorize"]]


def build(force=False, verbose=False, allow_fallback=None):
    """Compile csrc/liblgsim.so.  The flags actually used are written next to it (liblgsim.build.json) and echoed by bench.py
    in its JSON line, so the scheduling strategy of a measured binary is on record.

    The iterative-ilp scheduler has crashed clang (SIGSEGV in the register allocator) on some shapes of this kernel during
    development.  By default such a failure FAILS the build.  `allow_fallback=True` (or LG_ALLOW_FLAG_FALLBACK=1; the driver
    hook __graft_entry__.build() opts in) retries with the max-ilp strategy and then the default scheduler (within ~1 % / ~3 %
    on the bench) and records `"fallback": true` plus the compiler's error in the sidecar."""
    if not force and not needs_build():
        return OUT
    import json
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if allow_fallback is None:
        allow_fallback = os.environ.get("LG_ALLOW_FLAG_FALLBACK", "0") == "1"
    attempts = [EXTRA_FLAGS]
    if allow_fallback and "LG_HIPCC_FLAGS" not in os.environ:
        attempts += FALLBACK_FLAGS
    first_err, r = "", None
    for flags in attempts:
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17", *flags,
               "-o", OUT] + [os.path.join(CSRC, s) for s in SRC]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode == 0:
            fell = flags is not attempts[0]
            if fell:
                import sys
                print("build.py: primary flags failed, compiled with FALLBACK flags " + " ".join(flags), file=sys.stderr)
            ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout.splitlines()
            with open(SIDECAR, "w") as f:
                json.dump({"flags": ["--offload-arch=gfx950", "-O3", *flags], "fallback": fell,
                           "primary_flags": EXTRA_FLAGS, "primary_error": first_err[-1500:] if fell else "",
                           "hipcc": next((l for l in ver if "HIP version" in l), ver[0] if ver else "")}, f, indent=1)
            break
        first_err = first_err or r.stderr
    if r.returncode != 0:
        raise RuntimeError("hipcc failed" + ("" if allow_fallback else " (no flag fallback: set LG_ALLOW_FLAG_FALLBACK=1 to retry with the "
                           "max-ilp / default scheduler)") + ":\n" + first_err[-4000:])
    if verbose:
        print(r.stderr)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
