"""Builds csrc/liblgsim.so with hipcc for gfx950 (in-tree, so it travels to the GPU box)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SRC = ["lg_kernel.hip"]
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
    return any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # iterative-ilp has crashed the register allocator (SIGSEGV in clang) on some shapes of this kernel: fall back to the
    # max-ilp strategy (within ~1 % on the bench), then to the default scheduler, rather than fail the build
    attempts = [EXTRA_FLAGS]
    if "LG_HIPCC_FLAGS" not in os.environ:
        attempts += [["-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp"], ["-fno-slp-vectorize"]]
    err = ""
    for flags in attempts:
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17", *flags,
               "-o", OUT] + [os.path.join(CSRC, s) for s in SRC]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode == 0:
            if flags is not attempts[0]:
                import sys
                print("build.py: compiled with fallback flags " + " ".join(flags), file=sys.stderr)
            break
        err = r.stderr
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + err[-4000:])
    if verbose:
        print(r.stderr)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
