#!/bin/bash
# Static instruction mix of the headline kernel (quad_sim_kernel<4,true,12,1,3>) and of its sub-step loop, from the compiler's ISA.
# usage: tools/isa_loop_stats.sh [extra hipcc flags]
set -e
D=${ISA_DIR:-/tmp/isa}
mkdir -p $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp -DLG_GROUP=${LG_GROUP:-18} "$@" \
    -S --cuda-device-only -o $D/g.s $(dirname $0)/../hcr_genesis_lr_cl_amd/csrc/lg_inst.hip 2>/dev/null
python3 - $D/g.s <<'PY'
import re, sys
lines = open(sys.argv[1]).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z15quad_sim_kernel.*:', l))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start:end]
def is_inst(l):
    t = l.strip()
    return bool(t) and not t.startswith((';', '.', '_Z')) and not t.endswith(':')
def stats(ls, name):
    ins = [l.split()[0] for l in ls if is_inst(l)]
    c = lambda f: sum(1 for i in ins if f(i))
    print(f"{name}: {len(ins)} instructions | VALU {c(lambda i: i.startswith('v_'))} (pk {c(lambda i: i.startswith('v_pk_'))}, mov_dpp {c(lambda i: i == 'v_mov_b32_dpp')}, "
          f"cndmask {c(lambda i: i.startswith('v_cndmask'))}, accvgpr {c(lambda i: 'accvgpr' in i)}) | SALU {c(lambda i: i.startswith('s_') and i != 's_nop')} | s_nop {c(lambda i: i == 's_nop')} | "
          f"scratch {c(lambda i: i.startswith('scratch_') or i.startswith('buffer_'))}")
stats(body, "kernel")
# the sub-step loop: the outermost loop header with the longest span
hdr = [(i, re.search(r'(\.LBB\d+_\d+):', body[i]).group(1)) for i in range(len(body)) if 'Loop Header: Depth=1' in body[i] and re.match(r'^\.LBB', body[i])]
best = None
for i, lab in hdr:
    j = max((k for k in range(i, len(body)) if re.search(r'in Loop: Header=' + lab.replace('.L', '') + r'\b', body[k])), default=i)
    # extend to the end of that last block
    k = j + 1
    while k < len(body) and not body[k].startswith('.LBB'): k += 1
    if best is None or k - i > best[1] - best[0]: best = (i, k)
if best: stats(body[best[0]:best[1]], "sub-step loop")
for l in lines:
    if re.search(r'\.(vgpr_count|agpr_count|sgpr_spill_count|vgpr_spill_count|private_segment_fixed_size):', l): print(l.strip())
    if '.name:' in l and 'quad_sim' in l: print(l.strip())
PY
