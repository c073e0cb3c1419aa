"""Wait states spent in s_nop by the headline kernel's sub-step loop (and the whole kernel), from the pass's output next to the object.
usage: python tools/isa_nops.py [group] [kernel-name-prefix]"""
import re, sys, os
from collections import Counter
g = sys.argv[1] if len(sys.argv) > 1 else "18"
pref = sys.argv[2] if len(sys.argv) > 2 else "_Z15quad_sim_kernel"
lines = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hcr_genesis_lr_cl_amd", "csrc", "obj", f"lg_inst_{g}.fix.s")).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith(pref) and l.rstrip().endswith(":") or re.match(re.escape(pref) + r".*:", l))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
hdrs = {}
for i, l in enumerate(body):
    m = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=1", l)
    if m: hdrs.setdefault(m.group(1), []).append(i)
h = max(hdrs, key=lambda k: hdrs[k][-1] - hdrs[k][0])
a, b = hdrs[h][0], hdrs[h][-1]
while b < len(body) and not body[b].startswith(".LBB"): b += 1
def stat(ls, name):
    ws = Counter(); n = 0; valu = 0
    for l in ls:
        t = l.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"): continue
        n += 1
        if t.startswith("v_"): valu += 1
        if t.startswith("s_nop"):
            k = int(t.split()[1]) + 1
            ws["pass" if "(pass)" in t else "compiler"] += k
    print(f"{name}: {n} instructions, VALU {valu}, nop wait states {dict(ws)} -> issue slots {n + sum(ws.values()) - sum(1 for l in ls if l.strip().startswith('s_nop'))}")
print(lines[start][:80])
stat(body, "kernel")
stat(body[a:b], "sub-step loop")
