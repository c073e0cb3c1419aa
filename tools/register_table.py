"""Register / scratch table of every kernel instantiation of the library: compiles each group of csrc/lg_inst.hip with
-Rpass-analysis=kernel-resource-usage (device pass only, nothing is linked) and prints one line per kernel.
usage: python tools/register_table.py [> profiles/rNN_register_table.md]"""
import os, re, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hcr_genesis_lr_cl_amd import build as b


def one(g):
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", *b.EXTRA_FLAGS, f"-DLG_GROUP={g}",
           "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(b.CSRC, "lg_inst.hip"), "-o", "/dev/null"]
    return g, subprocess.run(cmd, capture_output=True, text=True).stderr


rows = []
with ThreadPoolExecutor(max_workers=8) as ex:
    for g, err in ex.map(one, range(b.N_GROUPS)):
        cur = {}
        for line in err.splitlines():
            m = re.search(r"remark: (?:.*?: )?\s*(Function Name|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (.*)$", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2).replace("[-Rpass-analysis=kernel-resource-usage]", "").strip()
            if k == "Function Name":
                cur = {"group": g, "name": v}
                rows.append(cur)
            else:
                cur[k] = v
print("| group | kernel | VGPR | AGPR | SGPR spills | VGPR spills | scratch B/lane | waves/SIMD | LDS B |")
print("|---|---|---|---|---|---|---|---|---|")
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().replace("(KParams)", "")
    print(f"| {r['group']} | `{name}` | {r.get('VGPRs')} | {r.get('AGPRs')} | {r.get('SGPRs Spill')} | {r.get('VGPRs Spill')} | "
          f"{r.get('ScratchSize [bytes/lane]')} | {r.get('Occupancy [waves/SIMD]')} | {r.get('LDS Size [bytes/block]')} |")
