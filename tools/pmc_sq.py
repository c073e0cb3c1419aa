"""Summarise one rocprofv3 --pmc pass of SQ counters for the step kernel(s): a markdown table (profiles/<name>.md) and an entry of
profiles/sq_counters.json, which bench.py reads for roofline.valu (VALU instructions per control step).
usage: python tools/pmc_sq.py <counter_collection.csv> <task> <out.md>"""
import csv, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from step_kernels import STEP_KERNELS, key_of, profile_stamp
path, task, out = sys.argv[1:4]
key, subs = key_of(task), STEP_KERNELS[task]
vals = defaultdict(lambda: defaultdict(list))
with open(path) as f:
    for row in csv.DictReader(f):
        k = row.get("Kernel_Name", "")
        if any(s in k for s in subs):
            vals[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
main_k = [k for k in vals if subs[0] in k]
steps = max(len(v) for k in main_k for v in vals[k].values())
tot = defaultdict(float)
for k, cs in vals.items():
    for c, v in cs.items():
        half = len(v) // 2
        tot[c] += (sum(v[half:]) / max(len(v) - half, 1)) * (len(v) / steps)
waves = tot.get("SQ_WAVES", 0.0) or 1.0
lines = [f"# SQ counters per control step: `{key}` ({', '.join(subs)}; steady state, second half of {steps} steps)", "",
         "`rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -- python3 tools/prof_steady.py ...`", "",
         "| counter | per control step | per wave |", "|---|---|---|"]
for c in sorted(tot):
    lines.append(f"| {c} | {tot[c]:.4g} | {tot[c] / waves:.4g} |")
wc, av, wa = tot.get("SQ_WAVE_CYCLES"), tot.get("SQ_ACTIVE_INST_VALU"), tot.get("SQ_WAIT_ANY")
entry = {k: tot[k] for k in tot}
if wc:
    entry["valu_active_share"], entry["wait_share"] = av / wc, wa / wc
    lines += ["", f"VALU-active share of wave time: {100 * av / wc:.1f} %; parked on s_waitcnt / barrier: {100 * wa / wc:.1f} %; "
                  f"cycles per VALU instruction while active: {4 * av / tot['SQ_INSTS_VALU']:.2f} (SQ_*_CYCLES count quad-cycles)."]
open(out, "w").write("\n".join(lines) + "\n")
jp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "sq_counters.json")
try:
    data = json.load(open(jp))
except (OSError, ValueError):
    data = {}
entry.update(profile_stamp())
data[key] = entry
json.dump(data, open(jp, "w"), indent=1)
print("\n".join(lines))
