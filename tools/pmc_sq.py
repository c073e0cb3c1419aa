"""Summarise one rocprofv3 --pmc pass of SQ counters for the step kernel into a markdown table (profiles/r01_c_pmc.md).
usage: python tools/pmc_sq.py <counter_collection.csv> <kernel substring> <out.md>"""
import csv, sys
from collections import defaultdict

path, ksub, out = sys.argv[1:4]
vals = defaultdict(list)
with open(path) as f:
    for row in csv.DictReader(f):
        if ksub in row.get("Kernel_Name", ""):
            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
mean = {k: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for k, v in vals.items()}
waves = mean.get("SQ_WAVES", 0.0) or 1.0
lines = [f"# SQ counters of `{ksub}` (go2 flat, 4096 envs, steady state; mean over the last {len(next(iter(vals.values()))) // 2} launches)", "",
         "`rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -- python3 tools/prof_steady.py 4096 300 100`", "",
         "| counter | per launch | per wave |", "|---|---|---|"]
for k in sorted(mean):
    lines.append(f"| {k} | {mean[k]:.4g} | {mean[k] / waves:.4g} |")
wc, av, wa = mean.get("SQ_WAVE_CYCLES"), mean.get("SQ_ACTIVE_INST_VALU"), mean.get("SQ_WAIT_ANY")
if wc:
    lines += ["", f"VALU-active share of wave time: {100 * av / wc:.1f} %; parked on s_waitcnt / barrier: {100 * wa / wc:.1f} %; "
                  f"cycles per VALU instruction while active: {4 * av / mean['SQ_INSTS_VALU']:.2f} "
                  "(SQ_*_CYCLES count quad-cycles).  Waves per launch = one per SIMD (1024 SIMDs): achieved occupancy is 1 wave "
                  "of 8 per SIMD by construction (4096 envs x 16 lanes); the bound is the serial instruction chain of that wave."]
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
