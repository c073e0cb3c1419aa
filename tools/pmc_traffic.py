"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per pass, kernel trace only) into
profiles/hbm_traffic.json, which bench.py reads for roofline.traffic.

usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <task>

Several kernel substrings = a control step made of several launches (biped: physics launch + MDP launch; history tasks: plus
the amortised compaction kernel): their per-launch means are weighted by launches per control step and summed.

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) and cdna_hip_programming.md:1295: rocprofv3 reports
FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE tallies 128-B requests as 64 B, so the read side is doubled;
WRITE_SIZE is taken as reported.  (The guide calibrates this on 16-B-per-lane streams; these kernels issue dword loads
of 12-byte-strided rows, so the absolute is indicative, the guide says as much: the raw counters are kept alongside.)
"""
import csv, json, os, sys


def per_kernel(path, name):
    vals = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") == name:
                vals.setdefault(row.get("Kernel_Name", ""), []).append(float(row["Counter_Value"]))
    return vals


def main():
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from step_kernels import STEP_KERNELS, key_of, profile_stamp
    fetch_csv, write_csv, task = sys.argv[1:4]
    key, subs = key_of(task), STEP_KERNELS[task]
    fv, wv = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    # launches of the main (first) kernel define the number of control steps in the trace
    main_k = [k for k in fv if subs[0] in k]
    if not main_k:
        raise SystemExit(f"no rows for {subs[0]} in {fetch_csv}")
    steps = sum(len(fv[k]) for k in main_k)
    rd = wr = 0.0
    used = []
    for sub in subs:
        for k in fv:
            if sub in k:
                half = len(fv[k]) // 2                     # steady state: second half of the launches
                per_step = len(fv[k]) / steps
                rd += (sum(fv[k][half:]) / max(len(fv[k]) - half, 1)) * per_step
                wk = wv.get(k, [0.0])
                h2 = len(wk) // 2
                wr += (sum(wk[h2:]) / max(len(wk) - h2, 1)) * per_step
                used.append({"kernel": k[:90], "launches_per_step": round(per_step, 4)})
    rd_b, wr_b = rd * 1024.0 * 2.0, wr * 1024.0
    out_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "hbm_traffic.json")
    try:
        data = json.load(open(out_path))
    except (OSError, ValueError):
        data = {}
    data[key] = {"kernels": used, "bytes_per_launch": rd_b + wr_b, "read_bytes": rd_b, "write_bytes": wr_b,
                 "raw_bytes_per_step": (rd + wr) * 1024.0, "fetch_size_raw_kb": rd, "write_size_raw_kb": wr, "control_steps": steps,
                 "note": "per control step; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B); WRITE_SIZE as reported", **profile_stamp()}
    json.dump(data, open(out_path, "w"), indent=1)
    print(json.dumps(data[key]))


if __name__ == "__main__":
    main()
