"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per pass, kernel trace only) into
profiles/hbm_traffic.json, which bench.py reads for roofline.traffic.

usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel substring> <key>

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) and cdna_hip_programming.md:1295: rocprofv3 reports
FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE tallies 128-B requests as 64 B, so the read side is doubled;
WRITE_SIZE is taken as reported.  (The guide calibrates this on 16-B-per-lane streams; this kernel issues dword loads
of 12-byte-strided rows, so the absolute is indicative, the guide says as much.)
"""
import csv, json, os, sys


def mean_counter(path, kernel_sub, name):
    vals = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel_sub in row.get("Kernel_Name", "") and row.get("Counter_Name") == name:
                vals.append(float(row["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no rows for {name} / {kernel_sub} in {path}")
    tail = vals[len(vals) // 2:]          # steady state: second half of the launches
    return sum(tail) / len(tail), len(tail)


def main():
    fetch_csv, write_csv, ksub, key = sys.argv[1:5]
    fetch, nf = mean_counter(fetch_csv, ksub, "FETCH_SIZE")
    write, nw = mean_counter(write_csv, ksub, "WRITE_SIZE")
    rd, wr = fetch * 1024.0 * 2.0, write * 1024.0
    out_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "hbm_traffic.json")
    try:
        data = json.load(open(out_path))
    except (OSError, ValueError):
        data = {}
    data[key] = {"kernel": ksub, "bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr,
                 "fetch_size_raw_kb": fetch, "write_size_raw_kb": write, "launches_averaged": min(nf, nw),
                 "note": "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B); WRITE_SIZE as reported"}
    json.dump(data, open(out_path, "w"), indent=1)
    print(json.dumps(data[key]))


if __name__ == "__main__":
    main()
