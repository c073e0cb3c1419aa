#!/bin/bash
# Round-end measurement on the GPU box: bench line, rocprofv3 kernel stats, two PMC passes (HBM traffic).
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -c 2000 gpurun_out/bench.json
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats --output-format csv -- python3 bench.py --steps 1000 --no-cpu-baseline > gpurun_out/bench_prof.json 2> gpurun_out/prof_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- python3 tools/prof_steady.py 4096 300 100 > /dev/null 2> gpurun_out/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write --output-format csv -- python3 tools/prof_steady.py 4096 300 100 > /dev/null 2> gpurun_out/pmc_write.err
F=$(find gpurun_out/pmc_fetch -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/pmc_write -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py "$F" "$W" "quad_sim_kernel<4, true, 12u>" go2_flat_4096
cp profiles/hbm_traffic.json gpurun_out/hbm_traffic.json
S=$(find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1)
cp "$S" gpurun_out/kernel_stats.csv
head -3 gpurun_out/kernel_stats.csv | cut -c1-200
