#!/bin/bash
# Bench lines only (no PMC passes: they read the committed summaries): the four BASELINE tasks, the driver's 20-step command plain and
# under rocprofv3, the headline command under rocprofv3.   usage: tools/gpu_bench_lines.sh <tag>
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export TMPDIR=/tmp
TAG=${1:-r03}
mkdir -p gpurun_out
for T in go2 go2_wtw go2_ee tron1_pf_ee; do
  EXTRA=$([ $T = go2 ] && echo "--ppo-rollout 30" || ([ $T = go2_ee ] || [ $T = tron1_pf_ee ]) && echo "--no-cpu-baseline --ppo-rollout 10" || echo "--no-cpu-baseline")
  python bench.py --task $T $EXTRA > gpurun_out/${TAG}_bench_$T.json 2> gpurun_out/${TAG}_bench_$T.err || { echo "bench failed for $T"; continue; }
  python -c "import json;d=json.load(open('gpurun_out/${TAG}_bench_$T.json'));r=d['roofline'];print('$T', round(d['value']/1e6,2),'M env-steps/s', round(d['ms_per_step']*1e3,2), 'us/step; kernel timer', round(r['launch_us'],2),'us frac',round(r['frac'],4),'stale',r['traffic_counters']['stale'])"
done
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats --output-format csv -- python3 bench.py --steps 1000 --no-cpu-baseline --no-stream-copy > gpurun_out/bench_prof.json 2> gpurun_out/prof_stats.err
head -8 "$(find gpurun_out/prof_stats -name '*kernel_stats.csv' | head -1)" > gpurun_out/${TAG}_kernel_stats.csv
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_go2_driver_cmd.json 2> gpurun_out/${TAG}_bench_go2_driver_cmd.err
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_go2_driver_cmd_again.json 2>> gpurun_out/${TAG}_bench_go2_driver_cmd.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats_drv --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stream-copy > /dev/null 2> gpurun_out/prof_stats_drv.err
head -6 "$(find gpurun_out/prof_stats_drv -name '*kernel_stats.csv' | head -1)" > gpurun_out/${TAG}_kernel_stats_driver_cmd.csv
python -c "
import json
for f in ('','_again'):
    d=json.load(open('gpurun_out/${TAG}_bench_go2_driver_cmd'+f+'.json')); print('driver cmd'+f, round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,2), [round(x*1e3,1) for x in d['repeats_ms_per_step']])"
head -3 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-200
