#!/bin/bash
# Round measurement on the GPU box: bench lines for the four BASELINE tasks, rocprofv3 kernel stats of the headline command,
# PMC passes (HBM traffic: FETCH_SIZE / WRITE_SIZE in separate passes; SQ counters) per task.  Everything lands in gpurun_out/;
# the summaries to keep are copied into profiles/ (tracked).   usage: tools/gpu_profile.sh <tag, e.g. r02_a> [tasks...]
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
export TMPDIR=/tmp
TAG=${1:-r03}
export LG_PROFILE_TAG=$TAG
shift || true
TASKS=${@:-go2 go2_wtw go2_ee tron1_pf_ee}
mkdir -p gpurun_out profiles
for T in $TASKS; do
  echo "== $T: PMC passes"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_$T --output-format csv -- python3 tools/prof_steady.py 4096 300 128 $T > /dev/null 2> gpurun_out/pmc_fetch_$T.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write_$T --output-format csv -- python3 tools/prof_steady.py 4096 300 128 $T > /dev/null 2> gpurun_out/pmc_write_$T.err
  F=$(find gpurun_out/pmc_fetch_$T -name "*counter_collection.csv" | head -1)
  W=$(find gpurun_out/pmc_write_$T -name "*counter_collection.csv" | head -1)
  python tools/pmc_traffic.py "$F" "$W" $T || echo "pmc_traffic failed for $T"
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d gpurun_out/pmc_sq_$T --output-format csv -- python3 tools/prof_steady.py 4096 300 128 $T > /dev/null 2> gpurun_out/pmc_sq_$T.err
  S=$(find gpurun_out/pmc_sq_$T -name "*counter_collection.csv" | head -1)
  python tools/pmc_sq.py "$S" $T gpurun_out/${TAG}_pmc_sq_$T.md > /dev/null || echo "pmc_sq failed for $T"
done
cp profiles/hbm_traffic.json profiles/sq_counters.json gpurun_out/ 2>/dev/null || true
echo "== bench lines (read the PMC summaries just written)"
for T in $TASKS; do
  EXTRA=$([ $T = go2 ] && echo "--ppo-rollout 30" || ([ $T = go2_ee ] || [ $T = tron1_pf_ee ]) && echo "--no-cpu-baseline --ppo-rollout 10" || echo "--no-cpu-baseline")
  python bench.py --task $T $EXTRA > gpurun_out/${TAG}_bench_$T.json 2> gpurun_out/${TAG}_bench_$T.err || { echo "bench failed for $T"; continue; }
  python -c "import json;d=json.load(open('gpurun_out/${TAG}_bench_$T.json'));r=d['roofline'];print('$T', round(d['value']/1e6,2),'M env-steps/s', round(r['launch_us'],2),'us frac',round(r['frac'],4),'traffic',r['traffic'],'valu',(r['valu'] or {}).get('frac'))"
done
echo "== rocprofv3 kernel stats of the headline command"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats --output-format csv -- python3 bench.py --steps 1000 --no-cpu-baseline --no-stream-copy > gpurun_out/bench_prof.json 2> gpurun_out/prof_stats.err
S=$(find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1)
head -8 "$S" > gpurun_out/${TAG}_kernel_stats.csv
echo "== the driver's command (20 steps, 5 warm-up), plain and under rocprofv3"
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_go2_driver_cmd.json 2> gpurun_out/${TAG}_bench_go2_driver_cmd.err || echo "driver-cmd bench failed"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats_drv --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stream-copy > /dev/null 2> gpurun_out/prof_stats_drv.err
S=$(find gpurun_out/prof_stats_drv -name "*kernel_stats.csv" | head -1)
head -6 "$S" > gpurun_out/${TAG}_kernel_stats_driver_cmd.csv
cp profiles/hbm_traffic.json profiles/sq_counters.json gpurun_out/ 2>/dev/null || true
head -4 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-220
echo "copy gpurun_out/${TAG}_* + hbm_traffic.json + sq_counters.json into profiles/ (tracked)"
