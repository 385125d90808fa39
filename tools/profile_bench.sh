#!/bin/bash
# Usage: tools/profile_bench.sh <tag>      (run on the GPU box from the repository root)
# 1. rocprofv3 --kernel-trace --stats of the default bench.py (without the CPU baseline leg);
# 2. per variant (uniform, integer, integer with look-ups forced), two --pmc passes (FETCH_SIZE, WRITE_SIZE: each in its own run) for the HBM traffic of
#    the cost-only bench.  Summaries go to gpurun_out/<tag>_*.csv and gpurun_out/<tag>_traffic.json (copy the
#    latter to profiles/r03_traffic.json: bench.py reads it); raw traces are deleted.
set -e
tag=$1
export TMPDIR=/tmp
d=gpurun_out/prof_$tag
rm -rf "$d"; mkdir -p "$d"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$d/stats" --output-format csv -- python3 bench.py --no-cpu > "$d/bench.json" 2> "$d/bench.err"
f=$(find "$d/stats" -name '*kernel_stats.csv' | head -1)
python3 tools/prof_summary.py "$f" "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu ($tag)" > gpurun_out/${tag}_bench_default_stats.csv
rm -rf "$d/stats"
for v in uniform integer integer_table; do
  w=${v%_table}
  tab=packed; [ "$v" = integer_table ] && tab=table
  : > gpurun_out/${tag}_${v}_hbm_traffic.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $c -d "$d/pmc_$c" --output-format csv -- python3 bench.py --no-cpu --no-qubo --single --weights $w --table $tab > "$d/pmc_${v}_$c.log" 2>&1
    f=$(find "$d/pmc_$c" -name '*counter_collection.csv' | head -1)
    python3 tools/pmc_summary.py "$f" k_cost_v3 >> gpurun_out/${tag}_${v}_hbm_traffic.csv
    rm -rf "$d/pmc_$c"
  done
done
python3 tools/make_traffic_json.py "$tag" > gpurun_out/${tag}_traffic.json
cat gpurun_out/${tag}_bench_default_stats.csv gpurun_out/${tag}_*_hbm_traffic.csv gpurun_out/${tag}_traffic.json
tail -1 "$d/bench.json" > gpurun_out/${tag}_bench.json
