#!/bin/bash
# Usage: tools/prof_v2.sh <tag>     (on the GPU box, from the repository root)
# Kernel stats of tools/time_v2.py 4096 (the gate-ordered cost, row walk and visit by visit, 16-bit and fp64 tables)
# and the counter passes of the two row-walk kernels; summaries go to gpurun_out/<tag>_*.
set -e
tag=$1
export TMPDIR=/tmp
d=gpurun_out/prof_$tag
rm -rf "$d"; mkdir -p "$d"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$d/stats" --output-format csv -- python3 tools/time_v2.py 4096 > gpurun_out/${tag}_time_v2.txt 2> "$d/err"
f=$(find "$d/stats" -name '*kernel_stats.csv' | head -1)
python3 tools/prof_summary.py "$f" "rocprofv3 --kernel-trace --stats -- python3 tools/time_v2.py 4096 ($tag)" > gpurun_out/${tag}_time_v2_stats.csv
rm -rf "$d/stats"
cat gpurun_out/${tag}_time_v2.txt gpurun_out/${tag}_time_v2_stats.csv
bash tools/pmc_passes.sh ${tag}_k_cost_v2_walk k_cost_v2_walk tools/time_v2.py 4096
bash tools/pmc_passes.sh ${tag}_k_cost_v2_pairs k_cost_v2_pairs tools/time_v2.py 4096
