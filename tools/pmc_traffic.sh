#!/bin/bash
# Usage: tools/pmc_traffic.sh <tag> <kernel-substring> <python script and args...>
# HBM traffic of one kernel from the L2 counters: FETCH_SIZE and WRITE_SIZE, each in a run of its own; per-dispatch
# averages go to gpurun_out/<tag>_hbm_traffic.csv (FETCH_SIZE is reported in units that need doubling on gfx950).
set -e
tag=$1; kern=$2; shift 2
export TMPDIR=/tmp
out=gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out"
: > gpurun_out/${tag}_hbm_traffic.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c -d "$out/$c" --output-format csv -- python3 "$@" > "$out/$c.log" 2>&1
  f=$(find "$out/$c" -name '*counter_collection.csv' | head -1)
  python3 tools/pmc_summary.py "$f" "$kern" >> gpurun_out/${tag}_hbm_traffic.csv
  rm -rf "$out/$c"
done
cat gpurun_out/${tag}_hbm_traffic.csv
