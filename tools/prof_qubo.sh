#!/bin/bash
# Usage: tools/prof_qubo.sh   (GPU box): kernel times of the QUBO energy / fields evaluation on the benchmark model
export TMPDIR=/tmp
rm -rf /tmp/qs
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d /tmp/qs --output-format csv -- python3 tools/time_qubo.py 20 fields > /tmp/qs.log 2>&1 || { tail -5 /tmp/qs.log; exit 1; }
grep -E "^(energy|fields)" /tmp/qs.log
f=$(find /tmp/qs -name '*kernel_stats.csv' | head -1)
python3 tools/prof_summary.py "$f" "rocprofv3 --kernel-trace --stats -- python3 tools/time_qubo.py 20 fields" | head -12
