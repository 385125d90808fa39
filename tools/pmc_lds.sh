#!/bin/bash
# Usage: tools/pmc_lds.sh <weights> <table>   (GPU box): LDS / VALU counters of the dense cost kernel on the
# benchmark workload of tools/tune_v3.py, one rocprofv3 --pmc pass.
export TMPDIR=/tmp
rm -rf /tmp/pmc_lds
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS -d /tmp/pmc_lds --output-format csv -- python3 tools/tune_v3.py "$1" "$2" > /tmp/pmc_lds.log 2>&1 || { tail -5 /tmp/pmc_lds.log; exit 1; }
f=$(find /tmp/pmc_lds -name '*counter_collection.csv' | head -1)
python3 tools/pmc_summary.py "$f" k_cost_v3
