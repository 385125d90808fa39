#!/bin/bash
# Usage: tools/pmc_passes.sh <tag> <kernel-substring> <python script and args...>
# Runs the rocprofv3 counter passes (each in its own run) and writes per-dispatch averages of the
# named kernel to gpurun_out/<tag>_pmc.csv.
set -e
tag=$1; kern=$2; shift 2
export TMPDIR=/tmp
out=gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out"
passes=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT"
 "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VALU SQ_IFETCH TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum"
)
: > gpurun_out/${tag}_pmc.csv
i=0
for c in "${passes[@]}"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $c -d "$out/p$i" --output-format csv -- python3 "$@" > "$out/p$i.log" 2>&1
  f=$(find "$out/p$i" -name '*counter_collection.csv' | head -1)
  python3 tools/pmc_summary.py "$f" "$kern" >> gpurun_out/${tag}_pmc.csv
  rm -rf "$out/p$i"
done
cat gpurun_out/${tag}_pmc.csv
