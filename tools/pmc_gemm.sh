#!/bin/bash
# Counter passes for the QUBO contraction kernel (each pass its own run): tools/pmc_gemm.sh <tag>
set -e
tag=$1
export TMPDIR=/tmp
out=gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out"
passes=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD"
)
: > gpurun_out/${tag}_pmc.csv
i=0
for c in "${passes[@]}"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $c -d "$out/p$i" --output-format csv -- python3 bench.py --no-cpu --single --steps 2 --warmup 1 --qubo-sweeps 1 > "$out/p$i.log" 2>&1
  f=$(find "$out/p$i" -name '*counter_collection.csv' | head -1)
  python3 tools/pmc_summary.py "$f" k_qubo_gemm >> gpurun_out/${tag}_pmc.csv
  rm -rf "$out/p$i"
done
cat gpurun_out/${tag}_pmc.csv
