#!/bin/bash
# SQ counters of the 16-step pass at 16384^2 and 4096^2 (two rocprofv3 --pmc passes each) -> gpurun_out/<dir>/sq_counters.txt
out=/root/repo/gpurun_out/$1; mkdir -p $out
: > $out/sq_counters.txt
for g in 16384 4096; do
  for ctr in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES" \
             "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_WAVES"; do
    echo "== $g: $ctr" >> $out/sq_counters.txt
    bash /root/repo/tools/tools_pmc.sh sq "$ctr" --grid $g --cols $g --materials uniform --boundary mur >> $out/sq_counters.txt 2>&1 || exit 1
  done
done
cat $out/sq_counters.txt
