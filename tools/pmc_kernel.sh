#!/bin/bash
# PMC counters of one kernel of a small python command, one rocprofv3 --pmc pass per counter group:
#   bash tools/pmc_kernel.sh <tag> <kernel substring> <python script and args...>   -> gpurun_out/pmc_<tag>.txt
export TMPDIR=/tmp
TAG=$1; SUB=$2; shift; shift
O=gpurun_out/pmck_$TAG; rm -rf $O; mkdir -p $O
OUT=gpurun_out/pmc_$TAG.txt; : > $OUT
for grp in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  rocprofv3 --pmc $grp -d $O/p -o run --output-format csv -- python3 "$@" > /dev/null 2> $O/err.txt
  f=$(find $O/p -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 tools/pmc_table.py "$f" "$SUB" >> $OUT; else echo "## failed: $grp" >> $OUT; tail -3 $O/err.txt >> $OUT; fi
  rm -rf $O/p
done
cat $OUT
