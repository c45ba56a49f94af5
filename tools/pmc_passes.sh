#!/bin/bash
# usage: pmc_passes.sh <outdir> -- <command...>   (separate rocprofv3 --pmc passes, kernel-trace only)
OUT=$1; shift; shift
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD" \
         "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_VMEM" \
         "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pass$i -o p -- "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed" >> $OUT/fail.log
done
