#!/bin/bash
# PMC passes for the dominant kernel (AlignConv f16, P3, B=8).  Counters only (no tracing domains);
# FETCH_SIZE and WRITE_SIZE in separate passes (TCC slot limits, MI355X_MICROARCH.md).
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_align2
mkdir -p $OUT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python $R/scripts/bench_ops.py --which align8 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
echo done
