#!/bin/bash
# PMC passes over the default bench command for the AlignConv (k_dcn_patch) and conv-tower (k_conv_f16)
# launches.  Counters only (+ kernel trace); FETCH_SIZE and WRITE_SIZE in separate passes (TCC slot
# limits, MI355X_MICROARCH.md HBM section).
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_bench
mkdir -p $OUT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/p$i.log 2>&1 || echo "pass $i failed"
  find $OUT/p$i -name "*kernel_trace.csv" -delete
done
python $R/scripts/pmc_summary.py $OUT k_dcn_patch > $OUT/summary_k_dcn_patch.txt
python $R/scripts/pmc_summary.py $OUT "k_conv_f16ILi9ELi4" > $OUT/summary_k_conv_f16_9_4.txt
python $R/scripts/pmc_summary.py $OUT "k_conv_f16ILi1ELi4" > $OUT/summary_k_conv_f16_1_4.txt
cat $OUT/summary_k_dcn_patch.txt
