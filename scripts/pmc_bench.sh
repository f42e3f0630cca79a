#!/bin/bash
# PMC passes over the default bench command for the AlignConv (k_dcn_patch) and conv-tower (k_conv_f16)
# launches.  Counters only (+ kernel trace); FETCH_SIZE and WRITE_SIZE in separate passes (TCC slot
# limits, MI355X_MICROARCH.md HBM section).  Writes gpurun_out/pmc_bench/traffic.json = the record bench.py
# reports as roofline.traffic once it is copied to profiles/rNN_traffic.json.
#   bash scripts/pmc_bench.sh [hbm]      hbm: only the FETCH_SIZE / WRITE_SIZE passes
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_bench
rm -rf $OUT; mkdir -p $OUT
SETS=("FETCH_SIZE" "WRITE_SIZE")
if [ "$1" != "hbm" ]; then
  SETS+=("TCC_HIT_sum TCC_MISS_sum"
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
         "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE")
fi
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-ops > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; exit 1; }
  find $OUT/p$i -name "*kernel_trace.csv" -delete
done
python $R/scripts/pmc_summary.py $OUT k_dcn_patch > $OUT/summary_k_dcn_patch.txt
python $R/scripts/pmc_summary.py $OUT "k_conv_f16ILi9ELi4" > $OUT/summary_k_conv_f16_9_4.txt
python $R/scripts/pmc_summary.py $OUT --traffic-json $OUT/traffic.json \
    align_conv_pyramid=k_dcn_patch,8,174592 conv_tower_pyramid=k_conv_f16ILi9ELi4ELi2,8,174592
cat $OUT/summary_k_dcn_patch.txt
