#!/bin/bash
# timing-only ablations of the fused deform-conv backward kernels (S2A_BWD_ABL bits, see dcn_bwd_ops.hip):
#   bash scripts/abl_bwd.sh <bwd16|bwd32> <bits> [<bits> ...]     kernel times from a rocprofv3 kernel trace
cd $GRAFT_REPO_ROOT
export S2A_ALLOW_MEASURE_BUILD=1 TMPDIR=/tmp
W=$1; shift
trap 'rm -f s2anet_amd/csrc/dcn_bwd_ops.o; make -C s2anet_amd/csrc -s' EXIT
for a in "$@"; do
  rm -f s2anet_amd/csrc/dcn_bwd_ops.o; make -C s2anet_amd/csrc -s EXTRA="-DS2A_BWD_ABL=$a" 2>&1 | grep error
  echo "[S2A_BWD_ABL=$a] $(timeout -k 10 200 python scripts/bench_ops.py --which $W 2>/dev/null | cut -c1-150)"
  O=$GRAFT_REPO_ROOT/gpurun_out/abl_bwd_$a; mkdir -p $O
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python $GRAFT_REPO_ROOT/scripts/bench_ops.py --which $W > $O/run.log 2>&1)
  S=$(ls $O/*kernel_stats.csv $O/*/*kernel_stats.csv 2>/dev/null | head -1)
  python -c "
import csv,sys
for r in csv.DictReader(open('$S')):
    if 'k_dcn_bwd' in r['Name']: print('    %-40s %8.1f us' % (r['Name'].split('::')[-1][:40], float(r['AverageNs'])/1e3))"
  rm -rf $O
done
