#!/bin/bash
# timing-only ablations of the fused f16 deform-conv backward kernels (k_dcn_bwd_input: S2A_BWD_ABL bits, see dcn_bwd_ops.hip)
cd $GRAFT_REPO_ROOT
export S2A_ALLOW_MEASURE_BUILD=1
trap 'rm -f s2anet_amd/csrc/dcn_bwd_ops.o; make -C s2anet_amd/csrc -s' EXIT
for a in "$@"; do
  rm -f s2anet_amd/csrc/dcn_bwd_ops.o; make -C s2anet_amd/csrc -s EXTRA="-DS2A_BWD_ABL=$a" 2>&1 | grep error
  echo "[S2A_BWD_ABL=$a] $(timeout -k 10 200 python scripts/bench_ops.py --which bwd16 2>/dev/null | cut -c1-190)"
done
