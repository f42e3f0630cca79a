#!/bin/bash
# phase stamps of the fused f32 weight-gradient kernel: measurement build, restored on exit
cd $GRAFT_REPO_ROOT
export S2A_ALLOW_MEASURE_BUILD=1
trap 'rm -f s2anet_amd/csrc/dcn_bwd_ops.o; make -C s2anet_amd/csrc -s' EXIT
# bash scripts/bwd32_stamps.sh [extra flags]        the f32 kernel;   bash scripts/bwd32_stamps.sh f16      the f16 kernel
if [ "$1" = "f16" ]; then
  rm -f s2anet_amd/csrc/dcn_bwd_ops.o; make -C s2anet_amd/csrc -s EXTRA="-DS2A_MEASURE -DS2A_MEASURE_F16W" 2>&1 | grep error
  timeout -k 10 200 python scripts/bwd16_stamps.py
  exit
fi
rm -f s2anet_amd/csrc/dcn_bwd_ops.o; make -C s2anet_amd/csrc -s EXTRA="-DS2A_MEASURE $1" 2>&1 | grep error
# (the stamps sit in the f32-instruction kernel; since round 6 the default f32 weight gradient is the three-plane kernel)
S2A_BWD_F32_WEIGHT=mfma32 timeout -k 10 200 python scripts/bwd32_stamps.py
