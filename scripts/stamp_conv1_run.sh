#!/bin/bash
# diagnostic build with in-kernel phase stamps for the full-width 1x1 layers; the shipped library is rebuilt afterwards
cd $GRAFT_REPO_ROOT
export S2A_ALLOW_MEASURE_BUILD=1
trap 'rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s' EXIT
rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s EXTRA="-DS2A_STAMP=1 -DS2A_STAMP_W2=3" 2>&1 | grep error
timeout -k 10 200 python scripts/stamps_conv1.py 2>&1 | grep -v amdgpu.ids | tail -10
