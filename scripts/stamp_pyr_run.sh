#!/bin/bash
cd $GRAFT_REPO_ROOT
export S2A_ALLOW_MEASURE_BUILD=1
trap 'rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s' EXIT
rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s EXTRA="-DS2A_STAMP=1" 2>&1 | grep error
S2A_DCN_MW8=0 timeout -k 10 200 python scripts/stamps_pyr.py 2>&1 | tail -3
rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s EXTRA="-DS2A_STAMP=1 -DS2A_STAMP_W2=8" 2>&1 | grep error
S2A_DCN_MW8=1 timeout -k 10 200 python scripts/stamps_pyr.py 2>&1 | tail -3
