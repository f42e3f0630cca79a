#!/bin/bash
# 1x1 layers: 128- / 64-position tiles and narrower out-channel groups (measurement build; the shipped library is rebuilt afterwards)
cd $GRAFT_REPO_ROOT
export S2A_ALLOW_MEASURE_BUILD=1
trap 'rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s' EXIT
rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s EXTRA=-DS2A_MEASURE 2>&1 | grep error
timeout -k 10 300 python scripts/bench_conv1.py 2>&1 | grep -v amdgpu.ids
