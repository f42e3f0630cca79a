#!/bin/bash
# phase stamps of k_nms_small (measurement build of rotated_ops, restored afterwards)
cd $GRAFT_REPO_ROOT
export S2A_ALLOW_MEASURE_BUILD=1
trap 'rm -f s2anet_amd/csrc/rotated_ops.o; make -C s2anet_amd/csrc -s' EXIT
rm -f s2anet_amd/csrc/rotated_ops.o; make -C s2anet_amd/csrc -s EXTRA=-DS2A_MEASURE 2>&1 | grep error
timeout -k 10 200 python scripts/nms_small_probe.py ${1:-5000} 2>&1 | grep -v amdgpu
