#!/bin/bash
cd $GRAFT_REPO_ROOT
export S2A_ALLOW_MEASURE_BUILD=1   # the objects built below carry measurement switches (s2anet_amd/_lib.py refuses them otherwise)
trap 'rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s' EXIT
rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s EXTRA=-DS2A_STAMP=1 2>&1 | grep error
for a in "8 128" "1 128"; do timeout -k 10 200 python scripts/stamps.py $a 2>&1 | tail -5; done
timeout -k 10 200 python scripts/stamps_pyr.py 2>&1 | tail -4
timeout -k 10 200 python scripts/stamps_pyr.py zeros 2>&1 | tail -4
