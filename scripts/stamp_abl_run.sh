#!/bin/bash
# stamped ablation builds of the pyramid AlignConv launch: cycles AND in-kernel clock per form, dense and zero data
cd $GRAFT_REPO_ROOT
export S2A_ALLOW_MEASURE_BUILD=1
trap 'rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s' EXIT
for abl in "$@"; do
  rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s EXTRA="-DS2A_STAMP=1 -DS2A_ABL=$abl" 2>&1 | grep error
  echo "== S2A_ABL=$abl"
  timeout -k 10 200 python scripts/stamps_pyr.py 2>&1 | tail -4 | cut -c1-230
  timeout -k 10 200 python scripts/stamps_pyr.py zeros 2>&1 | tail -4 | cut -c1-230
done
