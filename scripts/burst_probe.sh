#!/bin/bash
# is a tile's prologue / epilogue longer because all workgroups of a round hit the memory system together?  Same kernel, same
# tile, 16 / 64 / 256 workgroups in flight (one round each) and the bench pyramid
cd $GRAFT_REPO_ROOT
export S2A_ALLOW_MEASURE_BUILD=1
trap 'rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s' EXIT
rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s EXTRA=-DS2A_STAMP=1 2>&1 | grep error
for l in "1:32x64" "2:64x64" "8:64x64" ""; do S2A_STAMP_LAYOUT=$l S2A_DCN_HALF_TAIL=0 timeout -k 10 200 python scripts/stamps_pyr.py 2>&1 | grep "layout\|matrix wave\|clock" | cut -c1-200; done
