#!/bin/bash
# measurement build: device-side NMS totals + cull phase stamps of the detector's own segmented NMS call (bench step)
cd $GRAFT_REPO_ROOT
export S2A_ALLOW_MEASURE_BUILD=1   # the objects built below carry measurement switches (s2anet_amd/_lib.py refuses them otherwise)
restore() { rm -f s2anet_amd/csrc/rotated_ops.o; make -C s2anet_amd/csrc -s 2>&1 | grep -E "error" | head -3; }
trap restore EXIT
rm -f s2anet_amd/csrc/rotated_ops.o
make -C s2anet_amd/csrc -s EXTRA="-DS2A_MEASURE" 2>&1 | grep -E "error" | head -3
S2A_NMS_DEBUG=1 python bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-ops 2>&1 | grep "\[nms\]\|\[cull\]\|\[finish\]" | tail -6
