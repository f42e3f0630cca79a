#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2d; mkdir -p $O; cd $R
for args in "76770 5 2" "76770 100 2" "76770 5 1" "76770 100 1" "76770 5 0" "76770 100 0" "641280 1 0"; do
  timeout -k 5 60 scripts/repro/select_graph $args > $O/out.log 2>&1; rc=$?
  echo "== args [$args] rc=$rc: $(grep -v '^\s' $O/out.log | grep -v amdgpu.ids | tr '\n' ';' | cut -c1-230)"
done
exit 0
