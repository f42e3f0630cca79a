#!/bin/bash
# fused bottleneck tail: residual requested before the 3x3 GEMM (S2A_TAIL_EARLY_RES=1, shipped) vs behind the second GEMM (0); alternating builds
cd $GRAFT_REPO_ROOT
export S2A_ALLOW_MEASURE_BUILD=1
trap 'rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s' EXIT
for rep in 1 2; do
  for m in 1 0; do
    rm -f s2anet_amd/csrc/dcn_ops.o; make -C s2anet_amd/csrc -s EXTRA=-DS2A_TAIL_EARLY_RES=$m 2>&1 | grep error
    echo "EARLY=$m tail: $(timeout -k 10 200 python scripts/bench_tail.py 2>/dev/null | tail -1 | cut -c1-120)"
    echo "EARLY=$m bench: $(timeout -k 10 300 python bench.py --no-cpu-baseline --no-ops --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chips/s', d['value'])")"
  done
done
