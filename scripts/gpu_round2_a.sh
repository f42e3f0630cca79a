#!/bin/bash
# first GPU call of round 2: new parity tests, e2e diagnostics, NMS counter evidence (row d2), baseline bench line
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2a
mkdir -p $O
cd $R
echo "== new tests"; timeout -k 10 900 python -m pytest tests/test_gpu_e2e.py "tests/test_gpu_ops.py::test_pyramid_alignconv_and_refine" -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests.log
echo "== e2e diag"; timeout -k 10 600 python scripts/e2e_diag.py both > $O/diag.log 2>&1; echo "diag rc=$?"; tail -60 $O/diag.log
echo "== counters available"; (cd /tmp && TMPDIR=/tmp rocprofv3 -L > $O/counters.txt 2>&1); grep -c "" $O/counters.txt
echo "== nms pmc"
bash scripts/pmc_cmd.sh nms200k "scripts/bench_ops.py --which nms200k" "k_nms_cull k_nms_heavy k_nms_scan k_zero_words" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
  "SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU" \
  "FETCH_SIZE" "WRITE_SIZE" > $O/nms_pmc.log 2>&1; echo "pmc rc=$?"; tail -40 $O/nms_pmc.log
echo "== nms memory-copy trace"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/memcpy -o run -- python $R/scripts/bench_ops.py --which nms200k > $O/memcpy.log 2>&1); echo "memcpy rc=$?"
ls $O/memcpy $O/memcpy/* 2>/dev/null | head -20
python - <<PY
import csv, glob
for f in glob.glob("$O/memcpy/**/*memory_copy_trace.csv", recursive=True) + glob.glob("$O/memcpy/*memory_copy_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    print(f, len(rows), "copies")
    import collections
    agg = collections.Counter(); byt = collections.Counter()
    for r in rows:
        k = r.get("Direction") or r.get("Name") or "?"
        agg[k] += 1; byt[k] += int(r.get("Bytes", r.get("Size", 0)) or 0)
    for k in agg: print("  ", k, agg[k], "copies", byt[k], "bytes")
PY
echo "== bench"; timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json | head -c 3000
echo "== ops"; timeout -k 10 300 python scripts/bench_ops.py --which nms > $O/ops_nms.jsonl 2>&1; cat $O/ops_nms.jsonl
timeout -k 10 300 python scripts/bench_ops.py --which iou > $O/ops_iou.jsonl 2>&1; cat $O/ops_iou.jsonl
timeout -k 10 300 python scripts/bench_ops.py --which align > $O/ops_align.jsonl 2>&1; cat $O/ops_align.jsonl
