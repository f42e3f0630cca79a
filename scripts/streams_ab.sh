cd $GRAFT_REPO_ROOT
for cfg in "--streams 3" "--streams 3 --graph" "--streams 4" "--streams 4 --graph" "--streams 6 --graph" "--streams 2 --graph"; do
  echo "[$cfg] $(timeout -k 10 300 python bench.py --no-cpu-baseline $cfg 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
