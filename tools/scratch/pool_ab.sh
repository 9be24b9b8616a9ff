cd "${GRAFT_REPO_ROOT:-.}"
B="python bench.py --no-cpu-baseline --no-roofline-events --steps 20 --warmup 5"
for r in 1 2 3; do
 for e in "AIPT_TRACE_POOL=1" "AIPT_TRACE_POOL=0" "AIPT_TRACE_POOL=0 AIPT_TRACE_LANES=1"; do
  v=$(env $e $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])")
  echo "$e $v"
 done
done
for b in 2 4 8; do for e in "AIPT_TRACE_POOL=1" "AIPT_TRACE_POOL=0"; do
  v=$(env $e $B --batch $b 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])"); echo "batch $b $e $v"; done; done
