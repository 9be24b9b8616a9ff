#!/bin/bash
# tools/abenv.sh REPS "ENV=.. opts" ... : like ab.sh but option sets may start with VAR=value environment assignments
reps=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
declare -A vals
for r in $(seq 1 $reps); do
  i=0
  for o in "$@"; do
    v=$(env $o python bench.py --no-cpu-baseline --no-roofline-events --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['frame_by_frame']['value'], d['frame_by_frame']['prefetch']['value'])")
    vals[$i]="${vals[$i]:-} | $v"
    i=$((i+1))
  done
done
i=0
for o in "$@"; do echo "[$o] ${vals[$i]}"; i=$((i+1)); done
