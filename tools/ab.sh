#!/bin/bash
# A/B of bench.py option sets on the GPU box: tools/ab.sh REPS "opts A" "opts B" ...  -> frames/s of every run (interleaved), medians
reps=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
declare -A vals
for r in $(seq 1 $reps); do
  i=0
  for o in "$@"; do
    v=$(python bench.py --no-cpu-baseline --no-roofline-events --steps 20 --warmup 5 $o 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])")
    vals[$i]="${vals[$i]:-} $v"
    i=$((i+1))
  done
done
i=0
for o in "$@"; do
  echo "[$o] ${vals[$i]}  median $(echo ${vals[$i]} | tr ' ' '\n' | sort -n | awk '{a[NR]=$1} END{print a[int((NR+1)/2)]}')"
  i=$((i+1))
done
