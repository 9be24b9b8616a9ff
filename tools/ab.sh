#!/bin/bash
# A/B of bench.py option sets on the GPU box: tools/ab.sh REPS "opts A" "opts B" ...  -> frames/s of every run (interleaved), medians
# an option set that starts with LIB=NAME runs ab_variants/libaiptd_NAME.so (tools/build_variant.sh) instead of the product library
reps=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
declare -A vals
for r in $(seq 1 $reps); do
  i=0
  for o in "$@"; do
    lib=""; opts="$o"
    case "$o" in LIB=*) first="${o%% *}"; lib="ab_variants/libaiptd_${first#LIB=}.so"; opts="${o#"$first"}";; esac
    v=$(AIPT_LIB=${lib:+$PWD/$lib} python bench.py --no-cpu-baseline --no-roofline-events --steps 20 --warmup 5 $opts 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])")
    vals[$i]="${vals[$i]:-} $v"
    i=$((i+1))
  done
done
i=0
for o in "$@"; do
  echo "[$o] ${vals[$i]}  median $(echo ${vals[$i]} | tr ' ' '\n' | sort -n | awk '{a[NR]=$1} END{print a[int((NR+1)/2)]}')"
  i=$((i+1))
done
