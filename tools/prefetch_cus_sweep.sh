#!/bin/bash
# aipt_frame_prefetch: frames/s of bench.py --batch 1 --prefetch against the trace stream's share of the CUs (AIPT_PREFETCH_TRACE_CUS);
# DESIGN.md 5 "What stays": the denoiser's share per XCD has to be a multiple of 4
cd "${GRAFT_REPO_ROOT:-.}"
B="python bench.py --no-cpu-baseline --no-roofline-events --steps 40 --warmup 10 --batch 1 --prefetch"
for r in 1; do for c in 64 72 80 88 96 104 112 120 128 136; do
  v=$(AIPT_PREFETCH_TRACE_CUS=$c $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])"); echo "trace_cus $c $v"; done; done
