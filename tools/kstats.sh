#!/bin/bash
# rocprofv3 kernel statistics of one bench.py command on the GPU box (through gpurun): tools/kstats.sh TAG [bench args...]
#   -> gpurun_out/ks_TAG/k_kernel_stats.csv, and the aipt:: rows on stdout (one denoiser stream: a duration belongs to one kernel)
set -u
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/ks_$tag
rm -rf $O; mkdir -p $O
AIPT_DN_PIPELINE=${AIPT_DN_PIPELINE:-0} rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python bench.py --no-cpu-baseline --no-roofline-events --steps 20 --warmup 5 "$@" > $O/bench.log 2>&1
python - "$O/k_kernel_stats.csv" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "aipt::" in n:
        print(f'{n.replace("void aipt::", "").replace("aipt::", "").split("(")[0]:58s} calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"]) / 1e3:8.1f} us  min {float(r["MinNs"]) / 1e3:7.1f}  total {float(r["TotalDurationNs"]) / 1e6:8.2f} ms')
P
tail -1 $O/bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('frames/s', d['value'], 'ms/frame', d['ms_per_step'])"
rm -f $O/k_kernel_trace.csv
