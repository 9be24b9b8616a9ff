#!/bin/bash
# PMC passes over the batched trace alone (tools/trace_bench.py --batch 8 at 1280x720): VALU issue, L1 (TCP) accesses, L2
# requests of the bounce kernel, for the variants named in VARIANTS ("tag:ENV=VAL,ENV=VAL:extra trace_bench args").
# Each pass is its own rocprofv3 run with --kernel-trace only.  Output: gpurun_out/walk_<tag>_<pass>/
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
VARIANTS=${VARIANTS:-"pool::"}
for v in $VARIANTS; do
  tag=${v%%:*}; rest=${v#*:}; envs=${rest%%:*}; extra=${rest#*:}
  CMD="python tools/trace_bench.py --batch 8 --frames 4 --sizes 1280x720 ${extra//,/ }"
  pass() {
    p=$1; shift
    env ${envs//,/ } timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/walk_${tag}_$p -o p -- $CMD > $OUT/walk_${tag}_$p.log 2>&1
    echo "variant $tag pass $p rc=$?"
  }
  pass sq SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
  pass tcp TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
  pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
done
