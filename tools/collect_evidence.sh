#!/bin/bash
# Round evidence on the GPU box (gpurun): bench lines, rocprofv3 kernel stats, PMC passes -> gpurun_out/ev/.
# Each rocprofv3 --pmc pass is its own run with --kernel-trace only.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/ev
mkdir -p $O
B="python bench.py"
$B > $O/bench_default.json 2> $O/bench_default.err
$B --batch 1 --no-cpu-baseline > $O/bench_unpipelined.json 2>/dev/null
for c in 0 1 3 4; do $B --config $c --no-cpu-baseline --steps 40 > $O/bench_config$c.json 2>/dev/null; done
$B --no-cpu-baseline --bn running --hidden reset > $O/bench_running_reset.json 2>/dev/null
$B --no-cpu-baseline --impl f32 --steps 30 > $O/bench_f32exact.json 2>/dev/null
$B --no-cpu-baseline --trace-flags 35 --batch 1 --steps 30 > $O/bench_sort_material.json 2>/dev/null
$B --no-cpu-baseline --prefetch > $O/bench_prefetch.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats -o k -- $B --no-cpu-baseline > $O/kstats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_unpipelined -o k -- $B --no-cpu-baseline --batch 1 > $O/kstats_unpipelined.log 2>&1
P="$B --steps 8 --warmup 4 --no-cpu-baseline --no-roofline-events"
pass() { tag=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- $P > $O/pmc_$tag.log 2>&1; echo "pass $tag rc=$?"; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
ls $O
