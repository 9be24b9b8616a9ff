#!/bin/bash
# PMC passes over the mesh workload (run on the GPU box through gpurun).  Each pass is a separate rocprofv3 run with
# --kernel-trace only (never combined with other trace domains).  Output: gpurun_out/pmc_<tag>/p_counter_collection.csv
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out
CMD="python bench.py --mesh ${MESH:-262144} --steps 4 --warmup 2 --no-cpu-baseline --no-roofline-events ${BENCH_EXTRA:-}"
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1
pass() {
  tag=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$tag -o p -- $CMD > $OUT/pmc_$tag.log 2>&1
  echo "pass $tag rc=$?"
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
ls $OUT/pmc_*/ 
