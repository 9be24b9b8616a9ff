#!/bin/bash
# PMC passes over the denoiser's conv kernels (bench.py with one denoiser stream, so that a chip-wide counter belongs to one
# kernel): issue / wait / MFMA / LDS / L1 / L2 / HBM counters, each pass its own rocprofv3 run with --kernel-trace only.
#   gpurun -- 'TAG=r tools/prof_conv_pmc.sh'   ->  gpurun_out/convpmc_<TAG>/table.txt (+ .json): per-kernel averages
# Extra environment for the bench (kernel variants) goes in ENVS="A=1 B=2".
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${TAG:-r}
O=gpurun_out/convpmc_$TAG
rm -rf $O; mkdir -p $O
P="python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-roofline-events"
pass() { tag=$1; shift; env AIPT_DN_PIPELINE=0 ${ENVS:-} timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- $P > $O/pmc_$tag.log 2>&1; echo "pass $tag rc=$?"; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python tools/pmc_table.py $O/pmc_*/p_counter_collection.csv --kernel conv3x3 --json $O/table.json > $O/table.txt
rm -rf $O/pmc_*/
cat $O/table.txt
