#!/bin/bash
# Three PMC passes over the conv kernels for one library build (A/B of kernel variants in CYCLES): TAG=name ENVS="AIPT_LIB=..." tools/prof_conv_pmc_short.sh
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${TAG:-r}
O=gpurun_out/convpmc_$TAG
rm -rf $O; mkdir -p $O
P="python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-roofline-events"
pass() { tag=$1; shift; env AIPT_DN_PIPELINE=0 ${ENVS:-} timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- $P > $O/pmc_$tag.log 2>&1; echo "pass $tag rc=$?"; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
python tools/pmc_table.py $O/pmc_*/p_counter_collection.csv --kernel conv3x3_f16x3r --json $O/table.json > $O/table.txt
rm -rf $O/pmc_*/
cat $O/table.txt
