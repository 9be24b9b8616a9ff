#!/bin/bash
# Round evidence on the GPU box (gpurun): bench lines, rocprofv3 kernel stats, PMC passes -> gpurun_out/ev/.
# Each rocprofv3 --pmc pass is its own run with --kernel-trace only.  Afterwards (build container):
#   tools/install_evidence.sh rNN     copies the summaries into profiles/
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/ev
rm -rf $O; mkdir -p $O
B="python bench.py"
# PMC: the bench command with one denoiser stream and one trace lane, so that a counter belongs to one kernel (two kernels in flight share the
# chip-wide counters); a bounce launch covers 16 frames either way
P="$B --steps 20 --warmup 20 --no-cpu-baseline --no-roofline-events"   # (every batched trace launch holds 20 frames, as in the driver's command)
pass() { tag=$1; shift; AIPT_DN_PIPELINE=0 AIPT_TRACE_LANES=1 timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- $P > $O/pmc_$tag.log 2>&1; echo "pass $tag rc=$?"; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
python tools/pmc_summarize.py --frames-per-launch 20 --out $O/pmc_dominant.json $O/pmc_fetch/p_counter_collection.csv $O/pmc_write/p_counter_collection.csv \
    'trace_bounce<false,true>' 'trace_bounce<true,true>' 'conv3x3_f16x3r<false,12,3,false,4,false>' 'conv3x3_f16x3r<false,8,3,false,4,true>' \
    'conv3x3_f16x3<1,8,false,false,1>' 'conv3x3_f16x3<1,4,false,false,3>' 'conv3x3_quad<3,3,false>' 'conv3x3_quad<3,3,true>'
python tools/pmc_table.py $O/pmc_*/p_counter_collection.csv --kernel trace_bounce --json $O/pmc_trace.json > /dev/null
python tools/pmc_table.py $O/pmc_*/p_counter_collection.csv --kernel 'conv3x3_f16x3' --json $O/pmc_conv.json > /dev/null
# the bench lines read roofline.traffic from profiles/pmc_dominant.json: the fresh one
cp $O/pmc_dominant.json profiles/pmc_dominant.json
$B > $O/bench_default.json 2> $O/bench_default.err
$B --no-cpu-baseline --layers > /dev/null 2> $O/layers.txt
$B --batch 1 --no-cpu-baseline > $O/bench_frame_by_frame.json 2>/dev/null
$B --batch 1 --prefetch --no-cpu-baseline > $O/bench_frame_by_frame_prefetch.json 2>/dev/null
AIPT_DN_PIPELINE=0 $B --no-cpu-baseline > $O/bench_one_denoiser_stream.json 2>/dev/null
AIPT_TRACE_LANES=1 $B --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_one_trace_lane.json 2>/dev/null
for c in 0 1 3 4; do $B --config $c --no-cpu-baseline > $O/bench_config$c.json 2>/dev/null; done
$B --no-cpu-baseline --bn running --hidden reset > $O/bench_running_reset.json 2>/dev/null
$B --no-cpu-baseline --impl f32 --steps 32 > $O/bench_f32exact.json 2>/dev/null
$B --no-cpu-baseline --trace-flags 35 --batch 1 --steps 30 > $O/bench_sort_material.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats -o k -- $B --no-cpu-baseline > $O/kstats.log 2>&1
AIPT_DN_PIPELINE=0 AIPT_TRACE_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_one_stream -o k -- $B --no-cpu-baseline > $O/kstats_one_stream.log 2>&1
# round 4: the driver's own command first (what BENCH_rNN.json reproduces), 300-frame pans of configs[1] / [2] (BASELINE.json),
# the drift study ON the register-staged kernel, the two-stream timeline with the call queued behind a spin
$B --steps 20 --warmup 5 > $O/bench_driver_command.json 2>/dev/null
$B --steps 300 --warmup 32 --no-cpu-baseline > $O/bench_config2_300frames.json 2>/dev/null
$B --config 1 --steps 300 --warmup 32 --no-cpu-baseline > $O/bench_config1_300frames.json 2>/dev/null
python tools/drift_probe.py 384 640 16 565 > $O/drift_384x640_default_selection.json 2>/dev/null
python tools/drift_probe.py 192 320 32 565 0 > $O/drift_192x320_r_minpix0.json 2>/dev/null
python tools/drift_probe.py 192 320 32 7 0 > $O/drift_192x320_r_minpix0_seed7.json 2>/dev/null
rocprofv3 --kernel-trace --output-format csv -d $O/timeline -o k -- $B --no-cpu-baseline --no-roofline-events --steps 20 --warmup 5 --gate-ms 60 > $O/timeline.log 2>&1
python tools/timeline_summary.py $O/timeline/k_kernel_trace.csv > $O/timeline_two_streams.txt 2>&1
rm -rf $O/timeline
find $O -name "*_kernel_stats.csv" | head; ls $O
# keep the merge small: raw counter CSVs stay on the box
rm -rf $O/pmc_*/ 
