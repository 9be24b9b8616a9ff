#!/bin/bash
# gpurun_out/ev (tools/collect_evidence.sh) -> profiles/<round>_*: bench lines, kernel stats, PMC summaries
set -eu
R=${1:?round tag, e.g. r02}
E=gpurun_out/ev
for f in $E/bench_*.json; do n=$(basename $f); cp $f profiles/${R}_$n; done
cp $E/layers.txt profiles/${R}_layers.txt
cp $(find $E/kstats -name "*kernel_stats.csv" | head -1) profiles/${R}_kernel_stats.csv
cp $(find $E/kstats_one_stream -name "*kernel_stats.csv" | head -1) profiles/${R}_kernel_stats_one_denoiser_stream.csv
cp $E/pmc_dominant.json profiles/pmc_dominant.json
cp $E/pmc_trace.json profiles/${R}_pmc_trace.json
cp $E/pmc_conv.json profiles/${R}_pmc_conv.json
for f in conv_ablate.txt ubench_valu_mfma.txt timeline_two_streams.txt; do [ -f $E/$f ] && cp $E/$f profiles/${R}_$f; done
for f in $E/drift_*.json; do [ -f $f ] && cp $f profiles/${R}_$(basename $f); done
ls profiles | grep "^${R}_"
