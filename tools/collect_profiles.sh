#!/bin/bash
# Copies the judged summaries of one round's measurement runs (gpurun_out/<tag>/, written by tools/profile_round.sh) into profiles/:
#   bash tools/collect_profiles.sh r03 c2 c3 c4 c5
TAG=$1; shift
for CFG in "$@"; do
  SRC=gpurun_out/$TAG
  [ -f $SRC/bench_$CFG.json ] && cp $SRC/bench_$CFG.json profiles/${TAG}_bench_${CFG}_bf16.json
  [ -f $SRC/per_call_$CFG.txt ] && cp $SRC/per_call_$CFG.txt profiles/${TAG}_bench_${CFG}_bf16_per_call.txt
  [ -f $SRC/kt_$CFG/kt_kernel_stats.csv ] && cp $SRC/kt_$CFG/kt_kernel_stats.csv profiles/${TAG}_bench_${CFG}_bf16_kernel_stats.csv
  [ -f $SRC/pmc_traffic_${CFG}_bf16.json ] && cp $SRC/pmc_traffic_${CFG}_bf16.json profiles/pmc_traffic_${CFG}_bf16.json
  [ -f $SRC/pmc_sq_${CFG}_bf16.json ] && cp $SRC/pmc_sq_${CFG}_bf16.json profiles/${TAG}_pmc_sq_${CFG}_bf16.json
  [ -f $SRC/pmc_lds_${CFG}_bf16.json ] && cp $SRC/pmc_lds_${CFG}_bf16.json profiles/${TAG}_pmc_lds_${CFG}_bf16.json
  [ -s $SRC/timeline_$CFG.txt ] && cp $SRC/timeline_$CFG.txt profiles/${TAG}_timeline_${CFG}.txt
done
ls -la profiles | tail -30
