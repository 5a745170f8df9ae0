#!/bin/bash
# Regenerates the measurement artefacts of one round on the GPU box (run through gpurun from the repository root):
#   bash tools/profile_round.sh r02 c2            -> gpurun_out/<tag>/...   (copy the summaries into profiles/ afterwards)
# One rocprofv3 pass per counter group (TCC FETCH_SIZE / WRITE_SIZE need their own passes, MI355X_MICROARCH.md HBM section);
# --pmc is never combined with any trace domain other than --kernel-trace.
set -e -o pipefail
TAG=${1:-r02}; CFG=${2:-c2}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
rocprofv3 --kernel-trace --stats -d $OUT/kt_$CFG -o kt --output-format csv -- python3 bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-feed-profile > $OUT/kt_$CFG.log 2>&1
echo "kernel trace done"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_${C}_$CFG -o pmc --output-format csv -- python3 bench.py --config $CFG --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-feed-profile > $OUT/pmc_${C}_$CFG.log 2>&1
  echo "$C done"
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $OUT/pmc_sq_$CFG -o pmc --output-format csv -- python3 bench.py --config $CFG --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-feed-profile > $OUT/pmc_sq_$CFG.log 2>&1
echo "SQ done"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $OUT/pmc_lds_$CFG -o pmc --output-format csv -- python3 bench.py --config $CFG --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-feed-profile > $OUT/pmc_lds_$CFG.log 2>&1
echo "LDS done"
python3 tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE_$CFG $OUT/pmc_WRITE_SIZE_$CFG $OUT/pmc_traffic_${CFG}_bf16.json $CFG > $OUT/pmc_traffic_$CFG.txt
# the bench line LAST: it quotes roofline.traffic from profiles/pmc_traffic_<config>_bf16.json, which must carry the fingerprint of the
# sources it runs from -- the counter passes above have just produced it
cp $OUT/pmc_traffic_${CFG}_bf16.json profiles/pmc_traffic_${CFG}_bf16.json
python3 bench.py --config $CFG --steps 50 --warmup 10 --detail $OUT/per_call_$CFG.txt > $OUT/bench_$CFG.log 2>&1
grep '^{' $OUT/bench_$CFG.log | tail -1 > $OUT/bench_$CFG.json
echo "bench done"
python3 tools/pmc_sq.py $OUT/pmc_sq_$CFG $OUT/pmc_sq_${CFG}_bf16.json > $OUT/pmc_sq_$CFG.txt
python3 tools/pmc_sq.py $OUT/pmc_lds_$CFG $OUT/pmc_lds_${CFG}_bf16.json > $OUT/pmc_lds_$CFG.txt
python3 tools/trace_timeline.py $OUT/kt_$CFG > $OUT/timeline_$CFG.txt 2>&1 || true
# keep only the summaries of the raw traces (the per-dispatch CSVs are tens of MB)
find $OUT -name '*kernel_trace.csv' -size +8M -delete || true
echo "all done"
