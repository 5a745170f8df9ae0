set -e -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_c3; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
export P2P_HIST_BWD3=0
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $OUT/sq -o pmc --output-format csv -- python3 bench.py --config c3 --steps 3 --warmup 2 --no-cpu-baseline --no-profile > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace -d $OUT/lds -o pmc --output-format csv -- python3 bench.py --config c3 --steps 3 --warmup 2 --no-cpu-baseline --no-profile > $OUT/lds.log 2>&1
python3 tools/pmc_sq.py $OUT/sq $OUT/sq.json > $OUT/sq.txt
python3 tools/pmc_sq.py $OUT/lds $OUT/lds.json > $OUT/lds.txt
find $OUT -name '*kernel_trace.csv' -size +8M -delete || true
find $OUT -name '*counter_collection.csv' -size +8M -delete || true
grep -i "hist" $OUT/sq.txt $OUT/lds.txt
