# probe, round 5 (run through gpurun from the repository root), last form: the early part of Adam on an optimizer stream beside the rest of
# the tail -- with tools/exp/tail_overlap.patch applied (P2P_TAIL_OVERLAP does not exist in the tree).  Earlier forms of this script ran the
# A/B matrices of profiles/r05_exp_small_batch.txt (fused block on / off, K-split targets, norm kernel switches, batch sweep).
set -e
mkdir -p gpurun_out/r05n
O=gpurun_out/r05n
timeout -k 10 900 python -m pytest tests/test_train_step_gpu.py tests/test_dp_gpu.py tests/test_models_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
run() {  # config, tag, env...
  cfg=$1; tag=$2; shift; shift
  st=300; [ $cfg = c1 ] || st=100
  env "$@" timeout -k 10 300 python bench.py --config $cfg --steps $st --warmup 20 --no-cpu-baseline --no-feed-profile --no-profile > $O/bench_${cfg}_$tag.json 2>$O/bench_${cfg}_$tag.err
  python -c "
import json
d=json.loads(open('$O/bench_${cfg}_$tag.json').read().strip().splitlines()[-1])
print('$cfg $tag', d['value'], d['ms_per_step'])"
}
for cfg in c1 c2 c3 c4 c5; do
  run $cfg on A=1
  run $cfg off P2P_TAIL_OVERLAP=0
  run $cfg on2 A=1
  run $cfg off2 P2P_TAIL_OVERLAP=0
done
echo done
