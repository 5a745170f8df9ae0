# small-batch probes, round 5 (run through gpurun from the repository root)
set -e
mkdir -p gpurun_out/r05n
O=gpurun_out/r05n
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_train_step_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
run() {  # config, tag, env...
  cfg=$1; tag=$2; shift; shift
  st=300; [ $cfg = c1 ] || st=100
  env "$@" timeout -k 10 300 python bench.py --config $cfg --steps $st --warmup 20 --no-cpu-baseline --no-feed-profile --detail $O/percall_${cfg}_$tag.txt > $O/bench_${cfg}_$tag.json 2>$O/bench_${cfg}_$tag.err
  python -c "
import json
d=json.loads(open('$O/bench_${cfg}_$tag.json').read().strip().splitlines()[-1])
k=d['kernel_ms_per_step']
print('$cfg $tag', d['value'], d['ms_per_step'], 'igemm', k.get('p2p_igemm'), 'norm_fwd', k.get('p2p_norm_act_fwd'), 'norm_bwd', k.get('p2p_norm_act_bwd'), 'serial', d.get('serialised_kernel_ms'))"
}
run c1 a A=1
run c1 b A=1
run c2 a A=1
echo done
