# probes, round 5 (run through gpurun from the repository root): workgroups wanted by the register-resident InstanceNorm kernels
set -e
mkdir -p gpurun_out/r05n
O=gpurun_out/r05n
run() {  # config, tag, env...
  cfg=$1; tag=$2; shift; shift
  st=300; [ $cfg = c1 ] || st=100
  env "$@" timeout -k 10 300 python bench.py --config $cfg --steps $st --warmup 20 --no-cpu-baseline --no-feed-profile --detail $O/percall_${cfg}_$tag.txt > $O/bench_${cfg}_$tag.json 2>$O/bench_${cfg}_$tag.err
  python -c "
import json
d=json.loads(open('$O/bench_${cfg}_$tag.json').read().strip().splitlines()[-1])
k=d['kernel_ms_per_step']
print('$cfg $tag', d['value'], d['ms_per_step'], 'norm_fwd', k.get('p2p_norm_act_fwd'), 'norm_bwd', k.get('p2p_norm_act_bwd'), 'serial', d.get('serialised_kernel_ms'))"
}
for w in 512 1024 2048 4096 256 512; do
  run c2 w$w P2P_NORM_REG_WGS=$w
done
run c2 off P2P_NORM_FWD_REG=0 P2P_NORM_BWD_REG=0
run c4 w512 P2P_NORM_REG_WGS=512
run c4 w2048 P2P_NORM_REG_WGS=2048
run c5 w512 P2P_NORM_REG_WGS=512
run c5 w2048 P2P_NORM_REG_WGS=2048
run c5 off P2P_NORM_FWD_REG=0 P2P_NORM_BWD_REG=0
echo done
