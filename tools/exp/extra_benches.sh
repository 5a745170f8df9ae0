set -e
O=gpurun_out/r05; mkdir -p $O
python3 bench.py --config c1 --steps 300 --warmup 20 --detail $O/per_call_c1.txt > $O/bench_c1.log 2>&1; grep '^{' $O/bench_c1.log | tail -1 > $O/bench_c1.json
python3 bench.py --config c2 --steps 20 --warmup 5 > $O/bench_c2_driver.log 2>&1; grep '^{' $O/bench_c2_driver.log | tail -1 > $O/bench_c2_driver.json
python3 bench.py --config c2 --dtype f32 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2_f32.log 2>&1; grep '^{' $O/bench_c2_f32.log | tail -1 > $O/bench_c2_f32.json
python3 bench.py --config c2 --host-batches --steps 100 --warmup 10 --no-cpu-baseline --no-profile --no-feed-profile > $O/bench_c2_host.log 2>&1; grep '^{' $O/bench_c2_host.log | tail -1 > $O/bench_c2_host.json
python3 bench.py --config c1 --host-batches --steps 300 --warmup 20 --no-cpu-baseline --no-profile --no-feed-profile > $O/bench_c1_host.log 2>&1; grep '^{' $O/bench_c1_host.log | tail -1 > $O/bench_c1_host.json
for b in 1 2 8 16 32 64 128; do python3 bench.py --config c2 --batch $b --steps 200 --warmup 20 --no-cpu-baseline --no-profile --no-feed-profile 2>/dev/null | grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch', $b, d['value'], d['ms_per_step'])"; done > $O/batch_sweep.txt
cat $O/batch_sweep.txt
python3 -c "
import json
for n in ('c1','c2_driver','c2_f32','c2_host','c1_host'):
    d=json.load(open('$O/bench_%s.json'%n)); print(n, d['value'], d['ms_per_step'], d.get('host_issue_ms_per_step'), d.get('host_issue_ms_per_step_eager'), d.get('ms_per_step_eager_issue'), d.get('ms_per_step_one_stream'), d.get('serialised_kernel_ms'))"
