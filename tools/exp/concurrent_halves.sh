#!/bin/bash
# Experiment: how much does the chip gain from two (four) independent half- (quarter-) batch steps in flight at once?
# N processes, each the c2 step at batch 256 / N, started together; per-process ms/step and the combined images/s.
cd "$(dirname "$0")/../.."
for N in 1 2 4; do
  B=$((256 / N))
  for i in $(seq 1 $N); do
    python bench.py --batch $B --steps 300 --warmup 30 --no-cpu-baseline --no-profile > gpurun_out/conc_${N}_$i.json 2>/dev/null &
  done
  wait
  python3 - <<PY
import json
tot=0
for i in range(1,$N+1):
    d=json.loads(open("gpurun_out/conc_${N}_%d.json"%i).read().strip().splitlines()[-1])
    tot+=d["value"]; print("N=$N proc",i,"B=$B ms/step",d["ms_per_step"])
print("N=$N combined images/s", round(tot))
PY
done
