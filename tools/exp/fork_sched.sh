# EXPERIMENT (round 5): which weight gradients share a fork of the side stream?  P2P_FORK_AT = layers at which the pending weight
# gradients are issued behind ONE fork (engine._wgrad, experimental knob); P2P_FORK_GROUP=n = every n-th.  c2, same box, alternating.
mkdir -p gpurun_out/r05o
T1="G.up4,G.up1,G.down4,G.down1"
T2="G.up3,G.down6,G.down4,G.down1"
T3="G.up4,G.up3,G.up2,G.up1,G.down6,G.down5,G.down4,G.down1"
T4="G.up4,G.down1"
T5="G.up5,G.up2,G.down5,G.down1"
T6="G.up4,G.up1,G.down4,G.down3,G.down2,G.down1"
T7="G.up3,G.down4,G.down1"
T8="G.up4,G.down6,G.down1"
for i in 1 2; do
  P2P_STOP_EVENT_FORKS=0 P2P_FORK_GROUP=1 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-profile --no-feed-profile > gpurun_out/r05o/t_c2_T0_$i.log 2>&1
  for n in 1 2 3 4 5 6 7 8; do
    eval "AT=\$T$n"
    P2P_STOP_EVENT_FORKS=0 P2P_FORK_GROUP=99 P2P_FORK_AT=$AT python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-profile --no-feed-profile > gpurun_out/r05o/t_c2_T${n}_$i.log 2>&1
  done
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r05o/t_*.log")):
    for l in open(f):
        if l.startswith("{"):
            r=json.loads(l); print(f.split("/")[-1], r["value"], r["ms_per_step"], r["losses"][:1])
PY
