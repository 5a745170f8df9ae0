# sweep of the packed (few-channel) weight-gradient layers: workgroups per launch, double buffer, strip height
cd $GRAFT_REPO_ROOT
export UB_ONLY=last,D.last,D.down,down1 UB_REPS=30
for th in 8 4; do for db in 0 1; do for want in 256 512 768 1024; do
  echo "== TH_PACK=$th DBUF_PACK=$db WANT_PACK=$want"
  P2P_WS_TH_PACK=$th P2P_WS_DBUF_PACK=$db P2P_WS_WANT_PACK=$want python3 tools/ubench/wgrad_layers.py 2>&1 | grep -v amdgpu.ids
done; done; done
echo "== power/clock while the c2 bench loops"
(for i in $(seq 1 12); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr '\n' ' '; echo; sleep 0.5; done) &
python3 bench.py --config c2 --steps 2500 --warmup 10 --no-cpu-baseline --no-profile 2>&1 | grep -v amdgpu.ids
wait
