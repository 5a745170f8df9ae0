set -e
cd $GRAFT_REPO_ROOT
B="python3 bench.py --config c2 --steps 60 --warmup 10 --no-cpu-baseline --no-profile"
for rep in 1 2; do
echo "== base"; $B | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
echo "== A last,up6,up5,up4"; P2P_EXP_DEFER=G.last,G.up6,G.up5,G.up4 $B | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
echo "== B A+D"; P2P_EXP_DEFER=G.last,G.up6,G.up5,G.up4,D.last,D.down $B | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
echo "== C B+up3,up2,up1"; P2P_EXP_DEFER=G.last,G.up6,G.up5,G.up4,D.last,D.down,G.up3,G.up2,G.up1 $B | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
echo "== D only D"; P2P_EXP_DEFER=D.last,D.down $B | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
echo "== E all"; P2P_EXP_DEFER=G.last,G.up6,G.up5,G.up4,D.last,D.down,G.up3,G.up2,G.up1,G.down6,G.down5,G.down4,G.down3,G.down2,G.down1 $B | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
