# Same-box A/B against an earlier round's tree (run through gpurun from the repository root).  Prepare here first:
#   git worktree add -f tools/exp/_r02_tree <commit> && (cd tools/exp/_r02_tree && python -c "import __graft_entry__ as g; g.build()")
# (the worktree is untracked and travels with the snapshot; remove it afterwards: git worktree remove --force tools/exp/_r02_tree)
cd $GRAFT_REPO_ROOT
run() { (cd $1 && timeout -k 10 300 python bench.py --config $2 --steps $3 --warmup 20 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-8s %-4s %9.1f img/s %.4f ms' % ('$4', '$2', d['value'], d['ms_per_step']))") || exit 1; }
for i in 1 2 3; do run tools/exp/_r02_tree c2 200 r02; run . c2 200 r03; done
for c in c3 c4 c5 c1; do for i in 1 2; do run tools/exp/_r02_tree $c 100 r02; run . $c 100 r03; done; done
