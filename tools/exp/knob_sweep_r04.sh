#!/bin/bash
# c2 step time against the grid heuristics that the round-4 kernels may have moved (same box, one run each, baseline first and last)
run() { env "$@" python bench.py --config c2 --steps 60 --warmup 10 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s %9.1f img/s  %.4f ms' % (sys.argv[1], d['value'], d['ms_per_step']))" "$*"; }
run X=0
run P2P_SPLITK_TARGET=128
run P2P_SPLITK_TARGET=384
run P2P_SPLITK_TARGET=512
run P2P_WGEMM_WANT_PIPE=128
run P2P_WGEMM_WANT_PIPE=448
run P2P_WGEMM_PIPE_MAXWG=1024
run P2P_BRIG_MIN_WG=1
run P2P_BRIG_MIN_WG=257
run P2P_IGEMM_BIG=512
run P2P_IGEMM_BIG=128
run P2P_WS_WANT=512
run X=1
