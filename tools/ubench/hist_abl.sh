#!/bin/bash
# Ablations of the histogram kernels (hist.hip, -DP2P_HIST_ABL=n): which resource holds the batch time.
#   bash tools/ubench/hist_abl.sh build      # here (no GPU needed): tools/ubench/libp2p_hist_abl{1..4}.so from the built objects
#   bash tools/ubench/hist_abl.sh            # on the GPU box: times the product library and the four ablations on the c3 shape
# 1 = no matrix products (fragment reads kept), 2 = no kernel-row evaluation, 3 = a quarter of the backward pairing sums,
# 4 = no single-wave work (per-pixel logs, per-pixel result).  Results are wrong by construction; only the times mean something.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd $ROOT
CS=palette_and_histo_gan_amd/csrc
if [ "$1" = build ]; then
  O=$(ls $CS/*.o | grep -v "/hist.o")
  for a in 1 2 3 4; do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DP2P_HIST_ABL=$a -Iinclude -c $CS/hist.hip -o /tmp/hist_abl$a.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ubench/libp2p_hist_abl$a.so $O /tmp/hist_abl$a.o -ldl
  done
  exit 0
fi
for a in 0 1 2 3 4; do
  if [ $a = 0 ]; then L=""; else L=$ROOT/tools/ubench/libp2p_hist_abl$a.so; fi
  echo "== P2P_HIST_ABL=$a"
  P2P_LIB=$L python tools/ubench/hist_layers.py 256 64 2>&1 | grep "N="
done
