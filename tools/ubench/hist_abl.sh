#!/bin/bash
# ablations of the histogram kernels (hist.hip, P2P_HIST_ABL): which resource holds the batch time
for a in 0 1 2 3 4; do
  if [ $a = 0 ]; then L=""; else L=$PWD/tools/ubench/libp2p_hist_abl$a.so; fi
  echo "== P2P_HIST_ABL=$a"
  P2P_LIB=$L python tools/ubench/hist_layers.py 256 64 2>&1 | grep "N="
done
