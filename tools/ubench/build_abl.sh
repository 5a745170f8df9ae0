#!/bin/bash
# Diagnostic variants of the library: igemm.hip / brig.hip rebuilt with P2P_ABL=1 (staging only), 2 (MFMA only), and for brig.hip
# 3 (never wait for the DMA), 4 / 5 (staging of the input blocks / the weights only); everything else reused.
set -e
cd "$(dirname "$0")/../.."
CS=palette_and_histo_gan_amd/csrc
python -c "import __graft_entry__ as g; g.build()" > /dev/null
for v in 1 2 3 4 5; do      # igemm.hip knows 1 and 2 only (3-5: the product kernel)
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DP2P_ABL=$v -c $CS/igemm.hip -o tools/ubench/igemm_abl$v.o
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DP2P_ABL=$v -c $CS/brig.hip -o tools/ubench/brig_abl$v.o
  objs=$(ls $CS/*.o | grep -v igemm.o | grep -v brig.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ubench/libp2p_abl$v.so $objs tools/ubench/igemm_abl$v.o tools/ubench/brig_abl$v.o -ldl
done
ls -la tools/ubench/*.so
