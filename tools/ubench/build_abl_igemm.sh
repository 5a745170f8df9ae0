#!/bin/bash
# Diagnostic variants of the library with igemm.hip rebuilt under P2P_ABL=1 (staging only) and 2 (LDS reads + MFMAs only).
set -e
cd "$(dirname "$0")/../.."
CS=palette_and_histo_gan_amd/csrc
python -c "import __graft_entry__ as g; g.build()" > /dev/null
for v in 1 2; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-uninitialized -DP2P_ABL=$v -c $CS/igemm.hip -o tools/ubench/igemm_abl$v.o &
done
wait
for v in 1 2; do
  objs=$(ls $CS/*.o | grep -v igemm.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ubench/libp2p_abl$v.so $objs tools/ubench/igemm_abl$v.o -ldl
done
ls -la tools/ubench/*.so
