#!/bin/bash
# usage (build container): bash tools/w4w_attrib.sh   -> csrc/liblgm_hip_w4w<n>.so for the attribution builds of the
# F(4x4) weight-gradient kernel (W4W_EXP bits: 1 no Yt transform, 2 no Xt transform, 4 no raw loads, 8 no operand reads);
# then on the GPU box:  for n in 0 1 2 3 4 7 8 15; do LGM_LIB=$PWD/lightning-generative-models_amd/csrc/liblgm_hip_w4w$n.so python tools/wino4_wgrad_bench.py 256 "64->64 @32"; done
set -e
cd "$(dirname "$0")/../lightning-generative-models_amd/csrc"
for n in 1 2 3 4 7 8 15; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value -DW4W_EXP=$n -c winograd4_wgrad.hip -o /tmp/w4w_$n.o
  objs=$(ls *.o | grep -v winograd4_wgrad.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o liblgm_hip_w4w$n.so $objs /tmp/w4w_$n.o
done
cp liblgm_hip.so liblgm_hip_w4w0.so
