#!/bin/bash
# Builds abtest/libabl<N>.so = the library with gemm_x3.hip compiled with -DX3_ABLATE=N (see gemm_x3.hip) for the
# ablation timing of tools/x3_ablate.py.  Usage: tools/x3_ablate.sh 0 1 2 4 8 ...
set -e
cd "$(dirname "$0")/../promptir_amd/csrc"
make -s
mkdir -p ../../abtest
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DX3_ABLATE=$n -c gemm_x3.hip -o ../../abtest/gemm_x3_abl$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../abtest/libabl$n.so gemm.o stencil.o stencil_wave.o gdfn_bwd.o norm.o mdta.o prompt.o tile.o misc.o bias.o ../../abtest/gemm_x3_abl$n.o
done
