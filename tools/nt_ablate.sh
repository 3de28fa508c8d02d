#!/bin/bash
# Builds abtest/libnt<N>.so = the library with gemm.hip compiled with -DNT_ABLATE=N (see gemm.hip) for tools/nt_ablate.py.
set -e
cd "$(dirname "$0")/../promptir_amd/csrc"
make -s
mkdir -p ../../abtest
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DNT_ABLATE=$n -c gemm.hip -o ../../abtest/gemm_nt$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../abtest/libnt$n.so ../../abtest/gemm_nt$n.o gemm_x3.o stencil.o stencil_wave.o gdfn_bwd.o norm.o mdta.o prompt.o tile.o misc.o bias.o
done
