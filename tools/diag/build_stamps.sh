#!/bin/bash
# Diagnostic build with per-phase cycle stamps (-DCAVE_STAMPS).  Never ship or time this build:
# read its phase SHARES only (tools/diag/run_stamps.py).  A unity build (one translation unit: the stamp
# buffer is a single __device__ array) of every kernel shape: ~3.5 minutes (the host code references every launch
# function, so a subset of the shapes does not load).  EXTRA_FLAGS=-DCAVE_STAMPS_FINE adds the per-block-step stamps of
# the band / dense factorisations; OUT names the library.
cd "$(dirname "$0")/../.." || exit 1
ALL="k_dense_w1 k_dense_w2 k_dense_w4 k_dense_w8 k_pack_w1 k_pack_w2 k_pack_w4 k_pack_w8 k_packed_w1 k_packed_w2 k_packed_w4 k_packed_w8 k_large_dense k_large_pack k_large_packed_w1 k_large_packed_w2 k_large_packed_w4 k_step"
U=tools/diag/_unity_stamps.hip
: > "$U"
for u in cave_hip $ALL; do echo "#include \"../../cave_amd/csrc/$u.hip\"" >> "$U"; done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DCAVE_STAMPS ${EXTRA_FLAGS} "$U" \
  -o "${OUT:-tools/diag/libcave_hip_stamps.so}"
