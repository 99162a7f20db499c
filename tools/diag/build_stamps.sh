#!/bin/bash
# Diagnostic build with per-phase cycle stamps (-DCAVE_STAMPS).  Never ship or time this build:
# read its phase SHARES only (tools/diag/run_stamps.py).
cd "$(dirname "$0")/../.." && hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -std=c++17 -fPIC -shared -DCAVE_STAMPS \
  cave_amd/csrc/cave_hip.hip -o tools/diag/libcave_hip_stamps.so
