#!/bin/bash
# A/B builds of the library: recompile the named units with extra flags and link them with the other (already built)
# objects into cave_amd/libcave_hip_<NAME>.so; load it with CAVE_LIB=... (tools/diag/step_check.py, step_pack_share.py)
# or CAVE_SO=... (large_profile.py).     usage: tools/diag/build_variant.sh NAME "-DFOO=1 ..." k_step [k_packed_w1 ...]
cd "$(dirname "$0")/../.." || exit 1
name=$1; flags=$2; shift 2
B=cave_amd/csrc/build; V=$B/variant_$name; mkdir -p $V
objs=""
for o in $B/*.o; do u=$(basename $o .o); skip=0; for x in "$@"; do [ "$x" = "$u" ] && skip=1; done; [ $skip = 0 ] && objs="$objs $o"; done
for u in "$@"; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c cave_amd/csrc/$u.hip -o $V/$u.o & done; wait
for u in "$@"; do objs="$objs $V/$u.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $objs -o cave_amd/libcave_hip_$name.so && echo built cave_amd/libcave_hip_$name.so
