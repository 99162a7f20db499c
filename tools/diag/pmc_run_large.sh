#!/bin/bash
# rocprofv3 passes for the large-cone kernels (configs 3 and 4, distinct cones); kernel trace and PMC in separate runs.
#   WHICH="tsp50 tsp100 sp30" PROF_TAG=r04 [CAVE_SO=variant.so] bash tools/diag/pmc_run_large.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for W in ${WHICH:-tsp50 tsp100 sp30}; do
  OUT=$R/gpurun_out/prof_${PROF_TAG:-r04}_$W
  mkdir -p $OUT
  CMD="python3 $R/tools/diag/large_profile.py $W"
  pass() {
    local name=$1; shift
    timeout -k 10 250 rocprofv3 "$@" --output-format csv -d $OUT/$name -- $CMD > $OUT/$name.txt 2> $OUT/$name.err
    local rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$W $name: KILLED at its time limit -- stopping"; exit 1; fi
    [ $rc -ne 0 ] && echo "$W $name failed (rc $rc)"
    return 0
  }
  pass trace --kernel-trace --stats
  pass pmc_fetch --pmc FETCH_SIZE
  pass pmc_write --pmc WRITE_SIZE
  [ -z "$QUICK" ] && pass pmc_sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
  tail -2 $OUT/trace.txt
done
