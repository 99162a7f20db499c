#!/bin/bash
# rocprofv3 passes for the large-cone kernels (configs 3 and 4, distinct cones); kernel trace and PMC in separate runs
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for W in tsp100 sp30; do
  OUT=$R/gpurun_out/prof_r02_$W
  mkdir -p $OUT
  CMD="python3 $R/tools/diag/large_profile.py $W"
  timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.txt 2> $OUT/trace.err || echo "trace failed"
  timeout -k 10 250 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.txt 2> $OUT/pmc_fetch.err || echo "pmc fetch failed"
  timeout -k 10 250 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.txt 2> $OUT/pmc_write.err || echo "pmc write failed"
  timeout -k 10 250 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.txt 2> $OUT/pmc_sq.err || echo "pmc sq failed"
  tail -1 $OUT/trace.txt
done
