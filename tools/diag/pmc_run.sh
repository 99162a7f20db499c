#!/bin/bash
# rocprofv3 passes for the bench command (kernel trace + PMC in separate runs, as the guide prescribes)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${PROF_DIR:-prof_r02}
mkdir -p $OUT
CMD="python3 $R/bench.py --steps 30 --warmup 5 --cpu-sample 0 --no-extras ${BENCH_ARGS:-}"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.json 2> $OUT/trace.err || echo "trace failed"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || echo "pmc fetch failed"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.json 2> $OUT/pmc_write.err || echo "pmc write failed"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err || echo "pmc sq failed"
ls -R $OUT | head -40
