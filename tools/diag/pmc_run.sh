#!/bin/bash
# rocprofv3 passes for the bench command (kernel trace + PMC in separate runs, as the guide prescribes).
# A pass whose counter set the profiler rejects is reported and skipped; a pass that is KILLED (timeout) ends the script.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${PROF_DIR:-prof_r04}
mkdir -p $OUT
CMD="python3 $R/bench.py --steps 30 --warmup 5 --cpu-sample 0 --no-extras ${BENCH_ARGS:-}"
pass() {  # name, rocprofv3 options...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 "$@" --output-format csv -d $OUT/$name -- $CMD > $OUT/$name.json 2> $OUT/$name.err
  local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name: KILLED at its time limit -- stopping"; exit 1; fi
  [ $rc -ne 0 ] && echo "$name failed (rc $rc): $(tail -2 $OUT/$name.err | tr '\n' ' ')"
  return 0
}
pass trace --kernel-trace --stats
pass pmc_fetch --pmc FETCH_SIZE
pass pmc_write --pmc WRITE_SIZE
pass pmc_sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
# instruction side: instruction-cache requests / hits / misses, fetches, issue stalls, scalar work
pass pmc_ic --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH
pass pmc_inst --pmc SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_VALU
pass pmc_ic2 --pmc SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_INPUT_VALID_READYB SQ_IFETCH_LEVEL SQ_BUSY_CYCLES
ls $OUT
