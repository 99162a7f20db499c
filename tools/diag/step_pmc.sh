#!/bin/bash
# instruction / cycle counters of the step kernel's pack-only, solve-only and fused launches (tools/diag/step_pmc.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${PROF_DIR:-prof_step}
mkdir -p $OUT
pass() { local name=$1; shift
  timeout -k 10 200 rocprofv3 "$@" --output-format csv -d $OUT/$name -- python3 $R/tools/diag/step_pmc.py > $OUT/$name.log 2>&1
  local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name: KILLED"; exit 1; fi
  [ $rc -ne 0 ] && echo "$name failed rc $rc: $(tail -2 $OUT/$name.log | tr '\n' ' ')"; return 0; }
pass sq1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY
pass sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU
python3 - <<PY
import csv, glob, collections
for name in ("sq1", "sq2"):
    fs = glob.glob("$OUT/%s/*/*_counter_collection.csv" % name)
    if not fs: print(name, "no csv"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "cone_step" in r["Kernel_Name"]:
            acc[(r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k in sorted(acc): print(name, "grid", k[0], k[1], "mean %.0f over %d" % (sum(acc[k]) / len(acc[k]), len(acc[k])))
PY
