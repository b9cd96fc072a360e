#!/bin/bash
# PMC pass over the layernorm micro-benchmark (counters only, kernel-trace for names)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_ln
mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/sq1 -- python3 tools/bench_ln.py > $OUT/sq1.log 2>&1 || tail -5 $OUT/sq1.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -- python3 tools/bench_ln.py > $OUT/sq2.log 2>&1 || tail -5 $OUT/sq2.log
python3 - <<PY
import csv, glob, collections
for d in ('sq1','sq2'):
    for f in glob.glob('$OUT/%s/**/*counter_collection.csv' % d, recursive=True):
        acc = collections.defaultdict(float); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if 'layernorm' in r['Kernel_Name']:
                acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
        for k in acc: print(d, k, acc[k] / n[k], n[k])
PY
