#!/bin/bash
# one PMC pass (instruction mix) over a short single-stream bench run
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcq
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/sq1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --streams 1 > $OUT/sq1.log 2>&1 || tail -5 $OUT/sq1.log
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt
python3 - <<PY
import json
d=json.load(open('$OUT/summary.json'))
for k,v in d.items():
    if 'SQ_INSTS_VALU' in v and v['dispatches']>=3 and 'rocclr' not in k and 'at::' not in k:
        print('%-40s disp %4d VALU %7.2fM SALU %6.2fM LDS %6.2fM waves %7d' % (k[:40], v['dispatches'], v['SQ_INSTS_VALU']/1e6, v['SQ_INSTS_SALU']/1e6, v['SQ_INSTS_LDS']/1e6, v['SQ_WAVES']))
PY
