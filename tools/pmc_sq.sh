#!/bin/bash
# SQ counters of the tile kernels (two --pmc passes, rocprofv3 with --kernel-trace only): where the wave cycles go
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $OUT/p1 -- python3 $R/tools/leg_probe.py ${1:-head} ${2:-} > $OUT/p1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/p2 -- python3 $R/tools/leg_probe.py ${1:-head} ${2:-} > $OUT/p2.log 2>&1 || exit 2
python3 - <<PY
import csv, glob, collections
for d in ('p1','p2'):
    for fn in glob.glob('$OUT/'+d+'/*/*counter_collection.csv'):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.defaultdict(set)
        for r in csv.DictReader(open(fn)):
            k=r['Kernel_Name'].split('(')[0].split('<')[0].split()[-1]
            if not k.startswith('k_tile'): continue
            agg[k][r['Counter_Name']]+=float(r['Counter_Value']); calls[k].add(r['Dispatch_Id'])
        for k in agg:
            print(k, {c: '%.4g'%(v/len(calls[k])) for c,v in sorted(agg[k].items())})
PY
