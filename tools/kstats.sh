#!/bin/bash
# per-kernel average durations of one perf_probe run (rocprofv3 --kernel-trace --stats): tools/kstats.sh <noRef> <tag>
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$2 -- python3 $R/tools/perf_probe.py $1 > /dev/null 2>&1
cd $R; python3 - $2 <<PY
import csv,glob,sys
for fn in glob.glob("gpurun_out/"+sys.argv[1]+"/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(fn)):
        if float(r["AverageNs"])>2e4: print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e6,3))
PY
