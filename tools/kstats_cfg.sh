#!/bin/bash
# per-kernel totals of one config_probe run: tools/kstats_cfg.sh <config> <size> <tag>
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$3 -- python3 $R/tools/config_probe.py $1 $2 > $R/gpurun_out/$3.log 2>&1
cd $R; tail -4 gpurun_out/$3.log; python3 - $3 <<PY
import csv,glob,sys
for fn in glob.glob("gpurun_out/"+sys.argv[1]+"/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(fn)):
        if float(r["TotalDurationNs"])>2e5: print(r["Name"][:64], r["Calls"], 'total ms', round(float(r["TotalDurationNs"])/1e6,2), 'avg', round(float(r["AverageNs"])/1e6,3))
PY
