#!/bin/bash
# HBM traffic counters of the bench workload only (two --pmc passes): tools/pmc_traffic.sh <tag> [noRef]; summary printed
TAG=${1:-tmp}
NOREF=${2:-7}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --noRef $NOREF --no-cpu --no-extra > /dev/null 2>&1 || exit 3
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --noRef $NOREF --no-cpu --no-extra > /dev/null 2>&1 || exit 4
cd $R && python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write | grep "k_tile\|k_fold\|k_mirror"
