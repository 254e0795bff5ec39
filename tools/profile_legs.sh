#!/bin/bash
# Kernel stats (rocprofv3 --kernel-trace --stats) and SQ counters (separate --pmc pass) of the bench legs other than the headline:
# C3 (getSparse, square 129^2), C4 (getH2 / near field, disc noRef 7), C5 (P2 + layers, noRef 6), P2 constant order, P1 with a
# general exponent, and the 97,537-DoF dense leg.  Run through gpurun from the repo root:  tools/profile_legs.sh <round-tag> [legs...]
set -o pipefail
TAG=${1:-r03}
shift
LEGS=${@:-c3 c4 c4serial c5 c5big p2 s04 big}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}_legs
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES"
for leg in $LEGS; do
  echo "== $leg" 
  # <leg>serial: the same leg with every phase on one stream (options PNL_NO_OVERLAP / PNL_NO_FORK): per-kernel durations that add up
  ARGS=$leg
  case $leg in *serial) ARGS="${leg%serial} serial";; esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${leg}_stats -- python3 $R/tools/leg_probe.py $ARGS > $OUT/${leg}_stats.log 2>&1 || { echo "stats pass of $leg failed"; tail -5 $OUT/${leg}_stats.log; exit 1; }
  rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/${leg}_sq -- python3 $R/tools/leg_probe.py $ARGS > $OUT/${leg}_sq.log 2>&1 || echo "sq pass of $leg failed"
  grep "^LEG" $OUT/${leg}_stats.log
done
