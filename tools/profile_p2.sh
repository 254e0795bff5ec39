#!/bin/bash
# Kernel stats and SQ counters of the P2 configurations at noRef 6 (constant order and BASELINE C5: three layers), run through
# gpurun from the repo root:  tools/profile_p2.sh <round-tag>.  Separate rocprofv3 passes for --stats and for each --pmc set.
set -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}_p2
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in p2 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${cfg}_stats -- python3 $R/tools/config_probe.py $cfg 6 > $OUT/${cfg}_stats.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/${cfg}_sq -- python3 $R/tools/config_probe.py $cfg 6 > $OUT/${cfg}_sq.log 2>&1 || exit 2
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $OUT/${cfg}_grbm -- python3 $R/tools/config_probe.py $cfg 6 > $OUT/${cfg}_grbm.log 2>&1 || echo "grbm pass failed"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${cfg}_fetch -- python3 $R/tools/config_probe.py $cfg 6 > /dev/null 2>&1 || echo "fetch pass failed"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${cfg}_write -- python3 $R/tools/config_probe.py $cfg 6 > /dev/null 2>&1 || echo "write pass failed"
  grep "rep 2\|kernel ms" $OUT/${cfg}_stats.log
done
ls $OUT
