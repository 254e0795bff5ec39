#!/bin/bash
# memory-side and SQ counters of one probe run, all kernels: tools/pmc_fold.sh <probe args...>   (separate --pmc passes)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_fold
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 $R/tools/config_probe.py "$@" > $OUT/f.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 $R/tools/config_probe.py "$@" > $OUT/w.log 2>&1 || exit 2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/s -- python3 $R/tools/config_probe.py "$@" > $OUT/s.log 2>&1 || exit 3
cd $R && python3 tools/pmc_summary.py $OUT/f $OUT/w $OUT/s | grep -v "k_singular\|k_boundary\|k_wl_\|k_scatter"
