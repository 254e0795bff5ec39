#!/bin/bash
# Collects the judged artefacts of a round on the GPU box (run through gpurun from the repo root):
#   bench line, rocprofv3 kernel stats of the same command, HBM traffic counters in separate --pmc passes.
# Usage: tools/profile_round.sh <round-tag> [noRef] [sectors]   (sectors 12 at noRef 7 = the north star's 97,537 DoFs: key noRef7_s12)
set -o pipefail
TAG=${1:-r01}
NOREF=${2:-7}
SECTORS=${3:-6}
SFX=$([ "$SECTORS" = 6 ] && echo "" || echo "_s$SECTORS")
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG$SFX
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 10 --warmup 2 --noRef $NOREF --sectors $SECTORS $([ "$SECTORS" = 6 ] || echo --no-extra) > $OUT/bench_noRef$NOREF$SFX.json 2> $OUT/bench_noRef$NOREF$SFX.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 20 --warmup 1 --noRef $NOREF --sectors $SECTORS --no-cpu --no-extra > $OUT/bench_prof.json 2> $OUT/bench_prof.err || exit 2
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --noRef $NOREF --sectors $SECTORS --no-cpu --no-extra > /dev/null 2>&1 || exit 3
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --noRef $NOREF --sectors $SECTORS --no-cpu --no-extra > /dev/null 2>&1 || exit 4
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --noRef $NOREF --sectors $SECTORS --no-cpu --no-extra > /dev/null 2>&1 || echo "sq pass failed"
rocprofv3 --kernel-trace --pmc TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- python3 $R/bench.py --steps 2 --warmup 1 --noRef $NOREF --sectors $SECTORS --no-cpu --no-extra > /dev/null 2>&1 || echo "tcc pass failed"
ls -R $OUT | head -40
cat $OUT/bench_noRef$NOREF$SFX.json
