#!/bin/bash
# ablation probe of the uniform-tile kernel (debug bits of PNL_PURE_ABL: 16 no LDS cross adds, 32 no DPP column sums, 64 no flush)
cd $GRAFT_REPO_ROOT
for a in 0 16 32 64 112; do
  echo "== PNL_PURE_ABL=$a"
  PNL_PURE_ABL=$a python3 tools/perf_probe.py 6 2>&1 | grep -E "rep 2" | tail -1 | sed 's/.*phases//'
done
