#!/bin/bash
# ablation probe of the general tile kernel at noRef 7 (debug bits of PNL_ABLATE: 1 no LDS accumulate, 2 no evaluation,
# 4 no flush, 8 nothing classified, 16 every pair order 2, 64 no LDS cross adds in list A)
cd $GRAFT_REPO_ROOT
for a in 0 1 2 4 16 6 8; do
  echo "== PNL_ABLATE=$a"
  PNL_ABLATE=$a python3 tools/perf_probe.py 7 2>&1 | grep -E "rep 2" | tail -1 | sed 's/.*phases//'
done
