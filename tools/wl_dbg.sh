#!/bin/bash
cd $GRAFT_REPO_ROOT
for a in 0 1 2 3 4 7; do
  echo "== PNL_WL_DBG=$a"
  PNL_WL_DBG=$a python3 tools/perf_probe.py 6 2>&1 | grep -E "rep 2" | tail -1
done
