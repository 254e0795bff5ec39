#!/bin/bash
cd $GRAFT_REPO_ROOT
for a in 0 2 4 8 16 32; do
  echo "== PNL_ACC_PAD=$a"
  PNL_VERBOSE=1 PNL_ACC_PAD=$a python3 tools/perf_probe.py 6 2>&1 | grep -E "rep 2|lds=" | tail -3
done
