#!/bin/bash
# ablation probe of the tile kernel (debug bits of PNL_ABLATE, see k_tile_distant)
cd $GRAFT_REPO_ROOT
for a in 0 2 18 10 1 64 4; do
  echo "== PNL_ABLATE=$a"
  PNL_VERBOSE=1 PNL_ABLATE=$a python3 tools/perf_probe.py 6 2>&1 | grep -E "rep 2|pnl\]" | tail -2
done
