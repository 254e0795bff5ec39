#!/bin/bash
# ablation of the uniform P2 tile kernel with a PNL_DEBUG_ABLATE build (pynucleus_amd/libpnl_dbg.so): PNL_UNI_ABL bits 2: no flush, 4: no LDS accumulate
for abl in 0 2 4 6; do
  echo "== PNL_UNI_ABL=$abl"
  PNL_LIB=$PWD/pynucleus_amd/libpnl_dbg.so PNL_UNI_ABL=$abl timeout -k 10 120 python tools/config_probe.py p2 6 2>&1 | grep "rep 2"
done
