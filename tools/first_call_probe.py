#!/usr/bin/env python3
"""first-call latency of a second builder in a process that already holds a dense operator (what bench.py's legs see)"""
import sys, os, time, cProfile, pstats
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynucleus_amd import disc, P1_DoFMap, P2_DoFMap, PHYSICAL, getFractionalKernel
from pynucleus_amd.builder import nonlocalBuilder
b1 = nonlocalBuilder(P1_DoFMap(disc(7), PHYSICAL), getFractionalKernel(2, 0.5), {'target_order': 0.5})
t0 = time.time(); A1 = b1.getDense(); torch.cuda.synchronize(); print('P1 first getDense', round(time.time()-t0, 3), flush=True)
if len(sys.argv) > 1 and sys.argv[1] == 'free':
    del A1, b1
    torch.cuda.empty_cache()
b2 = nonlocalBuilder(P2_DoFMap(disc(6), PHYSICAL), getFractionalKernel(2, 0.5), {'target_order': 0.5})
pr = cProfile.Profile(); pr.enable()
t0 = time.time(); A2 = b2.getDense(); torch.cuda.synchronize(); print('P2 first getDense', round(time.time()-t0, 3), flush=True)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(12)
