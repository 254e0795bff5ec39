#!/usr/bin/env python3
"""cProfile of getSparse for BASELINE configs[2] (square, constant kernel, finite horizon): where the host time goes"""
import sys, time, cProfile, pstats
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from pynucleus_amd import uniformSquare, P1_DoFMap, NO_BOUNDARY, getKernel, INDICATOR
from pynucleus_amd.builder import nonlocalBuilder
N = int(sys.argv[1]) if len(sys.argv) > 1 else 129
mesh = uniformSquare(N)
dm = P1_DoFMap(mesh, NO_BOUNDARY)
b = nonlocalBuilder(dm, getKernel(2, kernel=INDICATOR, horizon=0.1), {}, zeroExterior=False)
A = b.getSparse(); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.time(); A = b.getSparse(); torch.cuda.synchronize(); t1 = time.time()
pr.disable()
print('getSparse wall %.3f s, device %.1f ms' % (t1-t0, A.info.get('interior_ms', float('nan'))))
pstats.Stats(pr).sort_stats('cumtime').print_stats(18)
