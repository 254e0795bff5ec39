#!/usr/bin/env python3
"""C3 (square, constant kernel, delta = 0.1): device time with re-triangulated cut elements against the barycentre rule
(no sub-simplex loops): how much of the finite-horizon assembly is cut-element geometry"""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from pynucleus_amd import uniformSquare, P1_DoFMap, NO_BOUNDARY, getKernel, INDICATOR
from pynucleus_amd.builder import nonlocalBuilder
N = int(sys.argv[1]) if len(sys.argv) > 1 else 129
mesh = uniformSquare(N)
dm = P1_DoFMap(mesh, NO_BOUNDARY)
for inter in (None, 'ball2_barycenter'):
    b = nonlocalBuilder(dm, getKernel(2, kernel=INDICATOR, horizon=0.1, interaction=inter), {}, zeroExterior=False)
    for rep in range(2):
        A = b.getSparse()
    print(inter, 'device %.1f ms' % A.info['interior_ms'], A.info['counters'], b.context().phase_ms())
    cnt = b.context().counters()
    off = b.tables.dist_off
    print('   orders (order: points, pairs):', {int(q): (int(off[q+1]-off[q]), int(c)) for q, c in sorted(cnt['orders'].items())}, 'touching', cnt['singular'])
