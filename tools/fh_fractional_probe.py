import sys, os
sys.path.insert(0, os.getcwd())
import torch
from pynucleus_amd import uniformSquare, P1_DoFMap, NO_BOUNDARY, getFractionalKernel
from pynucleus_amd.builder import nonlocalBuilder
mesh = uniformSquare(129); dm = P1_DoFMap(mesh, NO_BOUNDARY)
for s in (0.75, 0.4):
    b = nonlocalBuilder(dm, getFractionalKernel(2, s, horizon=0.1), {}, zeroExterior=False)
    for rep in range(2):
        A = b.getSparse()
    print('s', s, 'device %.1f ms' % A.info['interior_ms'], {k: round(v, 2) for k, v in b.context().phase_ms().items()})
