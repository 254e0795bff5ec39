#!/usr/bin/env python3
"""C3 (square 129^2, constant kernel, horizon 0.1, getSparse): device time and its phases.  tools/c3_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pynucleus_amd import uniformSquare, P1_DoFMap, NO_BOUNDARY, getKernel, INDICATOR
from pynucleus_amd.builder import nonlocalBuilder
dm = P1_DoFMap(uniformSquare(129), NO_BOUNDARY)
b = nonlocalBuilder(dm, getKernel(2, kernel=INDICATOR, horizon=0.1), {}, zeroExterior=False)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); A = b.getSparse(); torch.cuda.synchronize(); t1 = time.perf_counter()
print('getSparse %.1f ms, device %.2f ms' % (1e3*(t1-t0), A.info['interior_ms']), {k: round(v, 2) for k, v in A.info.get('phase_ms', {}).items()},
      {k: round(v, 2) for k, v in b.context().kernel_ms().items() if v})
