#!/usr/bin/env python3
"""cProfile of a repeated getH2 call (host-side planning + device assembly): tools/h2_profile.py [noRef] [s]"""
import sys
import os
import time
import cProfile
import pstats
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynucleus_amd import disc, P1_DoFMap, PHYSICAL, getFractionalKernel
from pynucleus_amd.builder import nonlocalBuilder

noRef = int(sys.argv[1]) if len(sys.argv) > 1 else 7
s = float(sys.argv[2]) if len(sys.argv) > 2 else 0.75
dm = P1_DoFMap(disc(noRef), PHYSICAL)
prm = {'target_order': 0.5, 'eta': 3.}
if os.environ.get('PNL_MINCLUSTER'):
    prm['minClusterSize'] = int(os.environ['PNL_MINCLUSTER'])
b = nonlocalBuilder(dm, getFractionalKernel(2, s), prm, zeroExterior=True)
print('refinement params', b.getH2RefinementParams())
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    h2 = b.getH2()
    torch.cuda.synchronize()
    print('getH2 call {}: {:.3f} s'.format(rep, time.time()-t0), flush=True)
    del h2
pr = cProfile.Profile()
pr.enable()
h2 = b.getH2()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(8)
import numpy as np
x = torch.randn(dm.num_dofs, dtype=torch.float64, device='cuda')
D = b.getDense()
yd = D.matvec(x); yh = h2.matvec(x)
print('h2', h2, 'rel err vs dense', float((yh-yd).norm()/yd.norm()))
torch.cuda.synchronize(); t0 = time.time()
for _ in range(20):
    yh = h2.matvec(x)
torch.cuda.synchronize(); print('matvec ms', 1e3*(time.time()-t0)/20)
