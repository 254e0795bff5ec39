import sys, os, time, cProfile, pstats
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from pynucleus_amd import disc, P1_DoFMap, PHYSICAL, getFractionalKernel
from pynucleus_amd.builder import nonlocalBuilder
dm = P1_DoFMap(disc(7), PHYSICAL)
b = nonlocalBuilder(dm, getFractionalKernel(2, 0.75), {'target_order': 0.5, 'eta': 3.}, zeroExterior=True)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); h2 = b.getH2(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print('getH2 %.1f ms, near field device %.2f ms' % (1e3*(t1-t0), h2.Anear.info['interior_ms']))
pr = cProfile.Profile(); pr.enable()
h2 = b.getH2(); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
