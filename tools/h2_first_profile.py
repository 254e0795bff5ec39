import sys, os, time, cProfile, pstats, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from pynucleus_amd import disc, P1_DoFMap, PHYSICAL, getFractionalKernel
from pynucleus_amd.builder import nonlocalBuilder
torch.cuda.init(); torch.zeros(1, device='cuda')
dm = P1_DoFMap(disc(7), PHYSICAL)
t0=time.time()
b = nonlocalBuilder(dm, getFractionalKernel(2, 0.75), {'target_order': 0.5, 'eta': 3.}, zeroExterior=True)
print('builder', round(time.time()-t0,3))
pr = cProfile.Profile(); pr.enable()
t0=time.time(); h2 = b.getH2(); torch.cuda.synchronize(); print('first getH2', round(time.time()-t0,3))
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
