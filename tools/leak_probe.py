import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from pynucleus_amd import disc, P1_DoFMap, P2_DoFMap, PHYSICAL, getFractionalKernel
from pynucleus_amd.builder import nonlocalBuilder
from pynucleus_amd.multigrid import fractionalHierarchy, multigrid
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]/2**20
b = nonlocalBuilder(P1_DoFMap(disc(6), PHYSICAL), getFractionalKernel(2, 0.4), {'target_order': 0.5})
A = b.getDense(); del A
f0 = free()
for i in range(40):
    A = b.getDense(); del A
f1 = free()
print('40 assemblies: free MiB %.0f -> %.0f' % (f0, f1))
H = fractionalHierarchy('disc', 5, getFractionalKernel(2, 0.5), {'target_order': 0.5})
rhs = np.asarray(H.finest['DoFMap'].assembleRHS(1.0))
mg = multigrid(H); mg.cg(rhs)
f2 = free()
for i in range(30):
    mg2 = multigrid(H); mg2.cg(rhs); del mg2
f3 = free()
print('30 multigrid objects: free MiB %.0f -> %.0f' % (f2, f3))
for i in range(10):
    Hh = fractionalHierarchy('disc', 4, getFractionalKernel(2, 0.3+0.01*i), {})
    del Hh
f4 = free()
print('10 hierarchies with new exponents: free MiB %.0f -> %.0f' % (f3, f4))
