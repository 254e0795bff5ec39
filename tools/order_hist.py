import sys, numpy as np, torch
sys.path.insert(0, ''+__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))+'')
from pynucleus_amd import disc, P1_DoFMap, PHYSICAL, getFractionalKernel
from pynucleus_amd.builder import nonlocalBuilder
mesh = disc(7); dm = P1_DoFMap(mesh, PHYSICAL)
b = nonlocalBuilder(dm, getFractionalKernel(2, 0.5), {'target_order': 0.5}, zeroExterior=True)
A = b.getDense(); cnt = A.info['counters']
off = np.asarray(b.tables.dist_off)
tot = 0
for q, c in sorted(cnt['orders'].items()):
    n = int(off[q+1]-off[q]); c = int(c); tot += c*n*n
    print(q, n, c, '%.3e evals' % (c*n*n))
print('total', '%.3e' % tot, cnt['numIntegrations'])
