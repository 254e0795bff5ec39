import sys, os, time, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from pynucleus_amd import disc, P2_DoFMap, P1_DoFMap, PHYSICAL, getFractionalKernel
from pynucleus_amd.builder import nonlocalBuilder
from pynucleus_amd.fractionalOrders import layersFractionalOrder
import bench
def leg(name, dm, kernel, dpe):
    b = nonlocalBuilder(dm, kernel, {'target_order': 0.5}, zeroExterior=True)
    for rep in range(3):
        torch.cuda.synchronize(); A = b.getDense(); torch.cuda.synchronize()
        info = A.info; del A
    fl = bench.flops_from_counters(info['counters'], dpe)
    ms = info['phase_ms']['total']
    print(name, round(ms, 2), 'frac', round(fl/ms/1e9/78.6, 3), {k: round(v, 2) for k, v in info['phase_ms'].items()}, {k: round(v, 2) for k, v in b.context().kernel_ms().items() if v})
mesh = disc(6)
orders = np.array([[0.3, 0.4, 0.5], [0.4, 0.5, 0.6], [0.5, 0.6, 0.7]])
leg('C5', P2_DoFMap(mesh, PHYSICAL), getFractionalKernel(2, layersFractionalOrder(2, np.array([-1., -0.3, 0.3, 1.]), orders)), 6)
leg('P2 s=0.4', P2_DoFMap(mesh, PHYSICAL), getFractionalKernel(2, 0.4), 6)
leg('P1 s=0.4 noRef7', P1_DoFMap(disc(7), PHYSICAL), getFractionalKernel(2, 0.4), 3)
