#!/usr/bin/env python3
"""One bench leg, three repetitions, one line 'LEG <name> {json}' (device ms, phases, kernel ms, SURVEY 8(d) fraction of the fp64 peak).
Used under rocprofv3 by tools/profile_legs.sh.   usage: leg_probe.py c3|c4|h2mv|c5|c5big|p2|s04|big|head [noRef] [serial]"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pynucleus_amd import (disc, uniformSquare, P1_DoFMap, P2_DoFMap, PHYSICAL, NO_BOUNDARY, getFractionalKernel, getKernel, INDICATOR)
from pynucleus_amd.builder import nonlocalBuilder
from pynucleus_amd.fractionalOrders import layersFractionalOrder
import bench

what = sys.argv[1]
size = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else None
if 'serial' in sys.argv[2:]:
    # one stream, one phase after the other: the per-kernel durations of a rocprofv3 trace then add up to the device time
    from pynucleus_amd import _lib
    _lib.set_option('PNL_NO_OVERLAP', 1)
    _lib.set_option('PNL_NO_FORK', 1)
if os.environ.get('PNL_PROBE_VERBOSE'):
    from pynucleus_amd import _lib
    _lib.set_option('PNL_VERBOSE', 1)                  # LDS bytes and resident workgroups per CU of the tile launches on stderr
PEAK = bench.FP64_VECTOR_PEAK_TFLOPS


def layers():
    orders = np.array([[0.3, 0.4, 0.5], [0.4, 0.5, 0.6], [0.5, 0.6, 0.7]])
    return layersFractionalOrder(2, np.array([-1., -0.3, 0.3, 1.]), orders)


def dense(name, dm, kernel, dpe, reps=3):
    b = nonlocalBuilder(dm, kernel, {'target_order': 0.5}, zeroExterior=True)
    for rep in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        A = b.getDense()
        torch.cuda.synchronize(); wall = time.perf_counter()-t0
        info = A.info
        del A
    fl = bench.flops_from_counters(info['counters'], dpe)
    ms = info['phase_ms']['total']
    print('LEG', name, json.dumps(dict(num_dofs=dm.num_dofs, device_ms=round(ms, 3), wall_ms=round(1e3*wall, 2), frac_fp64_peak=round(fl/ms/1e9/PEAK, 4),
                                       pairs=info['counters']['numAssembledCellPairs'],
                                       phases_ms={k: round(v, 3) for k, v in info['phase_ms'].items()},
                                       kernel_ms={k: round(v, 3) for k, v in b.dense_context().kernel_ms().items() if v})), flush=True)


if what == 'c5':
    dense('C5_P2_layers_noRef{}'.format(size or 6), P2_DoFMap(disc(size or 6), PHYSICAL), getFractionalKernel(2, layers()), 6)
elif what == 'c5big':
    dense('C5_P2_layers_97537dofs', P2_DoFMap(disc(6, sectors=12), PHYSICAL), getFractionalKernel(2, layers()), 6, reps=2)
elif what == 'p2':
    dense('P2_const_noRef{}'.format(size or 6), P2_DoFMap(disc(size or 6), PHYSICAL), getFractionalKernel(2, 0.5), 6)
elif what == 's04':
    dense('P1_s0.4_noRef{}'.format(size or 7), P1_DoFMap(disc(size or 7), PHYSICAL), getFractionalKernel(2, 0.4), 3)
elif what == 'head':
    dense('P1_s0.5_noRef{}'.format(size or 7), P1_DoFMap(disc(size or 7), PHYSICAL), getFractionalKernel(2, 0.5), 3)
elif what == 'big':
    dense('P1_s0.5_97537dofs', P1_DoFMap(disc(7, sectors=12), PHYSICAL), getFractionalKernel(2, 0.5), 3, reps=2)
elif what == 'c3':
    dm = P1_DoFMap(uniformSquare(size or 129), NO_BOUNDARY)
    b = nonlocalBuilder(dm, getKernel(2, kernel=INDICATOR, horizon=0.1), {}, zeroExterior=False)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        A = b.getSparse()
        torch.cuda.synchronize(); wall = time.perf_counter()-t0
    c = A.info['counters']
    ms = A.info['interior_ms']
    fl = bench.flops_from_counters(c, 3)
    print('LEG', 'C3_square{}_delta0.1'.format(size or 129), json.dumps(dict(num_dofs=dm.num_dofs, device_ms=round(ms, 3), wall_ms=round(1e3*wall, 2),
          frac_fp64_peak=round(fl/ms/1e9/PEAK, 4), pairs=c['numAssembledCellPairs'], evals=c['numIntegrations'],
          phases_ms={k: round(v, 3) for k, v in A.info.get('phase_ms', {}).items()},
          kernel_ms={k: round(v, 3) for k, v in b.dense_context().kernel_ms().items() if v})), flush=True)
elif what == 'h2mv':
    # the H2 matvec of C4 (near-field CSR product + upward pass, far-field interactions, downward pass): 200 products
    dm = P1_DoFMap(disc(size or 7), PHYSICAL)
    b = nonlocalBuilder(dm, getFractionalKernel(2, 0.75), {'target_order': 0.5, 'eta': 3.}, zeroExterior=True)
    h2 = b.getH2()
    x = torch.from_numpy(np.random.default_rng(0).standard_normal(dm.num_dofs)).cuda()
    for _ in range(5):
        y = h2.matvec(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        y = h2.matvec(x)
    torch.cuda.synchronize(); dt = (time.perf_counter()-t0)/200
    print('LEG', 'H2_matvec_noRef{}'.format(size or 7), json.dumps(dict(num_dofs=dm.num_dofs, matvec_ms=round(1e3*dt, 4), near_nnz=int(h2.Anear.nnz))), flush=True)
elif what == 'c4':
    dm = P1_DoFMap(disc(size or 7), PHYSICAL)
    b = nonlocalBuilder(dm, getFractionalKernel(2, 0.75), {'target_order': 0.5, 'eta': 3.}, zeroExterior=True)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h2 = b.getH2()
        torch.cuda.synchronize(); wall = time.perf_counter()-t0
    near = h2.Anear
    c = near.info['counters']
    ms = near.info['interior_ms']
    fl = bench.flops_from_counters(c, 3)
    print('LEG', 'C4_H2_noRef{}'.format(size or 7), json.dumps(dict(num_dofs=dm.num_dofs, getH2_ms=round(1e3*wall, 2), near_device_ms=round(ms, 3),
          frac_fp64_peak=round(fl/ms/1e9/PEAK, 4), pairs=c['numAssembledCellPairs'], host_s={k: round(v, 4) for k, v in getattr(h2, 'host_s', {}).items()},
          phases_ms={k: round(v, 3) for k, v in near.info.get('phase_ms', {}).items()},
          kernel_ms={k: round(v, 3) for k, v in b.dense_context().kernel_ms().items() if v})), flush=True)
