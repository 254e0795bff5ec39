#!/usr/bin/env python3
"""Quick timing probe of the dense assembly (not the bench contract): phases per noRef."""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from pynucleus_amd import disc, P1_DoFMap, PHYSICAL, getFractionalKernel
from pynucleus_amd.builder import nonlocalBuilder

s = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
for noRef in [int(a) for a in sys.argv[1].split(',')]:
    t0 = time.time()
    mesh = disc(noRef)
    dm = P1_DoFMap(mesh, PHYSICAL)
    kernel = getFractionalKernel(2, s)
    b = nonlocalBuilder(dm, kernel, {'target_order': 0.5}, zeroExterior=True)
    t1 = time.time()
    ctx = b.context()
    N, nc = dm.num_dofs, mesh.num_cells
    A = torch.zeros((N, N), dtype=torch.float64, device='cuda')
    t2 = time.time()
    for rep in range(3):
        A.zero_()
        torch.cuda.synchronize()
        t3 = time.time()
        ctx.assemble_dense(A.data_ptr(), N, True, 0, nc)
        ctx.synchronize()
        t4 = time.time()
        ms = ctx.phase_ms()
        cnt = ctx.counters()
        pairs = cnt['numAssembledCellPairs']
        print('noRef {} N {} nc {} rep {}: wall {:.2f} ms, phases {} -> {:.3e} pairs/s (device total)'.format(
            noRef, N, nc, rep, 1e3*(t4-t3), {k: round(v, 3) for k, v in ms.items()}, pairs/(1e-3*ms['total'])), flush=True)
    print('   setup: tables {:.2f}s, upload {:.2f}s; evals {} ({:.1f}/pair), orders {}'.format(
        t1-t0, t2-t1, cnt['numIntegrations'], cnt['numIntegrations']/pairs, dict(list(cnt['orders'].items())[:8])), 'uniform-tile pairs', cnt.get('uniformTilePairs'), flush=True)
    del A
