#!/usr/bin/env python3
"""Timing probe of the BASELINE.json configurations other than the bench workload (not the bench contract):
   C3 square / constant kernel / finite horizon / getSparse, C4 disc s=0.75 near field (assembleClusters),
   C5 disc P2 + variable order dense, plus disc P2 constant order.     pw: disc P1, non-symmetric order s(x) per quadrature point (twoDomainNonSym).  c1: the user's view of the headline path, mesh -> builder -> getDense, first and repeated call.  usage: config_probe.py [c1|c3|c4|c5|p2|pw] [size]"""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from pynucleus_amd import disc, uniformSquare, P1_DoFMap, P2_DoFMap, PHYSICAL, NO_BOUNDARY, getFractionalKernel, getKernel, INDICATOR
from pynucleus_amd.builder import nonlocalBuilder
from pynucleus_amd.fractionalOrders import layersFractionalOrder
from pynucleus_amd import clusters

what = sys.argv[1]
size = int(sys.argv[2]) if len(sys.argv) > 2 else None


def sync():
    torch.cuda.synchronize()


if what == 'c1':
    noRef = size or 6
    t0 = time.time(); mesh = disc(noRef); t1 = time.time()
    dm = P1_DoFMap(mesh, PHYSICAL); t2 = time.time()
    b = nonlocalBuilder(dm, getFractionalKernel(2, 0.5), {'target_order': 0.5}, zeroExterior=True); t3 = time.time()
    print('c1 noRef {} N {}: mesh {:.2f} s, dofmap {:.2f} s, builder (tables) {:.2f} s'.format(noRef, dm.num_dofs, t1-t0, t2-t1, t3-t2), flush=True)
    for rep in range(3):
        sync(); t0 = time.time()
        A = b.getDense()
        sync(); t1 = time.time()
        print('   getDense call {}: wall {:.3f} s, device {:.1f} ms, host phases {}'.format(rep, t1-t0, A.info['phase_ms']['total'],
                                                                                     {k: round(v, 3) for k, v in A.info.get('host_s', {}).items()}), flush=True)
        del A
elif what in ('p2', 'c5'):
    noRef = size or 5
    sectors = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    mesh = disc(noRef, sectors=sectors)
    dm = P2_DoFMap(mesh, PHYSICAL)
    if what == 'c5':
        orders = np.array([[0.3, 0.4, 0.5], [0.4, 0.5, 0.6], [0.5, 0.6, 0.7]])
        s = layersFractionalOrder(2, np.array([-1., -0.3, 0.3, 1.]), orders)
    else:
        s = 0.5
    b = nonlocalBuilder(dm, getFractionalKernel(2, s), {'target_order': 0.5}, zeroExterior=True)
    for rep in range(3):
        sync(); t0 = time.time()
        A = b.getDense()
        sync(); t1 = time.time()
        cnt = A.info['counters']; ms = A.info['phase_ms']
        print('{} noRef {} N {} nc {} rep {}: wall {:.1f} ms device {:.1f} ms, phases {} -> {:.3e} pairs/s'.format(
            what, noRef, dm.num_dofs, mesh.num_cells, rep, 1e3*(t1-t0), ms['total'], {k: round(v, 2) for k, v in ms.items()},
            cnt['numAssembledCellPairs']/(1e-3*ms['total'])), flush=True)
        del A
    print('   kernel ms', {k: round(v, 3) for k, v in b.context().kernel_ms().items()}, flush=True)
elif what == 'pw':
    from pynucleus_amd.fractionalOrders import smoothedLeftRightFractionalOrder
    noRef = size or 5
    mesh = disc(noRef)
    dm = P1_DoFMap(mesh, PHYSICAL)
    t0 = time.time()
    b = nonlocalBuilder(dm, getFractionalKernel(2, smoothedLeftRightFractionalOrder(0.25, 0.75)), {'target_order': 0.5}, zeroExterior=True)
    R = b.tables.pw_rules()
    print('pw noRef {} N {}: host tables {:.2f} s, {} near-rule keys, {} touching pairs'.format(noRef, dm.num_dofs, time.time()-t0,
                                                                                            len(R['keys']), R['pairs'].shape[0]), flush=True)
    for rep in range(3):
        sync(); t0 = time.time()
        A = b.getDense()
        sync(); t1 = time.time()
        cnt = A.info['counters']; ms = A.info['phase_ms']
        print('pw noRef {} rep {}: wall {:.1f} ms device {:.1f} ms, phases {} -> {:.3e} pairs/s, {:.3e} kernel evaluations/s'.format(
            noRef, rep, 1e3*(t1-t0), ms['total'], {k: round(v, 2) for k, v in ms.items()}, cnt['numAssembledCellPairs']/(1e-3*ms['total']),
            cnt['numIntegrations']/(1e-3*ms['total'])), flush=True)
        del A
elif what == 'c4':
    noRef = size or 6
    mesh = disc(noRef)
    dm = P1_DoFMap(mesh, PHYSICAL)
    b = nonlocalBuilder(dm, getFractionalKernel(2, float(__import__('os').environ.get('PNL_S', '0.75'))), {'target_order': 0.5, 'eta': 3.}, zeroExterior=True)
    t0 = time.time()
    rp = b.getH2RefinementParams()
    root, Pnear, Pfar = clusters.getNearFieldClusters(dm, rp['eta'], rp['minSize'], rp['maxLevels'])
    t1 = time.time()
    print('c4 noRef {} N {}: tree+admissibility {:.2f} s, {} near pairs, {} far pairs'.format(noRef, dm.num_dofs, t1-t0, len(Pnear),
                                                                                           sum(len(v) for v in Pfar.values())), flush=True)
    for rep in range(2):
        sync(); t0 = time.time()
        A = b.assembleClusters(Pnear)
        sync(); t1 = time.time()
        c = A.info['counters']
        print('   rep {}: assembleClusters wall {:.2f} s (host masks/pattern included), device interior {:.1f} ms, nnz {} ({:.1f}% of N^2), '
              'element pairs {} -> {:.3e} pairs/s on the device'.format(rep, t1-t0, A.info['interior_ms'], A.nnz, 100.*(2*A.nnz+dm.num_dofs)/dm.num_dofs**2,
                                                                        c['numAssembledCellPairs'], c['numAssembledCellPairs']/(1e-3*A.info['interior_ms'])), flush=True)
    x = torch.randn(dm.num_dofs, dtype=torch.float64, device='cuda')
    sync(); t0 = time.time()
    for _ in range(20):
        y = A.matvec(x)
    sync()
    print('   near-field SpMV: {:.3f} ms'.format(1e3*(time.time()-t0)/20))
    sync(); t0 = time.time()
    h2 = b.getH2()
    sync(); t1 = time.time()
    print('   getH2 (tree + near field + far-field setup): wall {:.2f} s; {}'.format(t1-t0, h2))
    for _ in range(3):
        y = h2.matvec(x)
    sync(); t0 = time.time()
    for _ in range(20):
        y = h2.matvec(x)
    sync()
    t_h2 = (time.time()-t0)/20
    if dm.num_dofs <= 50000:
        D = b.getDense()
        yd = D.matvec(x)
        sync(); t0 = time.time()
        for _ in range(20):
            yd = D.matvec(x)
        sync()
        t_d = (time.time()-t0)/20
        print('   H2 matvec {:.3f} ms vs dense GEMV {:.3f} ms; |(A_dense - A_h2) x| / |A_dense x| = {:.2e}'.format(
            1e3*t_h2, 1e3*t_d, float(torch.linalg.norm(y-yd)/torch.linalg.norm(yd))))
    else:
        print('   H2 matvec {:.3f} ms'.format(1e3*t_h2))
    from pynucleus_amd.solvers import cg
    rhs = torch.from_numpy(np.asarray(dm.assembleRHS(1.0))).to(x.device)
    sync(); t0 = time.time()
    u, its, res = cg(h2, rhs, tol=1e-8, maxiter=2000)
    sync()
    print('   CG-Jacobi on the H2 operator: {} iterations, {:.1f} ms, residual {:.2e}'.format(its, 1e3*(time.time()-t0), res[-1]))
elif what == 'c3':
    N = size or 129
    mesh = uniformSquare(N)
    dm = P1_DoFMap(mesh, NO_BOUNDARY)
    delta = 0.1
    b = nonlocalBuilder(dm, getKernel(2, kernel=INDICATOR, horizon=delta), {}, zeroExterior=False)
    for rep in range(2):
        sync(); t0 = time.time()
        A = b.getSparse()
        sync(); t1 = time.time()
        c = A.info['counters']
        print('c3 N {} (h={:.4f}, delta/h={:.1f}) rep {}: getSparse wall {:.2f} s, device {:.1f} ms, candidates {}, assembled {}, evals {}, nnz {} -> {:.3e} pairs/s on the device'.format(
            dm.num_dofs, mesh.h, delta/mesh.h, rep, t1-t0, A.info['interior_ms'], A.info['num_candidate_pairs'], c['numAssembledCellPairs'],
            c['numIntegrations'], A.nnz, c['numAssembledCellPairs']/(1e-3*A.info['interior_ms'])), flush=True)
