"""GPU parity: libpnl_hip.so (through the C ABI) against the CPU oracle, entry-wise.

Tolerance: the GPU sums the same local contributions in a different (atomic) order and evaluates
pow/rsqrt with its own libm, so entries agree to rounding, not bit-wise:
|A_gpu - A_oracle| <= 1e-11 * max|A_oracle| (fp64, stated in BASELINE.json as "within a stated fp64 tolerance").
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-11


def _build(domain, noRef, s, element='P1', zeroExterior=True, params=None):
    from pynucleus_amd import disc, interval, PHYSICAL, dofmapFactory, getFractionalKernel
    from pynucleus_amd.builder import nonlocalBuilder
    mesh = disc(noRef) if domain == 'disc' else interval(noRef)
    dm = dofmapFactory(element, mesh, PHYSICAL)
    kernel = getFractionalKernel(mesh.dim, s)
    return nonlocalBuilder(dm, kernel, params or {}, zeroExterior=zeroExterior)


def _compare(builder):
    from oracle.oracle import OracleProblem
    A = builder.getDense()
    Aref, cnt, _ = OracleProblem(builder.tables).get_dense()
    got = A.info['counters']
    for key in ('numCellPairs', 'numAssembledCellPairs', 'numIntegrations', 'numBoundaryPairs', 'numBoundaryIntegrations',
                'orders', 'singular'):
        assert got[key] == cnt[key], (key, got[key], cnt[key])
    Ag = A.toarray()
    scale = np.abs(Aref).max()
    err = np.abs(Ag-Aref).max()/scale
    assert err < TOL, err
    assert np.abs(Ag-Ag.T).max() <= 1e-13*scale
    return A, Aref


@pytest.mark.parametrize('noRef,s', [(2, 0.5), (3, 0.5), (4, 0.5), (3, 0.25), (3, 0.75)])
def test_disc_P1_dense(noRef, s):
    _compare(_build('disc', noRef, s, params={'target_order': 0.5}))


def test_disc_P1_no_exterior_zero_row_sums():
    """constants are in the kernel of (u(x)-u(y))(v(x)-v(y)): without the exterior term and with all
    vertices as DoFs the rows sum to zero (SURVEY 8c identity)"""
    from pynucleus_amd import disc, NO_BOUNDARY, P1_DoFMap, getFractionalKernel
    from pynucleus_amd.builder import nonlocalBuilder
    mesh = disc(3)
    dm = P1_DoFMap(mesh, NO_BOUNDARY)
    b = nonlocalBuilder(dm, getFractionalKernel(2, 0.5), {}, zeroExterior=False)
    A, Aref = _compare(b)
    rows = np.abs(A.toarray().sum(axis=1)).max()
    assert rows < 1e-10*np.abs(Aref).max()


@pytest.mark.parametrize('noRef,s', [(4, 0.25), (6, 0.75)])
def test_interval_P1_dense(noRef, s):
    _compare(_build('interval', noRef, s))


@pytest.mark.parametrize('noRef,s', [(2, 0.5), (3, 0.75)])
def test_disc_P2_dense(noRef, s):
    _compare(_build('disc', noRef, s, element='P2'))


def _build_variable(domain, noRef, sFun, element='P1', zeroExterior=True):
    from pynucleus_amd import disc, interval, PHYSICAL, dofmapFactory, getFractionalKernel
    from pynucleus_amd.builder import nonlocalBuilder
    mesh = disc(noRef) if domain == 'disc' else interval(noRef)
    dm = dofmapFactory(element, mesh, PHYSICAL)
    return nonlocalBuilder(dm, getFractionalKernel(mesh.dim, sFun), {}, zeroExterior=zeroExterior)


@pytest.mark.parametrize('case', ['leftRight_P1', 'layers_P1', 'leftRight_P2', 'varconst_1d', 'leftRight_1d', 'leftRight_noext'])
def test_variable_order_dense(case):
    """a16: piecewise-constant variable order (evalParams per element pair, near rules per singularity): the GPU assembles
    class by class, the oracle looks the class up per pair; BASELINE configs[4] uses P2 with such an order"""
    from pynucleus_amd.fractionalOrders import leftRightFractionalOrder, layersFractionalOrder, variableConstFractionalOrder
    if case == 'leftRight_P1':
        b = _build_variable('disc', 3, leftRightFractionalOrder(0.25, 0.75))
    elif case == 'layers_P1':
        orders = np.array([[0.3, 0.4, 0.5], [0.4, 0.5, 0.6], [0.5, 0.6, 0.7]])
        b = _build_variable('disc', 3, layersFractionalOrder(2, np.array([-1., -0.3, 0.3, 1.]), orders))
    elif case == 'leftRight_P2':
        b = _build_variable('disc', 2, leftRightFractionalOrder(0.25, 0.75), element='P2')
    elif case == 'varconst_1d':
        b = _build_variable('interval', 5, variableConstFractionalOrder(0.75))
    elif case == 'leftRight_1d':
        b = _build_variable('interval', 5, leftRightFractionalOrder(0.3, 0.6))
    else:
        b = _build_variable('disc', 3, leftRightFractionalOrder(0.5, 0.75), zeroExterior=False)
    _compare(b)


def test_gemv_and_cg():
    builder = _build('disc', 4, 0.5, params={'target_order': 0.5})
    A, Aref = _compare(builder)
    rng = np.random.default_rng(0)
    x = rng.standard_normal(A.num_rows)
    y = A*x
    yref = Aref@x
    assert np.abs(y-yref).max() <= 1e-12*np.abs(yref).max()*A.num_rows
    b = builder.dm.assembleRHS(1.0)
    u, its, res = A.solve_cg_jacobi(np.asarray(b), tol=1e-10, maxiter=2000)
    uref = np.linalg.solve(Aref, np.asarray(b))
    assert res <= 1e-10 and its < 2000
    assert np.abs(u-uref).max() <= 1e-7*np.abs(uref).max()


def _dist_worker(rank, world, port, out):
    import os
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)                       # one-GPU box: both ranks share the card, collectives over gloo
    from pynucleus_amd import disc, P1_DoFMap, PHYSICAL, getFractionalKernel
    from pynucleus_amd.builder import nonlocalBuilder
    from oracle.oracle import OracleProblem
    mesh = disc(4)
    dm = P1_DoFMap(mesh, PHYSICAL)
    b = nonlocalBuilder(dm, getFractionalKernel(2, 0.5), {'target_order': 0.5}, zeroExterior=True, comm=True)
    op = b.getDense(distributed=True)
    x = np.linspace(-1., 1., dm.num_dofs)
    y = op.matvec(x)
    Aref = OracleProblem(b.tables).get_dense()[0]
    yref = Aref@x
    e1 = float(np.abs(y-yref).max()/np.abs(yref).max())
    full = b.getDense()                            # all-reduced like the reference (NA:1449-1450)
    e2 = float(np.abs(full.toarray()-Aref).max()/np.abs(Aref).max())
    pairs = torch.tensor([op.info['counters']['numAssembledCellPairs']], dtype=torch.float64)
    dist.all_reduce(pairs)
    # the non-symmetric paths under the same split: order per quadrature point (cellNo1 ranges), piecewise non-symmetric table
    from pynucleus_amd.fractionalOrders import smoothedLeftRightFractionalOrder, leftRightFractionalOrder
    e3 = []
    for sF in (smoothedLeftRightFractionalOrder(0.25, 0.75, r=0.3), leftRightFractionalOrder(0.25, 0.75, 0.3, 0.6)):
        mesh3 = disc(3)
        dm3 = P1_DoFMap(mesh3, PHYSICAL)
        b3 = nonlocalBuilder(dm3, getFractionalKernel(2, sF), {}, zeroExterior=True, comm=True)
        A3 = b3.getDense().toarray()
        R3 = OracleProblem(b3.tables).get_dense()[0]
        e3.append(float(np.abs(A3-R3).max()/np.abs(R3).max()))
    if rank == 0:
        out.put((e1, e2, float(pairs.item()), mesh.num_cells, max(e3)))
    dist.destroy_process_group()


def test_two_ranks_share_the_pairs():
    """world size 2 (gloo, both ranks on the one GPU of the box): the tile deal + cell-range split assemble every pair
    exactly once, the distributed operator's matvec and the all-reduced matrix match the oracle"""
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = 29600+os.getpid() % 2000
    procs = [ctx.Process(target=_dist_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    e1, e2, pairs, nc, e3 = out.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert pairs == nc*(nc+1)//2
    assert e1 < 1e-11 and e2 < TOL and e3 < TOL


@pytest.mark.parametrize('case', ['smoothedLeftRight_disc', 'constantNonSym_disc', 'innerOuter_disc', 'smoothedLeftRight_interval',
                                  'linearLeftRight_interval', 'constantNonSym_noext', 'smoothedLeftRight_disc4',
                                  'smoothedLeftRight_disc_P2', 'innerOuter_disc_P2_noext', 'smoothedLeftRight_interval_P2',
                                  'constantNonSym_disc_P2'])
def test_pointwise_nonsymmetric_dense(case):
    """a16: non-symmetric kernels with an order s(x) per quadrature point (fractionalLaplacian{1,2}D_nonsym, both orientations of
    every pair, (2 dpe)^2 local matrices, near rules keyed by the pair's order): GPU == oracle entry-wise, same counters"""
    from pynucleus_amd import disc, interval, PHYSICAL, P1_DoFMap, P2_DoFMap, getFractionalKernel
    from pynucleus_amd.builder import nonlocalBuilder
    from pynucleus_amd.fractionalOrders import (smoothedLeftRightFractionalOrder, constantNonSymFractionalOrder,
                                                smoothedInnerOuterFractionalOrder, linearLeftRightFractionalOrder)
    from oracle.oracle import OracleProblem
    zeroExterior = True
    DoFMap = P2_DoFMap if '_P2' in case else P1_DoFMap
    # P2 (FL2:894-1184 / FL1:410-604 are element-agnostic): (2 x 6)^2 local matrices, merged-DoF rows 6 / 9 / 11
    if case == 'smoothedLeftRight_disc_P2':
        mesh, s = disc(3), smoothedLeftRightFractionalOrder(0.25, 0.75, r=0.3)
    elif case == 'innerOuter_disc_P2_noext':
        mesh, s, zeroExterior = disc(2), smoothedInnerOuterFractionalOrder(0.3, 0.6, r=0.2), False
    elif case == 'smoothedLeftRight_interval_P2':
        mesh, s = interval(5), smoothedLeftRightFractionalOrder(0.25, 0.75)
    elif case == 'constantNonSym_disc_P2':
        mesh, s = disc(2), constantNonSymFractionalOrder(0.4)
    elif case == 'smoothedLeftRight_disc':
        mesh, s = disc(3), smoothedLeftRightFractionalOrder(0.25, 0.75, r=0.3)
    elif case == 'smoothedLeftRight_disc4':              # 24 blocks of 64 cells: some tiles are uniform (k_pw_tile)
        mesh, s = disc(4), smoothedLeftRightFractionalOrder(0.25, 0.75, r=0.3)
    elif case == 'constantNonSym_disc':
        mesh, s = disc(2), constantNonSymFractionalOrder(0.4)
    elif case == 'innerOuter_disc':
        mesh, s = disc(3), smoothedInnerOuterFractionalOrder(0.3, 0.6, r=0.2)
    elif case == 'smoothedLeftRight_interval':
        mesh, s = interval(5), smoothedLeftRightFractionalOrder(0.25, 0.75)
    elif case == 'linearLeftRight_interval':
        mesh, s = interval(5), linearLeftRightFractionalOrder(0.6, 0.3, r=0.25)
    else:
        mesh, s, zeroExterior = disc(2), constantNonSymFractionalOrder(0.6), False
    dm = DoFMap(mesh, PHYSICAL)
    b = nonlocalBuilder(dm, getFractionalKernel(mesh.dim, s), {}, zeroExterior=zeroExterior)
    A = b.getDense()
    Aref, cnt, _ = OracleProblem(b.tables).get_dense()
    got = A.info['counters']
    for key in ('numCellPairs', 'numAssembledCellPairs', 'numIntegrations', 'numBoundaryPairs', 'numBoundaryIntegrations',
                'orders', 'singular'):
        assert got[key] == cnt[key], (key, got[key], cnt[key])
    Ag = A.toarray()
    scale = np.abs(Aref).max()
    assert np.abs(Ag-Aref).max() < TOL*scale, np.abs(Ag-Aref).max()/scale
    if case.startswith('smoothed'):
        assert np.abs(Aref-Aref.T).max() > 1e-6*scale          # genuinely non-symmetric
    if case == 'smoothedLeftRight_disc4':
        assert got['uniformTilePairs'] > 0
    if case == 'smoothedLeftRight_disc':
        # getDiagonal / getEntry of these kernels come from the dense device path
        d = b.getDiagonal()
        assert np.abs(np.asarray(d.diagonal)-np.diag(Aref)).max() < TOL*scale
        assert abs(b.getEntry(3, 11)-Aref[3, 11]) < TOL*scale and abs(b.getEntry(11, 3)-Aref[11, 3]) < TOL*scale


def test_pointwise_stored_errors_disc():
    """The reference's stored numbers for the non-symmetric path on the disc (noRef 5, N = 2977; the CPU oracle needs minutes
    for this size, the GPU path milliseconds): runFractional --domain disc --s constantNonSym(0.25) --problem constant ->
    Hs error 0.18399339204392906; --s twoDomainNonSym(0.25,0.75) --problem knownSolution -> L2 error 0.005965596537366911
    (compared by the reference at relTol 1e-2 / 3e-2)"""
    from math import gamma
    from pynucleus_amd import driverMesh, PHYSICAL, P1_DoFMap, getFractionalKernel
    from pynucleus_amd.builder import nonlocalBuilder
    from pynucleus_amd.fractionalOrders import smoothedLeftRightFractionalOrder, constantNonSymFractionalOrder
    from pynucleus_amd.quadrature import simplexXiaoGimbutas
    from tests.test_pointwise import known_solution_problem
    mesh = driverMesh('disc', 5)
    dm = P1_DoFMap(mesh, PHYSICAL)
    assert dm.num_dofs == 2977
    s = 0.25
    A = nonlocalBuilder(dm, getFractionalKernel(2, constantNonSymFractionalOrder(s)), {'target_order': 0.5}).getDense().toarray()
    b = np.asarray(dm.assembleRHS(1.0))
    u = np.linalg.solve(A, b)
    C = 2.**(-2.*s)*gamma(1.)/gamma((2+2.*s)/2.)/gamma(1.+s)
    hs = np.sqrt(abs(b@u-C*np.pi/(s+1)))
    assert abs(hs-0.18399339204392906) <= 2e-5*0.18399339204392906, hs          # observed 5.6e-6

    kernel = getFractionalKernel(2, smoothedLeftRightFractionalOrder(0.25, 0.75))
    A = nonlocalBuilder(dm, kernel, {'target_order': 0.5}).getDense().toarray()
    assert np.abs(A-A.T).max() > 1e-3*np.abs(A).max()
    rhs, sol, L2ex2 = known_solution_problem(2, kernel)
    b = np.asarray(dm.assembleRHS(rhs, qr=simplexXiaoGimbutas(3, 2, 2)))

    class Gauss2D:                                           # fem/PyNucleus_fem/quadrature.pyx:279-282, femCy.pyx:2646-2650
        nodes = np.array([[0.5, 0.0, 0.5], [0.5, 0.5, 0.0], [0.0, 0.5, 0.5]])
        weights = np.full(3, 1./3.)
        num_nodes = 3
    u = np.linalg.solve(A, b)
    z = np.asarray(dm.assembleRHS(sol, qr=Gauss2D()))
    M = dm.assembleMass()
    err = float(np.sqrt(abs(L2ex2-2*z@u+u@(M@u))))
    assert abs(err-0.005965596537366911) <= 3e-2*0.005965596537366911, err


@pytest.mark.parametrize('case', ['leftRight_disc', 'layers_disc', 'leftRight_interval', 'leftRight_noext'])
def test_piecewise_nonsymmetric_order(case):
    """a16: piecewise-constant order with s(l1, l2) != s(l2, l1): both orientations of every pair, each with the parameters of
    its orientation (NA:1411-1428); GPU == oracle entry-wise, same counters; the operator is symmetric (frozen parameters)"""
    from pynucleus_amd import disc, interval, PHYSICAL, P1_DoFMap, getFractionalKernel
    from pynucleus_amd.builder import nonlocalBuilder
    from pynucleus_amd.fractionalOrders import leftRightFractionalOrder, layersFractionalOrder
    from oracle.oracle import OracleProblem
    zeroExterior = True
    if case == 'leftRight_disc':
        mesh, s = disc(3), leftRightFractionalOrder(0.25, 0.75, 0.3, 0.6)
    elif case == 'layers_disc':
        orders = np.array([[0.3, 0.45, 0.5], [0.35, 0.5, 0.65], [0.6, 0.55, 0.7]])
        mesh, s = disc(3), layersFractionalOrder(2, np.array([-1., -0.3, 0.3, 1.]), orders)
    elif case == 'leftRight_interval':
        mesh, s = interval(5), leftRightFractionalOrder(0.3, 0.7, 0.4, 0.6)
    else:
        mesh, s, zeroExterior = disc(2), leftRightFractionalOrder(0.25, 0.75, 0.3, 0.6), False
    dm = P1_DoFMap(mesh, PHYSICAL)
    kernel = getFractionalKernel(mesh.dim, s)
    assert not kernel.symmetric
    b = nonlocalBuilder(dm, kernel, {}, zeroExterior=zeroExterior)
    A = b.getDense()
    Aref, cnt, _ = OracleProblem(b.tables).get_dense()
    got = A.info['counters']
    for key in ('numAssembledCellPairs', 'numIntegrations', 'numBoundaryPairs', 'numBoundaryIntegrations', 'orders', 'singular'):
        assert got[key] == cnt[key], (key, got[key], cnt[key])
    Ag = A.toarray()
    scale = np.abs(Aref).max()
    assert np.abs(Ag-Aref).max() < TOL*scale, np.abs(Ag-Aref).max()/scale
    assert np.abs(Aref-Aref.T).max() <= 1e-13*scale


@pytest.mark.parametrize('case', ['disc0', 'disc1', 'interval1', 'interval2_P1', 'empty_range', 'all_boundary'])
def test_edge_cases(case):
    """ragged / tiny inputs: fewer cells than one block, a single interior DoF, an empty cell range, a DoF map without any
    interior DoF"""
    import torch
    from pynucleus_amd import disc, interval, PHYSICAL, NO_BOUNDARY, P1_DoFMap, getFractionalKernel
    from pynucleus_amd.builder import nonlocalBuilder
    from oracle.oracle import OracleProblem
    if case == 'disc0':
        mesh = disc(0)
    elif case == 'disc1':
        mesh = disc(1)
    elif case == 'interval1':
        mesh = interval(1)
    else:
        mesh = interval(2) if case == 'interval2_P1' else disc(1)
    dm = P1_DoFMap(mesh, PHYSICAL)
    b = nonlocalBuilder(dm, getFractionalKernel(mesh.dim, 0.5), {}, zeroExterior=True)
    if case == 'empty_range':
        ctx = b.context()
        A = torch.zeros((dm.num_dofs, dm.num_dofs), dtype=torch.float64, device='cuda')
        ctx.assemble_dense(A.data_ptr(), A.stride(0), True, 3, 3)
        torch.cuda.synchronize()
        assert float(A.abs().max()) == 0. and ctx.counters()['numAssembledCellPairs'] == 0
        return
    if case == 'all_boundary':
        mesh = disc(0)                                   # 6 cells around one interior vertex: remove it too -> no DoF at all
        dm = P1_DoFMap(mesh, PHYSICAL)
        assert dm.num_dofs == 1
    A = b.getDense()
    Aref, cnt, _ = OracleProblem(b.tables).get_dense()
    assert A.info['counters']['numAssembledCellPairs'] == cnt['numAssembledCellPairs']
    assert np.abs(A.toarray()-Aref).max() <= TOL*np.abs(Aref).max()


def test_cells_without_locality_are_renumbered():
    """a mesh whose cells are numbered at random (straight out of a generator): the builder renumbers the cells along a Morton
    curve for the tile kernels' LDS sub-blocks; DoF numbers are those of the caller's DoFMap"""
    from pynucleus_amd import disc, PHYSICAL, P1_DoFMap, getFractionalKernel, nonlocalTables
    from pynucleus_amd.mesh import mesh2d
    from pynucleus_amd.builder import nonlocalBuilder, block_dof_count
    from oracle.oracle import OracleProblem
    m0 = disc(4)
    rng = np.random.default_rng(1)
    shuffled = mesh2d(m0.vertices.copy(), np.ascontiguousarray(m0.cells[rng.permutation(m0.num_cells)]))
    dm = P1_DoFMap(shuffled, PHYSICAL)
    assert block_dof_count(dm.dofs, 64) > 100
    kernel = getFractionalKernel(2, 0.5)
    b = nonlocalBuilder(dm, kernel, {'target_order': 0.5})
    assert hasattr(b.dm, 'cell_permutation') and block_dof_count(b.dm.dofs, 64) <= 64
    A = b.getDense().toarray()
    # the operator of the CALLER's numbering: the renumbered cells keep the orientation of every touching pair (which cell is
    # cellNo1 of the singular rule, pnl_set_cell_order), so nothing but the summation order depends on the renumbering
    Aref = OracleProblem(nonlocalTables(dm, kernel, {'target_order': 0.5})).get_dense()[0]
    assert np.abs(A-Aref).max() <= TOL*np.abs(Aref).max()
    # the oracle run on the renumbered cells differs through exactly that orientation, at the quadrature error of the singular
    # rules -- the dependence on the cell numbering the reference itself has
    Aint = OracleProblem(b.tables).get_dense()[0]
    assert TOL*np.abs(Aref).max() < np.abs(Aint-Aref).max() <= 1e-6*np.abs(Aref).max()
    # without the renumbering the sub-block does not fit: the library says so instead of computing something else
    b2 = nonlocalBuilder(dm, kernel, {'target_order': 0.5, 'reorderCells': False})
    with pytest.raises(NotImplementedError):
        b2.getDense()


@pytest.mark.parametrize('noRef,s,zeroExterior', [(5, 0.25, True), (6, 0.75, True), (4, 0.4, False)])
def test_interval_P2_dense(noRef, s, zeroExterior):
    """P2 on intervals (vertex + cell-midpoint DoFs): the configuration whose oracle is pinned to the reference's stored numbers
    to 1e-12 (tests/test_oracle_pinning.py::test_interval_stored_errors_exact)"""
    _compare(_build('interval', noRef, s, element='P2', zeroExterior=zeroExterior))


@pytest.mark.parametrize('element,noRef,s,zeroExterior', [('P0', 6, 0.25, True), ('P0', 4, 0.4, False), ('P3', 5, 0.25, True),
                                                          ('P3', 5, 0.75, True), ('P3', 3, 0.4, False)])
def test_interval_P0_P3_dense(element, noRef, s, zeroExterior):
    """P0 (one DoF per cell, no cancellation across elements: FL1:212-216) and P3 (two vertices + two cell DoFs) on intervals --
    the elements of the reference's fixtures --elementP0 / --elementP3; entries and integer counters against the oracle, whose
    numbers for these elements are pinned to the stored Hs errors (tests/test_oracle_pinning.py)"""
    _compare(_build('interval', noRef, s, element=element, zeroExterior=zeroExterior))


@pytest.mark.parametrize('noRef,s,zeroExterior', [(2, 0.25, True), (3, 0.4, True), (3, 0.3, False)])
def test_disc_P0_dense(noRef, s, zeroExterior):
    """P0 on triangles (the reference's fixture --domaindisc--elementP0): touching pairs merge no DoFs, their rules cancel nothing
    across elements"""
    _compare(_build('disc', noRef, s, element='P0', zeroExterior=zeroExterior, params={'target_order': 0.5}))


def test_disc_P0_stored_error_through_the_gpu():
    """runFractional --domain disc --s const(0.25) --element P0 --matrixFormat dense: stored Hs error 0.1403179566911808 (4.3e-6:
    the triangle rules, like the P1 pin) with the matrix assembled on the GPU, 6144 DoFs"""
    from math import gamma
    from pynucleus_amd import driverMesh, PHYSICAL, dofmapFactory, getFractionalKernel, nonlocalBuilder
    s = 0.25
    dm = dofmapFactory('P0', driverMesh('disc', 5), PHYSICAL)
    A = nonlocalBuilder(dm, getFractionalKernel(2, s), {}).getDense().toarray()
    b = np.asarray(dm.assembleRHS(1.0))
    u = np.linalg.solve(A, b)
    C = 2.**(-2.*s)*gamma(1.)/gamma((2+2.*s)/2.)/gamma(1.+s)          # u = C (1 - |x|^2)^s, (f, u) = C pi / (s + 1)
    hs = np.sqrt(abs(b@u-C*np.pi/(s+1.)))
    assert abs(hs-0.1403179566911808) <= 1e-5*0.1403179566911808, hs


@pytest.mark.parametrize('element,s,noRef,stored', [('P1', 0.25, 6, 0.09611243700804001), ('P2', 0.25, 5, 0.08454379705489531),
                                                    ('P2', 0.75, 5, 0.03250922885004246), ('P0', 0.25, 6, 0.0863469994893122),
                                                    ('P3', 0.25, 5, 0.061422967833697564), ('P3', 0.75, 5, 0.02241204241913628)])
def test_interval_stored_errors_through_the_gpu(element, s, noRef, stored):
    """the reference's stored Hs errors of runFractional --domain interval (reproducible to ~1e-12 without third-party tables)
    through the product path: assembly on the GPU, solve on the host"""
    from math import gamma, pi, sqrt
    from pynucleus_amd import driverMesh, PHYSICAL, dofmapFactory, getFractionalKernel, nonlocalBuilder
    dm = dofmapFactory(element, driverMesh('interval', noRef), PHYSICAL)
    A = nonlocalBuilder(dm, getFractionalKernel(1, s), {'target_order': dm.polynomialOrder+1.-s}).getDense().toarray()
    b = np.asarray(dm.assembleRHS(1.0))
    u = np.linalg.solve(A, b)
    C = 2.**(-2.*s)*gamma(0.5)/gamma((1+2.*s)/2.)/gamma(1.+s)
    hs = np.sqrt(abs(b@u-C*sqrt(pi)*gamma(s+1)/gamma(s+3/2)))
    assert abs(hs-stored) <= 1e-8*stored, (hs, stored)


# ---- bench-size cases (BASELINE.json configs[1] family): oracle on shards, size-independent properties on the whole ------
def test_disc_P1_dense_noRef5_with_uniform_tiles():
    """N = 3 025, 1.9e7 pairs (6 s of oracle): the first size at which most pairs run through the uniform-tile kernel"""
    A, _ = _compare(_build('disc', 5, 0.5, params={'target_order': 0.5}))
    cnt = A.info['counters']
    assert cnt.get('uniformTilePairs', 0) > 0.3*cnt['numAssembledCellPairs'], cnt


def test_bench_size_shard_against_oracle():
    """noRef 6 (N = 12 097, 3.0e8 pairs): the pairs whose first cell lies in a 96-cell slab that cuts through 64-cell blocks,
    GPU shard against oracle shard entry-wise (the reference's cellNo1 split, NA:1280-1285) -- mixed, uniform and
    range-filtered tiles at a size whose full oracle run would take two minutes"""
    import torch
    from oracle.oracle import OracleProblem
    b = _build('disc', 6, 0.5, params={'target_order': 0.5})
    N, nc = b.dm.num_dofs, b.mesh.num_cells
    c0, c1 = nc//2+17, nc//2+17+96
    ctx = b.context()
    A = torch.zeros((N, N), dtype=torch.float64, device='cuda')
    ctx.assemble_dense(A.data_ptr(), N, True, c0, c1)
    ctx.synchronize()
    cnt = ctx.counters()
    Aref, cref, _ = OracleProblem(b.tables).get_dense(c0, c1)
    for key in ('numAssembledCellPairs', 'numIntegrations', 'numBoundaryPairs', 'orders', 'singular'):
        assert cnt[key] == cref[key], (key, cnt[key], cref[key])
    err = np.abs(A.cpu().numpy()-Aref).max()/np.abs(Aref).max()
    assert err < TOL, err


def test_bench_size_properties():
    """the bench workload itself (noRef 7, N = 48 769, 4.83e9 pairs, 19 GB): pair count, symmetry, determinism of the counters,
    two cellNo1 shards (written with the multi-rank symmetric flush) adding up to the whole (linearity over the pair partition), and the known answer of the driver
    problem: with f = 1 the energy b.u converges to C(2, s) pi / (s + 1) (analytic solution C (1-|x|^2)^s), the discrete
    energy approaching it from below with the mesh"""
    import torch
    from math import gamma, pi
    s = 0.5
    b = _build('disc', 7, s, params={'target_order': 0.5})
    N, nc = b.dm.num_dofs, b.mesh.num_cells
    A = b.getDense()
    cnt = A.info['counters']
    assert cnt['numCellPairs'] == nc*(nc+1)//2
    assert sum(cnt['orders'].values())+sum(cnt['singular'].values()) == cnt['numAssembledCellPairs']
    M = A.A
    scale = float(M.abs().max())
    blk = 8192
    for i in range(0, N, blk):                       # symmetry, blockwise (no second 19 GB copy)
        assert float((M[i:i+blk, :]-M[:, i:i+blk].T).abs().max()) <= 1e-13*scale
    # shards: the sum of two cell ranges is the operator
    ctx = b.context()
    P = torch.zeros((N, N), dtype=torch.float64, device='cuda')
    half = nc//2+29
    SYMMETRIC_FLUSH = 2                                # PNL_FLAG_SYMMETRIC_FLUSH: both shards accumulate into one block, no mirror pass
    ctx.assemble_dense(P.data_ptr(), N, True, 0, half, SYMMETRIC_FLUSH)
    c_lo = ctx.counters()
    ctx.assemble_dense(P.data_ptr(), N, True, half, nc, SYMMETRIC_FLUSH)
    c_hi = ctx.counters()
    ctx.synchronize()
    assert c_lo['numAssembledCellPairs']+c_hi['numAssembledCellPairs'] == cnt['numAssembledCellPairs']
    assert c_lo['numIntegrations']+c_hi['numIntegrations'] == cnt['numIntegrations']
    for i in range(0, N, blk):
        assert float((P[i:i+blk]-M[i:i+blk]).abs().max()) <= 1e-12*scale
    del P
    # known answer
    rhs = np.asarray(b.dm.assembleRHS(1.0))
    u, its, res = A.solve_cg_jacobi(rhs, tol=1e-10, maxiter=500)
    assert its < 200 and res < 1e-9
    C2s = 2.**(-2.*s)*gamma(1.)/gamma(1.+s)**2        # C(d=2, s) = 2^{-2s} Gamma(d/2) / (Gamma((d+2s)/2) Gamma(1+s))
    exact = C2s*pi/(s+1.)
    energy = float(rhs@u)
    assert 0. < exact-energy < 4e-3*exact, (energy, exact)
    # ... at the rate of the method: |u - u_h|_Hs^2 = exact - energy = O(h) for s = 1/2 (boundary singularity), h ratio 4 to noRef 5
    b5 = _build('disc', 5, s, params={'target_order': 0.5})
    rhs5 = np.asarray(b5.dm.assembleRHS(1.0))
    u5 = b5.getDense().solve_cg_jacobi(rhs5, tol=1e-10, maxiter=500)[0]
    ratio = (exact-float(rhs5@u5))/(exact-energy)
    assert 3. < ratio < 5.5, ratio


class _Group(dict):
    """the part of h5py.Group the operators' HDF5write / HDF5read use (h5py is not installed here)"""

    def __init__(self):
        super().__init__()
        self.attrs = {}

    def create_dataset(self, name, data=None, **kwargs):
        self[name] = np.array(data, copy=True)

    def create_group(self, name):
        self[name] = _Group()
        return self[name]


def test_operator_files_round_trip():
    """HDF5write / HDF5read of the dense, CSR and SSS operators in the reference's layout (DenseLinearOperator_{SCALAR}.pxi:86-94,
    CSR_LinearOperator_{SCALAR}.pxi:268-290, SSS_LinearOperator_{SCALAR}.pxi:273-293): same datasets and attributes, the operator
    read back applies like the one written"""
    from pynucleus_amd import uniformSquare, disc, P1_DoFMap, PHYSICAL, NO_BOUNDARY, getFractionalKernel, getKernel, INDICATOR
    from pynucleus_amd.builder import nonlocalBuilder
    from pynucleus_amd.linear_operators import Dense_LinearOperator, CSR_LinearOperator
    rng = np.random.default_rng(0)
    b = nonlocalBuilder(P1_DoFMap(disc(2), PHYSICAL), getFractionalKernel(2, 0.5), {})
    A = b.getDense()
    g = _Group()
    A.HDF5write(g)
    assert g.attrs['type'] == 'dense' and g['data'].shape == A.shape
    A2 = Dense_LinearOperator.HDF5read(g, b.context())
    x = rng.standard_normal(A.num_rows)
    assert np.abs(A2.toarray()-A.toarray()).max() == 0.
    assert np.abs(A2*x-A*x).max() <= 1e-14*np.abs(A*x).max()      # another leading dimension: another summation order
    dm = P1_DoFMap(uniformSquare(17), NO_BOUNDARY)
    for params in ({}, {'forceUnsymmetric': True}):                 # SSS (default) and CSR
        bs = nonlocalBuilder(dm, getKernel(2, kernel=INDICATOR, horizon=0.2), params, zeroExterior=False)
        S = bs.getSparse()
        g = _Group()
        S.HDF5write(g)
        assert g.attrs['type'] in ('csr', 'sss') and set(g) >= {'indices', 'indptr', 'data'}
        S2 = CSR_LinearOperator.HDF5read(g, bs.context())
        assert type(S2) is type(S)
        x = rng.standard_normal(S.num_rows)
        assert np.abs(S2*x-S*x).max() <= 1e-14*np.abs(S*x).max()
        assert np.abs(S2.toarray()-S.toarray()).max() == 0.


def test_general_exponent_at_scale_properties():
    """s = 0.4 (no rsqrt shortcut: the table-driven power of the tile kernels, pnl_pow_tab) at 24,576 cells / 12,097 DoFs: symmetry,
    two cell-range shards adding up to the operator, the energy of the driver problem below the exact value and converging, and
    the tile kernels' power against the exp / ln path (option PNL_NO_POWTAB) entry by entry"""
    import os
    import torch
    from math import gamma, pi
    s = 0.4
    b = _build('disc', 6, s, params={'target_order': 0.5})
    N, nc = b.dm.num_dofs, b.mesh.num_cells
    A = b.getDense()
    cnt = A.info['counters']
    assert cnt['numCellPairs'] == nc*(nc+1)//2
    M = A.A
    scale = float(M.abs().max())
    assert float((M-M.T).abs().max()) <= 1e-13*scale
    ctx = b.context()
    P = torch.zeros((N, N), dtype=torch.float64, device='cuda')
    half = nc//2+17
    ctx.assemble_dense(P.data_ptr(), N, True, 0, half, 2)
    ctx.assemble_dense(P.data_ptr(), N, True, half, nc, 2)
    ctx.synchronize()
    assert float((P-M).abs().max()) <= 1e-12*scale
    del P
    rhs = np.asarray(b.dm.assembleRHS(1.0))
    u, its, res = A.solve_cg_jacobi(rhs, tol=1e-10, maxiter=800)
    exact = 2.**(-2.*s)*gamma(1.)/gamma(1.+s)**2*pi/(s+1.)
    energy = float(rhs@u)
    assert 0. < exact-energy < 1e-2*exact, (energy, exact)
    # the same operator with exp(e ln x) from the __constant__ tables
    from pynucleus_amd import _lib
    _lib.set_option('PNL_NO_POWTAB', '1')
    try:
        b2 = _build('disc', 6, s, params={'target_order': 0.5})
        A2 = b2.getDense()
        assert A2.info['counters']['numIntegrations'] == cnt['numIntegrations']
        assert float((A2.A-M).abs().max()) <= 1e-13*scale
    finally:
        _lib.set_option('PNL_NO_POWTAB', None)


@pytest.mark.parametrize('case', ['layers_P2', 'layers_P1', 'leftRight_P2', 'leftRight_nonsym_P1', 'layers_P2_noext'])
def test_label_blocks_dense(case):
    """C5: cell blocks that follow the interfaces of a piecewise-constant order (builder.label_blocks: straddling blocks split per
    label, filled with zero-volume padding cells) against the plain numbering and the oracle: same integer counters (the padding
    cells form no pair), entries at 1e-11, and the tile kernels see no multi-label block any more (more pairs in uniform tiles)"""
    from pynucleus_amd import disc, PHYSICAL, NO_BOUNDARY, P1_DoFMap, P2_DoFMap, getFractionalKernel
    from pynucleus_amd.builder import nonlocalBuilder
    from pynucleus_amd.fractionalOrders import leftRightFractionalOrder, layersFractionalOrder
    from oracle.oracle import OracleProblem
    layers = layersFractionalOrder(2, np.array([-1., -0.3, 0.3, 1.]), np.array([[0.3, 0.4, 0.5], [0.4, 0.5, 0.6], [0.5, 0.6, 0.7]]))
    zeroExterior, tag = True, PHYSICAL
    if case == 'layers_P2':
        mesh, DoFMap, s = disc(4), P2_DoFMap, layers
    elif case == 'layers_P1':
        mesh, DoFMap, s = disc(4), P1_DoFMap, layers
    elif case == 'leftRight_P2':
        mesh, DoFMap, s = disc(3), P2_DoFMap, leftRightFractionalOrder(0.25, 0.75, interface=0.1)
    elif case == 'leftRight_nonsym_P1':
        mesh, DoFMap, s = disc(4), P1_DoFMap, leftRightFractionalOrder(0.25, 0.75, 0.3, 0.6)
    else:
        mesh, DoFMap, s, zeroExterior, tag = disc(3), P2_DoFMap, layers, False, NO_BOUNDARY
    dm = DoFMap(mesh, tag)
    kernel = getFractionalKernel(2, s)
    b1 = nonlocalBuilder(dm, kernel, {'target_order': 0.5}, zeroExterior=zeroExterior)
    b0 = nonlocalBuilder(dm, kernel, {'target_order': 0.5, 'labelBlocks': False}, zeroExterior=zeroExterior)
    assert b1._blocked_dense() is not None and b0._blocked_dense() is None
    A1, A0 = b1.getDense(), b0.getDense()
    c1, c0 = A1.info['counters'], A0.info['counters']
    Aref, cnt, _ = OracleProblem(b1.tables).get_dense()
    for key in ('numCellPairs', 'numAssembledCellPairs', 'numIntegrations', 'numBoundaryPairs', 'numBoundaryIntegrations', 'orders', 'singular'):
        assert c1[key] == cnt[key], (key, c1[key], cnt[key])
        assert c0[key] == cnt[key], (key, c0[key], cnt[key])
    scale = np.abs(Aref).max()
    assert np.abs(A1.toarray()-Aref).max() < TOL*scale, np.abs(A1.toarray()-Aref).max()/scale
    assert np.abs(A0.toarray()-Aref).max() < TOL*scale
    if case != 'leftRight_nonsym_P1':                        # non-symmetric tables run without uniform tiles
        assert c1.get('uniformTilePairs', 0) >= c0.get('uniformTilePairs', 0)
    # repeated assembly on the blocked context gives the same operator
    A2 = b1.getDense()
    assert np.abs(A2.toarray()-A1.toarray()).max() <= 1e-14*scale


@pytest.mark.parametrize('case', ['disc_P1', 'disc_P2', 'interval_P1', 'disc_P1_noext'])
def test_fe_fractional_order_dense(case):
    """a16: feFractionalOrder (fractionalOrders.pyx:660-668, lookupExtended :541-587) -- the order is a P1 function on the mesh,
    s(x) per quadrature point.  GPU (s from the cell's vertex values and the barycentric coordinates of the point) == oracle (point
    location + evaluation, like the reference's cellFinder) entry-wise, same counters; a P1 function that is linear on the whole
    domain reproduces the linearStep order with the same range exactly."""
    from pynucleus_amd import disc, interval, PHYSICAL, NO_BOUNDARY, P1_DoFMap, P2_DoFMap, getFractionalKernel, feFractionalOrder, nonlocalTables
    from pynucleus_amd.builder import nonlocalBuilder
    from pynucleus_amd.fractionalOrders import linearLeftRightFractionalOrder
    from oracle.oracle import OracleProblem
    zeroExterior = case != 'disc_P1_noext'
    mesh = interval(5) if case.startswith('interval') else disc(3 if case == 'disc_P1' else 2)
    dm = (P2_DoFMap if case == 'disc_P2' else P1_DoFMap)(mesh, PHYSICAL)
    dms = P1_DoFMap(mesh, NO_BOUNDARY)                        # the space of the order: P1 on all vertices
    dofs, cells = np.asarray(dms.dofs), np.asarray(mesh.cells)

    def coefficients(fun):
        u = np.zeros(dms.num_dofs)
        u[dofs.ravel()] = fun(mesh.vertices[cells.ravel()])
        return u
    # a bumpy order between 0.3 and 0.7
    u = coefficients(lambda p: 0.5+0.2*np.sin(3.*p[:, 0])*np.cos(2.*p[:, -1]))
    b = nonlocalBuilder(dm, getFractionalKernel(mesh.dim, feFractionalOrder(u, 0.3, 0.7, dm=dms)), {}, zeroExterior=zeroExterior)
    A = b.getDense()
    Aref, cnt, _ = OracleProblem(b.tables).get_dense()
    got = A.info['counters']
    for key in ('numCellPairs', 'numAssembledCellPairs', 'numIntegrations', 'numBoundaryPairs', 'numBoundaryIntegrations', 'orders', 'singular'):
        assert got[key] == cnt[key], (key, got[key], cnt[key])
    scale = np.abs(Aref).max()
    assert np.abs(A.toarray()-Aref).max() < TOL*scale, np.abs(A.toarray()-Aref).max()/scale
    assert np.abs(Aref-Aref.T).max() > 1e-6*scale
    if case == 'disc_P1':
        # s(x) = 0.5 + 0.2 x_0 as a P1 function == linearStep(0.1, 0.9, r = 2) (sl + (x - interface + r) (sr - sl) / (2 r))
        ul = coefficients(lambda p: 0.5+0.2*p[:, 0])
        A1 = nonlocalBuilder(dm, getFractionalKernel(2, feFractionalOrder(ul, 0.1, 0.9, dm=dms)), {}).getDense().toarray()
        A2 = OracleProblem(nonlocalTables(dm, getFractionalKernel(2, linearLeftRightFractionalOrder(0.1, 0.9, r=2.)), {})).get_dense()[0]
        assert np.abs(A1-A2).max() < TOL*np.abs(A2).max()


@pytest.mark.parametrize('case', ['gaussian_1d', 'exponential_1d', 'gaussian_1d_P2', 'gaussian_2d', 'exponential_noext'])
def test_full_space_integrable_kernels_dense(case):
    """Gaussian / exponential kernels on the full space (kernelsCy.pyx:388-477): interior pairs with the branchy general kern_eval,
    exterior term with the Gauss-theorem twins PNL_GAUSSIAN_BOUNDARY / PNL_EXPONENTIAL_BOUNDARY (erfc in 1D; in 2D the 1/|x-y| of the
    normal factor is folded in); entries and counters against the oracle"""
    from pynucleus_amd import disc, interval, PHYSICAL, dofmapFactory, getKernel
    from pynucleus_amd.builder import nonlocalBuilder
    name = case.split('_')[0]
    kw = {'variance': 0.1} if name == 'gaussian' else {'exponentialRate': 8.}
    if case.endswith('2d'):
        if name == 'exponential':
            pytest.skip('the exponential kernel is normalised in 1D only (kernelNormalization.pyx:275-282)')
        mesh, element = disc(2), 'P1'
    else:
        mesh, element = interval(5), ('P2' if case.endswith('P2') else 'P1')
    dm = dofmapFactory(element, mesh, PHYSICAL)
    k = getKernel(mesh.dim, kernel=name, horizon=np.inf, **kw)
    _compare(nonlocalBuilder(dm, k, {}, zeroExterior=not case.endswith('noext')))


@pytest.mark.parametrize('name,kw,stored_l2,stored_linf', [('gaussian', {'variance': 0.1}, 0.0029565447289171816, 0.006737946999085467),
                                                           ('exponential', {'exponentialRate': 8.}, 0.00025530396949181036, 0.00033546262790251185)])
def test_full_space_integrable_fixtures_through_the_gpu(name, kw, stored_l2, stored_linf):
    """tests/cache_runNonlocal.py--domaininterval--kernelType{gaussian,exponential}--...--matrixFormatH2--...--interactionfullSpace--horizoninf
    through the product path, dense and H2 (the stored runs are H2): 'Linf error interpolated' to the last digits, 'L2 error
    interpolated' to 1e-6 / 1e-4 dense and H2"""
    from pynucleus_amd import driverMesh, P1_DoFMap, PHYSICAL, NO_BOUNDARY, getKernel
    from pynucleus_amd.builder import nonlocalBuilder
    from test_kernels import full_space_errors
    mesh = driverMesh('interval', 8)
    dm, dmA = P1_DoFMap(mesh, PHYSICAL), P1_DoFMap(mesh, NO_BOUNDARY)
    k = getKernel(1, kernel=name, horizon=np.inf, **kw)
    b = nonlocalBuilder(dm, k, {})
    l2, linf = full_space_errors(name, kw, k, dm, dmA, b.getDense().toarray())
    assert abs(l2-stored_l2) <= (1e-6 if name == 'gaussian' else 1e-4)*stored_l2 and abs(linf-stored_linf) <= 1e-11*stored_linf, (l2, linf)
    H = b.getH2()
    l2h, linfh = full_space_errors(name, kw, k, dm, dmA, H.toarray())
    assert abs(l2h-stored_l2) <= 1e-4*stored_l2 and abs(linfh-stored_linf) <= 1e-8*stored_linf, (l2h, linfh)


@pytest.mark.gpu
@pytest.mark.parametrize('element', ['P1', 'P2'])
def test_queued_assemblies_equal_a_single_one(element):
    """bench.py queues its steps without a host synchronisation in between: the side streams of a step (touching pairs, boundary
    term next to the fold pass) must be joined before the next step clears the per-cell diagonal blocks.  Three queued assemblies
    into one matrix leave what a single one leaves (block-slot path: every entry is overwritten)."""
    import torch
    from pynucleus_amd import disc, PHYSICAL, dofmapFactory, getFractionalKernel
    from pynucleus_amd.builder import nonlocalBuilder
    mesh = disc(5 if element == 'P1' else 4)
    dm = dofmapFactory(element, mesh, PHYSICAL)
    b = nonlocalBuilder(dm, getFractionalKernel(2, 0.5), {'target_order': 0.5}, zeroExterior=True)
    ctx = b.context()
    N, nc = dm.num_dofs, mesh.num_cells
    dev = torch.device('cuda', ctx.device)
    overwrites = ctx.dense_overwrites(0, nc)

    def run(k):
        A = torch.zeros((N, N), dtype=torch.float64, device=dev)
        for _ in range(k):
            if not overwrites:
                A.zero_()
            ctx.assemble_dense(A.data_ptr(), A.stride(0), True, 0, nc)
        torch.cuda.synchronize(dev)
        return A.cpu().numpy()
    one, three = run(1), run(3)
    assert np.abs(one-one.T).max() <= 1e-13*np.abs(one).max()
    assert np.abs(three-one).max() <= 1e-13*np.abs(one).max()


@pytest.mark.parametrize('case', ['P1_s0.5', 'P1_s0.3', 'P2_s0.75', 'P1_layers', 'P2_layers', 'P1_gaussian'])
def test_boundary_tiled_equals_per_pair_kernel(case):
    """Omega x Omega^c term of the dense path (NA:1430-1448): the tiled kernel (256 cells per workgroup, facets and rules in LDS, row sums
    of orders 2 and 3 in registers; pnl_bndtile.h) against the per-pair kernel (option PNL_BND_OLD) -- the same operator to rounding,
    the same numbers of boundary pairs and boundary kernel evaluations; both are compared with the oracle by the other tests of this
    file, every one of which assembles with zeroExterior"""
    from pynucleus_amd import disc, PHYSICAL, dofmapFactory, getFractionalKernel, getKernel, _lib
    from pynucleus_amd.builder import nonlocalBuilder
    from pynucleus_amd.fractionalOrders import layersFractionalOrder
    element = case[:2]
    if case.endswith('layers'):
        order = layersFractionalOrder(2, np.array([-1., -0.3, 0.3, 1.]), np.array([[0.3, 0.4, 0.5], [0.4, 0.5, 0.6], [0.5, 0.6, 0.7]]))
        kernel = getFractionalKernel(2, order)
    elif case.endswith('gaussian'):
        kernel = getKernel(2, kernel='gaussian', horizon=np.inf, variance=0.1)
    else:
        kernel = getFractionalKernel(2, float(case.split('_s')[1]))
    dm = dofmapFactory(element, disc(4 if element == 'P1' else 3), PHYSICAL)
    out = {}
    try:
        for old in (False, True):
            _lib.set_option('PNL_BND_OLD', 1 if old else None)
            A = nonlocalBuilder(dm, kernel, {'target_order': 0.5}, zeroExterior=True).getDense()
            out[old] = (A.toarray(), A.info['counters'])
    finally:
        _lib.set_option('PNL_BND_OLD', None)
    (A0, c0), (A1, c1) = out[False], out[True]
    assert c0['numBoundaryPairs'] == c1['numBoundaryPairs'] > 0 and c0['numBoundaryIntegrations'] == c1['numBoundaryIntegrations']
    assert np.abs(A0-A1).max() < 1e-13*np.abs(A1).max()


def test_lambda_fractional_order_dense_and_h2():
    """lambdaFractionalOrder (fractionalOrders.pyx:176-201), tabulated on the host: the callable that restates leftRight(0.25, 0.75, 0.3, 0.6)
    assembles the operator of leftRightFractionalOrder -- dense (both orientations of the non-symmetric table) and through getH2"""
    from pynucleus_amd import disc, P1_DoFMap, PHYSICAL, getFractionalKernel
    from pynucleus_amd.builder import nonlocalBuilder
    from pynucleus_amd.fractionalOrders import lambdaFractionalOrder, leftRightFractionalOrder
    dm = P1_DoFMap(disc(3), PHYSICAL)
    ref = leftRightFractionalOrder(0.25, 0.75, 0.3, 0.6)
    lam = lambdaFractionalOrder(2, 0.25, 0.75, False, lambda x, y: [[0.25, 0.3], [0.6, 0.75]][int(x[0] >= 0.)][int(y[0] >= 0.)])
    out = []
    for order in (ref, lam):
        b = nonlocalBuilder(dm, getFractionalKernel(2, order), {'eta': 3., 'minClusterSize': 8}, zeroExterior=True)
        A = b.getDense().toarray()
        x = np.cos(0.37*np.arange(dm.num_dofs))
        out.append((A, b.getH2().matvec(x)))
    assert np.abs(out[0][0]-out[1][0]).max() < 1e-13*np.abs(out[0][0]).max()
    # the labels are numbered in the order the tabulation meets them: the kernel blocks hang under the tree in another order, another
    # (equally valid) tree -- the two H2 operators agree to the accuracy of the far-field interpolation
    assert np.abs(out[0][1]-out[1][1]).max() < 1e-4*np.abs(out[0][1]).max()


@pytest.mark.parametrize('element,domain,noRef,s', [('P1', 'disc', 3, 0.75), ('P1', 'interval', 5, 0.25), ('P2', 'disc', 2, 0.5)])
def test_two_dofmaps_dense_block(element, domain, noRef, s):
    """nonlocalBuilder(dm, kernel, dm2=dmBC).getDense() (NA:879-901, 1366-1375; examples/example_InfHorizonDirichlet.py): the block
    (interior DoFs) x (boundary DoFs) of the operator over the combined map, against the oracle's operator of that map"""
    import torch
    from pynucleus_amd import disc, interval, PHYSICAL, dofmapFactory, getFractionalKernel
    from pynucleus_amd.builder import nonlocalBuilder
    from pynucleus_amd.local_matrix import nonlocalTables
    from oracle.oracle import OracleProblem
    mesh = disc(noRef) if domain == 'disc' else interval(noRef)
    dm = dofmapFactory(element, mesh, PHYSICAL)
    dmBC = dm.getComplementDoFMap()
    kernel = getFractionalKernel(mesh.dim, s)
    B = nonlocalBuilder(dm, kernel, {'target_order': 0.5} if domain == 'disc' else {}, zeroExterior=True, dm2=dmBC).getDense()
    assert B.shape == (dm.num_dofs, dmBC.num_dofs)
    both = dm.combine(dmBC)
    Aref = OracleProblem(nonlocalTables(both, kernel, {'target_order': 0.5} if domain == 'disc' else {}, zeroExterior=True)).get_dense()[0]
    ref = Aref[:dm.num_dofs, dm.num_dofs:]
    assert np.abs(B.toarray()-ref).max() <= TOL*np.abs(Aref).max()
    x = np.random.default_rng(4).standard_normal(dmBC.num_dofs)
    assert np.abs(B*x-ref@x).max() <= 1e-11*np.abs(Aref).max()*dmBC.num_dofs
    y = B.matvec(torch.from_numpy(x).cuda())
    assert np.abs(y.cpu().numpy()-ref@x).max() <= 1e-11*np.abs(Aref).max()*dmBC.num_dofs
    with pytest.raises(NotImplementedError):
        nonlocalBuilder(dm, kernel, {}, zeroExterior=True, dm2=dmBC).getH2()
