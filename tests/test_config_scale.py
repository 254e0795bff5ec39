"""The BASELINE.json configurations other than the bench workload at THEIR sizes (C3: square 129^2, constant kernel, horizon 0.1,
getSparse; C4: disc noRef 7, s = 0.75, H2; C5: disc noRef 6, P2, three-layer variable order, dense), checked through
size-independent properties and through shards the oracle can afford -- what tests/test_gpu_parity.py::test_bench_size_* do for
the bench workload (C2)."""
import numpy as np
import pytest

TOL = 1e-11


@pytest.mark.gpu
def test_c5_p2_layers_noRef6_shard_and_properties():
    """C5: 24 576 P2 cells, N = 48 769, layers(0.3 ... 0.7): a 40-cell cellNo1 shard that straddles 32-cell blocks and the
    layer boundary against the oracle entry-wise (five kernel classes, multi-class tiles, class-aware boundary pass, deferred
    boundary pairs); the whole operator: pair count, counter identities, symmetry, two shards adding up to the whole"""
    import torch
    from pynucleus_amd import disc, P2_DoFMap, PHYSICAL, getFractionalKernel
    from pynucleus_amd.builder import nonlocalBuilder
    from pynucleus_amd.fractionalOrders import layersFractionalOrder
    from oracle.oracle import OracleProblem
    orders = np.array([[0.3, 0.4, 0.5], [0.4, 0.5, 0.6], [0.5, 0.6, 0.7]])
    s = layersFractionalOrder(2, np.array([-1., -0.3, 0.3, 1.]), orders)
    dm = P2_DoFMap(disc(6), PHYSICAL)
    b = nonlocalBuilder(dm, getFractionalKernel(2, s), {'target_order': 0.5}, zeroExterior=True)
    N, nc = b.dm.num_dofs, b.mesh.num_cells
    lab = b.tables.cell_labels
    # a shard whose cells carry two labels
    change = np.nonzero(lab[1:] != lab[:-1])[0]
    c0 = int(change[len(change)//2])-19
    c1 = c0+40
    assert np.unique(lab[c0:c1]).shape[0] >= 2
    ctx = b.context()
    A = torch.zeros((N, N), dtype=torch.float64, device='cuda')
    ctx.assemble_dense(A.data_ptr(), N, True, c0, c1)
    ctx.synchronize()
    cnt = ctx.counters()
    Aref, cref, _ = OracleProblem(b.tables).get_dense(c0, c1)
    for key in ('numAssembledCellPairs', 'numIntegrations', 'numBoundaryPairs', 'numBoundaryIntegrations', 'orders', 'singular'):
        assert cnt[key] == cref[key], (key, cnt[key], cref[key])
    # rows of the shard's DoFs and their mirror images hold everything the shard wrote
    err = float((A.cpu()-torch.from_numpy(Aref)).abs().max())/np.abs(Aref).max()
    assert err < TOL, err
    del A, Aref
    # the whole operator (block-slot path, one launch over the classes)
    Aop = b.getDense()
    c = Aop.info['counters']
    assert c['numCellPairs'] == nc*(nc+1)//2
    assert sum(c['orders'].values())+sum(c['singular'].values()) == c['numAssembledCellPairs']
    M = Aop.A
    scale = float(M.abs().max())
    blk = 8192
    for i in range(0, N, blk):
        assert float((M[i:i+blk, :]-M[:, i:i+blk].T).abs().max()) <= 1e-13*scale
    P = torch.zeros((N, N), dtype=torch.float64, device='cuda')
    half = nc//2+13
    SYMMETRIC_FLUSH = 2
    ctx.assemble_dense(P.data_ptr(), N, True, 0, half, SYMMETRIC_FLUSH)
    c_lo = ctx.counters()
    ctx.assemble_dense(P.data_ptr(), N, True, half, nc, SYMMETRIC_FLUSH)
    c_hi = ctx.counters()
    ctx.synchronize()
    assert c_lo['numAssembledCellPairs']+c_hi['numAssembledCellPairs'] == c['numAssembledCellPairs']
    assert c_lo['numIntegrations']+c_hi['numIntegrations'] == c['numIntegrations']
    for i in range(0, N, blk):
        assert float((P[i:i+blk]-M[i:i+blk]).abs().max()) <= 1e-12*scale


@pytest.mark.gpu
def test_c4_h2_noRef7_properties():
    """C4: disc noRef 7 (98 304 cells, N = 48 769), s = 0.75: the cluster pairs tile the matrix exactly once, the H2 operator
    is symmetric in the energy sense, agrees with the dense operator within the reference's H2 tolerance and solves the
    driver problem to the same energy"""
    import torch
    from pynucleus_amd import disc, P1_DoFMap, PHYSICAL, getFractionalKernel
    from pynucleus_amd.builder import nonlocalBuilder
    from pynucleus_amd.h2 import H2Matrix
    from pynucleus_amd import solvers
    dm = P1_DoFMap(disc(7), PHYSICAL)
    b = nonlocalBuilder(dm, getFractionalKernel(2, 0.75), {'target_order': 0.5, 'eta': 3.}, zeroExterior=True)
    N = dm.num_dofs
    h2, Pnear = b.getH2(returnNearField=True)
    assert isinstance(h2, H2Matrix)
    covered = sum(len(cp.n1.dofs)*len(cp.n2.dofs) for cp in Pnear)
    covered += sum(len(cp.n1.dofs)*len(cp.n2.dofs) for lvl in h2.Pfar.values() for cp in lvl)
    assert covered == N*N
    assert h2.Anear.nnz < 0.1*N*N
    D = b.getDense()
    rng = np.random.default_rng(3)
    x = torch.as_tensor(rng.standard_normal(N), device='cuda')
    y = torch.as_tensor(rng.standard_normal(N), device='cuda')
    hx, hy, dx = h2.matvec(x), h2.matvec(y), D.matvec(x)
    # tests/test_nearField.py: epsRelH2 = 1e-1 entry-wise; cache_testDistOp: |(A_dense - A_h2) x| ~ 1e-4 of |A x|
    rel = float((hx-dx).norm()/dx.norm())
    assert rel < 5e-3, rel
    assert abs(float(y@hx)-float(x@hy)) <= 1e-10*abs(float(y@hx))+1e-10*float(hx.norm()*y.norm())
    rhs = np.asarray(dm.assembleRHS(1.0))
    ud = D.solve_cg_jacobi(rhs, tol=1e-9, maxiter=2000)[0]
    uh = solvers.cg(h2, rhs, tol=1e-9, maxiter=2000)[0]
    e_d, e_h = float(rhs@np.asarray(ud)), float(rhs@np.asarray(uh))
    assert abs(e_d-e_h) < 2e-3*e_d, (e_d, e_h)


@pytest.mark.gpu
def test_c3_square129_getSparse_properties():
    """C3: square with 129^2 vertices, constant kernel, horizon 0.1 (two cells wide: every touching pair is cut by the
    horizon): the sparse operator has zero row sums (no exterior term), is positive semi-definite in the energy sense, acts
    as -Laplace on quadratics in the interior, and the device pair generator visits exactly the host's pair list"""
    import torch
    from pynucleus_amd import uniformSquare, P1_DoFMap, NO_BOUNDARY, getKernel, INDICATOR
    from pynucleus_amd.builder import nonlocalBuilder
    from pynucleus_amd import clusters
    delta = 0.1
    mesh = uniformSquare(129, 129, 0., 0., 1., 1.)
    dm = P1_DoFMap(mesh, NO_BOUNDARY)
    b = nonlocalBuilder(dm, getKernel(2, kernel=INDICATOR, horizon=delta), {}, zeroExterior=False)
    A = b.getSparse()
    N = dm.num_dofs
    cnt = A.info['counters']
    ones = torch.ones(N, dtype=torch.float64, device='cuda')
    r = A.matvec(ones)
    scale = float(torch.as_tensor(A.diagonal).abs().max())
    assert float(r.abs().max()) <= 1e-11*scale
    rng = np.random.default_rng(1)
    x = torch.as_tensor(rng.standard_normal(N), device='cuda')
    assert float(x@A.matvec(x)) > 0.
    # -Laplace on x0^2 (normalised constant kernel): (A q)_I / int phi_I -> -2 away from the boundary layer of width delta
    X = dm.getDoFCoordinates()
    q = torch.as_tensor(X[:, 0]**2, device='cuda')
    mass = np.asarray(dm.assembleRHS(1.0))
    inner = (X[:, 0] > 2*delta) & (X[:, 0] < 1-2*delta) & (X[:, 1] > 2*delta) & (X[:, 1] < 1-2*delta)
    lap = (A.matvec(q).cpu().numpy()/mass)[inner]
    assert np.abs(lap+2.).max() < 0.08, np.abs(lap+2.).max()
    # the same operator from the host's explicit pair list (interactingCellPairs), entry for entry
    b2 = nonlocalBuilder(dm, getKernel(2, kernel=INDICATOR, horizon=delta), {'pairList': 'host'}, zeroExterior=False)
    A2 = b2.getSparse()
    # (the host route keeps the structural zeros of its coarser pattern: compare the operators, not the arrays)
    assert A2.info['counters']['numAssembledCellPairs'] == cnt['numAssembledCellPairs']
    assert A2.nnz >= A.nnz
    for _ in range(3):
        v = torch.as_tensor(rng.standard_normal(N), device='cuda')
        assert float((A.matvec(v)-A2.matvec(v)).abs().max()) <= 1e-11*scale*np.sqrt(N)


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['tiles', 'masks'])
def test_c4_near_field_subtree_shard_against_oracle(mode):
    """C4 at its bench size (disc noRef 7, 98 304 cells, N = 48 769, s = 0.75, eta = 3): the near-field blocks of ONE subtree of the
    bench's own cluster tree -- assembleClusters(Pnear, myRoot=subtree), the share of a rank that owns it (NA:3247-3260) -- against
    the oracle's masked loop nlo_assemble_pairs_masked + cluster-local boundary items over the same cluster pairs: entries at 1e-11
    (cluster-pair tiles, the default and the bench's path; the reference's element-pair masks), integer counters exact for the masks.
    What test_bench_size_shard_against_oracle does for C2: the whole near field (1.4e8 element pairs) would take the oracle a minute,
    a subtree of a few hundred DoFs takes a second."""
    from pynucleus_amd import disc, P1_DoFMap, PHYSICAL, getFractionalKernel, clusters
    from pynucleus_amd.builder import nonlocalBuilder
    from oracle.oracle import OracleProblem
    dm = P1_DoFMap(disc(7), PHYSICAL)
    params = {'target_order': 0.5, 'eta': 3.}
    if mode == 'masks':
        params['maxMasksNNZ'] = 10000000
    b = nonlocalBuilder(dm, getFractionalKernel(2, 0.75), params, zeroExterior=True)
    rp = b.getH2RefinementParams()
    root, Pnear, Pfar = clusters.getNearFieldClusters(b.dm, rp['eta'], rp['minSize'], rp['maxLevels'], refinementType=rp['refinementType'])
    assert sum(len(v) for v in Pfar.values()) > 1000
    node = root
    while node.get_num_dofs() > 400 and len(node.children):
        node = node.children[(node.get_num_dofs() // 7) % len(node.children)]        # some path into the interior of the tree
    assert 50 <= node.get_num_dofs() <= 400
    part = b.assembleClusters(Pnear, myRoot=node)
    inside = np.zeros(b.dm.num_dofs, dtype=bool)
    inside[np.asarray(node.get_dofs())] = True
    mine = [cp for cp in Pnear if inside[cp.n1.dofs[0]]]
    assert len(mine) >= 4
    indptr, indices = clusters.getSparseNearField(b.dm, mine, symmetric=False)
    assert np.array_equal(np.asarray(part.indptr), indptr) and np.array_equal(np.asarray(part.indices), indices)
    pairs, masks = clusters.buildMasksForClusters(b.dm, mine, symmetrize=True)
    bc, bf, bm = clusters.clusterBoundaryItems(b.dm, mine, symmetrize=True)
    data, _, cnt = OracleProblem(b.tables).assemble_clusters(pairs, masks, bc, bf, bm, indptr, indices, False, None)
    got = np.asarray(part.data)
    assert np.abs(got-data).max() < TOL*np.abs(data).max()
    # only rows of the subtree are written
    rows = np.repeat(np.arange(b.dm.num_dofs), np.diff(indptr))
    assert inside[rows].all()
    if mode == 'masks':
        c = part.info['counters']
        assert c['numCellPairs'] == cnt['numCellPairs'] == pairs.shape[0]
        assert c['numAssembledCellPairs'] == cnt['numAssembledCellPairs'] and c['numIntegrations'] == cnt['numIntegrations']


@pytest.mark.gpu
def test_c3_square129_strip_against_oracle():
    """C3 at its bench size (square with 129^2 vertices, 32 768 cells, constant kernel, horizon 0.1): the pairs whose first cell lies in a
    96-cell strip that cuts through 64-cell blocks, assembled by the bench's own route (block tiles within reach of the horizon through
    k_tile_distant<..., FH>, cut pairs in the tile, touching pairs through the far list; pnl_assemble_pairs_in_horizon_range) against
    the oracle's getDense loop over the same cellNo1 range (NA:1280-1285; REMOTE pairs ignored by getPanelType): entries at 1e-11,
    pair and kernel-evaluation counters exact.  Two strips add up to what one range of both assembles (linearity over the split)."""
    import torch
    from pynucleus_amd import uniformSquare, P1_DoFMap, NO_BOUNDARY, getKernel, INDICATOR
    from pynucleus_amd.builder import nonlocalBuilder
    from oracle.oracle import OracleProblem
    delta = 0.1
    dm = P1_DoFMap(uniformSquare(129, 129, 0., 0., 1., 1.), NO_BOUNDARY)
    b = nonlocalBuilder(dm, getKernel(2, kernel=INDICATOR, horizon=delta), {}, zeroExterior=False)
    S = b.getSparse()
    assert S.symmetric
    nc, N = b.mesh.num_cells, b.dm.num_dofs
    c0 = nc//2+5
    c1 = c0+96
    ctx = b.context()
    S._bind()
    data_ptr, diag_ptr = S._ptrs()

    def strip(lo, hi):
        S.data_dev.zero_()
        S.diag_dev.zero_()
        torch.cuda.synchronize()
        ctx.assemble_pairs_in_horizon(data_ptr, diag_ptr, lo, hi)
        ctx.synchronize()
        return S.data_dev.cpu().numpy().copy(), S.diag_dev.cpu().numpy().copy(), ctx.counters()
    data, diag, cnt = strip(c0, c1)
    Aref, cref, _ = OracleProblem(b.tables).get_dense(c0, c1)
    scale = np.abs(Aref).max()
    indptr, indices = np.asarray(S.indptr), np.asarray(S.indices)
    rows = np.repeat(np.arange(N), np.diff(indptr))
    assert np.abs(data-Aref[rows, indices]).max() < TOL*scale
    assert np.abs(diag-np.diag(Aref)).max() < TOL*scale
    # nothing of the oracle's strip lies outside the pattern (strict lower triangle + diagonal, mirrored)
    inside = np.abs(Aref[rows, indices]).sum()*2+np.abs(np.diag(Aref)).sum()
    assert abs(inside-np.abs(Aref).sum()) < 1e-9*inside
    assert cnt['numAssembledCellPairs'] == cref['numAssembledCellPairs'] and cnt['numIntegrations'] == cref['numIntegrations']
    assert cref['numAssembledCellPairs'] > 20000
    del Aref
    mid = c0+37
    d1, g1, k1 = strip(c0, mid)
    d2, g2, k2 = strip(mid, c1)
    assert np.abs(d1+d2-data).max() < 1e-12*scale and np.abs(g1+g2-diag).max() < 1e-12*scale
    assert k1['numAssembledCellPairs']+k2['numAssembledCellPairs'] == cnt['numAssembledCellPairs']
