"""H2 near field (assembleClusters, NA:1663-1964): host logic + oracle on the CPU, GPU parity through the C ABI.

Reference test being mirrored: tests/test_nearField.py (dense matrix vs near-field assembly with cluster pairs covering
all matrix blocks; 2D tolerances epsAbsDense = 5e-3, epsRelDense = 3e-2, lines 32-41)."""
import os
import sys

import numpy as np
import pytest


def _setup(noRef=2, s=0.75, element='P1', zeroExterior=True, domain='disc'):
    from pynucleus_amd import disc, interval, PHYSICAL, dofmapFactory, getFractionalKernel
    from pynucleus_amd.local_matrix import nonlocalTables
    mesh = disc(noRef) if domain == 'disc' else interval(noRef)
    dm = dofmapFactory(element, mesh, PHYSICAL)
    kernel = getFractionalKernel(mesh.dim, s)
    return dm, kernel, nonlocalTables(dm, kernel, {}, zeroExterior)


def _oracle_near(tables, Pnear, symmetric=True, symmetrize=False):
    from pynucleus_amd import clusters
    from oracle.oracle import OracleProblem
    dm = tables.dm
    indptr, indices = clusters.getSparseNearField(dm, Pnear, symmetric=symmetric)
    pairs, masks = clusters.buildMasksForClusters(dm, Pnear, symmetrize=symmetrize)
    bc, bf, bm = clusters.clusterBoundaryItems(dm, Pnear, symmetrize=symmetrize)
    gb = None
    if not tables.zeroExterior:
        c, f, m = clusters.globalBoundaryItems(dm, tables.bcells)
        gb = (c, f, m, -1.)
    data, diag, cnt = OracleProblem(tables).assemble_clusters(pairs, masks, bc, bf, bm, indptr, indices, symmetric, gb)
    return indptr, indices, data, diag, cnt


def _to_dense(N, indptr, indices, data, diag):
    A = np.zeros((N, N))
    rows = np.repeat(np.arange(N), np.diff(indptr))
    A[rows, indices] = data
    if diag is not None:
        A = A+A.T+np.diag(diag)
    return A


def test_tree_and_masks():
    from pynucleus_amd import clusters
    dm, kernel, T = _setup(3)
    root, Pnear, Pfar = clusters.getNearFieldClusters(dm, eta=3., minClusterSize=8)
    nfar = sum(len(v) for v in Pfar.values())
    assert nfar > 0 and len(Pnear) > 0
    # near + far cluster pairs tile the DoF x DoF matrix exactly once
    cover = np.zeros((dm.num_dofs, dm.num_dofs), dtype=np.int32)
    for cp in Pnear:
        cover[np.ix_(cp.n1.dofs, cp.n2.dofs)] += 1
    for lvl in Pfar.values():
        for cp in lvl:
            cover[np.ix_(cp.n1.dofs, cp.n2.dofs)] += 1
    assert (cover == 1).all()
    pairs, masks = clusters.buildMasksForClusters(dm, Pnear)
    assert (pairs[:, 0] <= pairs[:, 1]).all()
    assert np.unique(pairs[:, 0].astype(np.int64)*dm.mesh.num_cells+pairs[:, 1]).shape[0] == pairs.shape[0]
    assert (masks[:, 0] != 0).any() and (masks[:, 1:] == 0).all()          # P1: 21 bits
    # chunked iteration requests every entry exactly once
    chunks = list(clusters.iterMasksForClusters(dm, Pnear, maxNNZ=2000))
    assert len(chunks) > 1
    acc = {}
    for p, m in chunks:
        for (a, b), w in zip(map(tuple, p), m[:, 0]):
            assert acc.get((a, b), 0) & int(w) == 0
            acc[(a, b)] = acc.get((a, b), 0) | int(w)
    assert all(acc[tuple(p)] == int(w) for p, w in zip(pairs, masks[:, 0]))


@pytest.mark.parametrize('symmetric', [True, False])
def test_oracle_covering_cluster_equals_dense(symmetric):
    """one cluster pair (root, root): the near field is the whole operator with the same quadrature as getDense"""
    from pynucleus_amd import clusters
    from oracle.oracle import OracleProblem
    dm, kernel, T = _setup(2, 0.75)
    root, Pnear = clusters.coveringCluster(dm)
    indptr, indices, data, diag, cnt = _oracle_near(T, Pnear, symmetric)
    Anear = _to_dense(dm.num_dofs, indptr, indices, data, diag)
    Adense, _, _ = OracleProblem(T).get_dense()
    assert np.abs(Anear-Adense).max() <= 1e-12*np.abs(Adense).max()


def test_oracle_leaf_pairs_match_dense_reference_tolerance():
    """tests/test_nearField.py: all leaf x leaf pairs vs the dense matrix, abs 5e-3 / rel 3e-2 in 2D"""
    from pynucleus_amd import clusters
    from oracle.oracle import OracleProblem
    dm, kernel, T = _setup(3, 0.75)
    root, Pnear = clusters.allLeafPairs(dm, 2)
    assert len(Pnear) == 16
    indptr, indices, data, diag, cnt = _oracle_near(T, Pnear)
    Anear = _to_dense(dm.num_dofs, indptr, indices, data, diag)
    Adense, _, _ = OracleProblem(T).get_dense()
    err = np.abs(Anear-Adense)
    assert err.max() < 5e-3
    assert np.linalg.norm(Anear-Adense) < 3e-2*np.linalg.norm(Adense)


def test_oracle_regional_no_exterior():
    """zeroExterior=False: the global Omega x Omega^c term is subtracted again (NA:1896-1913)"""
    from pynucleus_amd import clusters
    from oracle.oracle import OracleProblem
    dm, kernel, T = _setup(2, 0.25, zeroExterior=False)
    root, Pnear = clusters.coveringCluster(dm)
    indptr, indices, data, diag, cnt = _oracle_near(T, Pnear)
    Anear = _to_dense(dm.num_dofs, indptr, indices, data, diag)
    Adense, _, _ = OracleProblem(T).get_dense()
    assert np.abs(Anear-Adense).max() <= 1e-11*np.abs(Adense).max()


@pytest.mark.parametrize('zeroExterior', [True, False])
def test_oracle_row_sharded_near_field_sums_to_full(zeroExterior):
    """cluster pairs row-partitioned over ranks, rank-local unsymmetric CSR with symmetrised masks; the sum of
    the rank-local matrices is the near-field matrix"""
    from pynucleus_amd import clusters
    dm, kernel, T = _setup(3, 0.75, zeroExterior=zeroExterior)
    root, Pnear, Pfar = clusters.getNearFieldClusters(dm, eta=3., minClusterSize=8)
    indptr, indices, data, diag, cnt = _oracle_near(T, Pnear, symmetric=False)
    full = _to_dense(dm.num_dofs, indptr, indices, data, None)
    for size in (2, 3):
        parts = clusters.partitionClusterPairs(Pnear, size)
        assert sorted(np.concatenate(parts).tolist()) == list(range(len(Pnear)))
        w = [sum(Pnear[k].n1.cells.shape[0]*Pnear[k].n2.cells.shape[0] for k in p) for p in parts]
        assert min(w) > 0.5*max(w)
        acc = np.zeros_like(full)
        for p in parts:
            ip, ix, d, _, _ = _oracle_near(T, [Pnear[k] for k in p], symmetric=False, symmetrize=True)
            acc += _to_dense(dm.num_dofs, ip, ix, d, None)
        assert np.abs(acc-full).max() <= 1e-12*np.abs(full).max()


# ---- GPU -------------------------------------------------------------------------------------------------------------
def _gpu_builder(noRef, s, element='P1', zeroExterior=True, domain='disc', params=None, mode=None):
    params = dict(params or {})
    if mode is not None:
        params['nearFieldAssembly'] = mode
    from pynucleus_amd import disc, interval, PHYSICAL, dofmapFactory, getFractionalKernel
    from pynucleus_amd.builder import nonlocalBuilder
    mesh = disc(noRef) if domain == 'disc' else interval(noRef)
    dm = dofmapFactory(element, mesh, PHYSICAL)
    return nonlocalBuilder(dm, getFractionalKernel(mesh.dim, s), params or {}, zeroExterior=zeroExterior)


def _gpu_vs_oracle(builder, Pnear, symmetric=True, tol=1e-11, counters=True):
    Anear = builder.assembleClusters(Pnear, forceUnsymmetricMatrix=not symmetric)
    indptr, indices, data, diag, cnt = _oracle_near(builder.tables, Pnear, symmetric)
    assert np.array_equal(Anear.indptr, indptr) and np.array_equal(Anear.indices, indices)
    scale = max(np.abs(data).max() if data.size else 0., np.abs(diag).max() if diag is not None else 0.)
    assert np.abs(Anear.data-data).max() <= tol*scale
    if symmetric:
        assert np.abs(Anear.diagonal-diag).max() <= tol*scale
    got = Anear.info['counters']
    if Anear.info.get('mode') == 'tiles':
        # the tiled decomposition visits ordered element pairs per cluster pair: its counts are its own
        assert got['numAssembledCellPairs'] >= cnt['numAssembledCellPairs']
        return Anear, _to_dense(builder.dm.num_dofs, indptr, indices, data, diag)
    for k in ('numCellPairs', 'numAssembledCellPairs'):
        # chunked assembly visits an element pair once per chunk that requests entries of it (like NA:1786-1791)
        assert (got[k] == cnt[k]) if counters else (got[k] >= cnt[k]), (k, got[k], cnt[k])
    return Anear, _to_dense(builder.dm.num_dofs, indptr, indices, data, diag)


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['tiles', 'masks'])
@pytest.mark.parametrize('noRef,s,element,symmetric', [(3, 0.75, 'P1', True), (3, 0.5, 'P1', False), (2, 0.25, 'P2', True),
                                                       (4, 0.5, 'P1', True)])
def test_gpu_near_field_vs_oracle(noRef, s, element, symmetric, mode):
    """both device decompositions of assembleClusters -- cluster-pair tiles with LDS sub-blocks (default) and the
    reference's element-pair masks -- against the oracle's masked assembly"""
    from pynucleus_amd import clusters
    b = _gpu_builder(noRef, s, element, mode=mode)
    root, Pnear, Pfar = clusters.getNearFieldClusters(b.dm, eta=3., minClusterSize=8)
    assert sum(len(v) for v in Pfar.values()) > 0
    Anear, Aref = _gpu_vs_oracle(b, Pnear, symmetric)
    x = np.random.default_rng(1).standard_normal(b.dm.num_dofs)
    y = Anear*x
    assert np.abs(y-Aref@x).max() <= 1e-12*np.abs(Aref).max()*b.dm.num_dofs
    assert np.abs(Anear.toarray()-Aref).max() <= 1e-11*np.abs(Aref).max()


@pytest.mark.gpu
def test_gpu_dense_after_near_field_on_one_builder():
    """the vertex order of the dense tile kernels is searched on host threads and joined by the first DENSE assembly
    (finalize / tile_order_ready); the near-field path never waits for it.  Either order of the two calls on one builder
    gives the operators a fresh builder gives."""
    import torch
    from pynucleus_amd import clusters
    b = _gpu_builder(4, 0.75, params={'target_order': 0.5})
    h2, Pnear = b.getH2(returnNearField=True)
    near = h2.Anear.toarray() if hasattr(h2.Anear, 'toarray') else None
    A = b.getDense().toarray()
    b2 = _gpu_builder(4, 0.75, params={'target_order': 0.5})
    A2 = b2.getDense().toarray()
    h2b = b2.getH2()
    assert np.abs(A-A2).max() <= 1e-13*np.abs(A2).max()
    if near is not None:
        assert np.abs(near-h2b.Anear.toarray()).max() <= 1e-13*np.abs(near).max()
    x = torch.from_numpy(np.random.default_rng(3).standard_normal(b.dm.num_dofs)).cuda()
    assert np.abs((h2.matvec(x)-h2b.matvec(x)).cpu().numpy()).max() <= 1e-12*np.abs(A2).max()*np.abs(x.cpu().numpy()).max()


@pytest.mark.gpu
def test_gpu_near_field_heavy_boundary_pairs():
    """target_order 3.5 makes the cluster-local Gauss-theorem term expensive (444 point pairs per (cell, facet) pair on average
    at noRef 3): most pairs exceed the 200 point pairs above which k_cluster_boundary hands them to k_boundary_items (one pair
    per wave) instead of integrating them in its lane -- both routes against the oracle's loop (NA:1842-1889)"""
    from pynucleus_amd import clusters
    b = _gpu_builder(3, 0.75, params={'target_order': 3.5}, mode='tiles')
    root, Pnear, Pfar = clusters.getNearFieldClusters(b.dm, eta=3., minClusterSize=8)
    Anear, Aref = _gpu_vs_oracle(b, Pnear, True)
    c = Anear.info['counters']
    assert c['numBoundaryIntegrations'] > 200*c['numBoundaryPairs'] > 0
    assert np.abs(Anear.toarray()-Aref).max() <= 1e-11*np.abs(Aref).max()


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['tiles', 'masks'])
def test_gpu_covering_cluster_equals_gpu_dense(mode):
    from pynucleus_amd import clusters
    b = _gpu_builder(3, 0.5, params={'target_order': 0.5}, mode=mode)
    root, Pnear = clusters.coveringCluster(b.dm)
    Anear, Aref = _gpu_vs_oracle(b, Pnear)
    Adense = b.getDense().toarray()
    assert np.abs(Anear.toarray()-Adense).max() <= 1e-11*np.abs(Adense).max()


@pytest.mark.gpu
def test_gpu_near_field_chunked_and_regional():
    from pynucleus_amd import clusters
    b = _gpu_builder(3, 0.25, zeroExterior=False, params={'maxMasksNNZ': 3000})
    root, Pnear = clusters.allLeafPairs(b.dm, 2)
    Anear, Aref = _gpu_vs_oracle(b, Pnear, counters=False)
    b2 = _gpu_builder(3, 0.25, zeroExterior=False, mode='tiles')
    A2, _ = _gpu_vs_oracle(b2, clusters.allLeafPairs(b2.dm, 2)[1])
    assert A2.info['mode'] == 'tiles' and np.abs(A2.toarray()-Aref).max() <= 1e-11*np.abs(Aref).max()
    Adense = b.getDense().toarray()
    assert np.abs(Anear.toarray()-Adense).max() < 5e-3
    assert np.linalg.norm(Anear.toarray()-Adense) < 3e-2*np.linalg.norm(Adense)


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['tiles', 'masks'])
def test_gpu_near_field_1d(mode):
    from pynucleus_amd import clusters
    b = _gpu_builder(5, 0.75, domain='interval', mode=mode)
    root, Pnear, Pfar = clusters.getNearFieldClusters(b.dm, eta=3., minClusterSize=4)
    _gpu_vs_oracle(b, Pnear)


@pytest.mark.gpu
def test_getH2_returns_near_field():
    b = _gpu_builder(4, 0.75, params={'eta': 3., 'minClusterSize': 16})
    h2, Pnear = b.getH2(returnNearField=True)
    assert h2.Anear.nnz > 0 and len(Pnear) > 0 and h2.plan.far.shape[0] > 0


@pytest.mark.gpu
@pytest.mark.parametrize('zeroExterior', [True, False])
def test_gpu_getDiagonal_getEntry(zeroExterior):
    """getDiagonal / getEntry (NA:2269-2289, 1538-1661) against the dense operator: same tolerance as the reference's
    tests/test_fracLapl.py:103-111 (rtol 2e-3), and exactly against the oracle's cluster assembly"""
    from pynucleus_amd import clusters
    b = _gpu_builder(3, 0.75, zeroExterior=zeroExterior)
    dm = b.dm
    d = b.getDiagonal().diagonal
    if zeroExterior:
        Pnear = clusters.singleDoFClusters(dm)
        indptr, indices, data, diag, cnt = _oracle_near(b.tables, Pnear)
        assert np.abs(d-diag).max() <= 1e-11*np.abs(diag).max()
        Adense = b.getDense().toarray()
        assert np.allclose(d, np.diag(Adense), rtol=2e-3)
        for I, J in [(0, 0), (5, 5), (3, 4), (10, 90), (dm.num_dofs-1, 0)]:
            e = b.getEntry(I, J)
            assert abs(e-Adense[I, J]) <= 2e-3*abs(Adense[I, J])+1e-6, (I, J, e, Adense[I, J])
    else:
        # no exterior term at all (NA:1599: only (supp)^2): strictly smaller than the zeroExterior diagonal
        b2 = _gpu_builder(3, 0.75, zeroExterior=True)
        assert (d < b2.getDiagonal().diagonal).all() and (d > 0).all()


def _dist_near_worker(rank, world, port, out):
    import os
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)                       # one-GPU box: both ranks share the card, collectives over gloo
    from pynucleus_amd import disc, P1_DoFMap, PHYSICAL, getFractionalKernel, clusters
    from pynucleus_amd.builder import nonlocalBuilder
    mesh = disc(3)
    dm = P1_DoFMap(mesh, PHYSICAL)
    b = nonlocalBuilder(dm, getFractionalKernel(2, 0.75), {'eta': 3., 'minClusterSize': 8}, zeroExterior=True, comm=True)
    op, Pnear, root = b.getH2(returnNearField=True, returnTree=True)
    x = np.linspace(-1., 1., dm.num_dofs)
    y = op.matvec(x)                                # near field (row-sharded) + far field (pairs dealt over the ranks), all-reduced
    indptr, indices, data, diag, cnt = _oracle_near(b.tables, Pnear, symmetric=False)
    Aref = _to_dense(dm.num_dofs, indptr, indices, data, None)
    e2 = float(np.abs(op.toarray()-Aref).max()/np.abs(Aref).max())
    nfar_all = sum(len(v) for v in op.Pfar.values())
    nfar_mine = torch.tensor([float(op.far.plan.far.shape[0])], dtype=torch.float64)
    dist.all_reduce(nfar_mine)
    assert 0 < op.far.plan.far.shape[0] < nfar_all and int(nfar_mine.item()) == nfar_all     # every admissible pair on one rank
    if rank == 0:
        from pynucleus_amd.quadrature import simplexXiaoGimbutas
        from oracle import h2_oracle
        m = op.far.plan.m
        Aref = Aref+h2_oracle.far_field_dense(dm, b.kernel, root, op.Pfar, m, simplexXiaoGimbutas(m+2, 2, 2))
    e1 = float(np.abs(y-Aref@x).max()/np.abs(Aref@x).max()) if rank == 0 else 0.
    if rank == 0:
        out.put((e1, e2, op.local.nnz, indices.shape[0]))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_row_sharded_near_field():
    """BASELINE configs[3] at world size 2 (gloo, both ranks on the one GPU of the box): rank-local near-field blocks,
    matvec = local SpMV + all-reduce of the N-vector"""
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = 29700+os.getpid() % 2000
    procs = [ctx.Process(target=_dist_near_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    e1, e2, nnz_local, nnz_full = out.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert 0 < nnz_local < nnz_full
    assert e1 < 1e-11 and e2 < 1e-11


def test_pattern_on_a_torch_device_equals_the_numpy_one():
    """getSparseNearField(device=...) (keys generated and sorted with torch, on the GPU in production) gives the arrays of
    the numpy route, SSS and CSR"""
    import torch
    from pynucleus_amd import disc, P1_DoFMap, PHYSICAL
    from pynucleus_amd import clusters
    dm = P1_DoFMap(disc(4), PHYSICAL)
    root, Pnear, Pfar = clusters.getNearFieldClusters(dm, 3., 16, 200)
    assert len(Pnear) > 10
    for symmetric in (True, False):
        a = clusters.getSparseNearField(dm, Pnear, symmetric)
        b = clusters.getSparseNearField(dm, Pnear, symmetric, device=torch.device('cpu'))
        assert b[0].dtype == b[1].dtype == torch.int32          # tensors on the device they were built on
        b = (b[0].numpy(), b[1].numpy())
        assert a[0].dtype == b[0].dtype == np.int32 and a[1].dtype == b[1].dtype == np.int32
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['tiles', 'masks'])
def test_gpu_assemble_clusters_by_subtree(mode):
    """assembleClusters(Pnear, myRoot=subtree) (NA:3247-3260, 1697-1712): the near field of the rank that owns `subtree` -- the
    cluster pairs whose row cluster lies in it, complete blocks in unsymmetric CSR; over the children of the root the parts
    add up to the near field, each part equals the oracle's blocks of its pairs"""
    from pynucleus_amd import clusters
    b = _gpu_builder(3, 0.75, zeroExterior=True)
    if mode == 'masks':
        b.params['maxMasksNNZ'] = 10000000
    dm = b.dm
    rp = dict(b.getH2RefinementParams(), minSize=8)
    root, Pnear, Pfar = clusters.getNearFieldClusters(dm, rp['eta'], rp['minSize'], rp['maxLevels'])
    full = b.assembleClusters(Pnear, forceUnsymmetricMatrix=True).toarray()
    scale = np.abs(full).max()
    total = np.zeros_like(full)
    kids = root.children
    assert len(kids) >= 2
    seen = 0
    for kid in kids:
        part = b.assembleClusters(Pnear, myRoot=kid)
        mine = [cp for cp in Pnear if cp.n1.dofs[0] in set(kid.dofs.tolist())]
        seen += len(mine)
        P = part.toarray()
        rows = np.zeros(dm.num_dofs, dtype=bool)
        rows[kid.dofs] = True
        assert np.abs(P[~rows]).max() == 0.                       # only rows of the subtree
        indptr, indices, data, diag, cnt = _oracle_near(b.tables, mine, symmetric=False, symmetrize=True)
        ref = _to_dense(dm.num_dofs, indptr, indices, data, None)
        assert np.abs(P-ref).max() <= 1e-11*scale
        total += P
    assert seen == len(Pnear)
    assert np.abs(total-full).max() <= 1e-12*scale


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['dense_P1', 'dense_P2', 'near_P1', 'near_P1_unsym', 'sparse_P1', 'sparse_fractional'])
def test_gpu_tile_loops_with_two_workgroups_against_the_oracle(case):
    """The tile kernels are persistent: a workgroup walks many tiles (the cell data of the next tile is staged before the flush of the
    current one, the flush zeroes the accumulators it has read, the barriers order the LDS only).  At test sizes every workgroup
    gets ONE tile; the option PNL_TILE_WGS caps the grid at two workgroups, so the tile loops themselves -- dense, cluster tiles
    of the near field, finite-horizon tiles -- are compared with the oracle entry by entry."""
    from oracle.oracle import OracleProblem
    from pynucleus_amd import _lib, clusters
    sys.path.insert(0, os.path.dirname(__file__))
    try:
        _lib.set_option('PNL_TILE_WGS', 2)
        if case.startswith('dense'):
            b = _gpu_builder(4 if case == 'dense_P1' else 3, 0.5, element=case[-2:], params={'target_order': 0.5})
            A = b.getDense()
            Aref, cnt, _ = OracleProblem(b.tables).get_dense()
            for key in ('numAssembledCellPairs', 'numIntegrations', 'orders', 'singular'):
                assert A.info['counters'][key] == cnt[key], key
            assert np.abs(A.toarray()-Aref).max() <= 1e-11*np.abs(Aref).max()
        elif case.startswith('near'):
            b = _gpu_builder(4, 0.75, mode='tiles')
            root, Pnear, Pfar = clusters.getNearFieldClusters(b.dm, eta=3., minClusterSize=8)
            Anear, Aref = _gpu_vs_oracle(b, Pnear, symmetric=(case == 'near_P1'))
            assert np.abs(Anear.toarray()-Aref).max() <= 1e-11*np.abs(Aref).max()
        else:
            from test_finite_horizon import _gpu_sparse
            b = _gpu_sparse(17, 0.2, 'indicator') if case == 'sparse_P1' else _gpu_sparse(9, 0.45, 'fractional', s=0.4)
            A = b.getSparse()
            Aref, cnt, _ = OracleProblem(b.tables).get_dense()
            assert np.abs(A.toarray()-Aref).max() <= 1e-11*np.abs(Aref).max()
            assert A.info['counters']['numAssembledCellPairs'] == cnt['numAssembledCellPairs']
    finally:
        _lib.set_option('PNL_TILE_WGS', None)


@pytest.mark.gpu
@pytest.mark.parametrize('s', [0.5, 0.75, 0.4])
def test_gpu_uniform_tiles_structured_and_generic_evaluators_agree(s):
    """The uniform-order tile kernels of P1 evaluate the symmetric 3- and 6-point rules through their orbit structure (w phi_b(y_j) =
    A_o + B_o delta: cross blocks from row / column / orbit sums of the kernel values); the option PNL_UNI_GENERIC sends the same tiles
    through the evaluator for arbitrary rules.  Same matrix to rounding, same pair counts, and both orders occur in uniform tiles."""
    from pynucleus_amd import _lib
    b = _gpu_builder(6, s, params={'target_order': 0.5})           # 24,576 cells: uniform tiles of orders 2 and 3
    A = b.getDense()
    D1, cnt1 = A.toarray().copy(), A.info['counters']
    del A
    uni = cnt1.get('uniformTilePairsByOrder', {})
    assert uni.get(2, 0) > 0 and uni.get(3, 0) > 0, uni
    try:
        _lib.set_option('PNL_UNI_GENERIC', 1)
        b2 = _gpu_builder(6, s, params={'target_order': 0.5})
        A2 = b2.getDense()
        D2, cnt2 = A2.toarray(), A2.info['counters']
    finally:
        _lib.set_option('PNL_UNI_GENERIC', None)
    assert cnt1['numAssembledCellPairs'] == cnt2['numAssembledCellPairs'] and cnt1['orders'] == cnt2['orders']
    assert np.abs(D1-D2).max() <= 1e-13*np.abs(D1).max()
    assert np.abs(D1-D2).max() > 0.                       # two evaluators, not one
