"""The C++ planner of the H2 / near-field assembly (csrc/pnl_plan.hip: cluster tree, admissibility recursion, node cells, tile
work lists, transfer matrices -- clusterMethodCy.pyx:354-663, 4046-4136, 2004-2073) against the numpy implementation in
clusters.py / h2.py that round 1 used: same tree, same pairs in the same order, identical work lists.  Host code only: runs
without a GPU."""
import os
import numpy as np
import pytest
from pynucleus_amd import disc, interval, P1_DoFMap, P2_DoFMap, PHYSICAL
from pynucleus_amd import clusters
from pynucleus_amd.h2 import h2Plan

PLAN_ARRAYS = ('node_chunk_off', 'chunk_cells', 'chunk_ndof', 'chunk_dofs', 'chunk_slot', 'tile_chunkA', 'tile_chunkB', 'tile_pair',
               'tile_flags', 'tile_dslotA', 'tile_dslotB', 'd_cell', 'd_pair', 'pair_foff', 'fvid', 'bt_slot', 'bt_cell', 'bt_facet',
               'pair_nodes', 'node_off', 'node_dofs')


def _both(dm, minSize, tile, m=4, eta=3.):
    out = {}
    old = os.environ.get('PNL_PLAN')
    try:
        for mode in ('numpy', 'native'):
            os.environ['PNL_PLAN'] = mode
            root, Pnear, Pfar = clusters.getNearFieldClusters(dm, eta, minSize, 200)
            out[mode] = (root, Pnear, Pfar, clusters.nearFieldPlan(dm, Pnear, tile=tile), h2Plan(dm, root, Pfar, m))
    finally:
        if old is None:
            os.environ.pop('PNL_PLAN', None)
        else:
            os.environ['PNL_PLAN'] = old
    return out['numpy'], out['native']


@pytest.mark.parametrize('case', ['disc5_P1', 'disc4_P2', 'interval_P1', 'disc3_P1_eta1'])
def test_native_planner_equals_numpy(case):
    if case == 'disc5_P1':
        dm, minSize, tile, eta = P1_DoFMap(disc(5), PHYSICAL), 16, 64, 3.
    elif case == 'disc4_P2':
        dm, minSize, tile, eta = P2_DoFMap(disc(4), PHYSICAL), 24, 32, 3.
    elif case == 'interval_P1':
        dm, minSize, tile, eta = P1_DoFMap(interval(8), PHYSICAL), 8, 64, 1.
    else:
        dm, minSize, tile, eta = P1_DoFMap(disc(3), PHYSICAL), 12, 64, 1.
    (r0, n0, f0, p0, h0), (r1, n1, f1, p1, h1) = _both(dm, minSize, tile, eta=eta)
    # near-field pairs: same clusters in the same order
    assert len(n0) == len(n1) and len(n0) > 0
    for a, b in zip(n0, n1):
        assert np.array_equal(a.n1.dofs, b.n1.dofs) and np.array_equal(a.n2.dofs, b.n2.dofs)
        assert np.array_equal(a.n1.cells, b.n1.cells) and np.array_equal(a.cellsInter, b.cellsInter) and np.array_equal(a.cellsUnion, b.cellsUnion)
        assert np.allclose(a.n1.box, b.n1.box, rtol=0, atol=0)
    # far-field pairs per level
    assert sorted(f0) == sorted(f1)
    for lvl in f0:
        assert len(f0[lvl]) == len(f1[lvl])
        for a, b in zip(f0[lvl], f1[lvl]):
            assert np.array_equal(a.n1.dofs, b.n1.dofs) and np.array_equal(a.n2.dofs, b.n2.dofs)
    # tile work lists
    for name in PLAN_ARRAYS:
        x, y = getattr(p0, name), getattr(p1, name)
        assert x.shape == y.shape and np.array_equal(x, y), name
    for s in range(3):
        assert np.array_equal(p0.sing_items[s], p1.sing_items[s])
    assert p0.nU == p1.nU and p0.num_dslots == p1.num_dslots
    # far-field plan: the node numbering differs (depth first / breadth first), the content does not
    assert h0.box.shape == h1.box.shape and h0.far.shape == h1.far.shape
    k0 = {tuple(np.round(h0.box[k].ravel(), 12)): k for k in range(h0.box.shape[0])}
    for k in range(h1.box.shape[0]):
        j = k0[tuple(np.round(h1.box[k].ravel(), 12))]
        assert np.abs(h0.transfer[j]-h1.transfer[k]).max() <= 1e-13
        assert h0.level[j] == h1.level[k]


def test_native_tree_partition_properties():
    """every level of the tree partitions the DoFs; children are the halves of a median split along the longest box edge"""
    dm = P1_DoFMap(disc(4), PHYSICAL)
    root, Pnear, Pfar = clusters.getNearFieldClusters(dm, 3., 16, 200)
    assert isinstance(root, clusters.native_node)
    level = [root]
    while level:
        d = np.concatenate([n.dofs for n in level])
        assert np.unique(d).shape[0] == d.shape[0]
        nxt = []
        for n in level:
            if n.is_leaf:
                continue
            a, b = n.children
            assert np.array_equal(np.sort(np.concatenate([a.dofs, b.dofs])), n.dofs)
            assert abs(a.dofs.shape[0]-b.dofs.shape[0]) <= max(2, n.dofs.shape[0]//8)
            nxt += [a, b]
        level = nxt
    # the near and far pairs cover every DoF pair exactly once
    N = dm.num_dofs
    cover = np.zeros((N, N), dtype=np.int32)
    for cp in Pnear:
        cover[np.ix_(cp.n1.dofs, cp.n2.dofs)] += 1
    for lvl in Pfar:
        for cp in Pfar[lvl]:
            cover[np.ix_(cp.n1.dofs, cp.n2.dofs)] += 1
    assert cover.min() == 1 and cover.max() == 1


@pytest.mark.parametrize('case', ['square_P1', 'square_P2', 'interval_P1', 'square_P1_unsym'])
def test_native_horizon_pattern_equals_sparse_products(case):
    """pnl_horizon_pattern against G = M Q M^T formed with scipy (what builder.getSparse did in round 1)"""
    import ctypes as C
    import scipy.sparse as sp
    from scipy.spatial import cKDTree
    from pynucleus_amd import uniformSquare, NO_BOUNDARY, dofmapFactory, _lib
    sym = True
    if case == 'square_P1':
        mesh, el, delta = uniformSquare(17), 'P1', 0.2
    elif case == 'square_P2':
        mesh, el, delta = uniformSquare(9), 'P2', 0.3
    elif case == 'interval_P1':
        mesh, el, delta = interval(6), 'P1', 0.11
    else:
        mesh, el, delta, sym = uniformSquare(13), 'P1', 0.17, False
    dm = dofmapFactory(el, mesh, NO_BOUNDARY)
    N, nc, dpe, nv = dm.num_dofs, mesh.num_cells, dm.dofs_per_element, mesh.num_vertices
    rows = np.repeat(np.arange(nc), dpe)
    d = np.asarray(dm.dofs).reshape(-1)
    m = d >= 0
    Cm = sp.csr_matrix((np.ones(int(m.sum()), dtype=np.int32), (rows[m], d[m])), shape=(nc, N))
    vp = cKDTree(mesh.vertices).query_pairs(delta*(1.+1e-9), output_type='ndarray')
    Q = sp.csr_matrix((np.ones(vp.shape[0], dtype=np.int32), (vp[:, 0], vp[:, 1])), shape=(nv, nv))
    Q = Q+Q.T+sp.identity(nv, dtype=np.int32, format='csr')
    B = sp.csr_matrix((np.ones(nc*mesh.cells.shape[1], dtype=np.int32),
                       (np.repeat(np.arange(nc), mesh.cells.shape[1]), np.asarray(mesh.cells).reshape(-1))), shape=(nc, nv))
    M = (Cm.T @ B).tocsr()
    M.data[:] = 1
    G = ((M @ Q) @ M.T).tocsr()
    if sym:
        G = sp.tril(G, k=-1, format='csr')
    G.sort_indices()
    L = _lib.load()
    verts = np.ascontiguousarray(mesh.vertices, dtype=np.float64)
    mcells = np.ascontiguousarray(mesh.cells, dtype=np.int32)
    dofs32 = np.ascontiguousarray(dm.dofs, dtype=np.int32)
    h = C.c_void_p()
    assert L.pnl_horizon_pattern(mesh.dim, nv, verts.ctypes.data, nc, mcells.ctypes.data, dpe, N, dofs32.ctypes.data, float(delta), int(sym),
                                 C.byref(h)) == 0
    indptr = np.zeros(N+1, dtype=np.int32)
    indices = np.zeros(int(L.pnl_pattern_nnz(h)), dtype=np.int32)
    L.pnl_pattern_get(h, indptr.ctypes.data, indices.ctypes.data)
    L.pnl_pattern_destroy(h)
    assert np.array_equal(indptr, G.indptr) and np.array_equal(indices, G.indices)


@pytest.mark.parametrize('symmetric', [True, False])
def test_native_near_pattern_equals_the_numpy_one(symmetric, monkeypatch):
    """pnl_near_pattern (bitmap per leaf over the cluster pairs of the native tree) against the sorted-keys construction"""
    from pynucleus_amd import disc, P1_DoFMap, P2_DoFMap, PHYSICAL, clusters
    for DoFMap, noRef in ((P1_DoFMap, 4), (P2_DoFMap, 3)):
        dm = DoFMap(disc(noRef), PHYSICAL)
        root, Pnear, Pfar = clusters.getNearFieldClusters(dm, 3., 8, 200)
        assert isinstance(Pnear[0].n1, clusters.native_node)
        a = clusters.getSparseNearField(dm, Pnear, symmetric)
        monkeypatch.setenv('PNL_PLAN', 'numpy')
        b = clusters.getSparseNearField(dm, Pnear, symmetric)
        monkeypatch.delenv('PNL_PLAN')
        assert a[0].dtype == b[0].dtype == np.int32 and a[1].dtype == b[1].dtype == np.int32
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        # a subset of the pairs (what a rank owns) and the all-leaf cover
        sub = Pnear[::3]
        a = clusters.getSparseNearField(dm, sub, symmetric)
        monkeypatch.setenv('PNL_PLAN', 'numpy')
        b = clusters.getSparseNearField(dm, sub, symmetric)
        monkeypatch.delenv('PNL_PLAN')
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_pattern_nnz_guard():
    """the CSR row pointer is int32; a pattern beyond the limit must be refused (PNL_ERR_UNSUPPORTED), not wrapped.
    pnl_pattern_set_max_nnz lowers the limit so that a small mesh forces the guard in both builders."""
    import ctypes as C
    from pynucleus_amd import _lib, uniformSquare, disc, P1_DoFMap, NO_BOUNDARY, PHYSICAL
    from pynucleus_amd import clusters
    L = _lib.load()
    mesh = uniformSquare(9)
    dm = P1_DoFMap(mesh, NO_BOUNDARY)
    verts = np.ascontiguousarray(mesh.vertices, dtype=np.float64)
    cells = np.ascontiguousarray(mesh.cells, dtype=np.int32)
    dofs = np.ascontiguousarray(dm.dofs, dtype=np.int32)

    def horizon():
        h = C.c_void_p()
        rc = L.pnl_horizon_pattern(2, mesh.num_vertices, verts.ctypes.data, mesh.num_cells, cells.ctypes.data, 3, dm.num_dofs,
                                   dofs.ctypes.data, 0.3, 0, C.byref(h))
        nnz = int(L.pnl_pattern_nnz(h)) if rc == 0 else -1
        if rc == 0:
            L.pnl_pattern_destroy(h)
        return rc, nnz
    rc, nnz = horizon()
    assert rc == 0 and nnz > 100
    old = L.pnl_pattern_set_max_nnz(nnz-1)
    try:
        assert old == 2**31-1
        assert horizon()[0] == _lib.PNL_ERR_UNSUPPORTED
        L.pnl_pattern_set_max_nnz(nnz)
        assert horizon() == (0, nnz)
        # near-field pattern of a cluster tree
        dm2 = P1_DoFMap(disc(3), PHYSICAL)
        root, Pnear, Pfar = clusters.getNearFieldClusters(dm2, 3., 8, 200)
        L.pnl_pattern_set_max_nnz(2**31-1)
        indptr, indices = clusters.getSparseNearField(dm2, Pnear, symmetric=False)
        n2 = int(np.asarray(indices).shape[0])
        L.pnl_pattern_set_max_nnz(n2-1)
        with pytest.raises(_lib.PnlError, match='2\\^31'):
            clusters.getSparseNearField(dm2, Pnear, symmetric=False)
    finally:
        L.pnl_pattern_set_max_nnz(2**31-1)
    assert L.pnl_pattern_set_max_nnz(0) == 2**31-1          # out-of-range values leave the limit unchanged


@pytest.mark.parametrize('refType', ['MEDIAN', 'GEOMETRIC', 'BARYCENTER'])
def test_refinement_types(refType):
    """refinementType of the cluster tree (NA:3033-3040, tree_node.refine CM:354-663): MEDIAN / GEOMETRIC / BARYCENTER split points;
    whatever the split, the leaves partition the DoFs, children partition their parent, near + far pairs tile the DoF x DoF
    matrix exactly once, and every far pair is admissible (eta dist >= max diam)"""
    dm = P1_DoFMap(disc(4), PHYSICAL)
    root, Pnear, Pfar = clusters.getNearFieldClusters(dm, 3., 12, 200, refinementType=refType)
    N = dm.num_dofs
    cover = np.zeros((N, N), dtype=np.int32)
    for cp in Pnear:
        cover[np.ix_(cp.n1.dofs, cp.n2.dofs)] += 1
    nfar = 0
    for lvl, pairs in Pfar.items():
        for cp in pairs:
            cover[np.ix_(cp.n1.dofs, cp.n2.dofs)] += 1
            nfar += 1
            d = clusters.distBoxes(cp.n1.box, cp.n2.box)
            diam = max(np.linalg.norm(cp.n1.box[:, 1]-cp.n1.box[:, 0]), np.linalg.norm(cp.n2.box[:, 1]-cp.n2.box[:, 0]))
            assert 3.*d >= diam-1e-12
    assert (cover == 1).all() and nfar > 0

    def walk(n):
        if n.is_leaf:
            return [n]
        kids = n.children
        assert sorted(np.concatenate([k.dofs for k in kids]).tolist()) == sorted(n.dofs.tolist())
        return [l for k in kids for l in walk(k)]
    leaves = walk(root)
    assert sorted(np.concatenate([l.dofs for l in leaves]).tolist()) == list(range(N))
    if refType == 'GEOMETRIC':
        # the root is halved along its longest edge
        ax = int(np.argmax(root.box[:, 1]-root.box[:, 0]))
        mid = 0.5*(root.box[ax, 0]+root.box[ax, 1])
        boxes, _ = clusters.getDoFBoxesAndCells(dm)
        c = 0.5*(boxes[:, ax, 0]+boxes[:, ax, 1])
        a, b = root.children
        assert (c[a.dofs] < mid).all() and (c[b.dofs] >= mid).all()
    if refType == 'BARYCENTER':
        ax = int(np.argmax(root.box[:, 1]-root.box[:, 0]))
        boxes, _ = clusters.getDoFBoxesAndCells(dm)
        c = 0.5*(boxes[:, ax, 0]+boxes[:, ax, 1])
        a, b = root.children
        # (the disc is symmetric: coordinates within rounding of the mean may fall on either side)
        assert (c[a.dofs] < c.mean()+1e-12).all() and (c[b.dofs] >= c.mean()-1e-12).all() and min(len(a.dofs), len(b.dofs)) > 0.3*N


# ---- planner on the device (csrc/pnl_plan_dev.hip): refinement and admissibility as level-synchronous sweeps ---------------------
@pytest.mark.gpu
@pytest.mark.parametrize('domain,noRef,element,eta,minSize,refinementType',
                         [('disc', 4, 'P1', 3., 8, 'MEDIAN'), ('disc', 6, 'P1', 3., 64, 'MEDIAN'), ('disc', 5, 'P2', 1.5, 24, 'MEDIAN'),
                          ('square', 5, 'P1', 3., 8, 'MEDIAN'), ('square', 5, 'P1', 2., 5, 'GEOMETRIC'), ('interval', 9, 'P1', 1., 4, 'MEDIAN'),
                          ('disc', 5, 'P1', 3., 16, 'GEOMETRIC'), ('disc', 7, 'P1', 3., 64, 'MEDIAN')])
def test_device_planner_equals_host_planner(domain, noRef, element, eta, minSize, refinementType):
    """tree_node.refine (clusterMethodCy.pyx:354-663) and getAdmissibleClusters (:4046-4136) on the device -- the same
    tree (node ranges, parents, children, levels, boxes, DoF permutation) and the same near / far lists IN THE SAME ORDER as the host
    planner, entry for entry; the structured square puts cluster pairs exactly on the admissibility threshold (both planners evaluate
    the box metrics without FMA contraction), GEOMETRIC splits leave unbalanced trees"""
    from pynucleus_amd import disc, interval, uniformSquare, PHYSICAL, dofmapFactory, clusters
    mesh = {'disc': lambda: disc(noRef), 'interval': lambda: interval(noRef), 'square': lambda: uniformSquare(2**noRef+1, None, -1., -1., 1., 1.)}[domain]()
    dm = dofmapFactory(element, mesh, PHYSICAL)
    H = clusters._nativeTree(dm, eta, minSize, 200, 1, refinementType=refinementType, planner='host')
    D = clusters._nativeTree(dm, eta, minSize, 200, 1, refinementType=refinementType, planner='device')
    assert H.planner == 'host' and D.planner == 'device'
    for name in ('range', 'parent', 'children', 'level', 'perm'):
        assert np.array_equal(getattr(H, name), getattr(D, name)), name
    assert np.array_equal(H.box, D.box)
    assert H.near.shape[0] > 0 and np.array_equal(H.near, D.near)
    assert np.array_equal(H.far, D.far)
    if domain != 'interval':
        assert H.far.shape[0] > 0


@pytest.mark.gpu
def test_device_planner_feeds_getH2():
    """getH2 with params['planner'] = 'device' and 'host': the same operator"""
    from pynucleus_amd import disc, P1_DoFMap, PHYSICAL, getFractionalKernel
    from pynucleus_amd.builder import nonlocalBuilder
    out = []
    for planner in ('device', 'host'):
        dm = P1_DoFMap(disc(4), PHYSICAL)
        b = nonlocalBuilder(dm, getFractionalKernel(2, 0.75), {'eta': 3., 'minClusterSize': 8, 'planner': planner}, zeroExterior=True)
        H = b.getH2()
        x = np.cos(0.37*np.arange(dm.num_dofs))
        out.append((H.matvec(x), np.asarray(H.Anear.indptr), np.asarray(H.Anear.indices), H.info['numFarPairs']))
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2]) and out[0][3] == out[1][3]
    assert np.abs(out[0][0]-out[1][0]).max() < 1e-13*np.abs(out[1][0]).max()


@pytest.mark.parametrize('domain,N,horizon', [('square', 17, 0.3), ('square', 33, 0.15), ('interval', 8, 0.2), ('interval', 8, 1.0)])
def test_native_planner_with_a_finite_horizon_equals_numpy(domain, N, horizon):
    """getAdmissibleClusters with the horizon of the l2 ball (clusterMethodCy.pyx:4069-4090, 4131-4135): cluster pairs beyond the horizon
    are dropped, pairs it may cut stay in the near field, near-field children merge only inside the horizon -- the C++ planner
    (pnl_tree_build_horizon) and the Python restatement give the same ordered lists; every admissible pair lies inside the horizon and
    no near / far pair lies beyond it"""
    from pynucleus_amd import uniformSquare, NO_BOUNDARY
    dm = P1_DoFMap(uniformSquare(N) if domain == 'square' else interval(N), NO_BOUNDARY)
    old = os.environ.get('PNL_PLAN')
    out = {}
    try:
        for mode in ('numpy', 'native'):
            os.environ['PNL_PLAN'] = mode
            out[mode] = clusters.getNearFieldClusters(dm, 3., 4, 200, horizon=horizon)
    finally:
        if old is None:
            os.environ.pop('PNL_PLAN', None)
        else:
            os.environ['PNL_PLAN'] = old
    (r0, n0, f0), (r1, n1, f1) = out['numpy'], out['native']
    assert len(n0) == len(n1) > 0
    for a, b in zip(n0, n1):
        assert np.array_equal(a.n1.dofs, b.n1.dofs) and np.array_equal(a.n2.dofs, b.n2.dofs)
    assert sorted(f0) == sorted(f1)
    for lvl in f0:
        assert [(tuple(a.n1.dofs), tuple(a.n2.dofs)) for a in f0[lvl]] == [(tuple(b.n1.dofs), tuple(b.n2.dofs)) for b in f1[lvl]]
        for cp in f1[lvl]:
            assert clusters.maxDistBoxes(np.asarray(cp.n1.box), np.asarray(cp.n2.box)) < horizon
    for cp in n1:
        assert clusters.distBoxes(np.asarray(cp.n1.box), np.asarray(cp.n2.box)) <= horizon
    if domain == 'interval':
        assert sum(len(v) for v in f1.values()) > 0
    # the blocks of the near and far pairs tile what interacts: every pair of DoFs closer than the horizon is covered exactly once
    cover = np.zeros((dm.num_dofs, dm.num_dofs), dtype=np.int32)
    for cp in list(n1)+[c for v in f1.values() for c in v]:
        cover[np.ix_(np.asarray(cp.n1.dofs), np.asarray(cp.n2.dofs))] += 1
    assert cover.max() == 1
    c = dm.getDoFCoordinates()
    d = np.sqrt(((c[:, None, :]-c[None, :, :])**2).sum(axis=2))
    assert cover[d < horizon-2.*dm.mesh.h].min() == 1
