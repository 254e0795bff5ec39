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
