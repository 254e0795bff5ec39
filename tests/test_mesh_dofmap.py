"""Mesh and DoF-map mirrors: sizes quoted in SURVEY.md section 8 for the reference's own disc mesh, numbering invariants."""
import numpy as np
import pytest
from pynucleus_amd import disc, interval, uniformSquare, P1_DoFMap, P2_DoFMap, PHYSICAL, NO_BOUNDARY


@pytest.mark.parametrize('noRef,nc,N', [(0, 6, 1), (3, 384, 169), (5, 6144, 2977), (6, 24576, 12097)])
def test_disc_sizes(noRef, nc, N):
    mesh = disc(noRef)
    dm = P1_DoFMap(mesh, PHYSICAL)
    assert mesh.num_cells == nc and dm.num_dofs == N
    assert mesh.boundaryEdges.shape[0] == 6*2**noRef
    if noRef:
        assert np.isclose(mesh.diam, 2*np.sqrt(2))             # bounding-box diagonal, H0 = diam/sqrt(8) = 1
    r = np.linalg.norm(mesh.vertices[mesh.boundaryVertices], axis=1)
    assert np.allclose(r, 1.)                                  # radial projection keeps the boundary on the circle
    assert np.isclose(mesh.volVector.sum(), mesh.volume) and mesh.volume < np.pi
    assert (mesh.hVector >= mesh.hmin-1e-15).all() and np.isclose(mesh.hVector.max(), mesh.h)


def test_disc_orientation_and_boundary_edges():
    mesh = disc(3)
    v = mesh.vertices[mesh.cells]
    det = (v[:, 1, 0]-v[:, 0, 0])*(v[:, 2, 1]-v[:, 0, 1])-(v[:, 1, 1]-v[:, 0, 1])*(v[:, 2, 0]-v[:, 0, 0])
    assert (det > 0).all()
    # boundary edges keep their cell's (counter-clockwise) orientation: the rotated edge vector points outwards
    e = mesh.vertices[mesh.boundaryEdges]
    n = np.stack([e[:, 1, 1]-e[:, 0, 1], e[:, 0, 0]-e[:, 1, 0]], axis=1)
    mid = e.mean(axis=1)
    assert ((n*mid).sum(axis=1) > 0).all()
    # every boundary edge belongs to exactly one cell
    cells = {tuple(sorted(c[[i, j]])) for c in mesh.cells for i, j in ((0, 1), (1, 2), (0, 2))}
    assert all(tuple(sorted(b)) in cells for b in mesh.boundaryEdges)


def test_refine_numbering():
    """children of cell i are cells 4i..4i+3; new vertices are numbered in order of first encounter"""
    mesh = disc(1)
    fine = mesh.refine()
    nv = mesh.num_vertices
    assert fine.num_cells == 4*mesh.num_cells
    assert (fine.cells[0::4, 0] == mesh.cells[:, 0]).all()
    assert fine.cells[0, 1] == nv and fine.cells[0, 2] == nv+1 and fine.cells[1, 1] == nv+2
    seen = []
    for c in fine.cells:
        for v in c:
            if v >= nv and v not in seen:
                seen.append(v)
    assert seen == sorted(seen)


def test_P1_dofmap_numbering():
    mesh = disc(2)
    dm = P1_DoFMap(mesh, PHYSICAL)
    assert dm.num_dofs+dm.num_boundary_dofs == mesh.num_vertices
    first = []
    for d in dm.dofs.ravel():
        if d >= 0 and d not in first:
            first.append(int(d))
    assert first == list(range(dm.num_dofs))                   # first-encounter order
    assert (dm.dofs[np.isin(mesh.cells, mesh.boundaryVertices)] < 0).all()
    dmn = P1_DoFMap(mesh, NO_BOUNDARY)
    assert dmn.num_dofs == mesh.num_vertices and (dmn.dofs >= 0).all()


def test_P2_dofmap():
    mesh = disc(1)
    dm = P2_DoFMap(mesh, NO_BOUNDARY)
    nedges = 3*mesh.num_cells-(3*mesh.num_cells-mesh.boundaryEdges.shape[0])//2
    assert dm.num_dofs == mesh.num_vertices+nedges
    bary = np.array([[.2, .3, .5], [.6, .1, .3]]).T
    assert np.allclose(dm.evalShapeFunctions(bary).sum(axis=0), 1.)
    assert np.allclose(dm.evalShapeFunctions(dm.nodes.T), np.eye(6))


def test_interval_and_square():
    m = interval(6)
    assert m.num_cells == 64 and np.isclose(m.h, 2/64) and m.boundaryVertices.tolist() == [0, 1]
    dm = P1_DoFMap(m, PHYSICAL)
    assert dm.num_dofs == 63
    sq = uniformSquare(5, 5, -1, -1, 1, 1)
    assert sq.num_cells == 32 and np.isclose(sq.volume, 4.)


def test_rhs_and_mass():
    mesh = disc(3)
    dm = P1_DoFMap(mesh, NO_BOUNDARY)
    b = dm.assembleRHS(1.0)
    assert np.isclose(np.asarray(b).sum(), mesh.volume)
    M = dm.assembleMass()
    assert np.isclose(M.sum(), mesh.volume)


def test_label_blocks_partition():
    """builder.label_blocks: every cell once, padding = zero-volume copies without DoFs, no block of T cells holds two labels,
    the largest block keeps its DoF count (host logic of the C5 path; the GPU side is tests/test_gpu_parity.py)"""
    from pynucleus_amd import disc, P1_DoFMap, P2_DoFMap, PHYSICAL
    from pynucleus_amd.fractionalOrders import layersFractionalOrder, leftRightFractionalOrder
    from pynucleus_amd.builder import label_blocks, block_dof_count, tile_cells
    orders = (layersFractionalOrder(2, np.array([-1., -0.3, 0.3, 1.]), np.array([[0.3, 0.4, 0.5], [0.4, 0.5, 0.6], [0.5, 0.6, 0.7]])),
              leftRightFractionalOrder(0.25, 0.75, interface=0.1))
    for DoFMap in (P1_DoFMap, P2_DoFMap):
        for noRef in (3, 4):
            mesh = disc(noRef)
            dm = DoFMap(mesh, PHYSICAL)
            T = tile_cells(dm.dofs_per_element, 2)
            for s in orders:
                lab = s.labels(mesh.getCellCenters())
                dmb = label_blocks(dm, lab)
                assert dmb is not dm
                perm, pad = dmb.cell_permutation, dmb.cell_is_padding
                assert sorted(perm[~pad].tolist()) == list(range(mesh.num_cells))
                assert (dmb.dofs[pad] == -1).all() and (dmb.mesh.volVector[pad] == 0.).all() and (dmb.mesh.volVector[~pad] > 0.).all()
                assert np.array_equal(dmb.dofs[~pad], dm.dofs[perm[~pad]]) and np.array_equal(dmb.mesh.cells, mesh.cells[perm])
                lab2 = lab[perm]
                nc2 = dmb.mesh.num_cells
                for b in range((nc2+T-1)//T):
                    assert len(np.unique(lab2[b*T:(b+1)*T])) == 1
                    # padding copies a cell of its own block
                    blk = slice(b*T, min(nc2, (b+1)*T))
                    assert set(perm[blk][pad[blk]].tolist()) <= set(perm[blk][~pad[blk]].tolist())
                assert block_dof_count(dmb.dofs, T) <= block_dof_count(dm.dofs, T)
                assert dmb.num_dofs == dm.num_dofs
    # one label: nothing to do
    dm = P1_DoFMap(disc(3), PHYSICAL)
    assert label_blocks(dm, np.zeros(dm.mesh.num_cells, dtype=np.int32)) is dm


@pytest.mark.parametrize('element,domain', [('P1', 'disc'), ('P2', 'disc'), ('P1', 'interval'), ('P0', 'disc')])
def test_complement_and_combined_dofmaps(element, domain):
    """getComplementDoFMap / combine (DoFMaps.pyx): the complement numbers exactly the boundary DoFs, the combined map the DoFs of both
    (the first map's first), and equals the map without boundary up to the numbering"""
    from pynucleus_amd import disc, interval, PHYSICAL, NO_BOUNDARY, dofmapFactory
    mesh = disc(2) if domain == 'disc' else interval(4)
    dm = dofmapFactory(element, mesh, PHYSICAL)
    dmc = dm.getComplementDoFMap()
    assert dmc.num_dofs == dm.num_boundary_dofs and dmc.num_boundary_dofs == dm.num_dofs
    assert ((dm.dofs >= 0) != (dmc.dofs >= 0)).all()
    both = dm.combine(dmc)
    allm = dofmapFactory(element, mesh, NO_BOUNDARY)
    assert both.num_dofs == dm.num_dofs+dmc.num_dofs == allm.num_dofs and (both.dofs >= 0).all()
    assert np.array_equal(both.dofs[dm.dofs >= 0], dm.dofs[dm.dofs >= 0])
    assert np.array_equal(both.dofs[dmc.dofs >= 0], dm.num_dofs+dmc.dofs[dmc.dofs >= 0])
    # the same partition of the local DoFs into global ones as the map without boundary
    rel = {}
    for a, b in zip(both.dofs.ravel(), allm.dofs.ravel()):
        assert rel.setdefault(int(a), int(b)) == int(b)
    assert len(set(rel.values())) == allm.num_dofs
