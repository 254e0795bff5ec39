"""DoF maps consumed by the nonlocal assembly path (host side, numpy).

Mirrors the numbering of the reference's DoFMap constructor
(/root/reference/fem/PyNucleus_fem/DoFMaps.pyx:145-313): boundary entities of
the requested tag get negative ids -1,-2,... first; interior DoFs are numbered
in order of first encounter while walking cells, vertices before edges.
Shape functions follow DoFMaps.pyx:1854-2025 (P1: barycentric coordinates,
P2: lambda_i(2 lambda_i-1) on vertices, 4 lambda_i lambda_j on edges
(0,1),(1,2),(2,0)).
"""
import numpy as np
from .mesh import INDEX, REAL, PHYSICAL, NO_BOUNDARY


class fe_vector(np.ndarray):
    """numpy vector that remembers its DoFMap (reference: fem vector_{SCALAR}.pxi)."""

    def __new__(cls, data, dm):
        obj = np.asarray(data, dtype=REAL).view(cls)
        obj.dm = dm
        return obj

    def __array_finalize__(self, obj):
        self.dm = getattr(obj, 'dm', None)

    def inner(self, other, *args):
        return float(np.dot(np.asarray(self), np.asarray(other)))

    def toarray(self):
        return np.asarray(self)


class DoFMap:
    polynomialOrder = 0

    def __init__(self, mesh, dofs_per_vertex, dofs_per_edge, dofs_per_cell, tag=None):
        self.mesh = mesh
        self.dim = mesh.dim
        self.tag = tag
        md = mesh.manifold_dim
        vpe = md+1
        epe = 3 if md == 2 else 0
        self.dofs_per_vertex = dofs_per_vertex
        self.dofs_per_edge = dofs_per_edge if epe > 0 else 0
        self.dofs_per_face = 0
        self.dofs_per_cell = dofs_per_cell
        self.dofs_per_element = vpe*dofs_per_vertex+epe*self.dofs_per_edge+dofs_per_cell
        assert dofs_per_vertex in (0, 1) and self.dofs_per_edge in (0, 1) and \
            (dofs_per_cell == 0 or (md == 1 and dofs_per_cell <= 2) or (dofs_per_vertex == 0 and dofs_per_cell == 1)), \
            'P0, P1, P2 and (on intervals) P3 maps are implemented'
        cells = mesh.cells
        nc = cells.shape[0]
        MAXI = np.iinfo(INDEX).max
        dofs = np.full((nc, self.dofs_per_element), -MAXI, dtype=INDEX)

        numB = -1
        vert_dof = np.full(mesh.num_vertices, MAXI, dtype=np.int64)
        if dofs_per_vertex > 0 and md > 0:
            bv = mesh.getBoundaryVerticesByTag(tag) if not _is_no_boundary(tag) else np.zeros(0, dtype=INDEX)
            vert_dof[bv] = numB-np.arange(bv.shape[0])
            numB -= bv.shape[0]
        edge_keys_b = None
        if self.dofs_per_edge > 0:
            be = mesh.getBoundaryEdgesByTag(tag) if not _is_no_boundary(tag) else np.zeros((0, 2), dtype=INDEX)
            be = be.astype(np.int64)
            edge_keys_b = be.min(axis=1)*mesh.num_vertices+be.max(axis=1)
            edge_dof_b = numB-np.arange(be.shape[0])
            numB -= be.shape[0]
        self.num_boundary_dofs = -numB-1

        # Interior entities in order of first encounter. Walking the cells, cell i
        # numbers its not-yet-numbered vertices (local order 0..k) and then its
        # not-yet-numbered edges (01, 12, 02) before cell i+1 is visited.
        c = cells.astype(np.int64)
        ent_keys = [c[:, k] for k in range(vpe)] if dofs_per_vertex > 0 else []
        n_vslots = len(ent_keys)
        if self.dofs_per_edge > 0:
            nvert = mesh.num_vertices
            for (a, b) in ((0, 1), (1, 2), (0, 2)):
                lo = np.minimum(c[:, a], c[:, b])
                hi = np.maximum(c[:, a], c[:, b])
                ent_keys.append(nvert+lo*nvert+hi)
        for k in range(dofs_per_cell):
            # DoFs in the interior of the cell (P0; P2 / P3 on intervals): numbered after the cell's vertices (DoFMaps.pyx:300-305)
            ent_keys.append(mesh.num_vertices*(mesh.num_vertices+1)+dofs_per_cell*np.arange(nc, dtype=np.int64)+k)
        keys = np.stack(ent_keys, axis=1)                     # [nc, slots], cell-major
        flat = keys.reshape(-1)
        is_boundary = np.zeros(flat.shape[0], dtype=bool)
        bvals = np.zeros(flat.shape[0], dtype=np.int64)
        if dofs_per_vertex > 0:
            vmask = np.zeros(keys.shape, dtype=bool)
            vmask[:, :n_vslots] = True
            vmask = vmask.reshape(-1)
            vb = vmask.copy()
            vb[vmask] = vert_dof[flat[vmask]] < 0
            is_boundary |= vb
            bvals[vb] = vert_dof[flat[vb]]
        if self.dofs_per_edge > 0 and edge_keys_b.shape[0] > 0:
            emask = np.zeros(keys.shape, dtype=bool)
            emask[:, n_vslots:] = True
            emask = emask.reshape(-1)
            srt = np.argsort(edge_keys_b)
            ek = flat-mesh.num_vertices
            pos = np.searchsorted(edge_keys_b[srt], ek)
            pos = np.minimum(pos, srt.shape[0]-1)
            hit = emask & (edge_keys_b[srt][pos] == ek)
            is_boundary |= hit
            bvals[hit] = edge_dof_b[srt[pos[hit]]]
        interior = ~is_boundary
        ids = np.empty(flat.shape[0], dtype=np.int64)
        ids[is_boundary] = bvals[is_boundary]
        if interior.any():
            uniq, first, inv = np.unique(flat[interior], return_index=True, return_inverse=True)
            order = np.argsort(first, kind='stable')
            rank = np.empty_like(order)
            rank[order] = np.arange(order.shape[0])
            ids[interior] = rank[inv]
            self.num_dofs = int(uniq.shape[0])
        else:
            self.num_dofs = 0
        # reorder columns: our slot order for edges was (01,12,02); element layout is
        # vertices, then edges (01), (12), (20)  -> same order
        dofs[:, :] = ids.reshape(nc, -1).astype(INDEX)
        self.dofs = np.ascontiguousarray(dofs)
        self._set_nodes()

    # ------------------------------------------------------------------
    def _set_nodes(self):
        raise NotImplementedError()

    def getComplementDoFMap(self):
        """DoFMaps.pyx getComplementDoFMap: the same element on the same mesh with the roles swapped -- its DoFs are the boundary DoFs
        of this map (boundary DoF -1-k becomes DoF k), the DoFs of this map are its boundary"""
        import copy
        dmc = copy.copy(self)
        dmc.dofs = np.ascontiguousarray((-1-self.dofs).astype(self.dofs.dtype))
        dmc.num_dofs, dmc.num_boundary_dofs = int(self.num_boundary_dofs), int(self.num_dofs)
        return dmc

    def combine(self, other):
        """DoFMaps.pyx combine: one map over the DoFs of both (those of ``self`` first, then those of ``other``); the two maps live on the
        same mesh with the same element and share no DoF"""
        import copy
        assert type(self) is type(other) and self.dofs.shape == other.dofs.shape
        assert self.mesh is other.mesh or np.array_equal(self.mesh.cells, other.mesh.cells)
        assert not ((self.dofs >= 0) & (other.dofs >= 0)).any(), 'the two DoFMaps share DoFs'
        dmc = copy.copy(self)
        d = np.where(self.dofs >= 0, self.dofs, np.where(other.dofs >= 0, self.num_dofs+other.dofs, self.dofs))
        dmc.dofs = np.ascontiguousarray(d.astype(self.dofs.dtype))
        dmc.num_dofs = int(self.num_dofs+other.num_dofs)
        dmc.num_boundary_dofs = int(len(np.unique(d[d < 0])))
        return dmc

    def cell2dof(self, cellNo, perCellNo):
        return int(self.dofs[cellNo, perCellNo])

    def getDoFCoordinates(self):
        """coordinates of the interior DoFs [num_dofs, dim]"""
        coords = np.zeros((self.num_dofs, self.mesh.dim), dtype=REAL)
        v = self.mesh.vertices[self.mesh.cells]                 # [nc, k, dim]
        x = np.einsum('pk,ckd->cpd', self.nodes, v)             # [nc, dpe, dim]
        d = self.dofs
        m = d >= 0
        coords[d[m]] = x[m]
        return coords

    def evalShapeFunctions(self, bary):
        """values of all local shape functions at barycentric points bary[k, n] -> [dpe, n]"""
        raise NotImplementedError()

    # -- small FE helpers used by the driver-level harness -----------------
    def zeros(self):
        return fe_vector(np.zeros(self.num_dofs), self)

    def ones(self):
        return fe_vector(np.ones(self.num_dofs), self)

    def interpolate(self, fun):
        return fe_vector(np.array([fun(x) for x in self.getDoFCoordinates()]), self)

    def _volume_rule(self):
        from .quadrature import simplexXiaoGimbutas
        return simplexXiaoGimbutas(2*self.polynomialOrder+2, self.mesh.dim, self.mesh.manifold_dim)

    def assembleRHS(self, fun, qr=None):
        """b_i = int fun phi_i (fem/PyNucleus_fem/femCy.pyx assembleRHS)."""
        if qr is None:
            qr = self._volume_rule()
        v = self.mesh.vertices[self.mesh.cells]
        x = np.einsum('kn,ckd->cnd', qr.nodes, v)               # [nc, n, dim]
        if callable(fun):
            f = np.array([[fun(p) for p in cell] for cell in x])
        else:
            f = np.full(x.shape[:2], float(fun))
        phi = self.evalShapeFunctions(qr.nodes)                 # [dpe, n]
        loc = np.einsum('cn,n,pn->cp', f, qr.weights, phi)*self.mesh.volVector[:, None]
        b = np.zeros(self.num_dofs)
        m = self.dofs >= 0
        np.add.at(b, self.dofs[m], loc[m])
        return fe_vector(b, self)

    def assembleMass(self, qr=None):
        import scipy.sparse as sp
        qr = self._volume_rule() if qr is None else qr
        phi = self.evalShapeFunctions(qr.nodes)
        Mloc = np.einsum('n,pn,qn->pq', qr.weights, phi, phi)
        nc, dpe = self.dofs.shape
        I = np.repeat(self.dofs[:, :, None], dpe, axis=2)
        J = np.repeat(self.dofs[:, None, :], dpe, axis=1)
        V = Mloc[None, :, :]*self.mesh.volVector[:, None, None]
        m = (I >= 0) & (J >= 0)
        return sp.csr_matrix((V[m], (I[m], J[m])), shape=(self.num_dofs, self.num_dofs))

    def L2norm_of_error(self, u, fun, order=None):
        """sqrt(int (u_h - fun)^2) over the mesh, boundary DoFs = 0."""
        from .quadrature import simplexXiaoGimbutas
        qr = simplexXiaoGimbutas(order or (2*self.polynomialOrder+4), self.mesh.dim, self.mesh.manifold_dim)
        v = self.mesh.vertices[self.mesh.cells]
        x = np.einsum('kn,ckd->cnd', qr.nodes, v)
        phi = self.evalShapeFunctions(qr.nodes)
        uu = np.concatenate((np.asarray(u), [0.]))
        d = np.where(self.dofs >= 0, self.dofs, self.num_dofs)
        uh = np.einsum('cp,pn->cn', uu[d], phi)
        f = np.array([[fun(p) for p in cell] for cell in x])
        return float(np.sqrt(np.einsum('cn,n,c->', (uh-f)**2, qr.weights, self.mesh.volVector)))

    def assembleNonlocal(self, kernel, matrixFormat='DENSE', **kwargs):
        """DoFMaps.pyx:808-900"""
        from .builder import nonlocalBuilder
        params = kwargs.pop('params', {})
        builder = nonlocalBuilder(self, kernel, params, **kwargs)
        fmt = matrixFormat.upper()
        if fmt == 'DENSE':
            return builder.getDense()
        elif fmt == 'DIAGONAL':
            return builder.getDiagonal()
        elif fmt == 'SPARSE':
            return builder.getSparse()
        elif fmt == 'H2':
            return builder.getH2()
        raise NotImplementedError(matrixFormat)

    def __repr__(self):
        return '{} with {} DoFs and {} boundary DoFs.'.format(type(self).__name__, self.num_dofs, self.num_boundary_dofs)


def _is_no_boundary(tag):
    return (not isinstance(tag, list)) and tag is not None and tag == NO_BOUNDARY


class P1_DoFMap(DoFMap):
    polynomialOrder = 1

    def __init__(self, mesh, tag=None):
        super().__init__(mesh, 1, 0, 0, tag)

    def _set_nodes(self):
        k = self.mesh.manifold_dim+1
        self.nodes = np.eye(k, dtype=REAL)

    def evalShapeFunctions(self, bary):
        return np.array(bary, dtype=REAL, copy=True)


class P2_DoFMap(DoFMap):
    polynomialOrder = 2

    def __init__(self, mesh, tag=None):
        assert mesh.manifold_dim in (1, 2), 'P2 is implemented on intervals and triangles'
        if mesh.manifold_dim == 1:
            super().__init__(mesh, 1, 0, 1, tag)             # vertices + the midpoint of the cell (DoFMaps.pyx P2_DoFMap, 1D)
        else:
            super().__init__(mesh, 1, 1, 0, tag)

    def _set_nodes(self):
        if self.mesh.manifold_dim == 1:
            self.nodes = np.array([[1., 0.], [0., 1.], [.5, .5]], dtype=REAL)
        else:
            self.nodes = np.array([[1., 0., 0.], [0., 1., 0.], [0., 0., 1.],
                                   [.5, .5, 0.], [0., .5, .5], [.5, 0., .5]], dtype=REAL)

    def evalShapeFunctions(self, bary):
        if self.mesh.manifold_dim == 1:
            l0, l1 = bary[0], bary[1]
            return np.stack([l0*(2*l0-1), l1*(2*l1-1), 4*l0*l1])
        l0, l1, l2 = bary[0], bary[1], bary[2]
        return np.stack([l0*(2*l0-1), l1*(2*l1-1), l2*(2*l2-1), 4*l0*l1, 4*l1*l2, 4*l0*l2])


class P0_DoFMap(DoFMap):
    """piecewise constants, one DoF per cell at its barycentre (fem/PyNucleus_fem/DoFMaps.pyx:1788-1807); no DoF lies on the boundary"""
    polynomialOrder = 0

    def __init__(self, mesh, tag=None):
        assert mesh.manifold_dim in (1, 2)
        super().__init__(mesh, 0, 0, 1, tag)

    def _set_nodes(self):
        k = self.mesh.manifold_dim+1
        self.nodes = np.full((1, k), 1./k, dtype=REAL)

    def evalShapeFunctions(self, bary):
        return np.ones((1, np.asarray(bary).shape[1]), dtype=REAL)


class P3_DoFMap(DoFMap):
    """continuous piecewise cubics on intervals: the two vertices, then the nodes at 1/3 and 2/3 of the cell
    (DoFMaps.pyx:2106-2121; shape functions :2044-2045 vertex, :2068-2069 edge (v1, v2) = 13.5 l1 l2 (l1 - 1/3))"""
    polynomialOrder = 3

    def __init__(self, mesh, tag=None):
        assert mesh.manifold_dim == 1, 'P3 is implemented on intervals'
        super().__init__(mesh, 1, 0, 2, tag)

    def _set_nodes(self):
        self.nodes = np.array([[1., 0.], [0., 1.], [2./3., 1./3.], [1./3., 2./3.]], dtype=REAL)

    def evalShapeFunctions(self, bary):
        l0, l1 = bary[0], bary[1]
        return np.stack([4.5*l0*(l0-1./3.)*(l0-2./3.), 4.5*l1*(l1-1./3.)*(l1-2./3.),
                         13.5*l0*l1*(l0-1./3.), 13.5*l1*l0*(l1-1./3.)])


def dofmapFactory(element, mesh, tag=None):
    element = element.upper() if isinstance(element, str) else 'P{}'.format(element)
    if element == 'P0':
        return P0_DoFMap(mesh, tag)
    elif element == 'P1':
        return P1_DoFMap(mesh, tag)
    elif element == 'P2':
        return P2_DoFMap(mesh, tag)
    elif element == 'P3':
        return P3_DoFMap(mesh, tag)
    raise NotImplementedError(element)
