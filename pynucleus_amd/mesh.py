"""Simplicial meshes feeding the nonlocal assembly path.

Host-side (numpy) mirror of the small part of the reference's mesh layer that
the hot path consumes: vertices / cells / volVector / hVector / h / hmin / diam,
uniform refinement with the same vertex and cell numbering, radial projection
for discs, boundary edges with the cell's orientation and the surface mesh.

Reference behaviour followed (file:line under /root/reference):
  fem/PyNucleus_fem/mesh.py:209-218   simpleInterval
  fem/PyNucleus_fem/mesh.py:121-185   uniformSquare
  fem/PyNucleus_fem/mesh.py:946-960   uniform_disc (hexagon fan + radial transformer)
  fem/PyNucleus_fem/meshCy.pyx:34-89  radialMeshTransformer
  fem/PyNucleus_fem/meshCy.pyx:506-577 refine (boundary edge / vertex bookkeeping)
  fem/PyNucleus_fem/meshCy.pyx:863-903 refineCy1D, :1052-1109 refineCy2DedgeVals
  fem/PyNucleus_fem/meshCy.pyx:1654-1790 hdeltaCy (h, hmin, vol, hVector)
  fem/PyNucleus_fem/meshCy.pyx:1791-1845 boundaryVertices / boundaryEdges
  fem/PyNucleus_fem/mesh.py:1658-1661 diam = bounding box diagonal
  fem/PyNucleus_fem/mesh.py:2055-2068 get_surface_mesh
"""
import numpy as np

INDEX = np.int32
REAL = np.float64
TAG = np.int8

# boundary tags, fem/PyNucleus_fem/mesh.py:32-40
PHYSICAL = TAG(0)
INTERIOR_NONOVERLAPPING = TAG(-1)
INTERIOR = TAG(-2)
NO_BOUNDARY = np.iinfo(TAG).min


class radialMeshTransformer:
    """Moves each new edge midpoint to the mean radius of the edge's endpoints
    (meshCy.pyx:34-89, radius=0 branch)."""

    def __call__(self, mesh, parents):
        # parents: (num_new, 2) endpoints of the edge each new vertex bisects,
        # new vertices are the last num_new rows of mesh.vertices
        n_new = parents.shape[0]
        if n_new == 0:
            return
        v = mesh.vertices
        first = v.shape[0]-n_new
        r1 = np.sqrt((v[parents[:, 0]]**2).sum(axis=1))
        r2 = np.sqrt((v[parents[:, 1]]**2).sum(axis=1))
        r = 0.5*r1+0.5*r2
        r3 = np.sqrt((v[first:]**2).sum(axis=1))
        v[first:] *= (r/r3)[:, None]


def _first_occurrence_rank(keys):
    """rank of each key by order of first occurrence (what a dict that hands out
    consecutive ids on first insert produces)."""
    uniq, first, inv = np.unique(keys, return_index=True, return_inverse=True)
    order = np.argsort(first, kind='stable')
    rank = np.empty_like(order)
    rank[order] = np.arange(order.shape[0])
    return rank[inv], uniq[order]


class meshBase:
    def __init__(self, vertices, cells):
        self.vertices = np.ascontiguousarray(vertices, dtype=REAL)
        self.cells = np.ascontiguousarray(cells, dtype=INDEX)
        self.dim = self.vertices.shape[1]
        self.manifold_dim = self.cells.shape[1]-1
        self.transformer = None
        self._info = None

    # -- sizes ---------------------------------------------------------
    @property
    def num_vertices(self):
        return self.vertices.shape[0]

    @property
    def num_cells(self):
        return self.cells.shape[0]

    @property
    def vertices_as_array(self):
        return self.vertices

    @property
    def cells_as_array(self):
        return self.cells

    def setMeshTransformation(self, transformer):
        self.transformer = transformer

    # -- h, hmin, volume (hdeltaCy) -------------------------------------
    def _compute(self):
        v = self.vertices[self.cells]          # [nc, k, dim]
        if self.manifold_dim == 1:
            hv = np.sqrt(((v[:, 1]-v[:, 0])**2).sum(axis=1))
            vol = hv.copy()
            hmin = hv.min() if hv.size else 0.
        elif self.manifold_dim == 2 and self.dim == 2:
            g0 = v[:, 2]-v[:, 1]
            g1 = v[:, 2]-v[:, 0]
            g2 = v[:, 1]-v[:, 0]
            # volume2Dnew(gradient[1:, :]) = |det| / 2
            vol = 0.5*np.abs(g1[:, 0]*g2[:, 1]-g1[:, 1]*g2[:, 0])
            e = np.stack([np.sqrt((g**2).sum(axis=1)) for g in (g0, g1, g2)], axis=1)
            hv = e.max(axis=1)
            hmin = e.min()
        elif self.manifold_dim == 0:
            hv = np.ones(self.num_cells)
            vol = np.ones(self.num_cells)
            hmin = 1.
        else:
            raise NotImplementedError()
        self._info = dict(h=float(hv.max()), hmin=float(min(hmin, 100.)), volume=float(vol.sum()),
                          volVector=np.ascontiguousarray(vol), hVector=np.ascontiguousarray(hv))

    def resetMeshInfo(self):
        self._info = None

    def _get(self, key):
        if self._info is None:
            self._compute()
        return self._info[key]

    h = property(lambda self: self._get('h'))
    hmin = property(lambda self: self._get('hmin'))
    volume = property(lambda self: self._get('volume'))
    volVector = property(lambda self: self._get('volVector'))
    hVector = property(lambda self: self._get('hVector'))

    @property
    def diam(self):
        return float(np.linalg.norm(self.vertices.max(axis=0)-self.vertices.min(axis=0), 2))

    def getCellCenters(self):
        return self.vertices[self.cells].mean(axis=1)

    def __repr__(self):
        return '{}(vertices={}, cells={})'.format(type(self).__name__, self.num_vertices, self.num_cells)


class mesh0d(meshBase):
    """Point cloud used as the surface of a 1D mesh (cells = single vertex ids)."""

    def __init__(self, vertices, cells):
        super().__init__(vertices, cells)


class mesh1d(meshBase):
    def __init__(self, vertices, cells):
        super().__init__(vertices, cells)
        if self.dim == 1:
            bv = _boundary_vertices_1d(self.cells)
            self.boundaryVertices = bv
            self.boundaryVertexTags = np.full(bv.shape[0], PHYSICAL, dtype=TAG)

    def getBoundaryVerticesByTag(self, tag=None):
        return _by_tag(self.boundaryVertices, self.boundaryVertexTags, tag)

    def refine(self):
        """refineCy1D (meshCy.pyx:863-903): one new vertex per cell, cells (c0,nv),(nv,c1)."""
        nv, nc = self.num_vertices, self.num_cells
        c = self.cells
        new_ids = nv+np.arange(nc, dtype=INDEX)
        lo = np.minimum(c[:, 0], c[:, 1])
        hi = np.maximum(c[:, 0], c[:, 1])
        vertices = np.empty((nv+nc, self.dim), dtype=REAL)
        vertices[:nv] = self.vertices
        vertices[nv:] = (self.vertices[lo]+self.vertices[hi])*0.5
        cells = np.empty((2*nc, 2), dtype=INDEX)
        cells[0::2, 0] = c[:, 0]
        cells[0::2, 1] = new_ids
        cells[1::2, 0] = new_ids
        cells[1::2, 1] = c[:, 1]
        m = mesh1d(vertices, cells)
        if self.dim == 1:
            m.boundaryVertices = self.boundaryVertices.copy()
            m.boundaryVertexTags = self.boundaryVertexTags.copy()
        if self.transformer is not None:
            self.transformer(m, np.stack([lo, hi], axis=1))
            m.setMeshTransformation(self.transformer)
        return m

    def get_surface_mesh(self, tag=None):
        bv = self.getBoundaryVerticesByTag(tag)
        return mesh0d(self.vertices, bv.reshape(-1, 1))


class mesh2d(meshBase):
    def __init__(self, vertices, cells):
        super().__init__(vertices, cells)
        be = _boundary_edges_2d(self.cells)
        self.boundaryEdges = be
        self.boundaryEdgeTags = np.full(be.shape[0], PHYSICAL, dtype=TAG)
        bv = _boundary_vertices_from_edges(be)
        self.boundaryVertices = bv
        self.boundaryVertexTags = np.full(bv.shape[0], PHYSICAL, dtype=TAG)

    def getBoundaryVerticesByTag(self, tag=None):
        return _by_tag(self.boundaryVertices, self.boundaryVertexTags, tag)

    def getBoundaryEdgesByTag(self, tag=None):
        return _by_tag(self.boundaryEdges, self.boundaryEdgeTags, tag)

    def refine(self):
        """refineCy2DedgeVals (meshCy.pyx:1052-1109) + boundary bookkeeping (:535-560).

        New vertex ids are handed out in order of first encounter of the sorted
        edges (c0,c1), (c0,c2), (c1,c2) while walking the cells; children are
        (c0,m01,m02), (c1,m12,m01), (c2,m02,m12), (m01,m12,m02).
        """
        nv, nc = self.num_vertices, self.num_cells
        c = self.cells.astype(np.int64)
        pairs = np.stack([c[:, [0, 1]], c[:, [0, 2]], c[:, [1, 2]]], axis=1).reshape(-1, 2)
        lo = pairs.min(axis=1)
        hi = pairs.max(axis=1)
        keys = lo*np.int64(nv)+hi
        rank, uniq = _first_occurrence_rank(keys)
        n_new = uniq.shape[0]
        mid = (nv+rank).reshape(nc, 3).astype(INDEX)     # m01, m02, m12
        parents = np.stack([uniq//nv, uniq % nv], axis=1)
        vertices = np.empty((nv+n_new, self.dim), dtype=REAL)
        vertices[:nv] = self.vertices
        vertices[nv:] = (self.vertices[parents[:, 0]]+self.vertices[parents[:, 1]])*0.5
        m01, m02, m12 = mid[:, 0], mid[:, 1], mid[:, 2]
        c32 = self.cells
        cells = np.empty((4*nc, 3), dtype=INDEX)
        cells[0::4] = np.stack([c32[:, 0], m01, m02], axis=1)
        cells[1::4] = np.stack([c32[:, 1], m12, m01], axis=1)
        cells[2::4] = np.stack([c32[:, 2], m02, m12], axis=1)
        cells[3::4] = np.stack([m01, m12, m02], axis=1)
        m = mesh2d.__new__(mesh2d)
        meshBase.__init__(m, vertices, cells)
        # boundary edges: (e0, nv), (nv, e1) keeping orientation and tags
        be = self.boundaryEdges.astype(np.int64)
        bkeys = be.min(axis=1)*np.int64(nv)+be.max(axis=1)
        pos = np.searchsorted(uniq[np.argsort(uniq)], bkeys)
        sorter = np.argsort(uniq)
        bmid = (nv+sorter[pos]).astype(INDEX)
        nbe = np.empty((2*be.shape[0], 2), dtype=INDEX)
        nbe[0::2, 0] = self.boundaryEdges[:, 0]
        nbe[0::2, 1] = bmid
        nbe[1::2, 0] = bmid
        nbe[1::2, 1] = self.boundaryEdges[:, 1]
        m.boundaryEdges = nbe
        m.boundaryEdgeTags = np.repeat(self.boundaryEdgeTags, 2)
        m.boundaryVertices = np.concatenate((self.boundaryVertices, bmid))
        m.boundaryVertexTags = np.concatenate((self.boundaryVertexTags, self.boundaryEdgeTags))
        if self.transformer is not None:
            self.transformer(m, parents)
            m.setMeshTransformation(self.transformer)
        return m

    def get_surface_mesh(self, tag=None):
        s = mesh1d(self.vertices, self.getBoundaryEdgesByTag(tag))
        s.setMeshTransformation(self.transformer)
        return s


def _by_tag(items, tags, tag):
    if tag is None or (isinstance(tag, list) and tag[0] is None):
        return items
    if not isinstance(tag, list):
        tag = [tag]
    idx = np.zeros(tags.shape[0], dtype=bool)
    for t in tag:
        idx |= (tags == t)
    return items[idx]


def _boundary_vertices_1d(cells):
    """meshCy.pyx:1791-1808: vertices that occur in exactly one cell."""
    ids, counts = np.unique(cells.ravel(), return_counts=True)
    return ids[counts == 1].astype(INDEX)


def _boundary_edges_2d(cells):
    """meshCy.pyx:1811-1845: edges (c0,c1),(c1,c2),(c2,c0) that occur once, listed in order
    of appearance and oriented as in their cell."""
    nv = int(cells.max())+1 if cells.size else 0
    c = cells.astype(np.int64)
    e = np.stack([c[:, [0, 1]], c[:, [1, 2]], c[:, [2, 0]]], axis=1).reshape(-1, 2)
    keys = e.min(axis=1)*nv+e.max(axis=1)
    uniq, first, counts = np.unique(keys, return_index=True, return_counts=True)
    sel = np.sort(first[counts == 1])
    return e[sel].astype(INDEX)


def _boundary_vertices_from_edges(bedges):
    if bedges.size == 0:
        return np.zeros((0), dtype=INDEX)
    return np.unique(bedges.ravel()).astype(INDEX)


# ---------------------------------------------------------------------
# mesh constructors

def simpleInterval(a=0., b=1., numCells=1):
    vertices = np.zeros((numCells+1, 1), dtype=REAL)
    cells = np.zeros((numCells, 2), dtype=INDEX)
    for i in range(numCells):
        vertices[i, 0] = a+(b-a)*(i/numCells)
        cells[i, 0] = i
        cells[i, 1] = i+1
    vertices[-1, 0] = b
    return mesh1d(vertices, cells)


def uniformSquare(N=2, M=None, ax=0, ay=0, bx=1, by=1):
    """Structured, uncrossed triangulation of [ax,bx]x[ay,by] (mesh.py:121-147)."""
    xVals = np.linspace(ax, bx, N)
    if M is None:
        M = max(int(np.around((by-ay)/(bx-ax)))*N, 2)
    yVals = np.linspace(ay, by, M)
    x, y = np.meshgrid(xVals, yVals)
    vertices = np.stack([x.flatten(), y.flatten()], axis=1)
    cells = []
    for i in range(M-1):
        for j in range(N-1):
            cells.append((i*N+j, i*N+j+1, (i+1)*N+j+1))
            cells.append((i*N+j, (i+1)*N+j+1, (i+1)*N+j))
    return mesh2d(np.array(vertices, dtype=REAL), np.array(cells, dtype=INDEX))


def uniform_disc(radius=1., sectors=6):
    """Hexagon fan; refinement projects new boundary-ring vertices radially (mesh.py:946-960).  sectors != 6 gives a fan of
    that many triangles (not a reference mesh: used to reach DoF counts between two refinement levels, e.g. 12 sectors
    refined 7 times = 97 921 interior vertices)."""
    points = [(0., 0.)]
    n = int(sectors)
    for i in range(n):
        points.append((radius*np.cos(i*2*np.pi/n), radius*np.sin(i*2*np.pi/n)))
    cells = []
    for i in range(1, len(points)-1):
        cells.append((0, i, i+1))
    cells.append((0, len(points)-1, 1))
    mesh = mesh2d(np.array(points, dtype=REAL), np.array(cells, dtype=INDEX))
    mesh.setMeshTransformation(radialMeshTransformer())
    return mesh


def disc(noRef=0, radius=1., sectors=6):
    """The reference's 'disc' domain for horizon=inf: uniform_disc refined noRef times
    (nl/PyNucleus_nl/nonlocalProblems.py:146-222, fem mesh.py:709-723)."""
    mesh = uniform_disc(radius, sectors)
    for _ in range(noRef):
        mesh = mesh.refine()
    return mesh


def interval(noRef=0, a=-1., b=1.):
    mesh = simpleInterval(a, b)
    for _ in range(noRef):
        mesh = mesh.refine()
    return mesh


def intervalWithInteraction(a, b, horizon, h=None, strictInteraction=True):
    """[a, b] in cells of size <= h with a collar of width `horizon` on both sides, the collar in as many cells of (at most) the
    interior size as cover it (PyNucleus_fem/mesh.py:229-256; strictInteraction=False rounds the collar up to whole cells)."""
    h = horizon if h is None else h
    n = int((b-a)/h)
    n += n*h < b-a
    inner = np.linspace(a, b, n+1)
    hi = inner[1]-inner[0]
    k = int(horizon/hi)
    k += k*hi < horizon-1e-8
    if not strictInteraction:
        horizon = k*hi
    nodes = np.concatenate((np.linspace(a-horizon, a, k+1)[:-1], inner, np.linspace(b, b+horizon, k+1)[1:]))
    cells = np.stack((np.arange(nodes.size-1), np.arange(1, nodes.size)), axis=1).astype(INDEX)
    return mesh1d(np.ascontiguousarray(nodes[:, None], dtype=REAL), cells)


def driverMesh(domain, noRef):
    """Mesh of the reference's runFractional driver for `--domain domain --noRef noRef`: the factory mesh is
    refined until a P1 space with PHYSICAL boundary has a DoF (nonlocalProblems.py:209-212) and then noRef
    more times by the level hierarchy (discretizedProblems.py:386-409, helpers.py:381-411)."""
    if domain == 'interval':
        mesh = simpleInterval(-1., 1.)
    elif domain == 'disc':
        mesh = uniform_disc(1.)
    else:
        raise NotImplementedError(domain)
    while mesh.num_vertices-mesh.boundaryVertices.shape[0] == 0:
        mesh = mesh.refine()
    for _ in range(noRef):
        mesh = mesh.refine()
    return mesh
