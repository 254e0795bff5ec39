"""Cluster tree, admissibility and near-field work lists for assembleClusters (host side, numpy).

Mirrors what the H2 near-field assembly of the reference consumes
(/root/reference/nl/PyNucleus_nl):
  clusterMethodCy.pyx:3922-3977  getDoFBoxesAndCells  (support boxes of the DoFs, DoF -> cells graph)
  clusterMethodCy.pyx:354-663    tree_node.refine     (here: the MEDIAN split along the longest box edge only)
  clusterMethodCy.pyx:4008-4136  queryAdmissibility / getAdmissibleClusters (eta criterion, near/far recursion and
                                 the merge of near-field children into their parent pair)
  nonlocalAssembly.pyx:374-392   nearFieldClusterPair.set_cells (cellsUnion, cellsInter)
  nonlocalAssembly.pyx:540-578   boundaryEdges of a cell set (orientation of the owning cell)
  nonlocalAssembly_{SCALAR}.pxi:260-391, 408-424   buildMasksForClusters / getElemElemSymMask
  nonlocalAssembly_{SCALAR}.pxi:3226-3289          getSparseNearField (CSR / SSS sparsity pattern)
The far field (Chebyshev interpolation, transfer operators, H2 matvec) is not built yet.
"""
import numpy as np


def getDoFBoxesAndCells(dm):
    """boxes[N, dim, 2] of the DoF supports and the DoF -> cells graph as CSR (indptr, indices); kept on the DoF map"""
    cached = getattr(dm, '_boxes_and_cells', None)
    if cached is not None and cached[0] is dm.dofs and cached[1] is dm.mesh:
        return cached[2]
    out = _getDoFBoxesAndCells(dm)
    dm._boxes_and_cells = (dm.dofs, dm.mesh, out)
    return out


def _getDoFBoxesAndCells(dm):
    mesh = dm.mesh
    v = mesh.vertices[mesh.cells]                       # [nc, k, dim]
    lo, hi = v.min(axis=1), v.max(axis=1)               # [nc, dim]
    N, dim = dm.num_dofs, mesh.dim
    boxes = np.empty((N, dim, 2))
    boxes[:, :, 0] = np.inf
    boxes[:, :, 1] = -np.inf
    cell_ids = np.repeat(np.arange(mesh.num_cells), dm.dofs_per_element)
    d = dm.dofs.reshape(-1)
    m = d >= 0
    for k in range(dim):
        np.minimum.at(boxes[:, k, 0], d[m], lo[cell_ids[m], k])
        np.maximum.at(boxes[:, k, 1], d[m], hi[cell_ids[m], k])
    order = np.lexsort((cell_ids[m], d[m]))
    dd, cc = d[m][order], cell_ids[m][order]
    keep = np.ones(dd.shape[0], dtype=bool)
    keep[1:] = (dd[1:] != dd[:-1]) | (cc[1:] != cc[:-1])
    dd, cc = dd[keep], cc[keep]
    indptr = np.zeros(N+1, dtype=np.int64)
    np.add.at(indptr, dd+1, 1)
    indptr = np.cumsum(indptr)
    return boxes, (indptr, cc.astype(np.int32))


class tree_node:
    def __init__(self, parent, dofs, boxes, coords, dof2cells, levelNo):
        self.parent = parent
        self.dofs = np.sort(np.asarray(dofs, dtype=np.int32))
        self.box = np.stack([boxes[self.dofs, :, 0].min(axis=0), boxes[self.dofs, :, 1].max(axis=0)], axis=1)
        self.children = []
        self.levelNo = levelNo
        self._boxes, self._coords, self._d2c = boxes, coords, dof2cells
        self._cells = None

    @property
    def is_leaf(self):
        return len(self.children) == 0

    def get_dofs(self):
        return self.dofs

    def get_num_dofs(self):
        return self.dofs.shape[0]

    @property
    def cells(self):
        """cells touching the cluster's DoFs (NA:2887-2898)"""
        if self._cells is None:
            indptr, indices = self._d2c
            parts = [indices[indptr[I]:indptr[I+1]] for I in self.dofs]
            self._cells = np.unique(np.concatenate(parts)) if parts else np.zeros(0, dtype=np.int32)
        return self._cells

    def refine(self, minSize, maxLevels):
        """MEDIAN split of the DoF coordinates along the longest edge of the cluster box into two children"""
        n = self.dofs.shape[0]
        if self.levelNo+1 >= maxLevels or n <= minSize or not self.is_leaf:
            return
        k = int(np.argmax(self.box[:, 1]-self.box[:, 0]))
        x = self._coords[self.dofs, k]
        med = np.median(x)
        left, right = self.dofs[x < med], self.dofs[x >= med]
        if left.shape[0] < minSize or right.shape[0] < minSize or left.shape[0] == n or right.shape[0] == n:
            return
        self.children = [tree_node(self, left, self._boxes, self._coords, self._d2c, self.levelNo+1),
                         tree_node(self, right, self._boxes, self._coords, self._d2c, self.levelNo+1)]

    def leaves(self):
        if self.is_leaf:
            yield self
        else:
            for c in self.children:
                yield from c.leaves()


class nearFieldClusterPair:
    def __init__(self, n1, n2):
        self.n1, self.n2 = n1, n2
        self._union = self._inter = None

    def set_cells(self):
        """nonlocalAssembly.pyx:382-392; computed on first use"""
        return self

    @property
    def cellsUnion(self):
        if self._union is None:
            self._union = np.union1d(self.n1.cells, self.n2.cells)
        return self._union

    @property
    def cellsInter(self):
        if self._inter is None:
            self._inter = np.intersect1d(self.n1.cells, self.n2.cells)
        return self._inter

    def __repr__(self):
        return 'nearFieldClusterPair({} x {} DoFs)'.format(self.n1.get_num_dofs(), self.n2.get_num_dofs())


class farFieldClusterPair:
    def __init__(self, n1, n2):
        self.n1, self.n2 = n1, n2


# ---------------------------------------------------------------------------------------------------------------------
# The tree, the admissibility recursion and the cells of the nodes come from libpnl_hip.so (csrc/pnl_plan.hip, host code:
# no GPU needed); the classes below are views on its arrays with the attributes of tree_node.  The numpy versions above
# stay as the reference the CPU tests compare with (PNL_PLAN=numpy selects them).
def _use_native():
    import os
    return os.environ.get('PNL_PLAN', 'native') != 'numpy'


REFINEMENT_TYPES = {'median': 0, 'MEDIAN': 0, 'geometric': 1, 'GEOMETRIC': 1, 'barycenter': 2, 'BARYCENTER': 2}


class _nativeTree:
    def __init__(self, dm, eta, minSize, maxLevels, mode, dof_block=None, mixed_block=-1, refinementType='MEDIAN', planner='host',
                 horizon=np.inf):
        import ctypes as C
        from . import _lib
        L = _lib.load()
        boxes, (ptr, idx) = getDoFBoxesAndCells(dm)
        self.dm, self.L = dm, L
        self.boxes, self.d2c = boxes, (ptr, idx)
        N, dim = dm.num_dofs, dm.mesh.dim
        b = np.ascontiguousarray(boxes, dtype=np.float64)
        p = np.ascontiguousarray(ptr, dtype=np.int64)
        ix = np.ascontiguousarray(idx, dtype=np.int32)
        h = C.c_void_p()
        self.dof_block = None if dof_block is None else np.ascontiguousarray(dof_block, dtype=np.int32)
        rc = _lib.PNL_ERR_UNSUPPORTED
        self.planner = 'host'
        if planner == 'device' and dof_block is None and not np.isfinite(horizon):
            # refinement and admissibility as level-synchronous sweeps on the GPU (csrc/pnl_plan_dev.hip): the same tree and lists; what
            # it does not take (BARYCENTER split) falls through to the host loops
            rc = L.pnl_tree_build_device(N, dim, b.ctypes.data, p.ctypes.data, ix.ctypes.data, dm.mesh.num_cells, float(eta), int(minSize),
                                         int(maxLevels), int(mode), REFINEMENT_TYPES[refinementType], C.byref(h))
            if rc == 0:
                self.planner = 'device'
            elif rc != _lib.PNL_ERR_UNSUPPORTED:
                raise RuntimeError('pnl_tree_build_device failed: {}'.format(rc))
        if rc == _lib.PNL_ERR_UNSUPPORTED:
            rc = L.pnl_tree_build_horizon(N, dim, b.ctypes.data, p.ctypes.data, ix.ctypes.data, dm.mesh.num_cells, float(eta), int(minSize),
                                          int(maxLevels), int(mode), None if dof_block is None else self.dof_block.ctypes.data,
                                          int(mixed_block), REFINEMENT_TYPES[refinementType], float(horizon), C.byref(h))
        if rc:
            raise RuntimeError('pnl_tree_build failed: {}'.format(rc))
        self.h = h
        sz = np.zeros(3, dtype=np.int64)
        L.pnl_tree_sizes(h, sz.ctypes.data)
        nn, nnear, nfar = int(sz[0]), int(sz[1]), int(sz[2])
        self.range = np.zeros((nn, 2), dtype=np.int32)
        self.parent = np.zeros(nn, dtype=np.int32)
        self.children = np.zeros((nn, 2), dtype=np.int32)
        self.level = np.zeros(nn, dtype=np.int32)
        self.box = np.zeros((nn, dim, 2))
        self.perm = np.zeros(N, dtype=np.int32)
        self.near = np.zeros((nnear, 2), dtype=np.int32)
        self.far = np.zeros((nfar, 3), dtype=np.int32)
        L.pnl_tree_get(h, self.range.ctypes.data, self.parent.ctypes.data, self.children.ctypes.data, self.level.ctypes.data,
                       self.box.ctypes.data, self.perm.ctypes.data, self.near.ctypes.data, self.far.ctypes.data)
        self._nodes = {}
        self._cells = {}

    def __del__(self):
        try:
            self.L.pnl_tree_destroy(self.h)
        except Exception:
            pass

    def node(self, k):
        n = self._nodes.get(k)
        if n is None:
            n = self._nodes[k] = native_node(self, int(k))
        return n

    def load_cells(self, ids):
        """cells of many nodes in one call"""
        ids = np.ascontiguousarray([k for k in ids if k not in self._cells], dtype=np.int32)
        if ids.shape[0] == 0:
            return
        off = np.zeros(ids.shape[0]+1, dtype=np.int64)
        self.L.pnl_tree_node_cells(self.h, ids.shape[0], ids.ctypes.data, off.ctypes.data, None)
        cells = np.zeros(int(off[-1]), dtype=np.int32)
        self.L.pnl_tree_node_cells(self.h, ids.shape[0], ids.ctypes.data, off.ctypes.data, cells.ctypes.data)
        for i, k in enumerate(ids):
            self._cells[int(k)] = cells[off[i]:off[i+1]]


class native_node(tree_node):
    """a node of the C++ tree with the attributes of tree_node (dofs sorted ascending, box [dim, 2], children, cells)"""

    def __init__(self, T, k):
        self._T, self._k = T, k
        self.levelNo = int(T.level[k])
        self.box = T.box[k]
        self._dofs = self._cells_ = None
        self._boxes, self._d2c = T.boxes, T.d2c
        self._coords = None

    @property
    def parent(self):
        p = int(self._T.parent[self._k])
        return self._T.node(p) if p >= 0 else None

    @property
    def children(self):
        c = self._T.children[self._k]
        return [self._T.node(int(c[0])), self._T.node(int(c[1]))] if c[0] >= 0 else []

    @property
    def is_leaf(self):
        return self._T.children[self._k, 0] < 0

    @property
    def dofs(self):
        if self._dofs is None:
            b, e = self._T.range[self._k]
            d = self._T.perm[b:e]
            self._dofs = d if self.is_leaf else np.sort(d)
        return self._dofs

    @property
    def cells(self):
        if self._k not in self._T._cells:
            self._T.load_cells([self._k])
        return self._T._cells[self._k]

    def refine(self, minSize, maxLevels):
        return                                               # the native tree is refined completely when it is built


def distBoxes(b1, b2):
    gap = np.maximum(0., np.maximum(b1[:, 0]-b2[:, 1], b2[:, 0]-b1[:, 1]))
    return float(np.sqrt((gap**2).sum()))


def diamBox(b):
    return float(np.sqrt(((b[:, 1]-b[:, 0])**2).sum()))


def getTree(dm, native=None):
    if (native if native is not None else False):
        return _nativeTree(dm, 3., 1 << 30, 1, -1).node(0)
    boxes, d2c = getDoFBoxesAndCells(dm)
    coords = boxes.mean(axis=2)
    root = tree_node(None, np.arange(dm.num_dofs, dtype=np.int32), boxes, coords, d2c, 0)
    return root


def maxDistBoxes(b1, b2):
    """interactionDomain.maxDistBoxes (interactionDomains.pyx:325-337), as written there"""
    s = 0.
    for i in range(b1.shape[0]):
        lo, hi = (b2[i, 0], b1[i, 1]) if b1[i, 0] > b2[i, 0] else (b1[i, 0], b2[i, 1])
        s += max(hi-lo, 0.)**2
    return float(np.sqrt(s))


def getAdmissibleClusters(n1, n2, eta, minSize, maxLevels, Pfar, Pnear, level=0, horizon=np.inf):
    """clusterMethodCy.pyx:4046-4136; returns whether far-field pairs were added below.  Finite horizon (l2 ball, :4069-4090): cluster
    pairs farther apart than the horizon do not interact, pairs the horizon may cut stay in the near field, near-field children
    are merged into one block only if the block fits into the horizon (:4131-4135)"""
    dist = distBoxes(n1.box, n2.box)
    admissible = eta*dist >= max(diamBox(n1.box), diamBox(n2.box))
    finite = np.isfinite(horizon)
    diamUnion = 0.
    if finite:
        if dist > horizon:
            return True
        if horizon <= maxDistBoxes(n1.box, n2.box):
            admissible = False
        diamUnion = diamBox(np.stack([np.minimum(n1.box[:, 0], n2.box[:, 0]), np.maximum(n1.box[:, 1], n2.box[:, 1])], axis=1))
    lenNear = len(Pnear)
    added = False
    if admissible:
        Pfar.setdefault(level, []).append(farFieldClusterPair(n1, n2))
        return True
    n1.refine(minSize, maxLevels)
    n2.refine(minSize, maxLevels)
    if (n1.is_leaf and n2.is_leaf) or level == maxLevels:
        Pnear.append(nearFieldClusterPair(n1, n2))
        return False
    elif n1.is_leaf:
        for t2 in n2.children:
            added |= getAdmissibleClusters(n1, t2, eta, minSize, maxLevels, Pfar, Pnear, level+1, horizon)
    elif n2.is_leaf:
        for t1 in n1.children:
            added |= getAdmissibleClusters(t1, n2, eta, minSize, maxLevels, Pfar, Pnear, level+1, horizon)
    else:
        for t1 in n1.children:
            for t2 in n2.children:
                added |= getAdmissibleClusters(t1, t2, eta, minSize, maxLevels, Pfar, Pnear, level+1, horizon)
    if not added and (not finite or diamUnion < horizon):
        # no far-field pair below: keep the whole block as one near-field pair (CM:4131-4135)
        del Pnear[lenNear:]
        Pnear.append(nearFieldClusterPair(n1, n2))
    return added


def dofKernelBlocks(dm, T):
    """getKernelBlocksAndJumps NA:2312-2352 for a piecewise-constant order: block of every DoF = the label shared by all its
    cells, or num_labels for DoFs on an interface (INTERFACE_DOF).  Returns (dof_block[num_dofs], mixed_block)."""
    lab = np.asarray(T.cell_labels, dtype=np.int64)
    L = int(T.num_labels)
    lo = np.full(dm.num_dofs+1, L, dtype=np.int64)
    hi = np.full(dm.num_dofs+1, -1, dtype=np.int64)
    d = np.where(dm.dofs >= 0, dm.dofs, dm.num_dofs).reshape(-1)
    rep = np.repeat(lab, dm.dofs.shape[1])
    np.minimum.at(lo, d, rep)
    np.maximum.at(hi, d, rep)
    blk = np.where(lo == hi, lo, L)[:dm.num_dofs]
    return blk.astype(np.int32), L


def getNearFieldClusters(dm, eta=3., minClusterSize=None, maxLevels=200, dof_block=None, mixed_block=-1, refinementType='MEDIAN', planner='host',
                         horizon=np.inf):
    """(root, Pnear, Pfar) for dm; both orientations (n1,n2) and (n2,n1) of off-diagonal pairs are listed, like the
    reference's recursion from (root, root).  dof_block / mixed_block (dofKernelBlocks): clusters are split by kernel block
    before anything else and only pairs of single-block clusters can be admissible (variable orders, NA:2619-2640)."""
    if minClusterSize is None:
        minClusterSize = max(dm.num_dofs//64, 8)
    if dof_block is not None and not _use_native():
        raise NotImplementedError('cluster trees by kernel block: native planner only')
    if _use_native():
        T = _nativeTree(dm, eta, minClusterSize, maxLevels, 1, dof_block, mixed_block, refinementType, planner, horizon)
        T.load_cells(np.unique(T.near))
        Pnear = [nearFieldClusterPair(T.node(a), T.node(b)) for a, b in T.near]
        Pfar = {}
        for a, b, lvl in T.far:
            Pfar.setdefault(int(lvl), []).append(farFieldClusterPair(T.node(a), T.node(b)))
        return T.node(0), Pnear, Pfar
    if REFINEMENT_TYPES[refinementType] != 0:
        raise NotImplementedError('refinementType {}: native planner only'.format(refinementType))
    root = getTree(dm)
    Pnear, Pfar = [], {}
    getAdmissibleClusters(root, root, eta, minClusterSize, maxLevels, Pfar, Pnear, 0, horizon)
    for cp in Pnear:
        cp.set_cells()
    return root, Pnear, Pfar


def coveringCluster(dm):
    """one near-field pair (root, root): assembleClusters must then reproduce the dense operator (tests/test_nearField.py:171-184)"""
    root = getTree(dm)
    cp = nearFieldClusterPair(root, root)
    cp.set_cells()
    return root, [cp]


# ---------------------------------------------------------------------------------------------------------------------
def getSparseNearField(dm, Pnear, symmetric=True, device=None):
    """sparsity pattern of the near field: CSR (indptr, indices); with symmetric=True only I > J is stored (SSS, the
    diagonal lives in its own vector).  With a torch device the keys of all blocks are generated, sorted and split into
    rows there (5e7 entries at 49k DoFs: 0.6 s of numpy sort on the host, tens of ms on the GPU)."""
    N = dm.num_dofs
    # cluster pairs of ONE native tree: the pattern in C++ (bitmap per leaf, threaded)
    if len(Pnear) and _use_native() and all(isinstance(cp.n1, native_node) and isinstance(cp.n2, native_node) for cp in Pnear):
        T = Pnear[0].n1._T
        if all(cp.n1._T is T and cp.n2._T is T for cp in Pnear):
            import ctypes as C
            pairs = np.ascontiguousarray([(cp.n1._k, cp.n2._k) for cp in Pnear], dtype=np.int32)
            h = C.c_void_p()
            rc = T.L.pnl_near_pattern(T.h, pairs.shape[0], pairs.ctypes.data, 1 if symmetric else 0, C.byref(h))
            if rc == -2:                                         # PNL_ERR_UNSUPPORTED
                from ._lib import PnlError
                raise PnlError('getSparseNearField: the near-field pattern has more than 2^31 - 1 stored entries (INDEX_t is 32 bits)')
            if rc == 0:
                nnz = int(T.L.pnl_pattern_nnz(h))
                if device is not None and getattr(device, 'type', 'cpu') == 'cuda':
                    # filled straight into pinned host memory (torch caches the allocation), copied asynchronously
                    import torch
                    ip = torch.empty(N+1, dtype=torch.int32, pin_memory=True)
                    ix = torch.empty(max(nnz, 1), dtype=torch.int32, pin_memory=True)
                    T.L.pnl_pattern_get(h, C.c_void_p(ip.data_ptr()), C.c_void_p(ix.data_ptr()))
                    T.L.pnl_pattern_destroy(h)
                    return ip.to(device, non_blocking=True), ix[:nnz].to(device, non_blocking=True)
                indptr = np.empty(N+1, dtype=np.int32)
                indices = np.empty(nnz, dtype=np.int32)
                T.L.pnl_pattern_get(h, indptr.ctypes.data, indices.ctypes.data)
                T.L.pnl_pattern_destroy(h)
                if device is not None:
                    import torch
                    return torch.from_numpy(indptr).to(device), torch.from_numpy(indices).to(device)
                return indptr, indices
    if device is not None and len(Pnear):
        import torch
        nr = np.array([len(cp.n1.dofs) for cp in Pnear], dtype=np.int64)
        ncol = np.array([len(cp.n2.dofs) for cp in Pnear], dtype=np.int64)
        rows_cat = torch.as_tensor(np.concatenate([np.asarray(cp.n1.dofs, dtype=np.int64) for cp in Pnear]), device=device)
        cols_cat = torch.as_tensor(np.concatenate([np.asarray(cp.n2.dofs, dtype=np.int64) for cp in Pnear]), device=device)
        size = torch.as_tensor(nr*ncol, device=device)
        start = torch.cumsum(size, 0)-size
        rp = torch.as_tensor(np.concatenate([[0], np.cumsum(nr)[:-1]]), device=device)
        cp_ = torch.as_tensor(np.concatenate([[0], np.cumsum(ncol)[:-1]]), device=device)
        nc_d = torch.as_tensor(ncol, device=device)
        total = int((nr*ncol).sum())
        blk = torch.repeat_interleave(torch.arange(len(Pnear), device=device), size, output_size=total)
        loc = torch.arange(total, device=device)-start[blk]
        r = torch.div(loc, nc_d[blk], rounding_mode='floor')
        I = rows_cat[rp[blk]+r]
        J = cols_cat[cp_[blk]+(loc-r*nc_d[blk])]
        del blk, loc, r
        keys = (I << 32) | J
        if symmetric:
            keys = keys[I > J]
        del I, J
        keys = torch.sort(keys).values
        if keys.numel() > 1 and bool((keys[1:] == keys[:-1]).any()):
            keys = torch.unique_consecutive(keys)
        rows = keys >> 32
        indptr = torch.zeros(N+1, dtype=torch.int64, device=device)
        indptr[1:] = torch.cumsum(torch.bincount(rows, minlength=N), 0)
        # int32 tensors on the device: the operators hand them to the library device-to-device and copy them to the host only
        # when somebody reads .indptr / .indices
        return indptr.to(torch.int32), (keys & 0xffffffff).to(torch.int32)
    parts = []
    for cp in Pnear:
        I = np.asarray(cp.n1.dofs, dtype=np.int64)[:, None]
        J = np.asarray(cp.n2.dofs, dtype=np.int64)[None, :]
        k = (I << 32) | J                                           # one key per entry of the block, no repeat / tile copies
        parts.append(k[I > J] if symmetric else k.ravel())
    keys = np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64)
    keys.sort()
    if keys.shape[0] > 1 and (keys[1:] == keys[:-1]).any():         # blocks of a cluster partition are disjoint as a rule
        keys = np.unique(keys)
    I = keys >> 32                                                  # (shifts instead of a 64-bit division per entry)
    J = (keys & 0xffffffff).astype(np.int32)
    indptr = np.zeros(N+1, dtype=np.int64)
    indptr[1:] = np.cumsum(np.bincount(I, minlength=N))
    return indptr.astype(np.int32), J


def elemElemSymMaskTable(dpe):
    """k(p, q) for p <= q over the 2*dpe local DoFs: flattened index of the symmetric local matrix"""
    n = 2*dpe
    k = np.full((n, n), -1, dtype=np.int64)
    c = 0
    for p in range(n):
        for q in range(p, n):
            k[p, q] = c
            c += 1
    return k


def _masksOfClusterPair(dm, cp, ktab, symmetrize=False):
    """(keys = c1*nc + c2 with c1 <= c2, mask words) requested by one near-field cluster pair; symmetrize also requests
    the entries of the transposed pair (n2, n1) -- see partitionClusterPairs"""
    dpe = dm.dofs_per_element
    nc = dm.mesh.num_cells
    E = (2*dpe)*(2*dpe+1)//2
    dofs = dm.dofs
    in1 = np.zeros(dm.num_dofs+1, dtype=bool)
    in2 = np.zeros(dm.num_dofs+1, dtype=bool)
    in1[cp.n1.dofs] = True
    in2[cp.n2.dofs] = True
    cu = cp.cellsUnion
    d = np.where(dofs[cu] >= 0, dofs[cu], dm.num_dofs)              # [ncu, dpe]
    m1 = in1[d]                                                    # cellMasks1: local DoF in cluster 1
    m2 = in2[d]
    pos = -np.ones(nc, dtype=np.int64)
    pos[cu] = np.arange(cu.shape[0])
    c1 = np.repeat(cp.n1.cells, cp.n2.cells.shape[0])
    c2 = np.tile(cp.n2.cells, cp.n1.cells.shape[0])
    swap = c1 > c2
    a, b = np.where(swap, c2, c1), np.where(swap, c1, c2)
    pa, pb = pos[a], pos[b]
    cm1 = np.concatenate([m1[pa], m1[pb]], axis=1)                  # cellMask1 over 2 dpe local DoFs
    cm2 = np.concatenate([m2[pa], m2[pb]], axis=1)
    ok = cm1.any(axis=1) & cm2.any(axis=1)
    a, b, cm1, cm2 = a[ok], b[ok], cm1[ok], cm2[ok]
    # getElemElemSymMask: bit k(p,q), p <= q, set if p in cellMask1 and q in cellMask2
    words = np.zeros((a.shape[0], 4), dtype=np.uint64)
    for p in range(2*dpe):
        for q in range(p, 2*dpe):
            k = int(ktab[p, q])
            bit = cm1[:, p] & cm2[:, q]
            if symmetrize:
                bit = bit | (cm2[:, p] & cm1[:, q])
            words[:, k//64] |= bit.astype(np.uint64) << np.uint64(k % 64)
    assert E <= 256
    return a.astype(np.int64)*nc+b, words


def _mergeMasks(keys, masks, nc):
    keys = np.concatenate(keys)
    masks = np.concatenate(masks)
    order = np.argsort(keys, kind='stable')
    keys, masks = keys[order], masks[order]
    uniq, first = np.unique(keys, return_index=True)
    merged = np.bitwise_or.reduceat(masks, first, axis=0) if keys.shape[0] else masks
    pairs = np.stack([uniq//nc, uniq % nc], axis=1).astype(np.int32)
    return np.ascontiguousarray(pairs), np.ascontiguousarray(merged)


def iterMasksForClusters(dm, Pnear, maxNNZ=10000000, symmetrize=False):
    """yields (pairs[np, 2] with c1 <= c2, masks[np, 4] uint64) for consecutive groups of cluster pairs holding about
    maxNNZ element pairs each -- the reference's chunked loop NA:1786-1791 over buildMasksForClusters NA:260-391
    (symmetric cells and local matrix): which entries of the symmetric local matrix of each element pair are requested
    by some cluster pair of the group (OR-merged)."""
    nc = dm.mesh.num_cells
    ktab = elemElemSymMaskTable(dm.dofs_per_element)
    keys, masks, n = [], [], 0
    for cp in Pnear:
        k, m = _masksOfClusterPair(dm, cp, ktab, symmetrize)
        keys.append(k)
        masks.append(m)
        n += k.shape[0]
        if n > maxNNZ:
            yield _mergeMasks(keys, masks, nc)
            keys, masks, n = [], [], 0
    if keys:
        yield _mergeMasks(keys, masks, nc)


def _masksOfClusterPairNonsym(dm, cp):
    """(keys = c1*nc + c2 over ORDERED pairs, mask words) requested by one near-field cluster pair for a non-symmetric local matrix:
    buildMasksForClusters with useSymmetricCells == symmetricLocalMatrix == False (NA:322-349) walks cellsUnion x cellsUnion, keeps
    the pairs whose 2 dpe local DoFs hold a DoF of n1 and a DoF of n2, and sets bit p (2 dpe) + q when local DoF p lies in n1 and
    local DoF q in n2 (getElemElemMask NA:425-440)."""
    dpe = dm.dofs_per_element
    n2 = 2*dpe
    assert n2*n2 <= 256
    nc = dm.mesh.num_cells
    in1 = np.zeros(dm.num_dofs+1, dtype=bool)
    in2 = np.zeros(dm.num_dofs+1, dtype=bool)
    in1[cp.n1.dofs] = True
    in2[cp.n2.dofs] = True
    cu = np.asarray(cp.cellsUnion)
    d = np.where(dm.dofs[cu] >= 0, dm.dofs[cu], dm.num_dofs)
    m1, m2 = in1[d], in2[d]                                       # [ncu, dpe]
    a1, a2 = m1.any(axis=1), m2.any(axis=1)
    ok = (a1[:, None] | a1[None, :]) & (a2[:, None] | a2[None, :])
    ii, jj = np.nonzero(ok)
    cm1 = np.concatenate([m1[ii], m1[jj]], axis=1)                # cellMask1 over the 2 dpe local DoFs of (c1, c2)
    cm2 = np.concatenate([m2[ii], m2[jj]], axis=1)
    bits = (cm1[:, :, None] & cm2[:, None, :]).reshape(ii.shape[0], n2*n2)
    full = np.zeros((ii.shape[0], 256), dtype=np.uint8)
    full[:, :n2*n2] = bits
    words = np.packbits(full, axis=1, bitorder='little').view(np.uint64).reshape(-1, 4)
    return cu[ii].astype(np.int64)*nc+cu[jj], words


def iterMasksForClustersNonsym(dm, Pnear, maxNNZ=10000000):
    """yields (pairs[np, 2] ORDERED, masks[np, 4] uint64 over the (2 dpe)^2 local entries) for consecutive groups of cluster pairs
    of about maxNNZ element pairs: the chunked loop NA:1786-1791 for non-symmetric kernels (masks of one group OR-merged)"""
    nc = dm.mesh.num_cells
    keys, masks, n = [], [], 0
    for cp in Pnear:
        k, m = _masksOfClusterPairNonsym(dm, cp)
        keys.append(k)
        masks.append(m)
        n += k.shape[0]
        if n > maxNNZ:
            yield _mergeMasks(keys, masks, nc)
            keys, masks, n = [], [], 0
    if keys:
        yield _mergeMasks(keys, masks, nc)


def buildMasksForClusters(dm, Pnear, symmetrize=False):
    """all cluster pairs in one group"""
    out = list(iterMasksForClusters(dm, Pnear, maxNNZ=1 << 62, symmetrize=symmetrize))
    if not out:
        return np.zeros((0, 2), dtype=np.int32), np.zeros((0, 4), dtype=np.uint64)
    return out[0]


def boundaryFacetsOfCells(mesh, cellIds):
    """facets (vertex ids, oriented as in their cell) that belong to exactly one cell of the set
    (nonlocalAssembly.pyx:540-578 in 2D, :505-533 in 1D)"""
    cells = mesh.cells[cellIds].astype(np.int64)
    nv = mesh.num_vertices
    if mesh.manifold_dim == 2:
        e = np.stack([cells[:, [0, 1]], cells[:, [1, 2]], cells[:, [2, 0]]], axis=1).reshape(-1, 2)
        keys = e.min(axis=1)*nv+e.max(axis=1)
        uniq, first, counts = np.unique(keys, return_index=True, return_counts=True)
        return e[np.sort(first[counts == 1])].astype(np.int32)
    ids, counts = np.unique(cells.reshape(-1), return_counts=True)
    return ids[counts == 1].astype(np.int32).reshape(-1, 1)


def clusterBoundaryItems(dm, Pnear, symmetrize=False):
    """work list of the cluster-local Gauss-theorem term (NA:1842-1889): for every near-field pair with common cells,
    (cell in cellsInter) x (facet of the boundary of cellsUnion) with the mask of the local entries whose DoFs lie in
    (n1, n2) -- getElemSymMaskCluster NA:463-478.  Returns cells[ni], facets[ni, dim], masks[ni] (uint32 bit field over
    the dpe(dpe+1)/2 entries)."""
    mesh = dm.mesh
    dpe = dm.dofs_per_element
    out_c, out_f, out_m = [], [], []
    # items are not OR-merged like the element pairs: symmetrise only the pairs whose transposed pair is not in the list
    have = {(id(cp.n1), id(cp.n2)) for cp in Pnear}
    for cp in Pnear:
        ci = cp.cellsInter
        if ci.shape[0] == 0:
            continue
        sym = symmetrize and (id(cp.n2), id(cp.n1)) not in have
        facets = boundaryFacetsOfCells(mesh, cp.cellsUnion)
        in1 = np.zeros(dm.num_dofs+1, dtype=bool)
        in2 = np.zeros(dm.num_dofs+1, dtype=bool)
        in1[cp.n1.dofs] = True
        in2[cp.n2.dofs] = True
        d = np.where(dm.dofs[ci] >= 0, dm.dofs[ci], dm.num_dofs)
        m1, m2 = in1[d], in2[d]
        mask = np.zeros(ci.shape[0], dtype=np.uint32)
        k = 0
        for p in range(dpe):
            for q in range(p, dpe):
                bit = m1[:, p] & m2[:, q]
                if sym:
                    bit = bit | (m2[:, p] & m1[:, q])
                mask |= (bit.astype(np.uint32) << np.uint32(k))
                k += 1
        keep = mask != 0
        ci, mask = ci[keep], mask[keep]
        out_c.append(np.repeat(ci, facets.shape[0]))
        out_f.append(np.tile(facets, (ci.shape[0], 1)))
        out_m.append(np.repeat(mask, facets.shape[0]))
    if not out_c:
        return np.zeros(0, dtype=np.int32), np.zeros((0, mesh.dim), dtype=np.int32), np.zeros(0, dtype=np.uint32)
    return (np.concatenate(out_c).astype(np.int32), np.ascontiguousarray(np.concatenate(out_f), dtype=np.int32),
            np.concatenate(out_m).astype(np.uint32))


def _clusterCellMasks(dm, cp, ci, sym):
    """getElemSymMaskCluster NA:463-478 for the cells ci: bit k of the dpe(dpe+1)/2 local entries (p <= q) is set when
    DoF p lies in n1 and DoF q in n2 (or the other way round when the transposed pair is folded in)"""
    dpe = dm.dofs_per_element
    in1 = np.zeros(dm.num_dofs+1, dtype=bool)
    in2 = np.zeros(dm.num_dofs+1, dtype=bool)
    in1[cp.n1.dofs] = True
    in2[cp.n2.dofs] = True
    d = np.where(dm.dofs[ci] >= 0, dm.dofs[ci], dm.num_dofs)
    m1, m2 = in1[d], in2[d]
    mask = np.zeros(ci.shape[0], dtype=np.uint32)
    k = 0
    for p in range(dpe):
        for q in range(p, dpe):
            bit = m1[:, p] & m2[:, q]
            if sym:
                bit = bit | (m2[:, p] & m1[:, q])
            mask |= (bit.astype(np.uint32) << np.uint32(k))
            k += 1
    return mask


def facetTables(mesh):
    """fv[nc, nf, dim]: the facets of every cell (vertex ids, oriented as in the cell: the normal (dy, -dx) points out of it),
    keys[nc, nf]: orientation-free facet keys, nbr[nc, nf]: the cell on the other side or -1"""
    cells = mesh.cells.astype(np.int64)
    nc, nv = cells.shape[0], mesh.num_vertices
    if mesh.manifold_dim == 2:
        fv = np.stack([cells[:, [0, 1]], cells[:, [1, 2]], cells[:, [2, 0]]], axis=1)
        keys = fv.min(axis=2)*nv+fv.max(axis=2)
    else:
        fv = cells[:, :, None]
        keys = cells.copy()
    nf = keys.shape[1]
    flat = keys.reshape(-1)
    order = np.argsort(flat, kind='stable')
    sk = flat[order]
    i = np.nonzero(sk[1:] == sk[:-1])[0]
    nbr = np.full(flat.shape[0], -1, dtype=np.int64)
    nbr[order[i]] = order[i+1]//nf
    nbr[order[i+1]] = order[i]//nf
    return fv, keys, nbr.reshape(nc, nf)


def variableBoundaryItems(dm, Pnear, T, zeroExterior=True, symmetrize=False, clusterBoundary=True, globalBoundary=True):
    """Boundary items of the near field of a piecewise-constant variable order, grouped by (kernel class, sign):

    (i)   cluster exterior (NA:2007-2049): (cell of cellsInter) x (facet of the surface of cellsUnion).  The reference moves
          the facet centre by evalShift along the outer normal (dy, -dx) before the kernel parameters are evaluated
          (NA:2034-2042): the order is the one between the cell and the region OUTSIDE the facet -- here the label of the
          cell across the facet, or the label of the domain-boundary facet.
    (ii)  interfaces of the order (NA:2050-2124): every interface facet (getKernelBlocksAndJumps NA:2354-2384) none of whose
          two cells lies in cellsUnion, against every cell of cellsInter: +1 with the order of the region the normal points
          to, -1 with the order of the region behind it (in 1D the sign follows the side the cell lies on, NA:2079-2083).
    (iii) without zeroExterior the global Omega x Omega^c term with -1 (NA:2126-2156), the order between the cell and the
          boundary facet.

    Returns a list of (class, fac, cells[ni], facets[ni, dim], masks[ni])."""
    mesh = dm.mesh
    dim = mesh.dim
    L = np.asarray(T.cell_labels, dtype=np.int64)
    cls_of = np.asarray(T.cls_of, dtype=np.int64)
    fv, keys, nbr = facetTables(mesh)
    nc, nf = nbr.shape
    nv = mesh.num_vertices
    # label of the region outside every (cell, facet)
    outlab = L[np.maximum(nbr, 0)]
    bc = np.asarray(T.bcells, dtype=np.int64).reshape(-1, dim)
    if bc.shape[0]:
        bkeys = bc.min(axis=1)*nv+bc.max(axis=1) if dim == 2 else bc[:, 0]
        o = np.argsort(bkeys)
        cb, jb = np.nonzero(nbr < 0)
        pos = np.searchsorted(bkeys[o], keys[cb, jb])
        assert (bkeys[o][pos] == keys[cb, jb]).all()
        outlab[cb, jb] = np.asarray(T.facet_labels, dtype=np.int64)[o[pos]]
    # interfaces: facets between cells of different labels, taken from the side of the lower cell number (the reference compares
    # the cells' own orders, NA:2326-2329, 2360; items whose two classes coincide cancel and are dropped below)
    jc, jj = np.nonzero((nbr > np.arange(nc)[:, None]) & (L[:, None] != outlab))
    jn = nbr[jc, jj]
    jfv = fv[jc, jj]                                                  # [nj, dim]
    groups = {}

    def add(k, fac, cells, facets, masks):
        # k: class per item
        for kk in np.unique(k):
            sel = k == kk
            g = groups.setdefault((int(kk), float(fac)), ([], [], []))
            g[0].append(cells[sel])
            g[1].append(facets[sel])
            g[2].append(masks[sel])

    xc = mesh.vertices[mesh.cells].mean(axis=1)
    inU = np.zeros(nc, dtype=bool)
    have = {(id(cp.n1), id(cp.n2)) for cp in Pnear}
    for cp in (Pnear if clusterBoundary else []):
        ci = cp.cellsInter
        if ci.shape[0] == 0:
            continue
        sym = symmetrize and (id(cp.n2), id(cp.n1)) not in have
        mask = _clusterCellMasks(dm, cp, ci, sym)
        keep = mask != 0
        ci, mask = ci[keep].astype(np.int64), mask[keep]
        if ci.shape[0] == 0:
            continue
        U = np.asarray(cp.cellsUnion, dtype=np.int64)
        inU[U] = True
        nb = nbr[U]
        uc, uj = np.nonzero((nb < 0) | ~inU[np.maximum(nb, 0)])
        sf, sl = fv[U[uc], uj], outlab[U[uc], uj]
        ns = sf.shape[0]
        add(cls_of[L[ci][:, None], sl[None, :]].reshape(-1), 1., np.repeat(ci, ns), np.tile(sf, (ci.shape[0], 1)), np.repeat(mask, ns))
        js = np.nonzero(~inU[jc] & ~inU[jn])[0]
        inU[U] = False
        if js.shape[0]:
            nj = js.shape[0]
            kfar = cls_of[L[ci][:, None], L[jn[js]][None, :]]          # the region the normal of the facet (as in cell jc) points to
            knear = cls_of[L[ci][:, None], L[jc[js]][None, :]]
            if dim == 1:
                xv = mesh.vertices[jfv[js, 0], 0]
                sgn = np.where((xc[ci, 0][:, None] < xv[None, :]) == (xc[jc[js], 0] < xv)[None, :], 1., -1.)
            else:
                sgn = np.ones(kfar.shape)
            cells = np.repeat(ci, nj)
            facets = np.tile(jfv[js], (ci.shape[0], 1))
            masks = np.repeat(mask, nj)
            differ = (kfar != knear).reshape(-1)                        # equal classes cancel
            for s in (1., -1.):
                sel = differ & (sgn.reshape(-1) == s)
                if sel.any():
                    add(kfar.reshape(-1)[sel], s, cells[sel], facets[sel], masks[sel])
                    add(knear.reshape(-1)[sel], -s, cells[sel], facets[sel], masks[sel])
    if not zeroExterior and globalBoundary and bc.shape[0]:
        cells, facets, masks = globalBoundaryItems(dm, T.bcells)
        nb_ = bc.shape[0]
        fl = np.tile(np.asarray(T.facet_labels, dtype=np.int64), cells.shape[0]//nb_)
        add(cls_of[L[cells], fl], -1., cells, facets, masks)
    out = []
    for (k, fac), (c, f, m) in sorted(groups.items()):
        out.append((k, fac, np.concatenate(c).astype(np.int32), np.ascontiguousarray(np.concatenate(f), dtype=np.int32),
                    np.concatenate(m).astype(np.uint32)))
    return out


def globalBoundaryItems(dm, bcells):
    """every cell x every facet of the domain boundary with all entries requested (getElemSymMask NA:170-186): the
    global Omega x Omega^c term of NA:1896-1913 / 1945-1964"""
    nc, nb = dm.mesh.num_cells, bcells.shape[0]
    dpe = dm.dofs_per_element
    ok = dm.dofs >= 0
    mask = np.zeros(nc, dtype=np.uint32)
    k = 0
    for p in range(dpe):
        for q in range(p, dpe):
            mask |= (ok[:, p] & ok[:, q]).astype(np.uint32) << np.uint32(k)
            k += 1
    keep = np.nonzero(mask)[0].astype(np.int32)
    cells = np.repeat(keep, nb)
    facets = np.tile(np.ascontiguousarray(bcells, dtype=np.int32), (keep.shape[0], 1))
    return cells, facets, np.repeat(mask[keep], nb)


def allLeafPairs(dm, maxLevels, minSize=1):
    """every (leaf, leaf) pair of a tree refined maxLevels times: cluster pairs covering all matrix blocks, the set-up of
    the reference's dense-vs-cluster test (tests/test_nearField.py:131-163)"""
    root = getTree(dm)
    for _ in range(maxLevels):
        for n in list(root.leaves()):
            n.refine(minSize, maxLevels+1)
    leaves = list(root.leaves())
    Pnear = [nearFieldClusterPair(c, d) for c in leaves for d in leaves]
    for cp in Pnear:
        cp.set_cells()
    return root, Pnear


def dofClusterNode(dm, dofs, d2c=None):
    """a tree node holding the given DoFs and the cells of their supports, without boxes (getEntryCluster NA:2317-2360)"""
    n = tree_node.__new__(tree_node)
    n.parent, n.children, n.levelNo = None, [], 0
    n.dofs = np.sort(np.asarray(dofs, dtype=np.int32))
    n.box = None
    if d2c is None:
        _, d2c = getDoFBoxesAndCells(dm)
    n._boxes = n._coords = None
    n._d2c = d2c
    n._cells = None
    return n


def singleDoFClusters(dm):
    """[({I}, {I}) for every DoF I]: the cluster pairs of getDiagonalCluster (NA:2291-2309)"""
    _, d2c = getDoFBoxesAndCells(dm)
    out = []
    for I in range(dm.num_dofs):
        n = dofClusterNode(dm, [I], d2c)
        cp = nearFieldClusterPair(n, n)
        cp.set_cells()
        out.append(cp)
    return out


def partitionClusterPairs(Pnear, size):
    """Row-sharding of the near field over `size` ranks (the reference hangs one subtree per rank under the
    root, clusterMethodCy.pyx:1854-1896, and assembles the pairs whose row cluster n1 descends from it, NA:3247-3260):
    the cluster pairs are grouped by their row cluster n1, the groups are kept in tree order and cut into `size`
    contiguous parts of about equal work (|cells(n1)| * |cells(n2)| element pairs).  Returns a list of index arrays.

    A rank stores only the blocks n1 x n2 of its pairs (unsymmetric CSR).  The symmetric masked scatter writes an entry
    and its mirror image; the mirror image of a block entry lies in the block n2 x n1, which another rank may own, where
    it is dropped by the pattern (addToEntry semantics).  Each rank therefore builds its masks with symmetrize=True:
    an entry (p, q), p <= q, of an element pair is requested if (p in n1, q in n2) OR (p in n2, q in n1); the pattern
    keeps exactly the writes into the rank's own blocks, so every block is complete and the operator is the sum of
    the rank-local matrices."""
    order, seen = [], {}
    for k, cp in enumerate(Pnear):
        key = id(cp.n1)
        if key not in seen:
            seen[key] = len(order)
            order.append((int(cp.n1.dofs[0]) if cp.n1.dofs.shape[0] else 0, key))
    # tree order of the row clusters = order of first appearance in the recursion, which walks the tree depth first
    weight = np.zeros(len(order))
    group = np.zeros(len(Pnear), dtype=np.int64)
    for k, cp in enumerate(Pnear):
        g = seen[id(cp.n1)]
        group[k] = g
        weight[g] += float(cp.n1.cells.shape[0])*float(cp.n2.cells.shape[0])
    cum = np.cumsum(weight)
    total = cum[-1] if cum.shape[0] else 0.
    owner_of_group = np.minimum((cum-0.5*weight)/max(total, 1e-300)*size, size-1).astype(np.int64)
    owner = owner_of_group[group]
    return [np.nonzero(owner == r)[0] for r in range(size)]


# ---------------------------------------------------------------------------------------------------------------------
class nearFieldPlan:
    """Work lists of the tiled near-field assembly (the GPU's own decomposition of assembleClusters, NA:1663-1964).

    The reference records per element pair a 256-bit mask of requested entries (buildMasksForClusters NA:260-391) and
    scatters entry by entry.  Here every UNORDERED cluster pair {n1, n2} is processed once as a set of tiles

        (64-cell chunk of n1.cells) x (64-cell chunk of n2.cells)

    that accumulate the cross blocks of their ordered element pairs (X in n1.cells, Y in n2.cells) in an LDS sub-block
    whose rows / columns are the chunk's DoFs that belong to n1 / n2; a sub-block entry (I, J) is then written to the
    near-field matrix at (I, J) and (J, I).  An element pair with both cells in both cell sets is evaluated in both orders
    (each order feeds different entries); for n1 == n2 the unordered pairs are evaluated once and the symmetric write does
    the rest.  Contributions to the diagonal block of a cell X (both DoFs on X) are needed for X in cellsInter only and
    are summed per (cluster pair, cell) in a buffer D: an ordered pair (X, Y) adds X's part if X is in cellsInter, and Y's
    part if Y is in cellsInter and the pair (Y, X) is not enumerated itself (X not in n2.cells); the cluster-local
    Gauss-theorem term (NA:1842-1889) goes to the same buffer.  D is scattered at the end to the DoF pairs {I, J} of the
    cell that belong to {n1, n2}.  Element pairs that touch (singular rules) are listed per cluster pair and scattered
    entry-wise under the same membership rule -- that rule is all that is left of the masks."""

    def __init__(self, dm, Pnear, tile=64, maxChunkDofs=None):
        mesh = dm.mesh
        dpe = dm.dofs_per_element
        if maxChunkDofs is None:
            maxChunkDofs = 56 if dpe <= 3 else 96           # keeps two workgroups of the tile kernel per CU (LDS)
        nc = mesh.num_cells
        N = dm.num_dofs
        self.dm, self.tile = dm, tile
        # ---- distinct nodes, unordered pairs --------------------------------------------------------------------------
        nodes, nid = [], {}
        for cp in Pnear:
            for n in (cp.n1, cp.n2):
                if id(n) not in nid:
                    nid[id(n)] = len(nodes)
                    nodes.append(n)
        seen, pairs = set(), []
        for cp in Pnear:
            a, b = nid[id(cp.n1)], nid[id(cp.n2)]
            key = (min(a, b), max(a, b))
            if key in seen:
                continue
            seen.add(key)
            pairs.append((a, b, cp))
        self.nodes = nodes
        self.pair_nodes = np.array([(a, b) for a, b, _ in pairs], dtype=np.int32).reshape(-1, 2)
        self.node_off = np.zeros(len(nodes)+1, dtype=np.int32)
        self.node_off[1:] = np.cumsum([n.dofs.shape[0] for n in nodes])
        self.node_dofs = np.concatenate([n.dofs for n in nodes]).astype(np.int32) if nodes else np.zeros(0, dtype=np.int32)
        if _use_native():
            return self._build_native(dm, nodes, pairs, tile, maxChunkDofs)
        # ---- chunks of every node's cell list ---------------------------------------------------------------------------
        dofs = dm.dofs
        chunk_cells, chunk_ndof, chunk_dofl, chunk_slot = [], [], [], []
        self.node_chunk_off = np.zeros(len(nodes)+1, dtype=np.int32)
        cen = mesh.vertices[mesh.cells].mean(axis=1)
        lo, ext = cen.min(axis=0), np.maximum(cen.max(axis=0)-cen.min(axis=0), 1e-300)
        g = np.minimum(((cen-lo)/ext*65535.).astype(np.uint64), np.uint64(65535))
        morton = np.zeros(nc, dtype=np.uint64)
        for bit in range(16):
            for d in range(mesh.dim):
                morton |= ((g[:, d] >> np.uint64(bit)) & np.uint64(1)) << np.uint64(mesh.dim*bit+d)
        member = np.zeros(N+1, dtype=bool)                          # scratch flags, reset after every use (no O(N) work per node)
        for i, n in enumerate(nodes):
            # chunks of spatially compact cells (Morton order of the centres) share more DoFs: smaller LDS sub-blocks
            cells = n.cells[np.argsort(morton[n.cells], kind='stable')]
            member[n.dofs] = True
            # a chunk closes at `tile` cells or when one more cell would bring it above maxChunkDofs distinct DoFs (LDS budget)
            dall = dofs[cells]
            okall = (dall >= 0) & member[np.where(dall >= 0, dall, N)]
            member[n.dofs] = False
            s0 = 0
            while s0 < cells.shape[0]:
                s1 = min(s0+tile, cells.shape[0])
                while True:
                    cc = cells[s0:s1]
                    d, ok = dall[s0:s1], okall[s0:s1]
                    u = np.unique(d[ok])
                    if u.shape[0] <= maxChunkDofs or s1-s0 <= 1:
                        break
                    s1 = s0+max(1, (s1-s0)*maxChunkDofs//u.shape[0])
                slot = np.full((tile, dpe), -1, dtype=np.int16)
                slot[:cc.shape[0]][ok] = np.searchsorted(u, d[ok]).astype(np.int16)
                pad = np.full(tile, -1, dtype=np.int32)
                pad[:cc.shape[0]] = cc
                chunk_cells.append(pad)
                chunk_ndof.append(u.shape[0])
                chunk_dofl.append(u.astype(np.int32))
                chunk_slot.append(slot.T.copy())                        # [dpe, tile]
                s0 = s1
            self.node_chunk_off[i+1] = len(chunk_cells)
        self.nU = max(chunk_ndof) if chunk_ndof else 1
        nchunks = len(chunk_cells)
        self.chunk_cells = np.array(chunk_cells, dtype=np.int32).reshape(nchunks, tile)
        self.chunk_ndof = np.array(chunk_ndof, dtype=np.int32)
        self.chunk_dofs = np.zeros((nchunks, self.nU), dtype=np.int32)
        for k, u in enumerate(chunk_dofl):
            self.chunk_dofs[k, :u.shape[0]] = u
        self.chunk_slot = np.array(chunk_slot, dtype=np.int16).reshape(nchunks, dpe, tile)
        # ---- per pair: diagonal-block buffer slots, tiles, touching element pairs, boundary items --------------------------
        adj = _cellAdjacency(mesh)
        tA, tB, tP, tF, dsA, dsB = [], [], [], [], [], []
        sing = [[], [], []]                                               # by number of shared vertices - 1: (k, c1, c2)
        bt_slot, bt_cell, bt_facet = [], [], []
        pair_facets = []
        self.pair_foff = np.zeros(len(pairs)+1, dtype=np.int32)
        self.pair_dbase = np.zeros(len(pairs)+1, dtype=np.int64)
        d_cell, d_pair = [], []
        nV = mesh.dim+1
        pos = np.full(nc, -1, dtype=np.int64)                       # scratch arrays over the cells, reset after every pair
        in2 = np.zeros(nc, dtype=bool)
        ptr, idx = adj
        for k, (a, b, cp) in enumerate(pairs):
            n1, n2 = nodes[a], nodes[b]
            sym = a == b
            inter = np.intersect1d(n1.cells, n2.cells)
            dbase = int(self.pair_dbase[k])
            self.pair_dbase[k+1] = dbase+inter.shape[0]
            d_cell.append(inter)
            d_pair.append(np.full(inter.shape[0], k, dtype=np.int32))
            pos[inter] = dbase+np.arange(inter.shape[0])
            ca = np.arange(self.node_chunk_off[a], self.node_chunk_off[a+1])
            cb = np.arange(self.node_chunk_off[b], self.node_chunk_off[b+1])
            A, B = np.meshgrid(ca, cb, indexing='ij')
            A, B = A.reshape(-1), B.reshape(-1)
            if sym:
                keep = A <= B
                A, B = A[keep], B[keep]
            tA.append(A)
            tB.append(B)
            tP.append(np.full(A.shape[0], k, dtype=np.int32))
            tF.append(np.full(A.shape[0], 1 if sym else 0, dtype=np.int32))
            cellsA, cellsB = self.chunk_cells[A], self.chunk_cells[B]
            dsA.append(np.where(cellsA >= 0, pos[np.maximum(cellsA, 0)], -1))
            dsB.append(np.where(cellsB >= 0, pos[np.maximum(cellsB, 0)], -1))
            # touching element pairs {X, Y}, X in n1.cells, Y in n2.cells (folded, unique)
            in2[n2.cells] = True
            cnt = ptr[n1.cells+1]-ptr[n1.cells]
            X = np.repeat(n1.cells, cnt)
            # neighbours of all cells of n1 in one gather: position within the concatenated adjacency lists
            Y = idx[np.repeat(ptr[n1.cells]-(np.cumsum(cnt)-cnt), cnt)+np.arange(int(cnt.sum()))]
            m = in2[Y]
            in2[n2.cells] = False
            X, Y = X[m], Y[m]
            lo, hi = np.minimum(X, Y), np.maximum(X, Y)
            key = np.unique(lo.astype(np.int64)*nc+hi)
            lo, hi = (key//nc).astype(np.int32), (key % nc).astype(np.int32)
            common = (mesh.cells[lo][:, :, None] == mesh.cells[hi][:, None, :]).sum(axis=(1, 2))
            common = np.where(lo == hi, nV, common)
            for c in range(1, nV+1):
                mm = common == c
                if mm.any():
                    sing[c-1].append(np.stack([np.full(int(mm.sum()), k, dtype=np.int32), lo[mm], hi[mm]], axis=1))
            # cluster-local Gauss-theorem term: cells of cellsInter x boundary facets of cellsUnion; the device loops over the
            # facets per cell and skips the touching (cell, facet) pairs, which are listed here
            self.pair_foff[k+1] = self.pair_foff[k]
            if inter.shape[0]:
                union = np.union1d(n1.cells, n2.cells)
                facets = boundaryFacetsOfCells(mesh, union)
                pair_facets.append(facets)
                self.pair_foff[k+1] += facets.shape[0]
                cv = mesh.cells[inter]                                  # [m, nV]
                cand = np.nonzero(np.isin(cv, facets).any(axis=1))[0]
                if cand.shape[0]:
                    touch = (cv[cand][:, None, :, None] == facets[None, :, None, :]).any(axis=(2, 3))   # [ncand, nf]
                    ci, fi = np.nonzero(touch)
                    bt_slot.append(pos[inter[cand[ci]]])
                    bt_cell.append(inter[cand[ci]])
                    bt_facet.append(facets[fi])
            pos[inter] = -1
        cat = lambda L, dt, shape=None: (np.concatenate(L).astype(dt) if L else np.zeros((0,)+(shape or ()), dtype=dt))
        self.tile_chunkA, self.tile_chunkB = cat(tA, np.int32), cat(tB, np.int32)
        self.tile_pair, self.tile_flags = cat(tP, np.int32), cat(tF, np.int32)
        self.tile_dslotA = cat(dsA, np.int32, (tile,)).reshape(-1, tile)
        self.tile_dslotB = cat(dsB, np.int32, (tile,)).reshape(-1, tile)
        self.sing_items = [cat(s, np.int32, (3,)).reshape(-1, 3) for s in sing]
        self.d_cell, self.d_pair = cat(d_cell, np.int32), cat(d_pair, np.int32)
        self.num_dslots = int(self.pair_dbase[-1])
        self.fvid = cat(pair_facets, np.int32, (mesh.dim,)).reshape(-1, mesh.dim)
        self.bt_slot, self.bt_cell = cat(bt_slot, np.int32), cat(bt_cell, np.int32)
        self.bt_facet = cat(bt_facet, np.int32, (mesh.dim,)).reshape(-1, mesh.dim)
        # heavy tiles (more real cells) first
        if self.tile_chunkA.shape[0]:
            w = (self.chunk_cells[self.tile_chunkA] >= 0).sum(axis=1)*(self.chunk_cells[self.tile_chunkB] >= 0).sum(axis=1)
            order = np.argsort(-w, kind='stable')
            for name in ('tile_chunkA', 'tile_chunkB', 'tile_pair', 'tile_flags', 'tile_dslotA', 'tile_dslotB'):
                setattr(self, name, np.ascontiguousarray(getattr(self, name)[order]))

    def _build_native(self, dm, nodes, pairs, tile, maxChunkDofs):
        """the same lists from csrc/pnl_plan.hip (pnl_nfplan_build)"""
        import ctypes as C
        from . import _lib
        L = _lib.load()
        mesh = dm.mesh
        dpe, dim = dm.dofs_per_element, mesh.dim
        T = getattr(nodes[0], '_T', None) if nodes else None
        if T is not None:
            T.load_cells([n._k for n in nodes if isinstance(n, native_node)])
        node_cells = [np.asarray(n.cells, dtype=np.int32) for n in nodes]
        node_cell_off = np.zeros(len(nodes)+1, dtype=np.int64)
        node_cell_off[1:] = np.cumsum([c.shape[0] for c in node_cells])
        cells_cat = np.ascontiguousarray(np.concatenate(node_cells) if node_cells else np.zeros(0), dtype=np.int32)
        node_off = np.ascontiguousarray(self.node_off, dtype=np.int64)
        node_dofs = np.ascontiguousarray(self.node_dofs, dtype=np.int32)
        verts = np.ascontiguousarray(mesh.vertices, dtype=np.float64)
        mcells = np.ascontiguousarray(mesh.cells, dtype=np.int32)
        dofs = np.ascontiguousarray(dm.dofs, dtype=np.int32)
        pn = np.ascontiguousarray(self.pair_nodes, dtype=np.int32)
        h = C.c_void_p()
        rc = L.pnl_nfplan_build(dim, mesh.num_vertices, verts.ctypes.data, mesh.num_cells, mcells.ctypes.data, dpe, dm.num_dofs,
                                dofs.ctypes.data, len(nodes), node_off.ctypes.data, node_dofs.ctypes.data, node_cell_off.ctypes.data,
                                cells_cat.ctypes.data, pn.shape[0], pn.ctypes.data, int(tile), int(maxChunkDofs), C.byref(h))
        if rc:
            raise RuntimeError('pnl_nfplan_build failed: {}'.format(rc))
        try:
            sz = np.zeros(9, dtype=np.int64)
            L.pnl_nfplan_sizes(h, sz.ctypes.data)
            nchunks, nU, ntiles = int(sz[0]), int(sz[1]), int(sz[2])

            def get(which, shape, dt=np.int32):
                a = np.zeros(shape, dtype=dt)
                if a.size:
                    L.pnl_nfplan_get(h, which, a.ctypes.data)
                return a
            self.nU = nU
            self.node_chunk_off = get(0, len(nodes)+1)
            self.chunk_cells = get(1, (nchunks, tile))
            self.chunk_ndof = get(2, nchunks)
            self.chunk_dofs = get(3, (nchunks, nU))
            self.chunk_slot = get(4, (nchunks, dpe, tile), np.int16)
            self.tile_chunkA, self.tile_chunkB = get(5, ntiles), get(6, ntiles)
            self.tile_pair, self.tile_flags = get(7, ntiles), get(8, ntiles)
            self.tile_dslotA, self.tile_dslotB = get(9, (ntiles, tile)), get(10, (ntiles, tile))
            self.sing_items = [get(11+s, (int(sz[3+s]), 3)) for s in range(3)]
            self.num_dslots = int(sz[6])
            self.d_cell, self.d_pair = get(14, self.num_dslots), get(15, self.num_dslots)
            self.pair_foff = get(16, pn.shape[0]+1)
            self.fvid = get(17, (int(sz[7]), dim))
            self.bt_slot, self.bt_cell = get(18, int(sz[8])), get(19, int(sz[8]))
            self.bt_facet = get(20, (int(sz[8]), dim))
            self.pair_dbase = None
        finally:
            L.pnl_nfplan_destroy(h)

    @property
    def num_pairs(self):
        return self.pair_nodes.shape[0]


def _cellAdjacency(mesh):
    """CSR (ptr, idx) of the cells sharing at least one vertex with a cell (the cell itself included)"""
    nc, nV = mesh.num_cells, mesh.cells.shape[1]
    v = mesh.cells.reshape(-1)
    c = np.repeat(np.arange(nc), nV)
    order = np.argsort(v, kind='stable')
    v, c = v[order], c[order]
    vptr = np.zeros(mesh.num_vertices+1, dtype=np.int64)
    np.add.at(vptr, v+1, 1)
    vptr = np.cumsum(vptr)
    out_c, out_n = [], []
    deg = vptr[1:]-vptr[:-1]
    # pairs (cell, neighbour) through every shared vertex
    for d in np.unique(deg):
        vs = np.nonzero(deg == d)[0]
        if d == 0:
            continue
        block = c[(vptr[vs][:, None]+np.arange(d)[None, :])]           # [nvs, d]
        out_c.append(np.repeat(block, d, axis=1).reshape(-1))
        out_n.append(np.tile(block, (1, d)).reshape(-1))
    cc, nn = np.concatenate(out_c), np.concatenate(out_n)
    key = np.unique(cc.astype(np.int64)*nc+nn)
    cc, nn = key//nc, key % nc
    ptr = np.zeros(nc+1, dtype=np.int64)
    np.add.at(ptr, cc+1, 1)
    return np.cumsum(ptr), nn.astype(np.int64)
