"""nonlocalBuilder: the reference's assembly front end, backed by libpnl_hip.so.

Mirrors /root/reference/nl/PyNucleus_nl/nonlocalAssembly_{SCALAR}.pxi:878-3223
(class nonlocalBuilder: __init__ :879-901, setKernel :911-975, getDense :1262-1473) and the
module-level helpers nonlocalAssembly.pyx:362-372 (assembleNonlocalOperator).  The host side
only prepares tables and work lists; classification, quadrature and scatter run on the GPU.
There is no CPU fallback: a missing HIP library or GPU raises.
"""
import numpy as np
from .local_matrix import nonlocalTables
from . import _lib


class FakePLogger:
    """base/PyNucleus_base/performanceLogger.pyx: values + timers, no-op unless inspected"""

    def __init__(self):
        self.values = {}
        self.timings = {}

    def addValue(self, key, value):
        self.values[key] = value

    def addTimer(self, key, seconds):
        self.timings[key] = self.timings.get(key, 0.)+seconds


def tile_cells(dpe, dim=2):
    """cells per block of the GPU tile kernel (mirrors TILE_P1 / TILE_P2 in csrc/pnl_hip.hip)"""
    return 32 if (dpe == 6 or (dim == 1 and dpe == 3)) else 64


def upper_tiles(num_cells, T):
    """all block-tile pairs (ta <= tb), heavy near-diagonal ones first"""
    nb = (num_cells+T-1)//T
    return np.array([(a, a+d) for d in range(nb) for a in range(nb-d)], dtype=np.int32).reshape(-1, 2)


def tiles_of_rank(num_cells, T, rank, size):
    return np.ascontiguousarray(upper_tiles(num_cells, T)[rank::size])


def cell_range_of_rank(num_cells, rank, size):
    """the reference's MPI split of cellNo1 (NA:1280-1285)"""
    return int(np.ceil(num_cells*rank/size)), int(np.ceil(num_cells*(rank+1)/size))


def block_rows_of_rank(num_blocks, rank, size, costs=None):
    """contiguous range [a0, a1) of cell blocks owned by `rank`: block row a of the upper block triangle holds
    num_blocks - a tiles, the ranges hold equal numbers of tiles (tree_node.partition, clusterMethodCy.pyx:1854-1896, hangs
    one subtree per rank under the root; here the rows of the dense block are dealt by work)"""
    if costs is not None:
        # ranges of equal estimated WORK (Context.block_row_costs: tiles weighted by the kernel that takes them + per-cell work)
        cum = np.concatenate([[0.], np.cumsum(np.asarray(costs, dtype=np.float64))])
        cut = lambda k: int(np.searchsorted(cum, cum[-1]*k/float(size), side='left')) if 0 < k < size else (0 if k <= 0 else num_blocks)
        a0, a1 = cut(rank), cut(rank+1)
        return min(a0, num_blocks), min(max(a1, a0), num_blocks)
    cut = lambda k: int(round(num_blocks*(1.-np.sqrt(max(0., 1.-k/float(size))))))
    a0, a1 = cut(rank), (num_blocks if rank == size-1 else cut(rank+1))
    return min(a0, num_blocks), min(max(a1, a0), num_blocks)


def row_slab_of_rank(dm, T, rank, size, costs=None):
    """(cell_begin, cell_end, tiles, row DoFs, column DoFs) of the rank's one-sided row slab (include/pnl_hip.h,
    pnl_set_row_slab): rows = DoFs of its cells and of the cells touching them, columns = DoFs of its cells and of all
    later cells (+ the rows)."""
    mesh = dm.mesh
    nc = mesh.num_cells
    nb = (nc+T-1)//T
    a0, a1 = block_rows_of_rank(nb, rank, size, costs)
    c0, c1 = min(a0*T, nc), min(a1*T, nc)
    tiles = np.array([(a, b) for a in range(a0, a1) for b in range(a, nb)], dtype=np.int32).reshape(-1, 2)
    dofs = np.asarray(dm.dofs)
    cells = np.asarray(mesh.cells)
    if c1 <= c0:
        return c0, c1, tiles, np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32)
    mine = np.zeros(nc, dtype=bool)
    mine[c0:c1] = True
    vert = np.zeros(mesh.num_vertices, dtype=bool)
    vert[cells[c0:c1].ravel()] = True
    touching = vert[cells].any(axis=1) | mine
    rows = np.unique(dofs[touching].ravel())
    rows = rows[rows >= 0].astype(np.int32)
    cols = np.unique(np.concatenate([dofs[c0:].ravel(), rows]))
    cols = cols[cols >= 0].astype(np.int32)
    return c0, c1, tiles, rows, cols


def block_dof_count(dofs, T):
    """largest number of distinct DoFs of T consecutive cells: the side of the LDS sub-block a tile accumulates into"""
    nc = dofs.shape[0]
    worst = 0
    for b0 in range(0, nc, T):
        d = dofs[b0:b0+T].ravel()
        worst = max(worst, np.unique(d[d >= 0]).shape[0])
    return worst


def morton_order(centers):
    """cell permutation along a Z-order curve of the cell centres (16 bits per coordinate)"""
    c = np.asarray(centers, dtype=np.float64)
    lo, hi = c.min(axis=0), c.max(axis=0)
    q = np.minimum(((c-lo)/np.where(hi > lo, hi-lo, 1.)*65535.).astype(np.uint64), 65535)
    if c.shape[1] == 1:
        return np.argsort(q[:, 0], kind='stable')
    key = np.zeros(c.shape[0], dtype=np.uint64)
    for bit in range(16):
        for d in range(c.shape[1]):
            key |= ((q[:, d] >> np.uint64(bit)) & np.uint64(1)) << np.uint64(bit*c.shape[1]+d)
    return np.argsort(key, kind='stable')


def with_cell_locality(dm):
    """The tile kernels accumulate into an LDS sub-block whose side is the number of DoFs of a block of consecutive cells,
    so cells must be numbered with spatial locality (refined meshes are; meshes straight out of a generator need not be).
    Returns dm itself, or a shallow copy whose cells (and rows of dm.dofs) are renumbered along a Morton curve when that
    shrinks the blocks.  DoF numbers are unchanged; the operator changes only through the orientation of touching pairs (which
    cell comes first in the singular rule), i.e. at the quadrature error of those rules (1e-8 relative)."""
    import copy
    T = tile_cells(dm.dofs_per_element, dm.mesh.dim)
    mesh = dm.mesh
    if mesh.num_cells <= T:
        return dm
    n0 = block_dof_count(dm.dofs, T)
    if n0 <= (56 if dm.dofs_per_element <= 3 else 96):
        return dm
    perm = morton_order(mesh.getCellCenters())
    if block_dof_count(dm.dofs[perm], T) >= n0:
        return dm
    mesh2 = copy.copy(mesh)
    mesh2.cells = np.ascontiguousarray(mesh.cells[perm])
    mesh2.resetMeshInfo()
    dm2 = copy.copy(dm)
    dm2.mesh = mesh2
    dm2.dofs = np.ascontiguousarray(dm.dofs[perm])
    dm2.cell_permutation = perm                                  # new cell number -> cell number of the caller's mesh
    return dm2


def label_blocks(dm, labels, max_dofs=None):
    """Cell blocks that follow the interfaces of a piecewise-constant order.

    A tile of the dense path is uniform -- no per-pair classification, the fast kernels -- only if its two blocks of T consecutive
    cells carry one label each; a block that straddles an interface makes ALL its tiles multi-class.  Here every straddling block
    is split into one block per label: the fragments of consecutive straddling blocks are merged while they fit (T cells and the
    DoF count of the largest ordinary block, so that the LDS sub-blocks keep their size), and every new block is filled up to T
    cells with zero-volume copies of its own first cell (in-mesh padding: csrc/pnl_hip.hip finalize()).  Returns a shallow copy
    of dm whose mesh has the renumbered + padded cells (DoF numbers unchanged; dofs = -1 and volume 0 for the copies), or dm
    itself when no block straddles."""
    import copy
    T = tile_cells(dm.dofs_per_element, dm.mesh.dim)
    mesh = dm.mesh
    nc = mesh.num_cells
    labels = np.asarray(labels)
    dofs = np.asarray(dm.dofs)
    nb = (nc+T-1)//T
    if max_dofs is None:
        max_dofs = block_dof_count(dofs, T)
    order, is_dummy = [], []
    pend = {}                                   # label -> (cells, set of DoFs) of the open merged fragment

    def close(lab):
        cells, _ = pend.pop(lab)
        order.extend(cells)
        is_dummy.extend([False]*len(cells))
        pad = T-len(cells)
        order.extend([cells[0]]*pad)
        is_dummy.extend([True]*pad)

    nsplit = 0
    for b in range(nb):
        c0, c1 = b*T, min(nc, (b+1)*T)
        lb = labels[c0:c1]
        if (lb == lb[0]).all() and c1-c0 == T:
            # an ordinary block ends the run of straddling blocks: open fragments are closed first (keeps them compact)
            for lab in sorted(pend):
                close(lab)
            order.extend(range(c0, c1))
            is_dummy.extend([False]*T)
            continue
        if (lb == lb[0]).all():
            # the short block behind the last full one: stays where it is (the library pads behind the last cell)
            for lab in sorted(pend):
                close(lab)
            order.extend(range(c0, c1))
            is_dummy.extend([False]*(c1-c0))
            continue
        nsplit += 1
        for lab in np.unique(lb):
            frag = [c for c in range(c0, c1) if labels[c] == lab]
            fd = set(int(g) for g in dofs[frag].ravel() if g >= 0)
            if lab in pend:
                cells, dset = pend[lab]
                if len(cells)+len(frag) <= T and len(dset | fd) <= max_dofs:
                    pend[lab] = (cells+frag, dset | fd)
                    continue
                close(lab)
            pend[lab] = (frag, fd)
        # a label that this block does not hold ends its run: its fragment is closed
        for lab in sorted(set(pend)-set(int(x) for x in np.unique(lb))):
            close(lab)
    tail_short = nc % T != 0 and len(order) % T != 0
    for lab in sorted(pend):
        close(lab)
    if nsplit == 0:
        return dm
    order = np.asarray(order, dtype=np.int64)
    is_dummy = np.asarray(is_dummy, dtype=bool)
    if tail_short:
        # the short last block must stay last: move it behind the blocks closed after it
        k = (nc//T)*T
        pos = int(np.nonzero(order == k)[0][0])
        ntail = nc-k
        idx = np.r_[0:pos, pos+ntail:order.shape[0], pos:pos+ntail]
        order, is_dummy = order[idx], is_dummy[idx]
    mesh2 = copy.copy(mesh)
    mesh2.cells = np.ascontiguousarray(mesh.cells[order])
    mesh2.resetMeshInfo()
    mesh2._compute()
    mesh2._info['volVector'] = np.where(is_dummy, 0., mesh2._info['volVector'])
    dm2 = copy.copy(dm)
    dm2.mesh = mesh2
    d2 = np.array(dofs[order], copy=True)
    d2[is_dummy] = -1
    dm2.dofs = np.ascontiguousarray(d2)
    dm2.cell_permutation = order
    dm2.cell_is_padding = is_dummy
    dm2.num_split_blocks = nsplit
    return dm2


class nonlocalBuilder:
    def __init__(self, dm, kernel, params={}, zeroExterior=True, comm=None, PLogger=None, dm2=None, device=None, **kwargs):
        if 'boundary' in kwargs:
            zeroExterior = kwargs.pop('boundary')           # deprecated alias, NA:888-890
        # two DoFMaps (NA:879-901, 1366-1375): the rows are the DoFs of dm, the columns those of dm2 -- like the reference, the operator
        # is assembled over the combined map (dm.combine(dm2): the DoFs of dm first) and the block rows x columns is kept
        self._dm_pair = None
        if dm2 is not None:
            self._dm_pair = (int(dm.num_dofs), int(dm2.num_dofs))
            dm = dm.combine(dm2)
        self.PLogger = PLogger if PLogger is not None else FakePLogger()
        self.comm = comm
        self.params = dict(params)
        # cell numbers are internal to the assembly (cluster cells and pair lists refer to self.mesh)
        dm = with_cell_locality(dm) if self.params.get('reorderCells', True) else dm
        self.dm = dm
        self.mesh = dm.mesh
        self.device = device
        self._ctx = None
        self._geom_cache = {}
        self.setKernel(kernel, zeroExterior)

    def setKernel(self, kernel, zeroExterior=True):
        assert kernel.dim == self.dm.mesh.dim, "Kernel dimension must match dm.mesh dimension"
        self.kernel = kernel
        # NA:919-922
        self.zeroExterior = False if kernel.finiteHorizon else bool(zeroExterior)
        self.tables = nonlocalTables(self.dm, kernel, self.params, self.zeroExterior)
        self._uploaded = False
        self._blocked = None
        # operators of the previous kernel keep their data but must not re-run a device set-up with the new tables
        if getattr(self, '_ctx', None) is not None:
            self._ctx._kernel_epoch = getattr(self._ctx, '_kernel_epoch', 0)+1
            self._ctx._h2_owner = None

    # ------------------------------------------------------------------
    def _device_index(self):
        import torch
        if not torch.cuda.is_available():
            raise _lib.PnlError('no GPU visible: the nonlocal assembly path has no CPU implementation '
                                '(the CPU oracle under oracle/ is test infrastructure only)')
        if self.device is not None:
            return int(self.device)
        return torch.cuda.current_device()

    def context(self):
        import torch
        if self._ctx is None:
            self._ctx = _lib.Context(self._device_index())
        if not self._uploaded:
            self._ctx.upload_tables(self.tables)
            self._uploaded = True
        self._ctx.set_stream(torch.cuda.current_stream(self._ctx.device).cuda_stream)
        return self._ctx

    def _blocked_dense(self):
        """(context, number of cells) of the dense path of a piecewise-constant variable order in 2D on one GPU: its own context
        over the label-following cell blocks of label_blocks() -- the same DoFs, the same tables, the same operator; the tile
        kernels meet (nearly) no multi-class tile.  None when it does not apply (one label, no straddling block, 1D, comm,
        params['labelBlocks'] = False)."""
        T = self.tables
        if (not self.params.get('labelBlocks', True) or getattr(T, 'pointwise', False) or not T.classes or T.num_labels < 2
                or self.mesh.dim != 2 or self.comm is not None):
            return None
        if getattr(self, '_blocked', None) is None:
            import copy
            dmb = label_blocks(self.dm, T.cell_labels)
            if dmb is self.dm:
                self._blocked = False
                return None
            Tb = copy.copy(T)
            Tb.dm = dmb
            Tb.cell_labels = np.ascontiguousarray(np.asarray(T.cell_labels)[dmb.cell_permutation], dtype=np.int32)
            Tb.classes = []
            for c in T.classes:
                cb = copy.copy(c)
                cb.dm = dmb
                Tb.classes.append(cb)
            ctx = _lib.Context(self._device_index())
            ctx.upload_tables(Tb)
            self._blocked = (ctx, dmb, Tb)
        if self._blocked is False:
            return None
        import torch
        ctx, dmb, Tb = self._blocked
        ctx.set_stream(torch.cuda.current_stream(ctx.device).cuda_stream)
        return ctx, dmb.mesh.num_cells

    def dense_context(self):
        """the context getDense() assembles on (its kernel timers / counters): the label-blocked twin when there is one"""
        blocked = self._blocked_dense()
        return blocked[0] if blocked is not None else self.context()

    def _single_order_twin(self):
        """A variable-order kernel whose order takes one value (varconst) has one kernel block and no jumps
        (getKernelBlocksAndJumps, NA:2312-2384): its near field / H2 operator is the one of the constant-order kernel with
        the same near-field quadrature orders.  Returns that builder, or None."""
        T = self.tables
        if getattr(T, 'pointwise', False) or not self.kernel.variable or not T.classes or len(T.classes) != 1:
            return None
        if getattr(self, '_twin', None) is None:
            self._twin = nonlocalBuilder(self.dm, T.classes[0].kernel, self.params, zeroExterior=self.zeroExterior, comm=self.comm,
                                         PLogger=self.PLogger, device=self.device)
        return self._twin

    def _symmetric_only(self, what):
        if getattr(self.tables, 'pointwise', False):
            raise NotImplementedError('{} for non-symmetric kernels with an order per quadrature point (getDense only)'.format(what))

    def _rank_size(self):
        if self.comm is None:
            return 0, 1
        import torch.distributed as dist
        return dist.get_rank(self.comm if self.comm is not True else None), dist.get_world_size(self.comm if self.comm is not True else None)

    # ------------------------------------------------------------------
    def getDense(self, trySparsification=False, distributed=False):
        """NA:1262-1473; with two DoFMaps the block (DoFs of dm) x (DoFs of dm2) of the operator over the combined map (NA:1366-1375)"""
        if self._dm_pair is None:
            return self._getDense(trySparsification, distributed)
        if trySparsification or distributed:
            raise NotImplementedError('two DoFMaps: getDense() on one GPU')
        from .linear_operators import Dense_LinearOperator
        full = self._getDense()
        n1, n2 = self._dm_pair
        return Dense_LinearOperator(full.A[:n1, n1:n1+n2], full.ctx, full.info, symmetric=False)

    def _getDense(self, trySparsification=False, distributed=False):
        """Assemble the dense operator on the GPU (NA:1262-1473).

        comm=None: full matrix on this GPU.  With a torch.distributed group as comm the element pairs
        are split over the ranks (tiles dealt round-robin, balanced by construction instead of the
        reference's equal cellNo1 ranges NA:1280-1285); by default the parts are all-reduced into the
        full matrix on every rank exactly like the reference (NA:1449-1450); distributed=True keeps them
        separate and all-reduces the N-vector in matvec instead."""
        import torch
        from .linear_operators import Dense_LinearOperator, DistributedDense_LinearOperator
        sparsificationThreshold = 0.8                                # NA:1274
        if trySparsification and distributed:
            raise NotImplementedError('getDense(trySparsification=True, distributed=True)')
        if (trySparsification and self._rank_size()[1] == 1 and not self.zeroExterior and self.kernel.finiteHorizon
                and self.mesh.volume*(1.-sparsificationThreshold) > self.kernel.horizonValue**self.mesh.dim):
            # NA:1287-1348: a horizon that is small against the domain -- the operator is assembled into the sparsity pattern of the
            # element pairs getPanelType does not ignore (SSS for a symmetric local matrix), which is what getSparse builds and fills
            return self.getSparse()
        if self.kernel.finiteHorizon:
            # the reference's all-pairs loop visits every pair and ignores the REMOTE ones; the same matrix is obtained from
            # the pairs within the horizon (getSparse), stored densely
            ctx = self.context()
            S = self.getSparse()
            # scattered into the dense block on the device (no N x N array on the host)
            dev = torch.device('cuda', ctx.device)
            N = self.dm.num_dofs
            ctx.synchronize()
            indptr = torch.as_tensor(S.indptr, device=dev).to(torch.int64)
            cols = torch.as_tensor(S.indices, device=dev).to(torch.int64)
            rows = torch.repeat_interleave(torch.arange(N, device=dev), indptr[1:]-indptr[:-1], output_size=int(cols.numel()))
            A = torch.zeros((N, N), dtype=torch.float64, device=dev)
            data = S.data_dev[:cols.numel()]
            A[rows, cols] = data
            if S.symmetric:
                A[cols, rows] = data
                A.diagonal().copy_(S.diag_dev)
            op = Dense_LinearOperator(A, ctx, S.info, symmetric=bool(S.symmetric))
            return self._sparsified(op, sparsificationThreshold) if trySparsification else op
        ctx = self.context()
        if getattr(ctx, '_slab_owner', None) is not None:
            ctx.set_row_slab(np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32))   # a row slab of an earlier distributed operator
            ctx._slab_owner = None
        dev = torch.device('cuda', ctx.device)
        N = self.dm.num_dofs
        nc = self.mesh.num_cells
        blocked = self._blocked_dense()
        if blocked is not None:
            ctx, nc = blocked
        pointwise = bool(getattr(self.tables, 'pointwise', False))
        self.PLogger.addValue('useSymmetricCells', not pointwise)
        self.PLogger.addValue('useSymmetricLocalMatrix', not pointwise)
        rank, size = self._rank_size()
        # the block-slot path forms the operator in its own storage and writes every entry of A in one sweep: no zero fill
        overwrites = (not pointwise) and size == 1 and ctx.dense_overwrites(0, nc)
        # rows start on 64-byte lines when this rank holds the whole block: the fold / mirror / GEMV passes move whole lines
        ldA = ((N+7) & ~7) if size == 1 else N
        A = (torch.empty if overwrites else torch.zeros)((N, ldA), dtype=torch.float64, device=dev)[:, :N]
        if pointwise:
            # non-symmetric kernel, order per quadrature point (NA:1411-1428): the reference's cellNo1 split across ranks
            start, end = cell_range_of_rank(nc, rank, size)
            ctx.assemble_dense_pointwise(A.data_ptr(), A.stride(0), self.zeroExterior, start, end)
        elif size == 1:
            ctx.assemble_dense(A.data_ptr(), A.stride(0), self.zeroExterior, 0, nc)
        elif distributed:
            # row-owned storage: this rank's one-sided slab + its partial per-cell diagonal blocks, no N x N array anywhere
            del A
            from .linear_operators import DistributedSlab_LinearOperator
            group = None if self.comm is True else self.comm
            return DistributedSlab_LinearOperator.assemble(self, rank, size, group)
        else:
            tiles = self.tiles_for_rank(rank, size)
            start, end = cell_range_of_rank(nc, rank, size)
            ctx.assemble_dense_tiles(A.data_ptr(), A.stride(0), self.zeroExterior, tiles, start, end, flags=_lib.PNL_FLAG_SYMMETRIC_FLUSH)
        cnt = ctx.counters()
        ms = ctx.phase_ms()
        for k in ('numCellPairs', 'numAssembledCellPairs', 'numIntegrations'):
            self.PLogger.addValue(k, cnt[k])
        self.PLogger.addTimer('interior', 1e-3*(ms['tiles']+ms['tiles_uniform']+ms['worklist']+ms['singular']))
        self.PLogger.addTimer('zeroExterior', 1e-3*ms['boundary'])
        info = dict(counters=cnt, phase_ms=ms)
        if size == 1:
            op = Dense_LinearOperator(A, ctx, info, symmetric=not pointwise)
            return self._sparsified(op, sparsificationThreshold) if trySparsification else op
        group = None if self.comm is True else self.comm
        op = DistributedDense_LinearOperator(A, ctx, group, info)
        return op if distributed else op.reduce()

    def _sparsified(self, op, threshold):
        """NA:1451-1469: the dense operator becomes a CSR operator when more than `threshold` of its entries are explicit zeros.  The
        reference counts zeros row by row and stops at the first row that is not mostly zero; the ratio is taken over the rows counted."""
        import torch
        from .linear_operators import CSR_LinearOperator
        op.ctx.synchronize()
        A = op.A
        N = A.shape[0]
        zeros_row = (A == 0.).sum(dim=1)
        dense_rows = torch.nonzero(zeros_row <= threshold*A.shape[1])
        nr = int(dense_rows[0].item())+1 if dense_rows.numel() else N
        ratio = float(zeros_row[:nr].sum().item())/float(nr)/float(A.shape[1])
        if not ratio > threshold:
            return op
        nz = torch.nonzero(A != 0.)                                  # CSR_LinearOperator.from_dense: row-major, sorted columns
        counts = torch.bincount(nz[:, 0], minlength=N)
        indptr = torch.zeros(N+1, dtype=torch.int32, device=A.device)
        indptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
        S = CSR_LinearOperator(indptr, nz[:, 1].to(torch.int32), N, op.ctx, A.device)
        S.data_dev[:nz.shape[0]] = A[nz[:, 0], nz[:, 1]]
        S.info = dict(op.info, sparsified_from_dense=ratio)
        return S

    def tiles_for_rank(self, rank, size):
        """block-tile pairs (ta <= tb) of the upper block triangle owned by `rank`: the list is ordered by
        block distance (heavy near-diagonal tiles first) and dealt round-robin."""
        T = tile_cells(self.dm.dofs_per_element, self.mesh.dim)
        if self._ctx is not None:
            assert self._ctx.tile_cells() == T
        return tiles_of_rank(self.mesh.num_cells, T, rank, size)

    def getDiagonal(self):
        """NA:2269-2289 / getDiagonalCluster NA:2291-2309: the diagonal through the cluster pairs ({I}, {I}) -- element
        pairs inside the support of phi_I plus the Gauss-theorem term over the boundary of the support.  One masked
        assembly on the GPU into a diagonal-only SSS pattern; returns the diagonal as a numpy vector wrapped like the
        reference's diagonalOperator (``.data``, ``.diagonal``)."""
        if getattr(self.tables, 'pointwise', False):
            # order per quadrature point: the cluster path is not built for these kernels; the diagonal of the dense operator
            # (assembled on the device, O(N^2) work instead of the reference's O(N))
            from .linear_operators import diagonalOperator
            return diagonalOperator(self.getDense().diagonal)
        if self._single_order_twin() is not None:
            return self._single_order_twin().getDiagonal()
        from . import clusters
        from .linear_operators import diagonalOperator
        Pnear = clusters.singleDoFClusters(self.dm)
        Anear = self.assembleClusters(Pnear, _globalBoundary=False, _clusterBoundary=self.zeroExterior)
        return diagonalOperator(Anear.diagonal)

    def getEntry(self, I, J):
        """NA:1538-1661: the entry A[I, J] alone: element pairs of (supp phi_I u supp phi_J)^2 and, with zeroExterior,
        the Gauss-theorem term over the boundary of that union, assembled on the GPU into a one-entry pattern."""
        if getattr(self.tables, 'pointwise', False):
            # as getDiagonal: from the dense operator on the device
            A = self.getDense()
            A.ctx.synchronize()
            return float(A.A[int(I), int(J)].item())
        if self._single_order_twin() is not None:
            return self._single_order_twin().getEntry(I, J)
        import torch
        from . import clusters
        from .linear_operators import CSR_LinearOperator
        assert not self.kernel.finiteHorizon
        dm = self.dm
        _, d2c = clusters.getDoFBoxesAndCells(dm)
        if I == J:
            n = clusters.dofClusterNode(dm, [I], d2c)
            Pnear = [clusters.nearFieldClusterPair(n, n)]
        else:
            n1, n2 = clusters.dofClusterNode(dm, [I], d2c), clusters.dofClusterNode(dm, [J], d2c)
            Pnear = [clusters.nearFieldClusterPair(n1, n2), clusters.nearFieldClusterPair(n2, n1)]
        for cp in Pnear:
            cp.set_cells()
        ctx = self.context()
        indptr = np.zeros(dm.num_dofs+1, dtype=np.int32)
        indptr[I+1:] = 1
        A = CSR_LinearOperator(indptr, np.array([J], dtype=np.int32), dm.num_dofs, ctx, torch.device('cuda', ctx.device))
        self.assembleClusters(Pnear, Anear=A, _globalBoundary=False, _clusterBoundary=self.zeroExterior)
        return float(A.data[0])

    # -- the reference's remaining builder methods (thin: the work is in clusters.py / the methods above) ----------------------------
    def getDiagonalCluster(self):
        """NA:2291-2309: the diagonal through the cluster pairs ({I}, {I}) -- what getDiagonal does here for every kernel"""
        return self.getDiagonal()

    def getEntryCluster(self, I, J):
        """NA:1475-1537: one entry through the cluster pair of the two supports -- what getEntry does here"""
        return self.getEntry(I, J)

    def getTree(self):
        """NA:2541-2664 (one rank): root of the cluster tree over the DoFs"""
        from . import clusters
        return clusters.getTree(self.dm)

    def getAdmissibleClusters(self):
        """NA:2666-2905 (one rank): (Pnear, Pfar) for the kernel's horizon and the refinement parameters of getH2RefinementParams; a
        piecewise-constant order splits the clusters by kernel block first"""
        from . import clusters
        rp = self.getH2RefinementParams()
        blk = mixed = None
        if self.kernel.variable and not getattr(self.tables, 'pointwise', False):
            blk, mixed = clusters.dofKernelBlocks(self.dm, self.tables)
        horizon = float(self.kernel.horizonValue) if self.kernel.finiteHorizon else np.inf
        root, Pnear, Pfar = clusters.getNearFieldClusters(self.dm, rp['eta'], rp['minSize'], rp['maxLevels'], blk, -1 if mixed is None else mixed,
                                                          rp['refinementType'], horizon=horizon)
        return Pnear, Pfar

    def getCoveringClusters(self):
        """NA:2907-2981 for the whole mesh as one cluster pair: Pnear = [(root, root)]"""
        from . import clusters
        return clusters.coveringCluster(self.dm)[1]

    def getKernelBlocksAndJumps(self):
        """NA:2312-2384: blocks = {order of the cells around a DoF: set of DoFs}, key INTERFACE_DOF (numpy.inf) for the DoFs on an interface
        of the order; jumps = {(cell, cell) sorted: the vertex (1D) / the sorted vertex pair (2D) they share} for neighbouring cells of
        different order (the reference encodes both as 64-bit integers)"""
        if not (self.kernel.variable and not getattr(self.tables, 'pointwise', False)):
            return {float(getattr(self.kernel, 'sValue', np.nan)): set(range(self.dm.num_dofs))}, {}
        from . import clusters
        T, dm, mesh = self.tables, self.dm, self.mesh
        lab = np.asarray(T.cell_labels)
        sv = np.asarray(self.kernel.s.sVals)
        order_of_label = np.array([sv[l, l] for l in range(sv.shape[0])])
        blk, mixed = clusters.dofKernelBlocks(dm, T)
        blocks = {}
        for d, b in enumerate(blk):
            blocks.setdefault(np.inf if b == mixed else float(order_of_label[b]), set()).add(d)
        fv, keys, nbr = clusters.facetTables(mesh)
        jumps = {}
        for c in range(mesh.num_cells):
            for j in range(nbr.shape[1]):
                c2 = int(nbr[c, j])
                if c2 > c and order_of_label[lab[c]] != order_of_label[lab[c2]]:
                    f = tuple(sorted(int(v) for v in fv[c, j]))
                    jumps[(c, c2)] = f[0] if mesh.dim == 1 else f
        return blocks, jumps

    def _interaction_vertices(self):
        """the vertices in the coordinates the interaction set is a ball in (ellipse domains: T x; otherwise the mesh's own)"""
        T = getattr(self.kernel.interaction, 'transform', None)
        v = np.asarray(self.mesh.vertices, dtype=np.float64)
        return v if T is None else np.ascontiguousarray(v @ np.asarray(T).T)

    def interactingCellPairs(self):
        """all ordered cell pairs c1 <= c2 that the horizon does not separate for sure: |centre1 - centre2| <= delta + r1 + r2
        with r = largest vertex distance from the centre.  A superset of the pairs getRelativePosition does not call
        REMOTE; the reference enumerates a superset too (cells of covering cluster pairs, NA:1150-1170, CM:4139-4194) and
        the device drops the REMOTE ones in its classification (NO:515-517), so the assembled entries are the same."""
        from scipy.spatial import cKDTree
        mesh = self.mesh
        v = self._interaction_vertices()[mesh.cells]
        cen = v.mean(axis=1)
        rad = np.sqrt(((v-cen[:, None, :])**2).sum(axis=2)).max(axis=1)
        delta = self.kernel.horizonValue
        tree = cKDTree(cen)
        pr = tree.query_pairs(delta+2.*rad.max(), output_type='ndarray')
        if pr.shape[0]:
            d = np.sqrt(((cen[pr[:, 0]]-cen[pr[:, 1]])**2).sum(axis=1))
            pr = pr[d <= delta+rad[pr[:, 0]]+rad[pr[:, 1]]]
        lo, hi = np.minimum(pr[:, 0], pr[:, 1]), np.maximum(pr[:, 0], pr[:, 1])
        diag = np.arange(mesh.num_cells)
        pairs = np.stack([np.concatenate([diag, lo]), np.concatenate([diag, hi])], axis=1)
        order = np.lexsort((pairs[:, 1], pairs[:, 0]))
        return np.ascontiguousarray(pairs[order], dtype=np.int32)

    def getSparse(self, returnNearField=False):
        """Finite-horizon operator as a sparse matrix (NA:1062-1260): every element pair within the horizon is classified
        (REMOTE / INTERACT / CUT, getRelativePosition), integrated (cut pairs through the sub-simplex loops NO:790-847) and
        scattered without masks into the sparsity pattern of all DoF pairs that share such an element pair.  Symmetric
        storage (SSS) unless params['forceUnsymmetric']."""
        if self._dm_pair is not None:
            raise NotImplementedError('two DoFMaps: getDense only')
        self._symmetric_only('getSparse')
        import torch
        import scipy.sparse as sp
        from .linear_operators import CSR_LinearOperator, SSS_LinearOperator
        if not self.kernel.finiteHorizon:
            raise NotImplementedError('getSparse needs a finite horizon; use getDense / getH2 for horizon = inf')
        ctx = self.context()
        dev = torch.device('cuda', ctx.device)
        dm = self.dm
        N, nc, dpe = dm.num_dofs, self.mesh.num_cells, dm.dofs_per_element
        symmetric = not self.params.get('forceUnsymmetric', False)
        host_pairs = returnNearField or self.params.get('pairList', 'device') == 'host' or 'maxMasksNNZ' in self.params
        # the pattern depends on the mesh, the DoF map and the horizon only: repeated assemblies reuse it (0.4 s of the 0.47 s
        # of a getSparse call at 129^2 vertices are spent in these sparse products)
        cache = getattr(self, '_sparse_pattern', None)
        if cache is not None and cache[0] == (symmetric, host_pairs, float(self.kernel.horizonValue), repr(self.kernel.interaction)):
            indptr, indices, pairs = cache[1]
            return self._assembleSparse(indptr, indices, pairs, symmetric, host_pairs, returnNearField)
        rows = np.repeat(np.arange(nc), dpe)
        d = dm.dofs.reshape(-1)
        m = d >= 0
        C = sp.csr_matrix((np.ones(int(m.sum()), dtype=np.int32), (rows[m], d[m])), shape=(nc, N))
        if host_pairs:
            # explicit candidate list (a superset of the pairs within the horizon) and the pattern of all its DoF pairs:
            # G = C^T (P + P^T) C
            pairs = self.interactingCellPairs()
            Pm = sp.csr_matrix((np.ones(pairs.shape[0], dtype=np.int32), (pairs[:, 0], pairs[:, 1])), shape=(nc, nc))
            G = (C.T @ ((Pm+Pm.T) @ C)).tocsr()
        elif self.params.get('patternBuilder', 'native') == 'native' and self.mesh.dim <= 2:
            # two cells interact unless all their vertex distances are >= delta (getRelativePosition): the pattern is the
            # set of DoF pairs whose patches hold a vertex pair closer than delta; built by libpnl_hip.so (csrc/pnl_plan.hip)
            import ctypes as Ct
            L = _lib.load()
            verts = np.ascontiguousarray(self._interaction_vertices(), dtype=np.float64)
            mcells = np.ascontiguousarray(self.mesh.cells, dtype=np.int32)
            dofs32 = np.ascontiguousarray(dm.dofs, dtype=np.int32)
            h = Ct.c_void_p()
            rc = L.pnl_horizon_pattern(self.mesh.dim, self.mesh.num_vertices, verts.ctypes.data, nc, mcells.ctypes.data, dpe, N,
                                       dofs32.ctypes.data, float(self.kernel.horizonValue), int(symmetric), Ct.byref(h))
            if rc == _lib.PNL_ERR_UNSUPPORTED:
                raise _lib.PnlError('getSparse: the pattern has more than 2^31 - 1 stored entries (INDEX_t is 32 bits); use a smaller '
                                    'horizon or mesh, or getDense')
            if rc:
                raise RuntimeError('pnl_horizon_pattern failed: {}'.format(rc))
            indptr = np.zeros(N+1, dtype=np.int32)
            indices = np.zeros(int(L.pnl_pattern_nnz(h)), dtype=np.int32)
            L.pnl_pattern_get(h, indptr.ctypes.data, indices.ctypes.data if indices.shape[0] else None)
            L.pnl_pattern_destroy(h)
            pairs = None
            self._sparse_pattern = ((symmetric, host_pairs, float(self.kernel.horizonValue), repr(self.kernel.interaction)), (indptr, indices, pairs))
            return self._assembleSparse(indptr, indices, pairs, symmetric, host_pairs, returnNearField)
        else:
            # the same pattern through scipy sparse products, G = M Q M^T with M = DoF -> patch vertices (params['patternBuilder']
            # = 'scipy': the reference implementation the native builder is tested against)
            from scipy.spatial import cKDTree
            pairs = None
            nv = self.mesh.num_vertices
            vp = cKDTree(self._interaction_vertices()).query_pairs(self.kernel.horizonValue*(1.+1e-9), output_type='ndarray')
            Q = sp.csr_matrix((np.ones(vp.shape[0], dtype=np.int32), (vp[:, 0], vp[:, 1])), shape=(nv, nv))
            Q = Q+Q.T+sp.identity(nv, dtype=np.int32, format='csr')
            B = sp.csr_matrix((np.ones(nc*self.mesh.cells.shape[1], dtype=np.int32),
                               (np.repeat(np.arange(nc), self.mesh.cells.shape[1]), self.mesh.cells.reshape(-1))), shape=(nc, nv))
            M = (C.T @ B).tocsr()
            M.data[:] = 1
            G = ((M @ Q) @ M.T).tocsr()
        if symmetric:
            G = sp.tril(G, k=-1, format='csr')
        G.sort_indices()
        indptr, indices = G.indptr.astype(np.int32), G.indices.astype(np.int32)
        self._sparse_pattern = ((symmetric, host_pairs, float(self.kernel.horizonValue), repr(self.kernel.interaction)), (indptr, indices, pairs))
        return self._assembleSparse(indptr, indices, pairs, symmetric, host_pairs, returnNearField)

    def _assembleSparse(self, indptr, indices, pairs, symmetric, host_pairs, returnNearField):
        """device part of getSparse: integrate the pairs within the horizon into the given pattern"""
        import torch
        from .linear_operators import CSR_LinearOperator, SSS_LinearOperator
        ctx = self.context()
        dev = torch.device('cuda', ctx.device)
        N = self.dm.num_dofs
        A = (SSS_LinearOperator if symmetric else CSR_LinearOperator)(indptr, indices, N, ctx, dev)
        A._bind()
        data_ptr, diag_ptr = A._ptrs()
        totals = dict(numCellPairs=0, numAssembledCellPairs=0, numIntegrations=0)
        ms_total = 0.
        if host_pairs:
            maxNNZ = int(self.params.get('maxMasksNNZ', 10000000))
            for s0 in range(0, pairs.shape[0], maxNNZ):
                ctx.assemble_pairs_masked(pairs[s0:s0+maxNNZ], None, data_ptr, diag_ptr)
                cnt = ctx.counters()
                for k in totals:
                    totals[k] += cnt[k]
                ms_total += ctx.phase_ms()['total']
        else:
            # candidate pairs are generated on the device from the block tiles the horizon can reach
            ctx.assemble_pairs_in_horizon(data_ptr, diag_ptr)
            cnt = ctx.counters()
            for k in totals:
                totals[k] = cnt[k]
            ms_total = ctx.phase_ms()['total']
        ctx.synchronize()
        for k, v in totals.items():
            self.PLogger.addValue(k, v)
        self.PLogger.addTimer('interior - compute', 1e-3*ms_total)
        A.info = dict(counters=totals, interior_ms=ms_total, num_candidate_pairs=int(pairs.shape[0]) if pairs is not None else totals['numCellPairs'])
        return (A, pairs) if returnNearField else A

    def _planner(self):
        """'host' (C++ loops, csrc/pnl_plan.hip) or 'device' (level-synchronous sweeps on the GPU, csrc/pnl_plan_dev.hip) for the cluster
        tree and the admissible pairs -- the same tree and lists either way.  params['planner']; by default the device from 2^18 DoFs
        on: at 48,769 DoFs the host loops take 4.5 ms, the sweeps 10 ms (two dozen small launches per level), and the first call of a
        process loads the sort / scan kernels (0.25 s)"""
        p = self.params.get('planner', 'auto')
        if p == 'auto':
            return 'device' if self.dm.num_dofs >= (1 << 18) else 'host'
        return p

    def getH2RefinementParams(self):
        """NA:2979-3046: eta, leaf size, depth and refinement type of the cluster tree from params.  The default leaf size follows
        the GPU tile (a leaf of about one block of cells keeps the near-field tiles full); params['minClusterSize'] = 'reference'
        takes the reference's default interpolation_order(h)^dim // 2 (NA:3016-3024), a number takes that number."""
        p = self.params
        N = self.dm.num_dofs
        mcs = p.get('minClusterSize', None)
        if mcs == 'reference':
            loggamma = abs(np.log(0.25))
            sing = self.kernel.max_singularity
            io = max(np.ceil((2*self.tables.target_order+max(-sing, 2))*abs(np.log(self.mesh.h/self.mesh.diam))/loggamma/3.), 2)
            mcs = int(io**self.mesh.dim//2)
        elif mcs is None:
            mcs = max(self.dm.dofs_per_element*4, min(64, max(N//16, 8)))
        return dict(eta=p.get('eta', 3.), maxLevels=p.get('maxLevels', 200), minSize=int(mcs),
                    refinementType=p.get('refinementType', 'MEDIAN'))

    def getH2(self, returnNearField=False, returnTree=False, **kwargs):
        """NA:3094-3219: cluster tree, admissibility, near field (assembleClusters) and the Chebyshev-interpolated far field,
        all on the GPU; returns an H2Matrix whose matvec is near-field SpMV + upward pass + interactions + downward pass.
        Without an admissible pair the dense operator is returned (the reference's assembleDenseWhenH2Fails branch).  With a
        communicator (row-sharded near field) the near-field operator alone is returned: the far field is not
        distributed yet."""
        if self._dm_pair is not None:
            raise NotImplementedError('two DoFMaps: getDense only')
        if self._single_order_twin() is not None:
            return self._single_order_twin().getH2(returnNearField, returnTree, **kwargs)
        from . import clusters
        from .h2 import h2Plan, H2Matrix, interpolationOrder
        rp = self.getH2RefinementParams()
        far_class = None
        pointwise = bool(getattr(self.tables, 'pointwise', False))
        if pointwise and self.comm is not None:
            raise NotImplementedError('distributed H2 operator of a kernel with an order per quadrature point')
        if pointwise and hasattr(self.kernel.s.sFun, 'vertex_values'):
            # the far field evaluates s at interpolation nodes, which needs the reference's cell finder for a finite element order
            raise NotImplementedError('H2 operator of an order given as a finite element function')
        if self.kernel.variable and not pointwise:
            # kernel blocks (getKernelBlocksAndJumps NA:2312-2352): clusters of one block each, the interface DoFs stay in the near
            # field; the far field between two clusters uses the order between their blocks
            # (a non-symmetric order table s(l1, l2) != s(l2, l1) changes nothing here: cluster pairs are ordered, every one takes the
            # class of its orientation; the near field runs both orientations of every element pair, pnl_assemble_pairs_masked)
            horizon = np.inf
            if self.kernel.finiteHorizon:
                if not self.tables.has_boundary_tables:
                    raise NotImplementedError('H2 operator of a finite-horizon variable order: fractional kernels with the l2 ball')
                horizon = float(self.kernel.horizonValue)
            blk, mixed = clusters.dofKernelBlocks(self.dm, self.tables)
            root, Pnear, Pfar = clusters.getNearFieldClusters(self.dm, rp['eta'], rp['minSize'], rp['maxLevels'], blk, mixed, rp['refinementType'],
                                                              horizon=horizon)
            cls_of = self.tables.cls_of

            def far_class(cp):
                return int(cls_of[blk[cp.n1.dofs[0]], blk[cp.n2.dofs[0]]])
        else:
            # tree, admissible pairs, near-field tile plan, pattern and far-field plan depend on the mesh and the refinement
            # parameters only: kept on the builder (a second operator of the same DoFMap -- another kernel through setKernel, a
            # time step -- starts with the device work)
            horizon = np.inf
            if self.kernel.finiteHorizon:
                # getAdmissibleClusters with the horizon of the l2 ball (clusterMethodCy.pyx:4069-4090): far-field pairs lie inside it
                if not self.tables.has_boundary_tables:
                    raise NotImplementedError('H2 operator of a finite horizon: fractional kernels of constant order with the l2 ball')
                horizon = float(self.kernel.horizonValue)
            key = ('tree', rp['eta'], rp['minSize'], rp['maxLevels'], rp['refinementType'], horizon)
            # ... and on the DoF map itself (at most two parameter sets): a NEW builder on the same DoF map -- another kernel, the next
            # operator of a parameter study -- finds tree, tile plan, pattern and far-field plan there and starts with the device work
            store = self.dm.__dict__.setdefault('_pnl_geom', {}) if self.params.get('cacheGeometry', True) else {}
            if self._geom_cache.get('key') != key and key in store:
                self._geom_cache = store[key]
            if self._geom_cache.get('key') != key:
                # a new builder: the library's own set-up of the mesh (padded cell tables, adjacency lists: 16-19 ms at 98,304 cells)
                # runs on a host thread while this one builds the tree -- the two share nothing (ctypes releases the GIL)
                import threading
                ctx0, err = self.context(), []

                def warm():
                    try:
                        ctx0.tile_cells()
                    except Exception as e:              # reported by the assembly call that needs the tables
                        err.append(e)
                th = threading.Thread(target=warm)
                th.start()
                try:
                    tree = clusters.getNearFieldClusters(self.dm, rp['eta'], rp['minSize'], rp['maxLevels'], refinementType=rp['refinementType'],
                                                         planner=self._planner(), horizon=horizon)
                finally:
                    th.join()
                self._geom_cache = {'key': key, 'tree': tree}
                while len(store) >= 2:
                    store.pop(next(iter(store)))
                store[key] = self._geom_cache
            root, Pnear, Pfar = self._geom_cache['tree']
        rank, size = self._rank_size()
        if size > 1 and self.kernel.finiteHorizon:
            raise NotImplementedError('distributed H2 operator of a finite horizon')
        if sum(len(v) for v in Pfar.values()) == 0:
            h2 = self.getDense()
        elif size > 1 and self.params.get('localFarFieldIndexing', False):
            # rank-local data with halo exchange (the reference's assembleOnRoot=False, localFarFieldIndexing=True,
            # DistributedH2Matrix_localData CM:3368-3920): rows owned by subtrees, ghost x entries for the near field, cluster
            # coefficients for the far field, no N-vector collective
            from .distributed_h2 import DistributedH2Matrix_localData
            m = self.params.get('interpolation_order', None)
            if m is None:
                m = interpolationOrder(self.kernel, self.mesh, self.tables.target_order)
            h2 = DistributedH2Matrix_localData(self, root, Pnear, Pfar, m, far_class, None if self.comm is True else self.comm)
        elif size > 1:
            # row-sharded near field: this rank's cluster pairs into its own unsymmetric CSR; matvec = Bcast(x), local products
            # (near field + this rank's share of the far field), Allreduce(y)
            # (DistributedH2Matrix_globalData, clusterMethodCy.pyx:3127-3154)
            from .linear_operators import DistributedSparse_LinearOperator
            mine = clusters.partitionClusterPairs(Pnear, size)[rank]
            local = self.assembleClusters([Pnear[k] for k in mine], forceUnsymmetricMatrix=True, _symmetrizeMasks=True)
            m = self.params.get('interpolation_order', None)
            if m is None:
                m = interpolationOrder(self.kernel, self.mesh, self.tables.target_order)
            # the admissible pairs are dealt round-robin (they cost the same: one M x M product each); every rank runs the
            # upward pass on the broadcast x (replicated, O(N M)), its share of the interactions and the downward pass of what
            # it computed; the all-reduce of the N-vector sums near and far parts (DistributedH2Matrix_globalData, CM:3127-3154)
            Pfar_local, k = {}, 0
            for lvl in sorted(Pfar):
                for cp in Pfar[lvl]:
                    if k % size == rank:
                        Pfar_local.setdefault(lvl, []).append(cp)
                    k += 1
            far = H2Matrix(local, h2Plan(self.dm, root, Pfar_local, m, far_class), self.context(), root, Pfar_local)
            h2 = DistributedSparse_LinearOperator(local, None if self.comm is True else self.comm, far=far, Pfar=Pfar)
        else:
            # full CSR near field by default: its SpMV needs no atomics for the transposed half (0.16 ms against 0.39 ms with
            # SSS at 49k DoFs) and HBM is not the constraint; params['forceUnsymmetric'] = False keeps the reference's SSS
            # the near field is launched first; the far-field plan is built on the host while the device assembles
            Anear = self.assembleClusters(Pnear, forceUnsymmetricMatrix=bool(self.params.get('forceUnsymmetric', True)), _defer_info=True)
            m = self.params.get('interpolation_order', None)
            if m is None:
                m = interpolationOrder(self.kernel, self.mesh, self.tables.target_order)
            pkey = ('h2plan', id(root), m, far_class is None)
            if far_class is None and self._geom_cache.get('h2plan_key') == pkey:
                plan = self._geom_cache['h2plan']
            else:
                plan = h2Plan(self.dm, root, Pfar, m, far_class)
                if far_class is None:
                    self._geom_cache['h2plan_key'], self._geom_cache['h2plan'] = pkey, plan
            if getattr(Anear, '_finish', None) is not None:
                Anear._finish()
            h2 = H2Matrix(Anear, plan, self.context(), root, Pfar)
        out = (h2,)
        if returnNearField:
            out += (Pnear,)
        if returnTree:
            out += (root,)
        return out if len(out) > 1 else out[0]

    def assembleClusters(self, Pnear, forceUnsymmetricMatrix=False, Anear=None, jumps={}, myRoot=None, _clusterBoundary=True,
                         _globalBoundary=True, _symmetrizeMasks=False, _defer_info=False, **kwargs):
        """Near-field matrix of the cluster pairs Pnear (NA:1663-1964), assembled on the GPU.

        Host side: sparsity pattern (getSparseNearField NA:3226-3289), per element-pair 256-bit entry masks
        (buildMasksForClusters NA:260-391) and the item list of the cluster-local Gauss-theorem term
        (NA:1842-1889).  Device side: classification, quadrature and masked scatter into CSR / SSS.
        Without zeroExterior the global Omega x Omega^c term is subtracted again (NA:1896-1913)."""
        if self._dm_pair is not None:
            raise NotImplementedError('two DoFMaps: getDense only')
        if getattr(self.tables, 'pointwise', False):
            return self._assembleClustersPointwise(Pnear, Anear, myRoot, _clusterBoundary, _globalBoundary)
        if self._single_order_twin() is not None:
            return self._single_order_twin().assembleClusters(Pnear, forceUnsymmetricMatrix, Anear, jumps, myRoot, _clusterBoundary,
                                                              _globalBoundary, _symmetrizeMasks, **kwargs)
        if self.kernel.variable and self.kernel.finiteHorizon and (not self.tables.has_boundary_tables or not self.kernel.symmetric):
            raise NotImplementedError('near field of a finite-horizon variable order: fractional kernels with the l2 ball and a symmetric table')
        if self.kernel.variable and not self.kernel.symmetric:
            # both orientations of every element pair write both (I, J) and (J, I) of their entries: unsymmetric storage (the operator
            # itself is symmetric only if the table is)
            forceUnsymmetricMatrix = True
        import torch
        from . import clusters
        from .linear_operators import CSR_LinearOperator, SSS_LinearOperator
        # jumps (getKernelBlocksAndJumps NA:2312-2384) are a function of the mesh and the order: derived in
        # clusters.variableBoundaryItems, the argument is accepted for the reference's call signature
        if myRoot is not None:
            # NA:3247-3260 / :1697-1712: the near field of ONE rank -- the cluster pairs whose row cluster n1 lies in the subtree
            # myRoot (the reference hangs one subtree per rank under the root, clusterMethodCy.pyx:1854-1896), stored as
            # unsymmetric CSR with complete blocks n1 x n2; writes into rows of other subtrees are dropped by the pattern
            # (NA:2174, 2240).  The operator is the sum over the subtrees.
            mine = np.zeros(self.dm.num_dofs, dtype=bool)
            mine[np.asarray(myRoot.get_dofs() if hasattr(myRoot, 'get_dofs') else myRoot.dofs)] = True
            Pnear = [cp for cp in Pnear if cp.n1.dofs.shape[0] and mine[cp.n1.dofs[0]]]
            forceUnsymmetricMatrix, _symmetrizeMasks = True, True
        ctx = self.context()
        dev = torch.device('cuda', ctx.device)
        dm = self.dm
        symmetric = not forceUnsymmetricMatrix
        cached = self._geom_cache.get('tree') is not None and Pnear is self._geom_cache['tree'][1] and myRoot is None
        if Anear is None:
            pat_key = ('pattern', symmetric, ctx.device)          # device arrays: per device
            if cached and pat_key in self._geom_cache:
                indptr, indices = self._geom_cache[pat_key]
            else:
                indptr, indices = clusters.getSparseNearField(dm, Pnear, symmetric=symmetric, device=dev)
                if cached:
                    self._geom_cache[pat_key] = (indptr, indices)
            Anear = (SSS_LinearOperator if symmetric else CSR_LinearOperator)(indptr, indices, dm.num_dofs, ctx, dev)
        Anear._bind()
        data_ptr, diag_ptr = Anear._ptrs()
        mode = self.params.get('nearFieldAssembly', 'tiles' if 'maxMasksNNZ' not in self.params else 'masks')
        if mode == 'tiles' and not self.kernel.variable and not self.kernel.finiteHorizon:
            # the GPU's own decomposition: cluster-pair tiles with LDS sub-blocks, no masks (clusters.nearFieldPlan)
            tile = ctx.tile_cells()
            if cached and self._geom_cache.get('nfplan_tile') == tile:
                plan = self._geom_cache['nfplan']
            else:
                plan = clusters.nearFieldPlan(dm, Pnear, tile=tile)
                if cached:
                    self._geom_cache['nfplan_tile'], self._geom_cache['nfplan'] = tile, plan
            use_bnd = bool(self.tables.has_boundary_tables and _clusterBoundary)
            ctx.assemble_clusters_tiled(plan, use_bnd, data_ptr, diag_ptr)
            extra = use_bnd and not self.zeroExterior and _globalBoundary

            def finish():
                # reads the counters (synchronises the stream): deferred by getH2 until its far-field plan is built
                cnt = ctx.counters()
                ms = ctx.phase_ms()
                nitems = plan.bt_cell.shape[0]
                if extra:
                    cells, facets, bmasks = clusters.globalBoundaryItems(dm, self.tables.bcells)
                    nitems += int(cells.shape[0])
                    ctx.assemble_boundary_masked(cells, facets, bmasks, -1., data_ptr, diag_ptr)
                ctx.synchronize()
                self.PLogger.addTimer('interior', 1e-3*ms['total'])
                Anear.info = dict(counters=dict(cnt, numBoundaryItems=nitems, numTiles=int(plan.tile_chunkA.shape[0])), interior_ms=ms['total'],
                                  phase_ms=ms, mode='tiles')
                Anear._finish = None
            if _defer_info:
                Anear._finish = finish
            else:
                finish()
            return Anear
        maxNNZ = int(self.params.get('maxMasksNNZ', 10000000))
        totals = dict(numCellPairs=0, numAssembledCellPairs=0, numIntegrations=0)
        hist, sing = {}, {}
        ms_total = 0.
        # element pairs in chunks of cluster pairs like the reference's maxMasksNNZ loop (NA:1786-1791)
        for pairs, masks in clusters.iterMasksForClusters(dm, Pnear, maxNNZ, symmetrize=_symmetrizeMasks):
            ctx.assemble_pairs_masked(pairs, masks, data_ptr, diag_ptr)
            cnt = ctx.counters()
            for k in totals:
                totals[k] += cnt[k]
            for q, c in cnt['orders'].items():
                hist[q] = hist.get(q, 0)+c
            for q, c in cnt['singular'].items():
                sing[q] = sing.get(q, 0)+c
            ms_total += ctx.phase_ms()['total']
        nitems = 0
        if self.tables.has_boundary_tables and self.kernel.variable:
            # piecewise-constant order: cluster exterior with the order of the region outside every facet, the interfaces of
            # the order, the global term -- one launch per (kernel class, sign) (NA:1966-2156)
            # (a finite horizon: cluster surfaces and interfaces with the TRUNCATED twins of the classes -- facets beyond the horizon give
            # zero --, no global Omega x Omega^c term to take back; what lies beyond the horizon is subtracted below)
            for k, fac, cells, facets, bmasks in clusters.variableBoundaryItems(dm, Pnear, self.tables, self.zeroExterior or self.kernel.finiteHorizon,
                                                                                _symmetrizeMasks, _clusterBoundary, _globalBoundary):
                ctx.select_class(k)
                ctx.assemble_boundary_masked(cells, facets, bmasks, fac, data_ptr, diag_ptr)
                nitems += int(cells.shape[0])
            ctx.select_class(0)
            if self.kernel.finiteHorizon and _globalBoundary:
                self._subtractBeyondHorizon(Anear)
        elif self.tables.has_boundary_tables and _clusterBoundary:
            cells, facets, bmasks = clusters.clusterBoundaryItems(dm, Pnear, symmetrize=_symmetrizeMasks)
            nitems = int(cells.shape[0])
            ctx.assemble_boundary_masked(cells, facets, bmasks, 1., data_ptr, diag_ptr)
            if self.kernel.finiteHorizon and _globalBoundary:
                # NA:1915-1940: the cluster exteriors were integrated with the full-space twin; what lies beyond the horizon is a
                # constant per point -- the surface of the ball times the twin's value at the horizon -- times the mass matrix
                self._subtractBeyondHorizon(Anear)
            elif not self.zeroExterior and _globalBoundary:
                cells, facets, bmasks = clusters.globalBoundaryItems(dm, self.tables.bcells)
                nitems += int(cells.shape[0])
                ctx.assemble_boundary_masked(cells, facets, bmasks, -1., data_ptr, diag_ptr)
        ctx.synchronize()
        for k, v in totals.items():
            self.PLogger.addValue(k, v)
        self.PLogger.addTimer('interior', 1e-3*ms_total)
        Anear.info = dict(counters=dict(totals, orders=hist, singular=sing, numBoundaryItems=nitems), interior_ms=ms_total)
        return Anear


    def _subtractBeyondHorizon(self, Anear):
        """Anear -= vol * Gamma_b(horizon) * M on the pattern of the near field (NA:1915-1940: vol = 2 in 1D, 2 pi horizon in 2D, Gamma_b the
        boundary twin of the full-space kernel, M the mass matrix with the rule of degree 2 the reference takes)"""
        import torch
        import scipy.sparse as sp
        from .quadrature import simplexXiaoGimbutas
        from .linear_operators import SSS_LinearOperator
        dm, dim, delta = self.dm, self.mesh.dim, float(self.kernel.horizonValue)
        if dim not in (1, 2):
            raise NotImplementedError('near field of a finite horizon in {}D'.format(dim))
        vol = 2. if dim == 1 else 2.*np.pi*delta
        x, y = np.zeros(dim), np.zeros(dim)
        y[0] = delta
        qr = simplexXiaoGimbutas(2, dim, dim)
        if not self.kernel.variable:
            coeff = -vol*float(self.tables.boundaryKernelFull(x, y))
            M = (coeff*dm.assembleMass(qr)).tocoo()
        else:
            # piecewise-constant order (NA:2143-2156, horizonSurfaceIntegral nonlocalAssembly.pyx:132-175): the surface of the ball by
            # 2 points (1D) / 10 points of the circle (2D), the order between the point and each of them
            T = self.tables
            gam = np.array([float(c.boundaryKernelFull(x, y)) for c in T.classes])
            mesh = self.mesh
            v = mesh.vertices[mesh.cells]
            xq = np.einsum('kn,ckd->cnd', qr.nodes, v)                          # [nc, nq, dim]
            if dim == 1:
                off, w = np.array([[delta], [-delta]]), np.array([1., 1.])
            else:
                th = 2.*np.pi*np.arange(10)/10.
                off, w = delta*np.stack([np.cos(th), np.sin(th)], axis=1), np.full(10, 2.*np.pi/10.*delta)
            sFun = self.kernel.s
            lx = np.asarray(sFun.labels(xq.reshape(-1, dim))).reshape(xq.shape[:2])
            coef = np.zeros(xq.shape[:2])
            for k in range(off.shape[0]):
                ly = np.asarray(sFun.labels((xq+off[k]).reshape(-1, dim))).reshape(xq.shape[:2])
                coef -= w[k]*gam[np.asarray(T.cls_of)[lx, ly]]
            phi = dm.evalShapeFunctions(qr.nodes)
            loc = np.einsum('cn,n,pn,qn->cpq', coef, qr.weights, phi, phi)*mesh.volVector[:, None, None]
            dpe = dm.dofs.shape[1]
            I = np.repeat(dm.dofs[:, :, None], dpe, axis=2)
            J = np.repeat(dm.dofs[:, None, :], dpe, axis=1)
            m = (I >= 0) & (J >= 0)
            M = sp.csr_matrix((loc[m], (I[m], J[m])), shape=(dm.num_dofs, dm.num_dofs)).tocoo()
        n = dm.num_dofs
        indptr, indices = np.asarray(Anear.indptr), np.asarray(Anear.indices)
        pos = sp.csr_matrix((np.arange(1, indices.shape[0]+1, dtype=np.int64), indices, indptr), shape=(n, n))
        sym = isinstance(Anear, SSS_LinearOperator)
        Anear._bind()
        dev = Anear.data_dev.device
        self.context().synchronize()
        if sym:
            off = M.row > M.col                              # SSS: strict lower triangle + diagonal
            dg = M.row == M.col
            d = np.zeros(n)
            np.add.at(d, M.row[dg], M.data[dg])
            Anear.diag_dev += torch.from_numpy(d).to(dev)
        else:
            off = np.ones(M.row.shape[0], dtype=bool)
        r, c, v = M.row[off], M.col[off], M.data[off]
        p = np.asarray(pos[r, c]).reshape(-1)
        if (p == 0).any():
            raise RuntimeError('the near field does not hold every pair of DoFs that share a cell')
        upd = torch.zeros(indices.shape[0], dtype=torch.float64)
        upd.index_add_(0, torch.from_numpy(p-1), torch.from_numpy(np.ascontiguousarray(v)))
        Anear.data_dev[:indices.shape[0]] += upd.to(dev)
        torch.cuda.current_stream(dev).synchronize()

    def _assembleClustersPointwise(self, Pnear, Anear=None, myRoot=None, clusterBoundary=True, globalBoundary=True):
        """assembleClusters for the non-symmetric kernels with an order per quadrature point (NA:1776-1840 with symmetricCells ==
        symmetricLocalMatrix == False; cluster exterior NA:1966-2028 with local_matrix_surface = the pointwise boundary kernel, no
        shift of the facet centre and no interface terms for orders of one variable, NA:1966, 2623): every ORDERED element pair of
        cellsUnion x cellsUnion with a mask over its (2 dpe)^2 local entries, evaluated in its own orientation, into unsymmetric CSR."""
        import torch
        from . import clusters
        from .linear_operators import CSR_LinearOperator
        if myRoot is not None or self.comm is not None:
            raise NotImplementedError('distributed near field of a kernel with an order per quadrature point')
        ctx = self.context()
        dev = torch.device('cuda', ctx.device)
        dm, T = self.dm, self.tables
        if Anear is None:
            indptr, indices = clusters.getSparseNearField(dm, Pnear, symmetric=False, device=dev)
            Anear = CSR_LinearOperator(indptr, indices, dm.num_dofs, ctx, dev)
        Anear._bind()
        data_ptr, diag_ptr = Anear._ptrs()
        assert diag_ptr is None, 'non-symmetric kernels need unsymmetric (CSR) storage'
        mcells = np.asarray(self.mesh.cells)
        # cluster exterior (and, for the regional operator, the global term with -1): items and the orders of the touching ones first --
        # their near rules are keyed by the pair's order and may not be among the rules of the domain boundary
        groups = []
        if T.has_boundary_tables and clusterBoundary:
            groups.append((1., clusters.clusterBoundaryItems(dm, Pnear)))
        if T.has_boundary_tables and clusterBoundary and not self.zeroExterior and globalBoundary:
            groups.append((-1., clusters.globalBoundaryItems(dm, T.bcells)))
        prepared = []
        for fac, (cells, facets, bmasks) in groups:
            sv = np.maximum(T.cell_smax[cells], T.facet_order(facets)) if cells.shape[0] else np.zeros(0)
            common = (mcells[cells][:, :, None] == facets[:, None, :]).any(axis=2).sum(axis=1) if cells.shape[0] else np.zeros(0, dtype=np.int64)
            prepared.append((fac, cells, facets, bmasks, sv, common > 0))
        touching_sv = [sv[t] for _, _, _, _, sv, t in prepared if sv.shape[0]]
        if touching_sv and T.need_boundary_keys(np.concatenate(touching_sv)):
            ctx.upload_pointwise_rules(T)
        keys, bkeys = ctx._pw_keys
        maxNNZ = int(self.params.get('maxMasksNNZ', 10000000))
        totals = dict(numCellPairs=0, numAssembledCellPairs=0, numIntegrations=0)
        hist, sing = {}, {}
        ms_total = 0.
        for pairs, masks in clusters.iterMasksForClustersNonsym(dm, Pnear, maxNNZ):
            touching = (mcells[pairs[:, 0]][:, :, None] == mcells[pairs[:, 1]][:, None, :]).any(axis=(1, 2))
            rule = np.full(pairs.shape[0], -1, dtype=np.int32)
            svp = np.maximum(T.cell_smax[pairs[touching, 0]], T.cell_smax[pairs[touching, 1]])
            k = np.searchsorted(keys, svp)
            assert (k < keys.shape[0]).all() and (keys[np.minimum(k, keys.shape[0]-1)] == svp).all(), 'touching pair without a near rule'
            rule[touching] = k
            ctx.assemble_pairs_masked_pointwise(pairs, masks, rule, data_ptr)
            cnt = ctx.counters()
            for kk in totals:
                totals[kk] += cnt[kk]
            for q, c in cnt['orders'].items():
                hist[q] = hist.get(q, 0)+c
            for q, c in cnt['singular'].items():
                sing[q] = sing.get(q, 0)+c
            ms_total += ctx.phase_ms()['total']
        nitems = 0
        for fac, cells, facets, bmasks, sv, touching in prepared:
            if cells.shape[0] == 0:
                continue
            rule = np.full(cells.shape[0], -1, dtype=np.int32)
            k = np.searchsorted(bkeys, sv[touching])
            assert (k < bkeys.shape[0]).all() and (bkeys[np.minimum(k, bkeys.shape[0]-1)] == sv[touching]).all()
            rule[touching] = k
            ctx.assemble_boundary_masked_pointwise(cells, facets, bmasks, rule, sv, fac, data_ptr, None)
            nitems += int(cells.shape[0])
        ctx.synchronize()
        self.PLogger.addValue('useSymmetricCells', False)
        self.PLogger.addValue('useSymmetricLocalMatrix', False)
        for kk, v in totals.items():
            self.PLogger.addValue(kk, v)
        self.PLogger.addTimer('interior', 1e-3*ms_total)
        Anear.info = dict(counters=dict(totals, orders=hist, singular=sing, numBoundaryItems=nitems), interior_ms=ms_total, mode='masks')
        return Anear


def assembleNonlocalOperator(mesh, dm, s, horizon=None, params={}, zeroExterior=True, comm=None, **kwargs):
    """nonlocalAssembly.pyx:362-372"""
    from .kernels import getFractionalKernel
    kernel = getFractionalKernel(mesh.dim, s, horizon)
    builder = nonlocalBuilder(dm, kernel, params, zeroExterior, comm, **kwargs)
    return builder.getDense()
