"""Device-resident linear operators returned by the builder.

Host mirror of the reference's Dense_LinearOperator
(/root/reference/base/PyNucleus_base/DenseLinearOperator_{SCALAR}.pxi:8-95: ``data``,
``matvec`` -> dgemv, ``toarray``, ``diagonal``) and of the CG + Jacobi pair used by the
drivers (base/PyNucleus_base/solvers.pyx:229-245, 363-444).  The matrix lives in HBM
(torch is used for the allocation only); matvec and CG run in libpnl_hip.so.
"""
import numpy as np
import torch


def _as_dev(x, device):
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=torch.float64).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x), dtype=np.float64)).to(device)


class Dense_LinearOperator:
    def __init__(self, A_dev, ctx, info=None, symmetric=False):
        # row-major with a leading dimension >= number of columns (the builder pads rows to 64-byte lines)
        assert A_dev.dtype == torch.float64 and A_dev.stride(1) == 1 and A_dev.stride(0) >= A_dev.shape[1]
        self.A = A_dev
        self.ctx = ctx
        self.num_rows, self.num_columns = A_dev.shape
        self.shape = (self.num_rows, self.num_columns)
        self.info = info or {}
        # a symmetric operator is applied from its upper triangle alone (4 N^2 bytes per product instead of 8 N^2); set by the builder
        # for symmetric kernels only -- the block itself always holds the full matrix, like the reference's
        self.symmetric = bool(symmetric) and self.num_rows == self.num_columns

    # reference API -------------------------------------------------------
    @property
    def data(self):
        """the matrix as a numpy array: a snapshot of the device block, copied from HBM once and again only after the context
        ran another dense assembly (the block may have been assembled into again; Context.assembly_epoch) or after
        invalidate() / refresh() -- indexing A.data[i, j] in a loop does not move 8 N^2 bytes per access"""
        epoch = getattr(self.ctx, 'assembly_epoch', 0)
        if getattr(self, '_host', None) is None or self._host_epoch != epoch:
            self.ctx.synchronize()
            self._host, self._host_epoch = self.A.cpu().numpy(), epoch
        return self._host

    def invalidate(self):
        """drop the host snapshot (the device block was changed outside the context's assembly calls)"""
        self._host = None

    def refresh(self):
        self.invalidate()
        return self.data

    def toarray(self):
        return self.data

    @property
    def diagonal(self):
        self.ctx.synchronize()
        return torch.diagonal(self.A).cpu().numpy().copy()

    def matvec(self, x, y=None):
        """y = A x through the hand-written GEMV; accepts numpy or torch vectors"""
        dev = self.A.device
        xd = _as_dev(x, dev)
        assert xd.shape[0] == self.num_columns
        yd = torch.empty(self.num_rows, dtype=torch.float64, device=dev)
        torch.cuda.current_stream(dev).synchronize()
        if self.num_rows == self.num_columns:
            self.ctx.gemv(self.A.data_ptr(), self.A.stride(0), self.num_rows, xd.data_ptr(), yd.data_ptr(), 2 if self.symmetric else 0)
        else:
            # a rectangular block (two DoFMaps): y = 1 * A x + 0 * b
            self.ctx.gemv_axpby(self.A.data_ptr(), self.A.stride(0), self.num_rows, self.num_columns, xd.data_ptr(), 1., 0., None, yd.data_ptr())
        self.ctx.synchronize()
        if isinstance(x, torch.Tensor):
            if y is not None:
                y.copy_(yd)
                return y
            return yd
        out = yd.cpu().numpy()
        if y is not None:
            y[:] = out
            return y
        return out

    def __mul__(self, x):
        return self.matvec(x)

    dot = matvec

    def solve_cg_jacobi(self, b, x0=None, tol=1e-8, maxiter=1000):
        """Jacobi-preconditioned CG in HBM; returns (x, iterations, residual in the preconditioner norm)"""
        dev = self.A.device
        bd = _as_dev(b, dev)
        xd = torch.zeros_like(bd) if x0 is None else _as_dev(x0, dev).clone()
        torch.cuda.current_stream(dev).synchronize()
        its, res = self.ctx.cg_jacobi(self.A.data_ptr(), self.A.stride(0), self.num_rows, bd.data_ptr(), xd.data_ptr(), tol, maxiter)
        if isinstance(b, torch.Tensor):
            return xd, its, res
        return xd.cpu().numpy(), its, res

    # operator files: the reference's layout (DenseLinearOperator_{SCALAR}.pxi:86-94) on any h5py-like group (create_dataset,
    # attrs, item access); h5py itself is not a dependency of this package
    def HDF5write(self, node):
        node.create_dataset('data', data=np.ascontiguousarray(self.data))
        node.attrs['type'] = 'dense'

    @staticmethod
    def HDF5read(node, ctx, device=None):
        """the operator back in HBM, bound to the context ``ctx`` (a builder's ``context()``)"""
        dev = torch.device('cuda', ctx.device) if device is None else device
        A = torch.from_numpy(np.array(node['data'], dtype=np.float64)).to(dev)
        return Dense_LinearOperator(A, ctx)

    def __repr__(self):
        return '<{}x{} Dense_LinearOperator on {}>'.format(self.num_rows, self.num_columns, self.A.device)


class DistributedDense_LinearOperator(Dense_LinearOperator):
    """A = sum over ranks of the locally assembled parts; matvec = local GEMV + all-reduce of the
    N-vector (the reference's DistributedH2Matrix_globalData scheme, clusterMethodCy.pyx:3127-3154:
    Bcast(x), local matvec, Allreduce(y)) instead of all-reducing the N^2 matrix (NA:1449-1450)."""

    def __init__(self, A_dev, ctx, comm_group=None, info=None):
        super().__init__(A_dev, ctx, info)
        self.group = comm_group

    def matvec(self, x, y=None):
        import torch.distributed as dist
        dev = self.A.device
        xd = _as_dev(x, dev)
        dist.broadcast(xd, src=0, group=self.group)
        yd = torch.empty(self.num_rows, dtype=torch.float64, device=dev)
        torch.cuda.current_stream(dev).synchronize()
        self.ctx.gemv(self.A.data_ptr(), self.A.stride(0), self.num_rows, xd.data_ptr(), yd.data_ptr())
        self.ctx.synchronize()
        dist.all_reduce(yd, group=self.group)
        if isinstance(x, torch.Tensor):
            return yd
        return yd.cpu().numpy()

    def reduce(self):
        """all-reduce the matrix itself (what the reference's getDense does, NA:1449-1450)"""
        import torch.distributed as dist
        self.ctx.synchronize()
        dist.all_reduce(self.A, group=self.group)
        return Dense_LinearOperator(self.A, self.ctx, self.info)


class DistributedSlab_LinearOperator:
    """Row-owned distributed dense operator: every rank keeps the one-sided slab of its block rows,
    rows x (N - col0) doubles <= N^2 / P, and its partial per-cell diagonal blocks; matvec = local slab products + ONE
    all-reduce of the N-vector (DistributedH2Matrix_globalData.matvec, clusterMethodCy.pyx:3127-3154).  No collective in
    the assembly."""

    def __init__(self, slab, dblocks, rowdofs, coldofs, num_dofs, ctx, group, info=None):
        self.slab, self.dblocks = slab, dblocks
        self.rowdofs, self.coldofs = np.ascontiguousarray(rowdofs, dtype=np.int32), np.ascontiguousarray(coldofs, dtype=np.int32)
        self.num_rows = self.num_columns = int(num_dofs)
        self.shape = (self.num_rows, self.num_columns)
        self.ctx, self.group = ctx, group
        self.device = slab.device
        self.info = info or {}

    @classmethod
    def assemble(cls, builder, rank, size, group):
        from .builder import row_slab_of_rank, tile_cells
        ctx = builder.context()
        dm = builder.dm
        dev = torch.device('cuda', ctx.device)
        T = tile_cells(dm.dofs_per_element, builder.mesh.dim)
        nblocks = (dm.mesh.num_cells+T-1)//T
        c0, c1, tiles, rows, cols = row_slab_of_rank(dm, T, rank, size, ctx.block_row_costs(nblocks))
        N = dm.num_dofs
        ncols = max(int(cols.shape[0]), 1)
        slab = torch.zeros((max(rows.shape[0], 1), ncols), dtype=torch.float64, device=dev)
        dblocks = torch.zeros(ctx.diag_blocks_size(), dtype=torch.float64, device=dev)
        op = cls(slab, dblocks, rows, cols, N, ctx, group)
        if rows.shape[0]:
            ctx.set_row_slab(rows, cols)
            ctx._slab_owner = op
            ctx.assemble_dense_tiles(slab.data_ptr(), slab.stride(0), builder.zeroExterior, tiles, c0, c1)
            ctx.get_diag_blocks(dblocks.data_ptr())
            op.info = dict(counters=ctx.counters(), phase_ms=ctx.phase_ms(), cell_range=(c0, c1), num_tiles=int(tiles.shape[0]),
                           slab_rows=int(rows.shape[0]), slab_cols=int(ncols), slab_bytes=int(rows.shape[0])*int(ncols)*8)
        else:
            op.info = dict(counters=dict(numAssembledCellPairs=0, numIntegrations=0), cell_range=(c0, c1), num_tiles=0, slab_rows=0,
                           slab_cols=int(ncols), slab_bytes=0)
        return op

    def _bind(self):
        if getattr(self.ctx, '_slab_owner', None) is not self and self.rowdofs.shape[0]:
            self.ctx.set_row_slab(self.rowdofs, self.coldofs)
            self.ctx._slab_owner = self

    def _local(self, xd):
        yd = torch.zeros(self.num_rows, dtype=torch.float64, device=self.device)
        if self.rowdofs.shape[0]:
            self._bind()
            torch.cuda.current_stream(self.device).synchronize()
            self.ctx.slab_matvec(self.slab.data_ptr(), self.slab.stride(0), self.dblocks.data_ptr(), xd.data_ptr(), yd.data_ptr())
            self.ctx.synchronize()
        return yd

    def matvec(self, x, y=None):
        import torch.distributed as dist
        xd = _as_dev(x, self.device).contiguous()
        if dist.is_initialized():
            red = xd if dist.get_backend(self.group) != 'gloo' else xd.cpu()
            dist.broadcast(red, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            xd = red.to(self.device)
        yd = self._local(xd)
        if dist.is_initialized():
            red = yd if dist.get_backend(self.group) != 'gloo' else yd.cpu()
            dist.all_reduce(red, group=self.group)
            yd = red.to(self.device)
        if isinstance(x, torch.Tensor):
            return yd
        out = yd.cpu().numpy()
        if y is not None:
            y[:] = out
            return y
        return out

    __mul__ = matvec
    dot = matvec

    @property
    def diagonal(self):
        import torch.distributed as dist
        d = torch.zeros(self.num_rows, dtype=torch.float64, device=self.device)
        if self.rowdofs.shape[0]:
            self._bind()
            torch.cuda.current_stream(self.device).synchronize()
            self.ctx.slab_diagonal(self.slab.data_ptr(), self.slab.stride(0), self.dblocks.data_ptr(), d.data_ptr())
            self.ctx.synchronize()
        if dist.is_initialized():
            red = d if dist.get_backend(self.group) != 'gloo' else d.cpu()
            dist.all_reduce(red, group=self.group)
            d = red.to(self.device)
        return d.cpu().numpy()

    def local_bytes(self):
        return int(self.slab.numel()*8+self.dblocks.numel()*8)

    def toarray(self):
        """the full matrix on every rank (tests, small problems only): N products with unit vectors would be wasteful, so
        the local part is expanded on the host and summed over the ranks"""
        import torch.distributed as dist
        N = self.num_rows
        X = torch.eye(N, dtype=torch.float64, device=self.device)
        cols = [self._local(X[:, j].contiguous()) for j in range(N)]
        A = torch.stack(cols, dim=1)
        if dist.is_initialized():
            red = A if dist.get_backend(self.group) != 'gloo' else A.cpu()
            dist.all_reduce(red, group=self.group)
            A = red
        return A.cpu().numpy()

    def __repr__(self):
        return '<{}x{} DistributedSlab_LinearOperator: {} rows x {} columns on {}>'.format(self.num_rows, self.num_columns,
                                                                                        self.rowdofs.shape[0], self.coldofs.shape[0], self.device)


class CSR_LinearOperator:
    """Near-field matrix in HBM, CSR layout (base/PyNucleus_base/CSR_LinearOperator_{SCALAR}.pxi:20-60: ``indptr``,
    ``indices``, ``data``).  The pattern is fixed by getSparseNearField; ``data`` is filled by
    pnl_assemble_pairs_masked / pnl_assemble_boundary_masked with the reference's addToEntry semantics."""
    symmetric = False

    def __init__(self, indptr, indices, num_dofs, ctx, device):
        # the pattern either as host arrays or as int32 torch tensors on the device (getSparseNearField builds it there);
        # device patterns reach the library device-to-device and are copied to the host when .indptr / .indices is read
        self._pattern_dev = None
        if isinstance(indptr, torch.Tensor):
            self._pattern_dev = (indptr.to(torch.int32).contiguous(), indices.to(torch.int32).contiguous())
            self._indptr = self._indices = None
            self._nnz = int(indices.numel())
        else:
            self._indptr = np.ascontiguousarray(indptr, dtype=np.int32)
            self._indices = np.ascontiguousarray(indices, dtype=np.int32)
            self._nnz = int(self._indices.shape[0])
        self.num_rows = self.num_columns = int(num_dofs)
        self.shape = (self.num_rows, self.num_columns)
        self.ctx = ctx
        self.device = device
        self.data_dev = torch.zeros(max(self._nnz, 1), dtype=torch.float64, device=device)
        self.diag_dev = None
        self.info = {}

    @property
    def indptr(self):
        if self._indptr is None:
            self._indptr = self._pattern_dev[0].cpu().numpy()
        return self._indptr

    @property
    def indices(self):
        if self._indices is None:
            self._indices = self._pattern_dev[1].cpu().numpy()
        return self._indices

    @property
    def nnz(self):
        return self._nnz

    @property
    def data(self):
        self.ctx.synchronize()
        return self.data_dev[:self.nnz].cpu().numpy()

    def _ptrs(self):
        return self.data_dev.data_ptr(), (self.diag_dev.data_ptr() if self.diag_dev is not None else None)

    def _bind(self):
        """make this operator's pattern the one resident in the context"""
        if getattr(self.ctx, '_pattern_owner', None) is not self:
            if self._pattern_dev is not None and self._pattern_dev[0].device.type == 'cuda':
                self.ctx.upload_sparsity_device(*self._pattern_dev)
            else:
                self.ctx.upload_sparsity(self.indptr, self.indices)
            self.ctx._pattern_owner = self

    def toarray(self):
        N = self.num_rows
        A = np.zeros((N, N))
        rows = np.repeat(np.arange(N), np.diff(self.indptr))
        A[rows, self.indices] = self.data
        return A

    # operator files: CSR_LinearOperator_{SCALAR}.pxi:268-290 / SSS_LinearOperator_{SCALAR}.pxi:273-293 on an h5py-like group
    def HDF5write(self, node):
        node.create_dataset('indices', data=np.ascontiguousarray(self.indices))
        node.create_dataset('indptr', data=np.ascontiguousarray(self.indptr))
        node.create_dataset('data', data=np.ascontiguousarray(self.data))
        if self.symmetric:
            node.create_dataset('diagonal', data=np.ascontiguousarray(self.diagonal))
            node.attrs['type'] = 'sss'
        else:
            node.attrs['type'] = 'csr'
            node.attrs['num_rows'] = self.num_rows
            node.attrs['num_columns'] = self.num_columns

    @classmethod
    def HDF5read(cls, node, ctx, device=None):
        dev = torch.device('cuda', ctx.device) if device is None else device
        indptr = np.array(node['indptr'], dtype=np.int32)
        kind = node.attrs['type']
        kind = kind.decode() if isinstance(kind, bytes) else str(kind)
        op_cls = SSS_LinearOperator if kind == 'sss' else CSR_LinearOperator
        B = op_cls(indptr, np.array(node['indices'], dtype=np.int32), indptr.shape[0]-1, ctx, dev)
        data = np.array(node['data'], dtype=np.float64)
        B.data_dev[:data.shape[0]] = torch.from_numpy(data).to(dev)
        if kind == 'sss':
            B.diag_dev.copy_(torch.from_numpy(np.array(node['diagonal'], dtype=np.float64)).to(dev))
        else:
            assert B.num_rows == int(node.attrs['num_rows'])
        return B

    @property
    def diagonal(self):
        """stored diagonal entries (zero where the pattern has none), read off the CSR arrays"""
        N = self.num_rows
        rows = np.repeat(np.arange(N, dtype=np.int64), np.diff(self.indptr))
        pos = np.nonzero(self.indices == rows)[0]
        d = np.zeros(N)
        if pos.shape[0]:
            self.ctx.synchronize()
            d[rows[pos]] = self.data_dev[torch.from_numpy(pos).to(self.device)].cpu().numpy()
        return d

    def matvec(self, x, y=None):
        self._bind()
        xd = _as_dev(x, self.device)
        assert xd.shape[0] == self.num_columns
        yd = torch.empty(self.num_rows, dtype=torch.float64, device=self.device)
        torch.cuda.current_stream(self.device).synchronize()
        d, g = self._ptrs()
        self.ctx.spmv(d, g, xd.data_ptr(), yd.data_ptr())
        self.ctx.synchronize()
        if isinstance(x, torch.Tensor):
            return yd
        out = yd.cpu().numpy()
        if y is not None:
            y[:] = out
            return y
        return out

    def __mul__(self, x):
        return self.matvec(x)

    dot = matvec

    def __repr__(self):
        return '<{}x{} {} with {} stored entries on {}>'.format(self.num_rows, self.num_columns, type(self).__name__, self.nnz, self.device)


class SSS_LinearOperator(CSR_LinearOperator):
    """Symmetric sparse skyline storage (base/PyNucleus_base/SSS_LinearOperator_{SCALAR}.pxi:23-60): strict lower
    triangle in CSR (``indptr``, ``indices``, ``data``) plus the ``diagonal`` vector."""
    symmetric = True

    def __init__(self, indptr, indices, num_dofs, ctx, device):
        super().__init__(indptr, indices, num_dofs, ctx, device)
        self.diag_dev = torch.zeros(self.num_rows, dtype=torch.float64, device=device)

    @property
    def diagonal(self):
        self.ctx.synchronize()
        return self.diag_dev.cpu().numpy()

    def toarray(self):
        L = CSR_LinearOperator.toarray(self)
        return L+L.T+np.diag(self.diagonal)


class diagonalOperator:
    """base/PyNucleus_base/linear_operators.pyx diagonalOperator: ``data`` = the diagonal; matvec scales"""

    def __init__(self, diag):
        self.data = np.ascontiguousarray(diag, dtype=np.float64)
        self.num_rows = self.num_columns = self.data.shape[0]
        self.shape = (self.num_rows, self.num_columns)

    @property
    def diagonal(self):
        return self.data

    def matvec(self, x, y=None):
        out = self.data*np.asarray(x)
        if y is not None:
            y[:] = out
            return y
        return out

    __mul__ = matvec

    def toarray(self):
        return np.diag(self.data)


class DistributedSparse_LinearOperator:
    """Row-sharded near field: every rank holds the CSR blocks of its cluster pairs; A = sum over ranks.  matvec follows
    the reference's DistributedH2Matrix_globalData (clusterMethodCy.pyx:3127-3154): Bcast(x), local product, Allreduce(y)
    -- an N-vector over RCCL (or gloo), never the matrix."""

    def __init__(self, local, comm_group=None, far=None, Pfar=None):
        self.local = local
        self.group = comm_group
        self.far = far               # H2Matrix over the local near field with this rank's share of the admissible pairs
        self.Pfar = Pfar             # all admissible pairs (every rank knows the whole tree)
        self.num_rows, self.num_columns = local.num_rows, local.num_columns
        self.shape = local.shape
        self.info = local.info
        self.device = local.device

    @property
    def diagonal(self):
        """sum over the ranks of the local diagonals (every diagonal entry belongs to exactly one rank's blocks)"""
        import torch.distributed as dist
        d = torch.from_numpy(np.ascontiguousarray(self.local.diagonal, dtype=np.float64))
        if dist.get_backend(self.group) != 'gloo':
            d = d.to(self.device)
        dist.all_reduce(d, group=self.group)
        return d.cpu().numpy()

    def matvec(self, x, y=None):
        import torch.distributed as dist
        dev = self.local.device
        xd = _as_dev(x, dev)
        backend = dist.get_backend(self.group)
        if backend == 'gloo':
            xh = xd.cpu()
            dist.broadcast(xh, src=0, group=self.group)
            yh = (self.far if self.far is not None else self.local).matvec(xh.to(dev)).cpu()
            dist.all_reduce(yh, group=self.group)
            yd = yh.to(dev)
        else:
            dist.broadcast(xd, src=0, group=self.group)
            yd = (self.far if self.far is not None else self.local).matvec(xd)
            dist.all_reduce(yd, group=self.group)
        if isinstance(x, torch.Tensor):
            return yd
        out = yd.cpu().numpy()
        if y is not None:
            y[:] = out
            return y
        return out

    __mul__ = matvec
    dot = matvec

    def toarray(self):
        """the full matrix (all-reduce of the dense images; tests only)"""
        import torch.distributed as dist
        A = torch.from_numpy(self.local.toarray())
        dist.all_reduce(A, group=self.group)
        return A.numpy()
