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
    def __init__(self, A_dev, ctx, info=None):
        assert A_dev.dtype == torch.float64 and A_dev.is_contiguous()
        self.A = A_dev
        self.ctx = ctx
        self.num_rows, self.num_columns = A_dev.shape
        self.shape = (self.num_rows, self.num_columns)
        self.info = info or {}
        self._host = None

    # reference API -------------------------------------------------------
    @property
    def data(self):
        """the matrix as a numpy array (copied from HBM on first use)"""
        if self._host is None:
            self.ctx.synchronize()
            self._host = self.A.cpu().numpy()
        return self._host

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
        self.ctx.gemv(self.A.data_ptr(), self.A.stride(0), self.num_rows, xd.data_ptr(), yd.data_ptr())
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
        self._host = None
        return Dense_LinearOperator(self.A, self.ctx, self.info)
