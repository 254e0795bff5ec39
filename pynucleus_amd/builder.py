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


def tile_cells(dpe):
    """cells per block of the GPU tile kernel (mirrors TILE_P1 / TILE_P2 in csrc/pnl_hip.hip)"""
    return 32 if dpe == 6 else 64


def upper_tiles(num_cells, T):
    """all block-tile pairs (ta <= tb), heavy near-diagonal ones first"""
    nb = (num_cells+T-1)//T
    return np.array([(a, a+d) for d in range(nb) for a in range(nb-d)], dtype=np.int32).reshape(-1, 2)


def tiles_of_rank(num_cells, T, rank, size):
    return np.ascontiguousarray(upper_tiles(num_cells, T)[rank::size])


def cell_range_of_rank(num_cells, rank, size):
    """the reference's MPI split of cellNo1 (NA:1280-1285)"""
    return int(np.ceil(num_cells*rank/size)), int(np.ceil(num_cells*(rank+1)/size))


class nonlocalBuilder:
    def __init__(self, dm, kernel, params={}, zeroExterior=True, comm=None, PLogger=None, dm2=None, device=None, **kwargs):
        if 'boundary' in kwargs:
            zeroExterior = kwargs.pop('boundary')           # deprecated alias, NA:888-890
        if dm2 is not None:
            raise NotImplementedError('assembly with two DoFMaps')
        self.PLogger = PLogger if PLogger is not None else FakePLogger()
        self.comm = comm
        self.params = dict(params)
        self.dm = dm
        self.mesh = dm.mesh
        self.device = device
        self._ctx = None
        self.setKernel(kernel, zeroExterior)

    def setKernel(self, kernel, zeroExterior=True):
        assert kernel.dim == self.dm.mesh.dim, "Kernel dimension must match dm.mesh dimension"
        self.kernel = kernel
        # NA:919-922
        self.zeroExterior = False if kernel.finiteHorizon else bool(zeroExterior)
        self.tables = nonlocalTables(self.dm, kernel, self.params, self.zeroExterior)
        self._uploaded = False

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

    def _rank_size(self):
        if self.comm is None:
            return 0, 1
        import torch.distributed as dist
        return dist.get_rank(self.comm if self.comm is not True else None), dist.get_world_size(self.comm if self.comm is not True else None)

    # ------------------------------------------------------------------
    def getDense(self, trySparsification=False, distributed=False):
        """Assemble the dense operator on the GPU (NA:1262-1473).

        comm=None: full matrix on this GPU.  With a torch.distributed group as comm the element pairs
        are split over the ranks (tiles dealt round-robin, balanced by construction instead of the
        reference's equal cellNo1 ranges NA:1280-1285); by default the parts are all-reduced into the
        full matrix on every rank exactly like the reference (NA:1449-1450); distributed=True keeps them
        separate and all-reduces the N-vector in matvec instead."""
        import torch
        from .linear_operators import Dense_LinearOperator, DistributedDense_LinearOperator
        ctx = self.context()
        dev = torch.device('cuda', ctx.device)
        N = self.dm.num_dofs
        nc = self.mesh.num_cells
        self.PLogger.addValue('useSymmetricCells', True)
        self.PLogger.addValue('useSymmetricLocalMatrix', True)
        A = torch.zeros((N, N), dtype=torch.float64, device=dev)
        rank, size = self._rank_size()
        if size == 1:
            ctx.assemble_dense(A.data_ptr(), A.stride(0), self.zeroExterior, 0, nc)
        else:
            tiles = self.tiles_for_rank(rank, size)
            start, end = cell_range_of_rank(nc, rank, size)
            ctx.assemble_dense_tiles(A.data_ptr(), A.stride(0), self.zeroExterior, tiles, start, end)
        cnt = ctx.counters()
        ms = ctx.phase_ms()
        for k in ('numCellPairs', 'numAssembledCellPairs', 'numIntegrations'):
            self.PLogger.addValue(k, cnt[k])
        self.PLogger.addTimer('interior', 1e-3*(ms['tiles']+ms['worklist']+ms['singular']))
        self.PLogger.addTimer('zeroExterior', 1e-3*ms['boundary'])
        info = dict(counters=cnt, phase_ms=ms)
        if size == 1:
            return Dense_LinearOperator(A, ctx, info)
        group = None if self.comm is True else self.comm
        op = DistributedDense_LinearOperator(A, ctx, group, info)
        return op if distributed else op.reduce()

    def tiles_for_rank(self, rank, size):
        """block-tile pairs (ta <= tb) of the upper block triangle owned by `rank`: the list is ordered by
        block distance (heavy near-diagonal tiles first) and dealt round-robin."""
        T = tile_cells(self.dm.dofs_per_element)
        if self._ctx is not None:
            assert self._ctx.tile_cells() == T
        return tiles_of_rank(self.mesh.num_cells, T, rank, size)

    def getDiagonal(self):
        raise NotImplementedError('getDiagonal: assemble the dense operator and take .diagonal')

    def getSparse(self, returnNearField=False):
        raise NotImplementedError('finite-horizon sparse assembly is not implemented on the GPU path yet')

    def getH2(self, **kwargs):
        raise NotImplementedError('H2 assembly is not implemented on the GPU path yet')

    def assembleClusters(self, Pnear, **kwargs):
        raise NotImplementedError('cluster (near-field) assembly is not implemented on the GPU path yet')


def assembleNonlocalOperator(mesh, dm, s, horizon=None, params={}, zeroExterior=True, comm=None, **kwargs):
    """nonlocalAssembly.pyx:362-372"""
    from .kernels import getFractionalKernel
    kernel = getFractionalKernel(mesh.dim, s, horizon)
    builder = nonlocalBuilder(dm, kernel, params, zeroExterior, comm, **kwargs)
    return builder.getDense()
