"""H2 matrix: near field + Chebyshev-interpolated far field (host side).

Mirrors /root/reference/nl/PyNucleus_nl/clusterMethodCy.pyx: H2Matrix :2240-2320 (matvec = near field + upward pass +
far-field interactions + downward pass), tree_node.prepareTransferOperators :1071-1090 with transferMatrixBuilder
:2004-2073, the interpolation order of getH2RefinementParams (nonlocalAssembly_{SCALAR}.pxi:2990-3000).  The kernel
interpolants, the leaf values and the three passes run in libpnl_hip.so (pnl_h2_setup / pnl_h2_matvec); the transfer
matrices (pure geometry, a few hundred M x M blocks) are built here.
"""
import ctypes as C
import numpy as np


def interpolationOrder(kernel, mesh, target_order):
    """NA:2995-2998: max(ceil((2 target_order + max(-singularity, 2)) |ln(hmin/diam)| / ln 4 / 3), 2)"""
    loggamma = abs(np.log(0.25))
    sing = kernel.max_singularity
    return int(max(np.ceil((2*target_order+max(-sing, 2))*abs(np.log(mesh.hmin/mesh.diam))/loggamma/3.), 2))


def chebNodes(a, b, m):
    eta = np.cos((2.0*np.arange(m, 0, -1)-1.0)/(2.0*m)*np.pi)
    return (b-a)*0.5*(eta+1.0)+a


def lagrangeMatrix(nodes, x):
    """L[l, k] = l-th Lagrange polynomial on `nodes` at x[k]"""
    m = nodes.shape[0]
    L = np.ones((m, x.shape[0]))
    for l in range(m):
        for k in range(m):
            if k != l:
                L[l] *= (x-nodes[k])/(nodes[l]-nodes[k])
    return L


def transferMatrix(boxP, boxC, m):
    """T[I, J] = L^parent_I(xi^child_J), tensor index = i_0 + m i_1 (transferMatrixBuilder.build, CM:2010-2073)"""
    dim = boxP.shape[0]
    T = np.ones((1, 1))
    for d in range(dim):
        Ld = lagrangeMatrix(chebNodes(boxP[d, 0], boxP[d, 1], m), chebNodes(boxC[d, 0], boxC[d, 1], m))     # [m, m]
        T = np.kron(Ld, T)                        # coordinate 0 fastest
    return T


class h2Plan:
    """flattened cluster tree + admissible pairs for pnl_h2_setup"""

    @staticmethod
    def flatten(root):
        """(nodes, parent, level) of the tree in depth-first order: the node numbering of the plan"""
        nodes, parent, level = [], [], []

        def walk(n, p, lvl):
            k = len(nodes)
            nodes.append(n)
            parent.append(p)
            level.append(lvl)
            for c in n.children:
                walk(c, k, lvl+1)
        walk(root, -1, 0)
        return nodes, parent, level

    def __init__(self, dm, root, Pfar, m, far_class=None, flat=None, far_pairs=None, leaf_mask=None):
        """far_pairs: explicit list of admissible pairs (default: all of Pfar, level by level); leaf_mask[node]: only these leaves
        take part in the upward / downward pass (a rank's own subtrees)"""
        from .quadrature import simplexXiaoGimbutas
        mesh = dm.mesh
        dim = mesh.dim
        nodes, parent, level = flat if flat is not None else h2Plan.flatten(root)
        self.nodes = nodes
        nid = {id(n): k for k, n in enumerate(nodes)}
        self.m, self.M = int(m), int(m)**dim
        self.parent = np.array(parent, dtype=np.int32)
        self.level = np.array(level, dtype=np.int32)
        self.nlevels = int(self.level.max())+1
        self.box = np.ascontiguousarray(np.stack([n.box for n in nodes]), dtype=np.float64)        # [nnodes, dim, 2]
        leaves = [k for k, n in enumerate(nodes) if n.is_leaf and (leaf_mask is None or leaf_mask[k])]
        self.partial_leaves = leaf_mask is not None
        assert len(leaves) > 0
        self.leaf_node = np.array(leaves, dtype=np.int32)
        self.leaf_dof_off = np.zeros(len(leaves)+1, dtype=np.int32)
        self.leaf_cell_off = np.zeros(len(leaves)+1, dtype=np.int32)
        self.leaf_dof_off[1:] = np.cumsum([nodes[k].dofs.shape[0] for k in leaves])
        self.leaf_cell_off[1:] = np.cumsum([nodes[k].cells.shape[0] for k in leaves])
        self.leaf_dofs = np.concatenate([nodes[k].dofs for k in leaves]).astype(np.int32)
        self.leaf_cells = np.concatenate([nodes[k].cells for k in leaves]).astype(np.int32)
        if far_pairs is None:
            far_pairs = [cp for lvl in sorted(Pfar) for cp in Pfar[lvl]]
        self.nid = nid
        far = [(nid[id(cp.n1)], nid[id(cp.n2)]) for cp in far_pairs]
        self.far = np.array(far, dtype=np.int32).reshape(-1, 2)
        # variable order: kernel class per admissible pair, far_class(cp) -> class
        self.far_class = None if far_class is None else np.array([far_class(cp) for cp in far_pairs], dtype=np.int32)
        self.transfer = np.zeros((len(nodes), self.M, self.M))
        from .clusters import _use_native
        if _use_native() and dim <= 2:
            from . import _lib
            rc = _lib.load().pnl_h2_transfer_matrices(len(nodes), dim, self.m, self.box.ctypes.data, self.parent.ctypes.data,
                                                      self.transfer.ctypes.data)
            assert rc == 0, rc
        else:
            for k, n in enumerate(nodes):
                if parent[k] >= 0:
                    self.transfer[k] = transferMatrix(nodes[parent[k]].box, n.box, self.m)
        # quadrature of the leaf values: order m + polynomial degree + 1 (CM:1236-1247, "Sauter Schwab p. 428")
        qr = simplexXiaoGimbutas(self.m+dm.polynomialOrder+1, dim, dim)
        self.qbary = np.zeros((qr.num_nodes, 3))
        self.qbary[:, :dim+1] = qr.nodes.T
        self.qw = np.ascontiguousarray(qr.weights, dtype=np.float64)
        self.qphi = np.ascontiguousarray(dm.evalShapeFunctions(qr.nodes).T, dtype=np.float64)

    def as_struct(self, keep):
        from ._lib import pnl_h2_plan

        def ptr(a, dt):
            a = np.ascontiguousarray(a, dtype=dt)
            keep.append(a)
            return a.ctypes.data

        P = pnl_h2_plan()
        P.nnodes, P.nleaves, P.nfar = len(self.nodes), self.leaf_node.shape[0], self.far.shape[0]
        P.m, P.nlevels, P.nq = self.m, self.nlevels, self.qw.shape[0]
        P.box = ptr(self.box, np.float64)
        for name in ('parent', 'level', 'leaf_node', 'leaf_dof_off', 'leaf_dofs', 'leaf_cell_off', 'leaf_cells', 'far'):
            setattr(P, name, ptr(getattr(self, name), np.int32))
        P.transfer = ptr(self.transfer, np.float64)
        P.qbary, P.qw, P.qphi = ptr(self.qbary, np.float64), ptr(self.qw, np.float64), ptr(self.qphi, np.float64)
        P.far_class = None if self.far_class is None else ptr(self.far_class, np.int32)
        P.partial_leaves = 1 if self.partial_leaves else 0
        return P


class H2Matrix:
    """y = Anear x + (far field through the cluster tree); clusterMethodCy.pyx:2240-2295"""

    def __init__(self, Anear, plan, ctx, root, Pfar):
        self.Anear, self.plan, self.ctx = Anear, plan, ctx
        self.tree, self.Pfar = root, Pfar
        self.num_rows, self.num_columns = Anear.num_rows, Anear.num_columns
        self.shape = Anear.shape
        self.device = Anear.device
        self.info = dict(Anear.info, interpolation_order=plan.m, numFarPairs=int(plan.far.shape[0]))
        keep = []
        P = plan.as_struct(keep)
        ctx.check(ctx.L.pnl_h2_setup(ctx.h, C.byref(P)))
        ctx._h2_owner = self
        self._epoch = getattr(ctx, '_kernel_epoch', 0)

    def matvec(self, x, y=None):
        import torch
        from .linear_operators import _as_dev
        if getattr(self.ctx, '_h2_owner', None) is not self:
            if getattr(self.ctx, '_kernel_epoch', 0) != self._epoch:
                raise RuntimeError('this H2Matrix belongs to a kernel the builder no longer holds (setKernel was called): '
                                   'its far field cannot be set up again; assemble a new operator')
            keep = []
            P = self.plan.as_struct(keep)
            self.ctx.check(self.ctx.L.pnl_h2_setup(self.ctx.h, C.byref(P)))
            self.ctx._h2_owner = self
        xd = _as_dev(x, self.device)
        yd = self.Anear.matvec(xd)
        torch.cuda.current_stream(self.device).synchronize()
        self.ctx.check(self.ctx.L.pnl_h2_matvec(self.ctx.h, C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr())))
        self.ctx.synchronize()
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
        return self.Anear.diagonal

    def toarray(self):
        """dense image through N matvecs (tests only)"""
        import torch
        N = self.num_columns
        A = np.zeros((self.num_rows, N))
        e = torch.zeros(N, dtype=torch.float64, device=self.device)
        for j in range(N):
            e.zero_()
            e[j] = 1.
            A[:, j] = self.matvec(e).cpu().numpy()
        return A

    def __repr__(self):
        return '<{}x{} H2Matrix: near field {} entries, {} far-field cluster pairs of order {}>'.format(
            self.num_rows, self.num_columns, self.Anear.nnz, self.plan.far.shape[0], self.plan.m)
