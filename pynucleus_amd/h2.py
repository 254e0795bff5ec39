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
        self._ensure_setup()
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

    # ---- operator files: the reference's layout (H2Matrix.HDF5write / HDF5read, clusterMethodCy.pyx:2449-2550; tree:
    # tree_node.HDF5writeNew / HDF5readNew :1575-1760) on any h5py-like group (create_group, create_dataset, attrs, item access)
    def _ensure_setup(self):
        """make this operator the one whose far field is set up in the context (matvec, the multigrid cycle, farFieldData and
        HDF5write all come through here).  An operator whose kernel the builder no longer holds cannot be set up again from the
        context's tables -- unless its interpolants and leaf values are stored (HDF5read), which need no kernel."""
        if getattr(self.ctx, '_h2_owner', None) is not self:
            if getattr(self, '_stored', None) is None and getattr(self.ctx, '_kernel_epoch', 0) != self._epoch:
                raise RuntimeError('this H2Matrix belongs to a kernel the builder no longer holds (setKernel was called): '
                                   'its far field cannot be set up again; assemble a new operator')
            keep = []
            P = self.plan.as_struct(keep)
            self.ctx.check(self.ctx.L.pnl_h2_setup(self.ctx.h, C.byref(P)))
            self.ctx._h2_owner = self
            if getattr(self, '_stored', None) is not None:
                for which, a in enumerate(self._stored):
                    self.ctx.check(self.ctx.L.pnl_h2_set(self.ctx.h, which, a.ctypes.data))

    def farFieldData(self):
        """(kernel interpolants [nfar, M, M], leaf values as a list of [ndofs, M] blocks in the order of plan.leaf_node)"""
        self._ensure_setup()
        pl = self.plan
        K = np.zeros((pl.far.shape[0], pl.M, pl.M))
        V = np.zeros((int(pl.leaf_dof_off[-1]), pl.M))
        self.ctx.check(self.ctx.L.pnl_h2_get(self.ctx.h, 0, K.ctypes.data))
        self.ctx.check(self.ctx.L.pnl_h2_get(self.ctx.h, 1, V.ctypes.data))
        return K, [V[pl.leaf_dof_off[l]:pl.leaf_dof_off[l+1]] for l in range(pl.leaf_node.shape[0])]

    def HDF5write(self, node, version=2, Pnear=None, refinementParams=None):
        if version != 2:
            raise NotImplementedError('H2 operator files: version 2 (tree_node.HDF5writeNew)')
        pl = self.plan
        nodes, nn, dim = pl.nodes, len(pl.nodes), pl.box.shape[1]
        node.attrs['type'] = 'h2'
        self.Anear.HDF5write(node.create_group('Anear'))
        node.attrs['version'] = version
        tree = node.create_group('tree')

        def graph(g, lists, ncols):
            indptr = np.zeros(nn+1, dtype=np.int32)
            indptr[1:] = np.cumsum([len(v) for v in lists])
            g.create_dataset('indices', data=np.concatenate([np.asarray(v, dtype=np.int32) for v in lists]+[np.zeros(0, dtype=np.int32)]))
            g.create_dataset('indptr', data=indptr)
            g.attrs['num_rows'], g.attrs['num_columns'], g.attrs['type'] = nn, int(ncols), 'sparseGraph'
        kids = [[] for _ in range(nn)]
        for k in range(nn):
            if pl.parent[k] >= 0:
                kids[pl.parent[k]].append(k)
        graph(tree.create_group('children'), kids, nn)
        tree.create_dataset('boxes', data=np.ascontiguousarray(pl.box))
        tree.create_dataset('interpolationOrders', data=np.full(nn, pl.m, dtype=np.int32))
        tr = tree.create_group('transferOperators')
        for k in range(nn):
            if pl.parent[k] >= 0:
                tr.create_dataset(str(k), data=np.ascontiguousarray(pl.transfer[k]))
        tree.attrs['valueSize'] = 1
        graph(tree.create_group('dofs'), [np.asarray(n.dofs) for n in nodes], self.num_rows)
        cells = [np.asarray(n.cells) if n.is_leaf else np.zeros(0, dtype=np.int32) for n in nodes]
        graph(tree.create_group('cells'), cells, int(max([c.max() for c in cells if c.shape[0]]+[-1]))+1)
        K, V = self.farFieldData()
        vals = tree.create_group('values')
        for l, k in enumerate(pl.leaf_node):
            vals.create_dataset(str(int(k)), data=np.ascontiguousarray(V[l][None, :, :]))      # [valueSize, ndofs, M]
        tree.attrs['dim'] = dim
        rp = tree.create_group('refinementParams')
        defaults = dict(maxLevels=200, maxLevelsMixed=200, minSize=1, minMixedSize=1, refType=0, splitEveryDim=False, eta=3.,
                        farFieldInteractionSize=pl.M, interpolation_order=pl.m, attemptRefinement=True, targetOrder=0., meshDiam=0.,
                        maxSingularity=0.)
        defaults.update(refinementParams or {})
        for key, v in defaults.items():
            rp.attrs[key] = v
        g = node.create_group('Pfar')
        g.create_dataset('kernelInterpolants', data=K.ravel())
        ids = np.zeros((pl.far.shape[0], 5), dtype=np.int32)
        ids[:, :2] = pl.far
        ids[:, 2:4] = pl.M
        ids[:, 4] = pl.level[pl.far[:, 0]]
        g.create_dataset('nodeIds', data=ids)
        if Pnear is not None:
            g2 = node.create_group('Pnear')
            for k, cp in enumerate(Pnear):
                gk = g2.create_group(str(k))
                gk.attrs['n1'], gk.attrs['n2'] = pl.nid[id(cp.n1)], pl.nid[id(cp.n2)]

    @staticmethod
    def HDF5read(node, ctx, returnPnear=False):
        """the operator back on the device of ``ctx`` -- a builder's ``context()`` for the mesh / DoF map the operator was assembled
        on (the far-field passes run in that context); near field, tree, transfer operators, kernel interpolants and leaf values
        are the stored ones"""
        from .linear_operators import CSR_LinearOperator
        Anear = CSR_LinearOperator.HDF5read(node['Anear'], ctx)
        tree = node['tree']
        indptr, indices = np.array(tree['children']['indptr']), np.array(tree['children']['indices'])
        nn = indptr.shape[0]-1
        boxes = np.array(tree['boxes'], dtype=np.float64)
        dip, dix = np.array(tree['dofs']['indptr']), np.array(tree['dofs']['indices'])
        cip, cix = np.array(tree['cells']['indptr']), np.array(tree['cells']['indices'])
        orders = np.array(tree['interpolationOrders'])
        assert (orders == orders[0]).all(), 'one interpolation order for all nodes'

        class stored_node:
            pass
        nodes = [stored_node() for _ in range(nn)]
        for k, n in enumerate(nodes):
            n.id, n.box = k, boxes[k]
            n.dofs = np.ascontiguousarray(dix[dip[k]:dip[k+1]], dtype=np.int32)
            n.cells = np.ascontiguousarray(cix[cip[k]:cip[k+1]], dtype=np.int32)
            n.children = [nodes[c] for c in indices[indptr[k]:indptr[k+1]]]
            n.is_leaf = len(n.children) == 0
            n.parent = None
        for n in nodes:
            for c in n.children:
                c.parent = n
        root = [n for n in nodes if n.parent is None][0]
        plan = h2Plan.__new__(h2Plan)
        flat_nodes, parent, level = h2Plan.flatten(root)
        plan.nodes, plan.nid = flat_nodes, {id(n): k for k, n in enumerate(flat_nodes)}
        dim = boxes.shape[1]
        plan.m, plan.M = int(orders[0]), int(orders[0])**dim
        plan.parent, plan.level = np.array(parent, dtype=np.int32), np.array(level, dtype=np.int32)
        plan.nlevels = int(plan.level.max())+1
        plan.box = np.ascontiguousarray(np.stack([n.box for n in flat_nodes]))
        leaves = [k for k, n in enumerate(flat_nodes) if n.is_leaf]
        plan.partial_leaves = False
        plan.leaf_node = np.array(leaves, dtype=np.int32)
        plan.leaf_dof_off = np.zeros(len(leaves)+1, dtype=np.int32)
        plan.leaf_cell_off = np.zeros(len(leaves)+1, dtype=np.int32)
        plan.leaf_dof_off[1:] = np.cumsum([flat_nodes[k].dofs.shape[0] for k in leaves])
        plan.leaf_cell_off[1:] = np.cumsum([flat_nodes[k].cells.shape[0] for k in leaves])
        plan.leaf_dofs = np.concatenate([flat_nodes[k].dofs for k in leaves]).astype(np.int32)
        plan.leaf_cells = np.concatenate([flat_nodes[k].cells for k in leaves]).astype(np.int32)
        ids = np.array(node['Pfar']['nodeIds'], dtype=np.int64)
        new_of_stored = np.array([plan.nid[id(nodes[k])] for k in range(nn)], dtype=np.int32)
        plan.far = np.ascontiguousarray(new_of_stored[ids[:, :2]].reshape(-1, 2), dtype=np.int32)
        plan.far_class = None
        plan.transfer = np.zeros((nn, plan.M, plan.M))
        for key in tree['transferOperators']:
            plan.transfer[new_of_stored[int(key)]] = np.array(tree['transferOperators'][key])
        # pnl_h2_setup wants a volume rule for the leaf values it is about to compute; they are replaced by the stored ones
        plan.qbary = np.zeros((1, 3)); plan.qbary[0, :dim+1] = 1./(dim+1)
        plan.qw = np.ones(1)
        plan.qphi = np.full((1, int(ctx.dofs_per_element)), 1./(dim+1))
        Pfar = {}
        for j in range(ids.shape[0]):
            class pair:
                pass
            cp = pair()
            cp.n1, cp.n2 = nodes[ids[j, 0]], nodes[ids[j, 1]]
            Pfar.setdefault(int(ids[j, 4]), []).append(cp)
        op = H2Matrix(Anear, plan, ctx, root, Pfar)
        K = np.ascontiguousarray(np.array(node['Pfar']['kernelInterpolants'], dtype=np.float64))
        V = np.ascontiguousarray(np.concatenate([np.array(tree['values'][str(int(nodes.index(flat_nodes[k])))])[0] for k in leaves]))
        op._stored = (K, V)
        for which, a in enumerate(op._stored):
            ctx.check(ctx.L.pnl_h2_set(ctx.h, which, a.ctypes.data))
        if returnPnear:
            Pnear = []
            for key in node['Pnear']:
                cp = type('pair', (), {})()
                cp.n1, cp.n2 = nodes[int(node['Pnear'][key].attrs['n1'])], nodes[int(node['Pnear'][key].attrs['n2'])]
                Pnear.append(cp)
            return op, Pnear
        return op

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
