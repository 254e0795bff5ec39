"""H2 operator with rank-local data and halo exchange (DistributedH2Matrix_localData, clusterMethodCy.pyx:3368-3920;
DistributedLinearOperator, :3157-3366): every rank owns whole subtrees of the cluster tree and with them rows of the operator,

* the near-field blocks of its row clusters (unsymmetric CSR, columns anywhere): the matvec fetches the GHOST entries of x it
  multiplies with from their owners (communicateNear :3276-3287, 3487-3498),
* the admissible pairs of its row clusters: the upward pass runs over its own leaves only, the coefficients cup[n2] of foreign
  column clusters come from their owners (communicateFar :3610-3647) -- M doubles per cluster instead of its DoFs,
* the few nodes above the subtrees ("top" nodes, fewer than 4 P of them) are shared: their coefficients are all-reduced
  (a few KB), their admissible pairs dealt round-robin.

No N-vector is broadcast or all-reduced: x and y stay distributed by rows.  The exchange lists follow from the replicated tree
and pair lists, so every rank computes them for all ranks without a set-up Alltoall (the reference exchanges them,
:3216-3221, 3427-3432)."""
import ctypes as C
import numpy as np
import torch


def subtree_owners(flat, size):
    """Ownership of the tree nodes: the nodes at level ceil(log2 P) + 1 (and leaves above it) are the roots of owned subtrees,
    dealt to the ranks in tree order in contiguous groups of about equal DoF count; nodes above them are shared (-1).
    Returns (owner[node], cut nodes)."""
    nodes, parent, level = flat
    nn = len(nodes)
    cut_level = max(1, int(np.ceil(np.log2(max(size, 1))))+1)
    is_cut = np.zeros(nn, dtype=bool)
    for k, n in enumerate(nodes):
        if level[k] == cut_level or (n.is_leaf and level[k] < cut_level):
            is_cut[k] = True
    cut = np.nonzero(is_cut)[0]
    w = np.array([nodes[k].dofs.shape[0] for k in cut], dtype=np.float64)
    cum = np.cumsum(w)
    # group g ends where the cumulative weight passes (g+1)/size of the total; every rank gets at least one subtree if it can
    owner_cut = np.minimum((cum-0.5*w)/cum[-1]*size, size-1).astype(np.int64)
    owner = np.full(nn, -1, dtype=np.int64)
    for k, r in zip(cut, owner_cut):
        owner[k] = r
    for k in range(nn):                                   # depth-first order: parents come first
        p = parent[k]
        if owner[k] < 0 and p >= 0 and owner[p] >= 0:
            owner[k] = owner[p]
    return owner, cut


class DistributedH2Matrix_localData:
    """see the module docstring; built by nonlocalBuilder.getH2() under a communicator with params['localFarFieldIndexing']"""

    def __init__(self, builder, root, Pnear, Pfar, m, far_class=None, group=None):
        import torch.distributed as dist
        from .h2 import h2Plan
        self.group = group
        self.rank, self.size = dist.get_rank(group), dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        rank, size = self.rank, self.size
        dm = builder.dm
        N = dm.num_dofs
        self.num_rows = self.num_columns = N
        self.shape = (N, N)
        ctx = self.ctx = builder.context()
        dev = self.device = torch.device('cuda', ctx.device)
        flat = h2Plan.flatten(root)
        nodes, parent, level = flat
        nid = {id(n): k for k, n in enumerate(nodes)}
        owner, cut = subtree_owners(flat, size)
        # every rank computes the ownership of ALL ranks, so the check below fails on every rank alike (no rank walks on into a
        # collective the others never reach)
        empty = sorted(set(range(size))-set(int(owner[k]) for k in cut))
        if empty:
            raise NotImplementedError('rank(s) {} would own no subtree ({} subtrees at the cut level for {} ranks: tree too shallow or '
                                      'DoF weights too skewed); use the all-reduce operator (localFarFieldIndexing=False)'.format(empty, len(cut), size))
        self.node_owner = owner
        dof_owner = np.full(N, -1, dtype=np.int64)
        for k in cut:
            dof_owner[nodes[k].dofs] = owner[k]
        assert (dof_owner >= 0).all()
        self.dof_owner = dof_owner
        self.owned = np.nonzero(dof_owner == rank)[0]
        # ---- near field: pairs by the owner of their row cluster --------------------------------------------------------------
        near_of = [[] for _ in range(size)]
        for cp in Pnear:
            r = owner[nid[id(cp.n1)]]
            if r < 0:
                raise NotImplementedError('a near-field block whose row cluster lies above the rank subtrees (tree too shallow for '
                                          '{} ranks): use the all-reduce operator'.format(size))
            near_of[r].append(cp)
        self.local = builder.assembleClusters(near_of[rank], forceUnsymmetricMatrix=True, _symmetrizeMasks=True)
        # ghost entries of x: columns of a rank's blocks that it does not own, grouped by owner (lists for ALL ranks: what I need
        # and what the others need from me)
        ghosts = []
        for r in range(size):
            cl = {nid[id(cp.n2)] for cp in near_of[r]}
            cols = np.unique(np.concatenate([nodes[k].dofs for k in cl])) if cl else np.zeros(0, dtype=np.int64)
            cols = cols[dof_owner[cols] != r]
            ghosts.append([cols[dof_owner[cols] == q] for q in range(size)])
        self._x_recv = [torch.as_tensor(ghosts[rank][q], dtype=torch.int64, device=dev) for q in range(size)]
        self._x_send = [torch.as_tensor(ghosts[q][rank], dtype=torch.int64, device=dev) for q in range(size)]
        self.num_ghosts = int(sum(g.shape[0] for g in ghosts[rank]))
        # ---- far field: pairs by the owner of the row cluster, top pairs round-robin ---------------------------------------------
        all_far = [cp for lvl in sorted(Pfar) for cp in Pfar[lvl]]
        far_of = [[] for _ in range(size)]
        need = [[set() for _ in range(size)] for _ in range(size)]            # need[r][q]: nodes owned by q whose cup rank r needs
        for i, cp in enumerate(all_far):
            a, b = nid[id(cp.n1)], nid[id(cp.n2)]
            r = owner[a] if owner[a] >= 0 else i % size
            far_of[r].append(cp)
            if owner[b] >= 0 and owner[b] != r:
                need[r][owner[b]].add(b)
        self.num_far_pairs = len(far_of[rank])
        self._c_recv = [torch.as_tensor(sorted(need[rank][q]), dtype=torch.int64, device=dev) for q in range(size)]
        self._c_send = [torch.as_tensor(sorted(need[q][rank]), dtype=torch.int64, device=dev) for q in range(size)]
        self.num_ghost_clusters = int(sum(len(need[rank][q]) for q in range(size)))
        self._top = torch.as_tensor(np.nonzero(owner < 0)[0], dtype=torch.int64, device=dev)
        leaf_mask = owner == rank
        self.plan = h2Plan(dm, root, Pfar, m, far_class, flat=flat, far_pairs=far_of[rank], leaf_mask=leaf_mask)
        self.Pfar, self.tree = Pfar, root
        self._setup()
        self.M = self.plan.M
        self.info = dict(self.local.info, interpolation_order=self.plan.m, numFarPairs=self.num_far_pairs, numGhosts=self.num_ghosts,
                         numGhostClusters=self.num_ghost_clusters, numTopNodes=int(self._top.numel()), numOwned=int(self.owned.shape[0]))
        self._owned_t = torch.as_tensor(self.owned, dtype=torch.int64, device=dev)

    def _setup(self):
        keep = []
        P = self.plan.as_struct(keep)
        self.ctx.check(self.ctx.L.pnl_h2_setup(self.ctx.h, C.byref(P)))
        self.ctx._h2_owner = self
        self._epoch = getattr(self.ctx, '_kernel_epoch', 0)

    # ---- communication ---------------------------------------------------------------------------------------------------------
    def _exchange(self, send_parts, recv_counts, width):
        """all-to-all of row blocks: send_parts[q] [n_q, width] to rank q, returns the list of received blocks"""
        import torch.distributed as dist
        send = torch.cat([p.reshape(-1, width) for p in send_parts]) if send_parts else torch.zeros((0, width), dtype=torch.float64, device=self.device)
        in_splits = [int(p.reshape(-1, width).shape[0]) for p in send_parts]
        out_splits = [int(c) for c in recv_counts]
        if self.backend == 'gloo':
            send_h = send.cpu()
            recv_h = torch.zeros((sum(out_splits), width), dtype=torch.float64)
            dist.all_to_all_single(recv_h, send_h, out_splits, in_splits, group=self.group)
            recv = recv_h.to(self.device)
        else:
            recv = torch.zeros((sum(out_splits), width), dtype=torch.float64, device=self.device)
            dist.all_to_all_single(recv, send.contiguous(), out_splits, in_splits, group=self.group)
        return torch.split(recv, out_splits)

    def _allreduce_rows(self, buf, rows):
        import torch.distributed as dist
        if rows.numel() == 0:
            return
        part = buf.index_select(0, rows)
        if self.backend == 'gloo':
            h = part.cpu()
            dist.all_reduce(h, group=self.group)
            part = h.to(self.device)
        else:
            dist.all_reduce(part, group=self.group)
        buf.index_copy_(0, rows, part)

    # ---- the operator ------------------------------------------------------------------------------------------------------------
    def matvec_owned(self, x_owned):
        """y restricted to this rank's DoFs from x restricted to this rank's DoFs (torch tensors on the device)"""
        dev = self.device
        ctx = self.ctx
        if getattr(ctx, '_h2_owner', None) is not self:
            if getattr(ctx, '_kernel_epoch', 0) != self._epoch:
                raise RuntimeError('this operator belongs to a kernel the builder no longer holds')
            self._setup()
        N = self.num_rows
        x = torch.zeros(N, dtype=torch.float64, device=dev)
        x[self._owned_t] = x_owned
        # ghosts of the near field (communicateNear)
        recv = self._exchange([x[idx].reshape(-1, 1) for idx in self._x_send], [idx.numel() for idx in self._x_recv], 1)
        for idx, r in zip(self._x_recv, recv):
            if idx.numel():
                x[idx] = r.reshape(-1)
        y = self.local.matvec(x)                                              # rows of my clusters
        # far field: upward pass over my leaves, coefficients of foreign column clusters from their owners (communicateFar)
        nn, M = len(self.plan.nodes), self.M
        cup = torch.empty((nn, M), dtype=torch.float64, device=dev)
        cdown = torch.empty((nn, M), dtype=torch.float64, device=dev)
        torch.cuda.current_stream(dev).synchronize()
        ctx.check(ctx.L.pnl_h2_upward(ctx.h, C.c_void_p(x.data_ptr()), C.c_void_p(cup.data_ptr())))
        ctx.synchronize()
        self._allreduce_rows(cup, self._top)
        recv = self._exchange([cup.index_select(0, idx) for idx in self._c_send], [idx.numel() for idx in self._c_recv], M)
        for idx, r in zip(self._c_recv, recv):
            if idx.numel():
                cup.index_copy_(0, idx, r)
        torch.cuda.current_stream(dev).synchronize()
        ctx.check(ctx.L.pnl_h2_interact(ctx.h, C.c_void_p(cup.data_ptr()), C.c_void_p(cdown.data_ptr())))
        ctx.synchronize()
        self._allreduce_rows(cdown, self._top)
        torch.cuda.current_stream(dev).synchronize()
        ctx.check(ctx.L.pnl_h2_downward(ctx.h, C.c_void_p(cdown.data_ptr()), C.c_void_p(y.data_ptr())))
        ctx.synchronize()
        return y[self._owned_t]

    def matvec(self, x, y=None):
        """convenience for callers that hold the whole vector on every rank: restrict, multiply, all-gather the owned parts"""
        import torch.distributed as dist
        from .linear_operators import _as_dev
        xd = _as_dev(x, self.device)
        yo = self.matvec_owned(xd[self._owned_t])
        counts = [int((self.dof_owner == q).sum()) for q in range(self.size)]
        if self.backend == 'gloo':
            parts = [torch.zeros(c, dtype=torch.float64) for c in counts]
            dist.all_gather(parts, yo.cpu(), group=self.group) if len(set(counts)) == 1 else self._gatherv(parts, yo.cpu())
        else:
            parts = [torch.zeros(c, dtype=torch.float64, device=self.device) for c in counts]
            dist.all_gather(parts, yo, group=self.group) if len(set(counts)) == 1 else self._gatherv(parts, yo)
        yd = torch.zeros(self.num_rows, dtype=torch.float64, device=self.device)
        for q, p in enumerate(parts):
            yd[torch.as_tensor(np.nonzero(self.dof_owner == q)[0], dtype=torch.int64, device=self.device)] = p.to(self.device)
        if isinstance(x, torch.Tensor):
            return yd
        out = yd.cpu().numpy()
        if y is not None:
            y[:] = out
            return y
        return out

    def _gatherv(self, parts, mine):
        """all-gather of pieces of different lengths (broadcast from every owner)"""
        import torch.distributed as dist
        for q in range(self.size):
            if q == self.rank:
                parts[q].copy_(mine)
            dist.broadcast(parts[q], src=q if self.group is None else dist.get_global_rank(self.group, q), group=self.group)     # src is a global rank

    __mul__ = matvec
    dot = matvec

    @property
    def diagonal(self):
        """diagonal of the near field (the far field has no diagonal entries), owned rows from every rank"""
        import torch.distributed as dist
        d = torch.zeros(self.num_rows, dtype=torch.float64)
        dl = np.asarray(self.local.diagonal)
        d[self.owned] = torch.from_numpy(dl[self.owned])
        if self.backend == 'gloo':
            dist.all_reduce(d, group=self.group)
            return d.numpy()
        dd = d.to(self.device)
        dist.all_reduce(dd, group=self.group)
        return dd.cpu().numpy()

    def __repr__(self):
        return '<{}x{} DistributedH2Matrix_localData rank {}/{}: {} owned DoFs, {} ghosts, {} ghost clusters, {} far pairs>'.format(
            self.num_rows, self.num_columns, self.rank, self.size, self.owned.shape[0], self.num_ghosts, self.num_ghost_clusters,
            self.num_far_pairs)
