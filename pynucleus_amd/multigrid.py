"""Solver side of the nonlocal operators on the GPU: geometric multigrid on a hierarchy of assembled
nonlocal operators, multigrid-preconditioned CG, the theta time stepper of the fractional heat equation.

Host-side mirror of
  * ``fractionalLevel`` / ``paramsForFractionalHierarchy`` (nl/PyNucleus_nl/helpers.py:312-411): one uniformly refined mesh
    per level, the operator assembled on every level (here: ``nonlocalBuilder.getDense`` on the device),
  * ``buildRestriction_{1,2}D_P1`` / ``_P2`` (multilevelSolver/PyNucleus_multilevelSolver/restriction_*.pxi): R = P^T, P = the
    coarse shape functions at the fine nodes,
  * ``multigrid`` (multigrid_{SCALAR}.pxi:86-390): V cycle, Jacobi smoother (omega = 2/3, one pre- and one post-sweep),
    direct coarse solve; ``asPreconditioner`` (:293-295),
  * ``CrankNicolson`` / ``ImplicitEuler`` (base/PyNucleus_base/timestepping.py:64-160) driving
    ``discretizedTransientProblem`` (nl/PyNucleus_nl/discretizedProblems.py:722-905).

The cycle, the CG loop and the time step run inside libpnl_hip.so (csrc/pnl_solver.hip: pnl_mg_cycle, pnl_mg_solve,
pnl_mg_cg, pnl_theta_step); this module builds the hierarchy, keeps the level data alive in HBM and hands pointers over.
No CPU fallback: without the HIP library / a GPU the constructors raise.
"""
import numpy as np
from . import _lib


# ---- transfer operators --------------------------------------------------------------------------------------------
def _vertex_dofs(dm):
    """DoF of every mesh vertex (-1: boundary) for a P1 map"""
    v2d = np.full(dm.mesh.num_vertices, -1, dtype=np.int64)
    v2d[dm.mesh.cells.ravel()] = dm.dofs.ravel()
    return v2d


# parent barycentrics of the vertices of the children of a uniformly refined cell, in the child numbering of mesh.refine()
# (1D: (c0, m), (m, c1); 2D: (c0, m01, m02), (c1, m12, m01), (c2, m02, m12), (m01, m12, m02))
_CHILDREN = {
    1: np.array([[[1., 0.], [.5, .5]], [[.5, .5], [0., 1.]]]),
    2: np.array([[[1., 0., 0.], [.5, .5, 0.], [.5, 0., .5]], [[0., 1., 0.], [0., .5, .5], [.5, .5, 0.]],
                 [[0., 0., 1.], [.5, 0., .5], [0., .5, .5]], [[.5, .5, 0.], [0., .5, .5], [.5, 0., .5]]]),
}


def buildProlongation(dm_coarse, dm_fine):
    """P (n_fine x n_coarse, scipy CSR) between the Lagrange spaces of a mesh and its uniform refinement: the coarse shape
    functions evaluated at the fine nodes, P[I, J] = phi_J(x_I).  For P1 that is 1 at a coincident vertex and 1/2 at the two
    ends of a bisected edge, for P2 the weights 1, 3/8, -1/8, 3/4, 1/2, 1/4 -- the transposes of buildRestriction_{1,2}D_P1 /
    _P2 (multilevelSolver/PyNucleus_multilevelSolver/restriction_*.pxi), which tabulate exactly these values; the elements of
    the two levels may differ (P1 coarse, P2 fine)."""
    import scipy.sparse as sp
    mc, mf = dm_coarse.mesh, dm_fine.mesh
    dim = mc.manifold_dim
    if dim not in _CHILDREN:
        raise NotImplementedError(dim)
    CH = _CHILDREN[dim]
    nchild = CH.shape[0]
    assert mf.num_cells == nchild*mc.num_cells and (mf.cells[0::nchild, 0] == mc.cells[:, 0]).all(), \
        'the fine mesh is not the uniform refinement of the coarse one'
    rows, cols, vals = [], [], []
    for ch in range(nchild):
        bary = dm_fine.nodes@CH[ch]                                  # parent barycentrics of the nodes of child ch
        W = np.asarray(dm_coarse.evalShapeFunctions(np.ascontiguousarray(bary.T)))   # [dpe_coarse, dpe_fine]
        W[np.abs(W) < 1e-14] = 0.
        fd = dm_fine.dofs[ch::nchild]                               # [nc, dpe_fine]
        cd = dm_coarse.dofs                                         # [nc, dpe_coarse]
        jj, ii = np.nonzero(W)
        for j, i in zip(jj, ii):
            m = (fd[:, i] >= 0) & (cd[:, j] >= 0)
            rows.append(fd[m, i]); cols.append(cd[m, j]); vals.append(np.full(int(m.sum()), W[j, i]))
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    # a node shared by several cells gets the same weight from each of them: keep one copy
    key = rows.astype(np.int64)*dm_coarse.num_dofs+cols
    _, first = np.unique(key, return_index=True)
    P = sp.csr_matrix((vals[first], (rows[first], cols[first])), shape=(dm_fine.num_dofs, dm_coarse.num_dofs))
    P.sort_indices()
    return P


def buildRestriction(dm_coarse, dm_fine):
    R = buildProlongation(dm_coarse, dm_fine).T.tocsr()
    R.sort_indices()
    return R


class _DevCSR:
    """CSR matrix in HBM (int32 indices, fp64 values) for pnl_csr_matvec"""

    def __init__(self, M, device):
        import torch
        M = M.tocsr()
        M.sort_indices()
        self.shape = M.shape
        self.indptr = torch.from_numpy(M.indptr.astype(np.int32)).to(device)
        self.indices = torch.from_numpy(M.indices.astype(np.int32)).to(device)
        self.data = torch.from_numpy(M.data.astype(np.float64)).to(device)
        self.host = M

    def matvec(self, ctx, x, alpha=1., beta=0., y=None):
        import torch
        if y is None:
            y = torch.zeros(self.shape[0], dtype=torch.float64, device=x.device)
        ctx.csr_matvec(self.shape[0], self.indptr.data_ptr(), self.indices.data_ptr(), self.data.data_ptr(), x.data_ptr(), alpha, beta,
                       y.data_ptr())
        return y

    def todense_dev(self):
        import torch
        return torch.from_numpy(self.host.toarray()).to(self.data.device)


# ---- hierarchy ------------------------------------------------------------------------------------------------------
def _seed_mesh(domain):
    from .mesh import simpleInterval, uniform_disc
    if domain == 'interval':
        mesh = simpleInterval(-1., 1.)
    elif domain == 'disc':
        mesh = uniform_disc(1.)
    else:
        raise NotImplementedError(domain)
    # the factory mesh is refined until a P1 space has a DoF (nonlocalProblems.py:209-212)
    while mesh.num_vertices-mesh.boundaryVertices.shape[0] == 0:
        mesh = mesh.refine()
    return mesh


class fractionalHierarchy:
    """Levels 0 .. noRef of uniformly refined meshes with the nonlocal operator assembled on every one of them
    (helpers.py:312-380 with 'assemble': 'ALL').  levels[l] = {'mesh', 'DoFMap', 'A' (Dense_LinearOperator in HBM),
    'M' (scipy CSR, buildMass), 'P', 'R' (scipy CSR; l > 0)}."""

    def __init__(self, domain, noRef, kernel, params=None, element='P1', buildMass=False, tag=None, device=None, mesh=None,
                 matrixFormat='dense', h2MinDoFs=2000):
        from .dofmap import dofmapFactory
        from .mesh import PHYSICAL
        from .builder import nonlocalBuilder
        if element not in ('P0', 'P1', 'P2', 'P3'):
            raise NotImplementedError('hierarchies are built for P0 .. P3 elements (restriction_*_P{{0,1,2,3}}.pxi); got {}'.format(element))
        self.kernel, self.params = kernel, dict(params or {})
        mesh = mesh if mesh is not None else _seed_mesh(domain)
        self.levels = []
        self._builders = []
        for lvl in range(noRef+1):
            if lvl > 0:
                mesh = mesh.refine()
            dm = dofmapFactory(element, mesh, PHYSICAL if tag is None else tag)
            b = nonlocalBuilder(dm, kernel, dict(self.params), device=device)
            # matrixFormat 'H2': the levels with at least h2MinDoFs DoFs carry the H2 operator (getH2), the coarse ones stay dense
            A = b.getH2() if (matrixFormat.upper() == 'H2' and dm.num_dofs >= h2MinDoFs) else b.getDense()
            b.context().synchronize()
            L = {'mesh': mesh, 'DoFMap': dm, 'A': A}
            if buildMass:
                L['M'] = dm.assembleMass()
            if lvl > 0:
                L['P'] = buildProlongation(self.levels[-1]['DoFMap'], dm)
                L['R'] = L['P'].T.tocsr()
            self.levels.append(L)
            self._builders.append(b)

    @property
    def finest(self):
        return self.levels[-1]

    def context(self):
        return self._builders[-1].context()

    def getLevelList(self):
        return self.levels


def buildTransientHierarchy(levels, alpha, beta):
    """levels with A <- alpha M + beta A (discretizedProblems.py:740-749), formed in HBM"""
    import torch
    from .linear_operators import Dense_LinearOperator
    out = []
    for L in levels:
        A = L['A']
        A.ctx.synchronize()
        T = (A.A*beta).contiguous()
        M = L['M'].tocoo()                                   # the mass matrix stays sparse: its entries are added on the device
        dev = A.A.device
        T.index_put_((torch.from_numpy(M.row.astype(np.int64)).to(dev), torch.from_numpy(M.col.astype(np.int64)).to(dev)),
                     torch.from_numpy(alpha*M.data.astype(np.float64)).to(dev), accumulate=True)
        N = {k: v for k, v in L.items() if k in ('P', 'R', 'mesh', 'DoFMap', 'M')}
        N['A'] = Dense_LinearOperator(T, A.ctx)
        out.append(N)
    return out


# ---- multigrid --------------------------------------------------------------------------------------------------------
class multigrid:
    """multigrid(hierarchy, smoother=('jacobi', {'omega': 2/3})) like the reference's solver class; ``hierarchy`` is a
    fractionalHierarchy or a list of level dicts with 'A' (Dense_LinearOperator), 'R', 'P'."""

    def __init__(self, hierarchy, smoother=('jacobi', {'omega': 2.0/3.0}), ctx=None, native=None):
        import torch
        levels = hierarchy.getLevelList() if hasattr(hierarchy, 'getLevelList') else list(hierarchy)
        if len(levels) < 1:
            raise AssertionError('empty hierarchy')
        name, sp = smoother if isinstance(smoother, tuple) else (smoother, {})
        if name not in ('jacobi', 'chebyshev'):
            raise NotImplementedError('smoother {}: Jacobi (library cycle) and Chebyshev (operator-agnostic cycle) are built'.format(name))
        self.smootherType, self.smootherParams = name, dict(sp)
        if name == 'chebyshev':
            native = False                                     # the polynomial smoother runs through the operators' matvecs
        self.omega = float(sp.get('omega', 2.0/3.0))
        self.presmoothingSteps = int(sp.get('presmoothingSteps', 1))
        self.postsmoothingSteps = int(sp.get('postsmoothingSteps', 1))
        from .linear_operators import Dense_LinearOperator
        self.levels = levels
        self.A = levels[-1]['A']
        self.maxIter = 50
        self.tolerance = 1e-8
        self.num_rows = self.A.num_rows
        self._mg = None
        # every level a dense operator: the cycle runs inside the library.  Otherwise (H2 / sparse levels) the same cycle is
        # driven from here over the operators' device matvecs (vectors stay in HBM, transfer operators through pnl_csr_matvec)
        from .h2 import H2Matrix
        from .linear_operators import CSR_LinearOperator, SSS_LinearOperator
        dense_below = all(isinstance(L['A'], Dense_LinearOperator) for L in levels[:-1])
        top = levels[-1]['A']
        # an H2 operator on the FINEST level (full CSR near field) runs inside the library cycle too (pnl_mg_level_desc.kind = 1)
        self._h2_top = top if (isinstance(top, H2Matrix) and len(levels) > 1 and isinstance(top.Anear, CSR_LinearOperator)
                               and not isinstance(top.Anear, SSS_LinearOperator)) else None
        self._native = dense_below and (isinstance(top, Dense_LinearOperator) or self._h2_top is not None) and native is not False
        if not self._native:
            self._h2_top = None
            self._setup_generic(ctx)
            return
        self.ctx = ctx or self.A.ctx
        self.device = self.A.A.device if self._h2_top is None else self.A.device
        # level data in HBM (kept alive by this object: the library only stores the pointers)
        self._keep = []
        descs = []
        for l, L in enumerate(levels):
            A = L['A']
            A.ctx.synchronize()
            d = _lib.pnl_mg_level_desc()
            d.n = A.num_rows
            if A is self._h2_top:
                near = A.Anear
                near._bind()
                ip, ix = near._pattern_dev if near._pattern_dev is not None else (torch.from_numpy(near.indptr).to(self.device),
                                                                                  torch.from_numpy(near.indices).to(self.device))
                diag = torch.from_numpy(np.ascontiguousarray(near.diagonal, dtype=np.float64)).to(self.device)
                self._keep += [ip, ix, diag]
                d.kind = 1
                d.near_indptr_dev, d.near_indices_dev, d.near_data_dev = ip.data_ptr(), ix.data_ptr(), near.data_dev.data_ptr()
            else:
                d.A_dev = A.A.data_ptr()
                d.ldA = A.A.stride(0)
                # a symmetric dense level is applied from its upper triangle (half the HBM bytes); below 8192 rows the
                # one-sided kernel's grid fills the chip better
                if getattr(A, 'symmetric', False) and A.num_rows >= 8192:
                    d.kind = 2
                diag = torch.diagonal(A.A).contiguous().clone()
                self._keep.append(diag)
            d.diag_dev = diag.data_ptr()
            if l > 0:
                R, P = _DevCSR(L['R'], self.device), _DevCSR(L['P'], self.device)
                assert R.shape == (levels[l-1]['A'].num_rows, A.num_rows) and P.shape == (A.num_rows, levels[l-1]['A'].num_rows)
                self._keep += [R, P]
                d.R_indptr_dev, d.R_indices_dev, d.R_data_dev = R.indptr.data_ptr(), R.indices.data_ptr(), R.data.data_ptr()
                d.P_indptr_dev, d.P_indices_dev, d.P_data_dev = P.indptr.data_ptr(), P.indices.data_ptr(), P.data.data_ptr()
            descs.append(d)
        A0 = levels[0]['A']
        # coarse solver: the inverse of the coarsest operator (a handful of DoFs), applied with the GEMV
        A0.ctx.synchronize()
        self._coarse_inv = torch.from_numpy(np.linalg.inv(A0.A.cpu().numpy())).to(self.device).contiguous()
        torch.cuda.current_stream(self.device).synchronize()
        self._set_stream()
        self._mg = self.ctx.mg_create(descs, self._coarse_inv.data_ptr(), self.omega, self.presmoothingSteps, self.postsmoothingSteps)

    def _setup_generic(self, ctx):
        import torch
        A = self.A
        self.device = getattr(A, 'device', None)
        if self.device is None:
            self.device = A.A.device
        self.ctx = ctx or A.ctx
        self._g = []
        for l, L in enumerate(self.levels):
            op = L['A']
            G = {'A': op, 'n': op.num_rows}
            if l > 0:
                if self.smootherType == 'chebyshev':
                    G['cheb'] = self._chebyshev_coefficients(op)
                else:
                    d = op.diagonal
                    d = d() if callable(d) else d
                    G['invD'] = self.omega/self._vec(d)
                G['R'], G['P'] = _DevCSR(L['R'], self.device), _DevCSR(L['P'], self.device)
            self._g.append(G)
        A0 = self.levels[0]['A']
        self._coarse_inv = torch.from_numpy(np.linalg.inv(np.asarray(A0.toarray()))).to(self.device).contiguous()

    def estimateSpectralRadius(self, op, eps=1e-3, kMax=50):
        """power method (base/PyNucleus_base/linalg.pyx:811-829) on the device.  Two deliberate differences: the reference starts from
        a random vector on the unit sphere and stops on an ABSOLUTE change of eps -- with the eigenvalues of these operators
        (0.1 and below) that is after two steps, and the estimate (root mean square of the spectrum) can lie below the largest
        eigenvalue by more than the 10 % the upper bound allows: the smoother then amplifies the top of the spectrum.  Here: a
        fixed pseudo-random start vector (numpy default_rng(0): reproducible), a RELATIVE change of eps = 1e-3 and up to 50
        steps."""
        import torch
        n = op.num_rows
        x0 = np.random.default_rng(0).standard_normal(n)
        x = torch.from_numpy(x0/np.linalg.norm(x0)).to(self.device)
        lold, lam, k = 0., 1., 0
        while abs(lam-lold) > eps*abs(lam) and k <= kMax:
            x = op.matvec(x)
            lold = lam
            lam = float(torch.linalg.norm(x))
            x = x/lam
            k += 1
        return lam

    def _chebyshev_coefficients(self, op):
        """chebyshevPreconditioner.__init__ (multilevelSolver/PyNucleus_multilevelSolver/smoothers.pyx:390-424): the polynomial
        with the Chebyshev roots of [rhoA lowerBound, rhoA upperBound], scaled to C(0) = 1; p(A) r = sum_k coeffs[k] A^(deg-1-k) r"""
        sp = self.smootherParams
        degree = int(sp.get('degree', 3))
        rhoA = float(sp.get('rhoA', 0.))
        if rhoA == 0.:
            rhoA = self.estimateSpectralRadius(op)
        a, b = rhoA*float(sp.get('lowerBound', 1.0/30.0)), rhoA*float(sp.get('upperBound', 1.1))
        std_roots = np.cos(np.pi*(np.arange(degree, dtype=np.float64)+0.5)/degree)
        scaled_poly = np.poly(0.5*(b-a)*(1+std_roots)+a)
        scaled_poly /= np.polyval(scaled_poly, 0.)
        return -scaled_poly[:-1]

    def _generic_smooth(self, l, b, x, steps, simple):
        G = self._g[l]
        for _ in range(steps):
            res = b if simple else b-G['A'].matvec(x)
            simple = False
            if 'cheb' in G:
                # chebyshevPreconditioner.matvec (smoothers.pyx:426-436): y = c_0 r; y = c_k r + A y
                c = G['cheb']
                y = c[0]*res
                for ck in c[1:]:
                    y = ck*res+G['A'].matvec(y)
                x.add_(y)
            else:
                x.addcmul_(G['invD'], res)

    def _generic_level(self, l, b, x, simple):
        """multigrid.solveOnLevel (multigrid_{SCALAR}.pxi:237-292) over operator matvecs; x is updated in place"""
        import torch
        if l == 0:
            x.copy_(self._coarse_inv@b)
            return
        G = self._g[l]
        self._generic_smooth(l, b, x, self.presmoothingSteps, simple)
        res = b-G['A'].matvec(x) if not (simple and self.presmoothingSteps == 0) else b
        self._set_stream()
        defect = G['R'].matvec(self.ctx, res.contiguous())
        solcg = torch.zeros(self._g[l-1]['n'], dtype=torch.float64, device=self.device)
        self.ctx.synchronize()
        self._generic_level(l-1, defect, solcg, True)
        self._set_stream()
        G['P'].matvec(self.ctx, solcg, alpha=1., beta=1., y=x)
        self.ctx.synchronize()
        self._generic_smooth(l, b, x, self.postsmoothingSteps, False)

    def _set_stream(self):
        import torch
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        if getattr(self, '_h2_top', None) is not None:
            # the library cycle applies the H2 operator that is set up in the context: make it this level's
            self._h2_top._ensure_setup()
            self._h2_top.Anear._bind()

    def __del__(self):
        try:
            if getattr(self, '_mg', None):
                self.ctx.mg_destroy(self._mg)
                self._mg = None
        except Exception:
            pass

    def _vec(self, v):
        import torch
        if isinstance(v, torch.Tensor):
            return v.to(device=self.device, dtype=torch.float64).contiguous()
        return torch.from_numpy(np.ascontiguousarray(np.asarray(v, dtype=np.float64))).to(self.device)

    def _ret(self, like, xd):
        import torch
        return xd if isinstance(like, torch.Tensor) else xd.cpu().numpy()

    def cycle(self, b, x=None):
        """one V cycle (solveOnLevel on the finest level); returns the new iterate"""
        import torch
        bd = self._vec(b)
        zero = x is None
        xd = torch.zeros_like(bd) if zero else self._vec(x).clone()
        if not self._native:
            torch.cuda.current_stream(self.device).synchronize()
            self._generic_level(len(self.levels)-1, bd, xd, zero)
            return self._ret(b, xd)
        self._set_stream()
        self.ctx.mg_cycle(self._mg, bd.data_ptr(), xd.data_ptr(), zero)
        self.ctx.synchronize()
        return self._ret(b, xd)

    def solve(self, b, x=None, tol=None, maxiter=None):
        """multigrid.solve: returns (x, iterations, residual norms)"""
        import torch
        bd = self._vec(b)
        zero = x is None
        xd = torch.zeros_like(bd) if zero else self._vec(x).clone()
        tol = self.tolerance if tol is None else tol
        maxiter = self.maxIter if maxiter is None else maxiter
        if not self._native:
            res = [float(torch.linalg.norm(bd if zero else bd-self.A.matvec(xd)))]
            its = 0
            while res[-1] > tol and its < maxiter:
                its += 1
                self._generic_level(len(self.levels)-1, bd, xd, zero)
                zero = False
                res.append(float(torch.linalg.norm(bd-self.A.matvec(xd))))
            return self._ret(b, xd), its, res
        self._set_stream()
        its, res = self.ctx.mg_solve(self._mg, bd.data_ptr(), xd.data_ptr(), tol, maxiter, zero)
        return self._ret(b, xd), its, res

    def cg(self, b, x=None, tol=1e-8, maxiter=100, A=None):
        """CG on A (default: the finest operator) preconditioned by one V cycle: (x, iterations, sqrt(r.Br) history)"""
        import torch
        bd = self._vec(b)
        zero = x is None
        if not self._native:
            from .solvers import cg as _cg
            xd, its, res = _cg(A if A is not None else self.A, bd, x0=None if zero else self._vec(x), tol=tol, maxiter=maxiter,
                               preconditioner=self.asPreconditioner())
            return self._ret(b, xd), its, res
        xd = torch.zeros_like(bd) if zero else self._vec(x).clone()
        self._set_stream()
        if A is not None and not hasattr(A, 'A'):
            raise NotImplementedError('cg(A=...) inside the library takes a dense operator (the default is the finest level)')
        Aptr, ld = (A.A.data_ptr(), A.A.stride(0)) if A is not None else (None, 0)
        its, res = self.ctx.mg_cg(self._mg, Aptr, ld, bd.data_ptr(), xd.data_ptr(), tol, maxiter, zero)
        return self._ret(b, xd), its, res

    def asPreconditioner(self, maxIter=1):
        """callable r -> B r: maxIter V cycles from a zero guess (multigridPreconditioner)"""
        def B(r):
            z = self.cycle(r)
            for _ in range(maxIter-1):
                z = self.cycle(r, z)
            return z
        return B

    def __str__(self):
        sm = ('Chebyshev (degree {})'.format(int(self.smootherParams.get('degree', 3))) if self.smootherType == 'chebyshev'
              else 'Jacobi ({}/{} sweeps, {:.3} damping)'.format(self.presmoothingSteps, self.postsmoothingSteps, self.omega))
        return 'V-cycle multigrid, {} levels, {}, DoFs {}'.format(len(self.levels), sm, [L['A'].num_rows for L in self.levels])


# ---- time stepping --------------------------------------------------------------------------------------------------------
class CrankNicolson:
    """Theta method for M u_t + S u = g(t) on the finest level of a hierarchy with mass matrices (timestepping.py:64-112):
    (M/dt + theta S) u_{k+1} = (M/dt) u_k - (1 - theta) S u_k + (1 - theta) g(t_k) + theta g(t_{k+1}).
    The system is solved by multigrid-preconditioned CG on the transient hierarchy alpha M + beta A
    (buildTransientSolver, discretizedProblems.py:751-768) inside pnl_theta_step."""

    def __init__(self, hierarchy, dt, theta=0.5, tol=1e-8, maxiter=100, smoother=('jacobi', {'omega': 2.0/3.0})):
        assert 0. <= theta <= 1. and dt > 0.
        levels = hierarchy.getLevelList() if hasattr(hierarchy, 'getLevelList') else list(hierarchy)
        self.dt, self.theta, self.tol, self.maxiter = float(dt), float(theta), tol, maxiter
        self.S = levels[-1]['A']
        self.device = self.S.A.device
        self.M = _DevCSR(levels[-1]['M'], self.device)
        self.transient = buildTransientHierarchy(levels, 1./self.dt, self.theta)
        self.solver = multigrid(self.transient, smoother=smoother)
        if not getattr(self.solver, '_native', False):
            # pnl_theta_step drives the library's own V cycle: dense levels and the Jacobi smoother
            raise NotImplementedError('time stepping needs the library multigrid (dense levels, Jacobi smoother); '
                                      'got smoother={!r} / a level that is not dense'.format(smoother))
        self.iterations = []

    def setRHS(self, g_t, g_tdt):
        """forcing of one step from the load vectors at t and t + dt (setRHS, timestepping.py:76-91)"""
        return (1.-self.theta)*np.asarray(g_t)+self.theta*np.asarray(g_tdt)

    def step(self, t, u, forcing):
        """advance u (a device vector, overwritten) from t to t + dt; returns t + dt"""
        import torch
        mg = self.solver
        f = mg._vec(forcing)
        assert isinstance(u, torch.Tensor) and u.device == self.device and u.dtype == torch.float64 and u.is_contiguous()
        mg._set_stream()
        its, res = mg.ctx.theta_step(mg._mg, self.S.A.data_ptr(), self.S.A.stride(0), self.M.indptr.data_ptr(), self.M.indices.data_ptr(),
                                     self.M.data.data_ptr(), self.dt, self.theta, f.data_ptr(), u.data_ptr(), self.tol, self.maxiter)
        self.iterations.append(its)
        return t+self.dt


class ImplicitEuler(CrankNicolson):
    def __init__(self, hierarchy, dt, **kwargs):
        super().__init__(hierarchy, dt, theta=1., **kwargs)


def determineTimeSteps(h, finalTime, timeStepperType='Crank-Nicolson'):
    """discretizedProblems.py:774-783: dt = sqrt(h) (Crank-Nicolson) or h (implicit Euler), rounded to divide finalTime"""
    dt = np.sqrt(h) if timeStepperType == 'Crank-Nicolson' else h
    n = int(np.around(finalTime/dt))
    return finalTime/n, n


def solveFractionalHeat(hierarchy, initial, load, finalTime=1.0, timeStepperType='Crank-Nicolson', theta=0.5, tol=1e-8, maxiter=100):
    """discretizedTransientProblem.solve (discretizedProblems.py:889-905): u_t + (-Laplace)^s u = f on the finest level,
    u(0) = interpolant of ``initial``; ``load(t)`` returns the load vector int f(t) phi_i.  Returns (times, [u_k] on the host,
    stepper)."""
    import torch
    L = hierarchy.finest if hasattr(hierarchy, 'finest') else hierarchy[-1]
    dm = L['DoFMap']
    dt, nt = determineTimeSteps(dm.mesh.h, finalTime, timeStepperType)
    stepper = (CrankNicolson(hierarchy, dt, theta=theta, tol=tol, maxiter=maxiter) if timeStepperType == 'Crank-Nicolson'
               else ImplicitEuler(hierarchy, dt, tol=tol, maxiter=maxiter))
    times = np.linspace(0., finalTime, nt+1)
    u0 = np.asarray(dm.interpolate(initial), dtype=np.float64)
    u = torch.from_numpy(u0.copy()).to(stepper.device)
    us = [u0]
    g_prev = np.asarray(load(times[0]))
    t = 0.
    for k in range(nt):
        g_next = np.asarray(load(times[k+1]))
        t = stepper.step(t, u, stepper.setRHS(g_prev, g_next))
        g_prev = g_next
        stepper.solver.ctx.synchronize()
        us.append(u.cpu().numpy().copy())
    assert abs(t-finalTime) < 1e-10
    return times, us, stepper
