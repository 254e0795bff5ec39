"""Host-side setup of the element-pair quadrature: everything the reference's
local-matrix objects precompute in their constructors, flattened into plain
arrays that are uploaded once to the GPU.

Reference behaviour followed (paths under /root/reference/nl/PyNucleus_nl):
  fractionalLaplacian2D.pyx:587-620   setKernel: target order, quad_order_diagonal(V)
  fractionalLaplacian2D.pyx:622-642   getQuadOrder (constants of the order formula)
  fractionalLaplacian2D.pyx:644-813   getNearQuadRule: merged-DoF PSI tables
  fractionalLaplacian2D.pyx:1207-1253 boundary setKernel / getQuadOrder
  fractionalLaplacian2D.pyx:1255-1314 boundary getNearQuadRule
  fractionalLaplacian1D.pyx:203-253, 255-330, 626-712   the same in 1D
  nonlocalOperator_{SCALAR}.pxi:66-109  precomputePermutations (DoF permutation table)
  nonlocalOperator_{SCALAR}.pxi:549-600, 988-1020 addQuadRule / addQuadRule_boundary
"""
from itertools import permutations
import numpy as np
from .quadrature import (COMMON_VERTEX, COMMON_EDGE, COMMON_FACE, simplexXiaoGimbutas, simplexDuffyTransformation,
                         singularityCancelationQuadRule1D, singularityCancelationQuadRule1D_boundary,
                         singularityCancelationQuadRule2D, singularityCancelationQuadRule2D_boundary)
from .kernels import GAUSSIAN, EXPONENTIAL, FRACTIONAL, ball2_retriangulation

MAX_PANEL = 120       # nonlocalOperator.pyx:107
QCAP_DEFAULT = 60     # highest distant order we tabulate up-front (the reference adds rules lazily)


def lehmer_rank(perm):
    n = len(perm)
    fact = [1, 1, 2, 6, 24]
    idx = 0
    for i in range(n):
        smaller = sum(1 for j in range(i) if perm[j] < perm[i])
        idx += (perm[i]-smaller)*fact[n-1-i]
    return idx


def dof_permutation_table(dm):
    """table[rank(perm), dofPerm] = dofOrig with nodes[dofPerm, j] == nodes[dofOrig, perm[j]]"""
    nV = dm.mesh.manifold_dim+1
    dpe = dm.dofs_per_element
    table = np.zeros((int(np.prod(range(1, nV+1))), dpe), dtype=np.int32)
    for p in permutations(range(nV)):
        r = lehmer_rank(p)
        for dofPerm in range(dpe):
            for dofOrig in range(dpe):
                if np.abs(dm.nodes[dofPerm, :]-dm.nodes[dofOrig, list(p)]).max() < 1e-10:
                    table[r, dofPerm] = dofOrig
                    break
            else:
                raise NotImplementedError()
    return table


class orderFormula:
    """order = max(ceil((c0 + a*L_other + b*Lmax - e*logdh_other)/(max(logdh_self,0)+den0)), 2)"""

    def __init__(self, c0, a, b, e, den0, clip_num):
        self.c0, self.a, self.b, self.e, self.den0, self.clip_num = float(c0), float(a), float(b), float(e), float(den0), bool(clip_num)

    def __call__(self, H0, h1, h2, d):
        """vectorised host evaluation (used for statistics and tests only)"""
        logdh1, logdh2 = np.log(d/h1), np.log(d/h2)
        L1, L2 = np.abs(np.log(h1/H0)), np.abs(np.log(h2/H0))
        Lm = np.maximum(L1, L2)
        n1, n2 = (np.maximum(logdh1, 0.), np.maximum(logdh2, 0.)) if self.clip_num else (logdh1, logdh2)
        p1 = np.ceil((self.c0+self.a*L2+self.b*Lm-self.e*n2)/(np.maximum(logdh1, 0.)+self.den0))
        p2 = np.ceil((self.c0+self.a*L1+self.b*Lm-self.e*n1)/(np.maximum(logdh2, 0.)+self.den0))
        return np.maximum(np.maximum(p1, 2.), np.maximum(p2, 2.)).astype(np.int64)


class singularRule:
    def __init__(self, qr, psi, phi0=None, phi1=None):
        self.nodes = np.ascontiguousarray(qr.nodes)
        self.weights = np.ascontiguousarray(qr.weights)
        self.psi = np.ascontiguousarray(psi)
        self.num_nodes = qr.num_nodes
        self.rows = psi.shape[0]
        # non-symmetric local matrices (FL2:965-1075): psi = phi0 - phi1 with the x and the y parts kept apart
        self.phi0, self.phi1 = phi0, phi1


class nonlocalTables:
    """All precomputed quadrature data for one (DoFMap, kernel, params) triple."""

    def __init__(self, dm, kernel, params=None, zeroExterior=True, qcap=QCAP_DEFAULT):
        params = params or {}
        mesh = dm.mesh
        self.dm, self.kernel = dm, kernel
        self.params = dict(params)                          # the caller's inputs, as given
        self.dim = dim = mesh.dim
        assert mesh.manifold_dim == dim and dim in (1, 2)
        assert kernel.dim == dim, 'Kernel dimension must match dm.mesh dimension'
        self.pointwise = bool(getattr(kernel, 'pointwise', False))
        if self.pointwise:
            self.variable = True
            self._setup_pointwise(dm, kernel, params, zeroExterior, qcap)
            return
        # piecewise-constant order with s(label1, label2) != s(label2, label1): the non-symmetric local-matrix classes and both
        # orientations of every pair (NA:1411-1428), each with the parameters evalParams finds for that orientation
        self.nonsym = bool(kernel.variable and not kernel.symmetric)
        self.variable = bool(kernel.variable)
        if self.variable:
            self._setup_variable(dm, kernel, params, zeroExterior, qcap)
            return
        self.classes = None
        self.dpe = dm.dofs_per_element
        self.num_dofs = dm.num_dofs
        self.hmin = mesh.hmin
        self.H0 = mesh.diam/np.sqrt(8)                      # NO:435
        self.dof_perm_table = dof_permutation_table(dm)
        self.qcap = qcap
        target_order = params.get('target_order', None)
        self.zeroExterior = bool(zeroExterior) and not kernel.finiteHorizon      # NA:919-922
        # the _nonsym constructors drop the caller's target_order and quad_order_diagonal (FL2:911, FL1:427 pass num_dofs in
        # their place); the boundary twins keep them
        interior_target = None if params.get('_nonsymInterior', False) else target_order
        interior_qd = None if params.get('_nonsymInterior', False) else params.get('quad_order_diagonal', None)

        sing = kernel.getSingularityValue()
        if dim == 2:
            self._setup2D(kernel, interior_target, interior_qd)
        else:
            self._setup1D(kernel, interior_target, interior_qd)
        self.singularityValue = sing
        self._distant_rules(qcap)
        # surface integrals (NA:953-955): the Gauss-theorem twin is built whenever it exists, not only for zeroExterior --
        # the cluster-local boundary term of assembleClusters (NA:1842-1889) needs it too
        # (the integrable kernels of the full space, Gaussian and exponential, have one as well: kernelsCy.pyx:418-477)
        # finite horizon, fractional kernel of constant order: the near field of the cluster method (NA:953-955, 1842-1889, 1915-1940) integrates
        # the exterior of a cluster pair with the twin of the SAME kernel on the full space (getModifiedKernel(horizon = inf), same
        # scaling) and subtracts the part beyond the horizon as a multiple of the mass matrix
        fh_near = kernel.finiteHorizon and kernel.kernelType == FRACTIONAL and not kernel.variable and isinstance(kernel.interaction, ball2_retriangulation) \
            and not hasattr(kernel.interaction, 'transform')
        self.has_boundary_tables = (kernel.kernelType in (FRACTIONAL, GAUSSIAN, EXPONENTIAL) and not kernel.finiteHorizon) or fh_near
        if self.zeroExterior and not self.has_boundary_tables:
            raise NotImplementedError('zeroExterior needs a fractional, Gaussian or exponential kernel on the full space')
        if self.has_boundary_tables:
            bk = kernel.getBoundaryKernel()
            if fh_near:
                # a class of a piecewise-constant order keeps the TRUNCATED twin (NA:1966-2156 integrate cluster surfaces and interfaces
                # with local_matrix_surface = kernel.getBoundaryKernel(): facets beyond the horizon drop out); the full-space twin gives
                # the value on the sphere (horizonSurfaceIntegral, nonlocalAssembly.pyx:132-175)
                self.boundaryKernelFull = kernel.getFullSpaceKernel().getBoundaryKernel()
                if not params.get('_classOfVariable', False):
                    bk = self.boundaryKernelFull
            if (kernel.min_singularity, kernel.max_singularity) != (sing, sing):
                # class table of a variable-order kernel: the boundary twin inherits the range of singularities too
                bk.min_singularity, bk.max_singularity = kernel.min_singularity+1., kernel.max_singularity+1.
            self.boundaryKernel = bk
            if dim == 2:
                self._setup2D_boundary(bk, target_order, params.get('quad_order_diagonal', None))
            else:
                self._setup1D_boundary(bk, target_order, params.get('quad_order_diagonal', None))
            self._boundary_mesh()

    # ------------------------------------------------------------------
    def _setup_variable(self, dm, kernel, params, zeroExterior, qcap):
        """Variable order, piecewise constant per element pair (NO:509-513: evalParams at the two cell centres before the
        panel is chosen; FL2:664, 688: near rules keyed by the singularity of the pair).  The order takes finitely many
        values: every distinct value is a CLASS with the tables of the corresponding constant-order kernel (whose
        near-field quadrature orders use the extreme singularities of the variable kernel); cells and boundary facets
        carry the label of their centre and cls_of[label1, label2] names the class of a pair."""
        sFun = kernel.s
        mesh = dm.mesh
        if hasattr(sFun, '_reps'):
            # lambdaFractionalOrder: the callable is tabulated over the points the assembly will ask about (cell centres, centres of the
            # boundary facets) before the table of its values is read
            surf = mesh.get_surface_mesh()
            sFun.labels(np.concatenate([mesh.vertices[mesh.cells].mean(axis=1), mesh.vertices[np.asarray(surf.cells)].mean(axis=1)]))
        vals = np.unique(sFun.sVals)
        self.class_s = vals
        self.cls_of = np.searchsorted(vals, sFun.sVals).astype(np.int32)          # [L, L]
        self.num_labels = int(sFun.numLabels)
        centers = mesh.vertices[mesh.cells].mean(axis=1)
        self.cell_labels = np.ascontiguousarray(sFun.labels(centers), dtype=np.int32)
        assert self.cell_labels.min() >= 0 and self.cell_labels.max() < self.num_labels
        cparams = dict(params or {})
        if not kernel.symmetric:
            cparams['_nonsymInterior'] = True
        if kernel.finiteHorizon:
            cparams['_classOfVariable'] = True
        self.classes = [nonlocalTables(dm, kernel.constantOrderKernel(sv), cparams, zeroExterior, qcap) for sv in vals]
        c0 = self.classes[0]
        for name in ('dm', 'dim', 'dpe', 'num_dofs', 'hmin', 'H0', 'dof_perm_table', 'qcap', 'zeroExterior', 'has_boundary_tables',
                     'dist_off', 'dist_bary', 'dist_w', 'dist_phi', 'bfacet_off', 'bfacet_bary', 'bfacet_w', 'target_order',
                     'quad_order_diagonal', 'quad_order_diagonalV'):
            setattr(self, name, getattr(c0, name))
        self.kernel = kernel
        self.singular = c0.singular
        if self.has_boundary_tables:
            self.bcells = c0.bcells
            fc = mesh.vertices[self.bcells].mean(axis=1)
            self.facet_labels = np.ascontiguousarray(sFun.labels(fc), dtype=np.int32)
        else:
            self.facet_labels = np.zeros(0, dtype=np.int32)

    # ------------------------------------------------------------------
    def _setup_pointwise(self, dm, kernel, params, zeroExterior, qcap):
        """Non-symmetric kernels with an order s(x) evaluated per quadrature point (fractionalLaplacian{1,2}D_nonsym,
        FL2:894-1184, FL1:410-604; kernel evaluation updateAndEvalFractional KC:596-622).  Per pair the reference sets the
        singularity from the largest order at the two centres and all vertices (evalParamsOnSimplices, KC:1825-1846): that
        value enters the order formula and keys the near-field rule (FL2:957, 990, 1075).  Here: per-cell / per-facet maxima,
        order-formula constants as functions of the pair's order, and -- on request -- the table of near rules over the
        distinct orders of the touching pairs.  The _nonsym constructors drop the caller's target_order and
        quad_order_diagonal (FL2:911, FL1:427 pass num_dofs in their place), the boundary twins keep them."""
        mesh = dm.mesh
        self.classes = None
        self.dpe = dm.dofs_per_element
        self.num_dofs = dm.num_dofs
        self.hmin = mesh.hmin
        self.H0 = mesh.diam/np.sqrt(8)
        self.dof_perm_table = dof_permutation_table(dm)
        self.qcap = qcap
        self.zeroExterior = bool(zeroExterior)
        dim = self.dim
        sF = kernel.s
        logh = abs(np.log(self.hmin/self.H0))
        if dim == 2:
            self.target_order = 0.5
            qd = max(np.ceil((self.target_order+1.+sF.max)/0.43*logh), 4)
            qdV = max(np.ceil((self.target_order+1.+sF.max)/0.7*logh), 4)
            self.quad_order_diagonal, self.quad_order_diagonalV = int(qd), int(qdV)
            self.sing_fac = 4.0
            self.pw_c0 = (0.5*self.target_order+0.5)*np.log(self.num_dofs*self.H0**2)
        else:
            self.target_order = dm.polynomialOrder+1-sF.min
            qd = max(np.ceil(((self.target_order+2.)*np.log(self.num_dofs*self.H0)+(2.*sF.max-1.)*logh)/0.8), 2)
            self.quad_order_diagonal = self.quad_order_diagonalV = int(qd)
            self.sing_fac = 1.0
            self.pw_c0 = (self.target_order+2.)*np.log(self.num_dofs*self.H0)
        self._distant_rules(qcap)
        self.order_type = int(sF.sFun.device_type)
        self.order_params = np.array(sF.sFun.device_params(), dtype=np.float64)
        verts = mesh.vertices[mesh.cells]                                  # [nc, nV, dim]
        fe = hasattr(sF.sFun, 'vertex_values')                             # feFractionalOrder: a P1 function, values at the vertices
        if fe:
            assert sF.sFun.vertex_values.shape[0] == mesh.num_vertices, 'feFractionalOrder: the order must live on the mesh of the assembly'
            self.order_vertex_values = np.ascontiguousarray(sF.sFun.vertex_values)
            cv = sF.sFun.cell_values(mesh.cells)
            self.cell_smax = np.maximum(cv.mean(axis=1), cv.max(axis=1))    # centre and vertices (evalParamsOnSimplices)
        else:
            self.cell_smax = np.maximum(sF.evalPoints(verts.mean(axis=1)), sF.evalPoints(verts).max(axis=1))
        self.has_boundary_tables = True
        self.boundaryKernel = bk = kernel.getBoundaryKernel()
        t_b = params.get('target_order', None)
        qd_b = params.get('quad_order_diagonal', None)
        if dim == 2:
            if t_b is None:
                t_b = 0.5
            if qd_b is None:
                qd_b = max(np.ceil((t_b+0.5+sF.max)/0.35*logh), 2)
            self.bsing_fac = -2.0
            self.pw_bc0 = (0.5*t_b+0.25)*np.log(self.num_dofs*self.H0**2)
        else:
            if t_b is None:
                t_b = dm.polynomialOrder+1-sF.min
            if qd_b is None:
                qd_b = max(np.ceil(((t_b+1.)*np.log(self.num_dofs*self.H0)+(2.*sF.max-1.)*logh)/0.8), 2)
            self.bsing_fac = 1.0
            self.pw_bc0 = (t_b+1.)*np.log(self.num_dofs*self.H0)
        self.bquad_order_diagonal = int(qd_b)
        self._boundary_mesh()
        fv = mesh.vertices[self.bcells]                                     # [nb, dim, dim]
        if fe:
            fvv = sF.sFun.cell_values(self.bcells)
            self.facet_smax = np.maximum(fvv.mean(axis=1), fvv.max(axis=1))
        else:
            self.facet_smax = np.maximum(sF.evalPoints(fv.mean(axis=1)), sF.evalPoints(fv).max(axis=1))
        self._pw_rules = None
        # the scaling C(s) over [s.min, s.max] as a Chebyshev series: a polynomial per quadrature point on the device instead
        # of two Gamma functions (the oracle keeps the formula)
        self.scaling_cheb = None
        if kernel.normalized and sF.max > sF.min:
            from numpy.polynomial import chebyshev as Ch
            mid, half = 0.5*(sF.max+sF.min), 0.5*(sF.max-sF.min)
            for deg in (15, 23, 31):
                c = Ch.chebinterpolate(lambda t: kernel.scalingOfOrder(mid+half*t), deg)
                t = np.linspace(-1., 1., 401)
                err = np.abs(Ch.chebval(t, c)/kernel.scalingOfOrder(mid+half*t)-1.).max()
                if err < 1e-14:
                    break
            if err < 1e-14:
                self.scaling_cheb = (mid, half, c)

    def pw_formula(self, sv, boundary=False):
        """order-formula constants of a pair whose largest order is sv (FL2:915-935, FL1:431-450; FL2:1226-1243, FL1:644-660)"""
        if not boundary:
            if self.dim == 2:
                return orderFormula(self.pw_c0, sv-1., 1., sv, 0.4, False)
            return orderFormula(self.pw_c0, 2.*sv-1., 0., 2.*sv, 0.8, False)
        if self.dim == 2:
            return orderFormula(self.pw_bc0, sv-1., 1., sv, 0.35, True)
        st = max(sv-0.5, 0.)
        return orderFormula(self.pw_bc0, 2.*st-1., 0., 2.*st, 0.8, False)

    def touching_pairs(self):
        """(c1 <= c2, number of shared vertices) of all cell pairs with a common vertex, and the same for (cell, boundary facet)"""
        from scipy.sparse import csr_matrix
        mesh = self.dm.mesh
        nc, nV = mesh.num_cells, self.dim+1
        B = csr_matrix((np.ones(nc*nV, dtype=np.int32), (np.repeat(np.arange(nc), nV), mesh.cells.ravel())),
                       shape=(nc, mesh.num_vertices))
        Cm = (B@B.T).tocoo()
        keep = Cm.row <= Cm.col
        pairs = np.stack([Cm.row[keep], Cm.col[keep], Cm.data[keep]], axis=1).astype(np.int32)
        nb = self.bcells.shape[0]
        F = csr_matrix((np.ones(nb*self.dim, dtype=np.int32), (np.repeat(np.arange(nb), self.dim), self.bcells.ravel())),
                       shape=(nb, mesh.num_vertices))
        Cb = (B@F.T).tocoo()
        bpairs = np.stack([Cb.row, Cb.col, Cb.data], axis=1).astype(np.int32)
        return pairs, bpairs

    def pw_rules(self):
        """near-field rules over the distinct orders of the touching pairs (the reference builds them lazily, keyed by the
        singularity value: FL2:957-1112, FL1:466-530; boundary FL2:1255-1314, FL1:672-712).
        Returns dict(keys, rules[slot] = (nodes[nk, 2nV, M], w[nk, M], phi0[nk, rows, M], phi1[nk, rows, M]),
        bkeys, brules[slot] = (nodes, w, phi), pairs[np, 4] = (c1, c2, common, key index), bpairs likewise)."""
        if self._pw_rules is not None:
            return self._pw_rules
        dim, nV = self.dim, self.dim+1
        pairs, bpairs = self.touching_pairs()
        sv = np.maximum(self.cell_smax[pairs[:, 0]], self.cell_smax[pairs[:, 1]])
        keys, kidx = np.unique(sv, return_inverse=True)
        bsv = np.maximum(self.cell_smax[bpairs[:, 0]], self.facet_smax[bpairs[:, 1]])
        # + the orders of touching (cell, facet) items that are not domain-boundary facets (cluster exterior of the near field)
        extra = getattr(self, '_pw_extra_bkeys', None)
        bkeys = np.unique(bsv) if extra is None else np.unique(np.concatenate([bsv, extra]))
        bkidx = np.searchsorted(bkeys, bsv)
        rules = {}
        dm_order = max(self.dm.polynomialOrder, 1)
        for slot in range(nV):
            panel = -(slot+1)
            acc = [[], [], [], []]
            for sval in keys:
                sing = -dim-2.*sval
                if dim == 2:
                    qr = singularityCancelationQuadRule2D(panel, 2.+sing, self.quad_order_diagonal, self.quad_order_diagonalV)
                else:
                    qr = singularityCancelationQuadRule1D(panel, 2.+sing, self.quad_order_diagonal, 2*dm_order)
                r = self._psi_tables({panel: qr})[panel]
                acc[0].append(r.nodes); acc[1].append(r.weights); acc[2].append(r.phi0); acc[3].append(r.phi1)
            rules[slot] = tuple(np.ascontiguousarray(np.stack(a)) for a in acc)
        brules = {}
        qd = self.bquad_order_diagonal
        for slot in range(dim):
            panel = -(slot+1)
            acc = [[], [], []]
            for sval in bkeys:
                sing = 1.-dim-2.*sval
                if dim == 2:
                    sg = sing if (panel == COMMON_VERTEX or sing > -2.+1e-3) else 2.+sing          # FL2:1271-1274
                    qr = singularityCancelationQuadRule2D_boundary(panel, sg, qd, qd)
                    phi = self.dm.evalShapeFunctions(qr.nodes[:3])
                else:
                    sg = sing if sing > -1.+1e-3 else 2.+sing                                       # FL1:686-689
                    qr = singularityCancelationQuadRule1D_boundary(COMMON_VERTEX, sg, qd, 1)
                    phi = self.dm.evalShapeFunctions(qr.nodes[:2])
                acc[0].append(qr.nodes); acc[1].append(qr.weights); acc[2].append(phi)
            brules[slot] = tuple(np.ascontiguousarray(np.stack(a)) for a in acc)
        # pairs of one rule next to each other: consecutive waves read the same tables
        po = np.lexsort((kidx, pairs[:, 2]))
        bo = np.lexsort((bkidx, bpairs[:, 2]))
        self._pw_rules = dict(keys=keys, rules=rules, bkeys=bkeys, brules=brules,
                              pairs=np.ascontiguousarray(np.column_stack([pairs, kidx])[po].astype(np.int32)),
                              bpairs=np.ascontiguousarray(np.column_stack([bpairs, bkidx])[bo].astype(np.int32)))
        return self._pw_rules

    def facet_order(self, facets):
        """largest order over the centre and the vertices of facets given by their vertex ids [n, dim] (the facet's share of
        evalParamsOnSimplices, KC:1825-1846)"""
        sF = self.kernel.s
        facets = np.asarray(facets)
        if hasattr(sF.sFun, 'vertex_values'):
            fvv = sF.sFun.cell_values(facets)
            return np.maximum(fvv.mean(axis=1), fvv.max(axis=1))
        fv = self.dm.mesh.vertices[facets]
        return np.maximum(sF.evalPoints(fv.mean(axis=1)), sF.evalPoints(fv).max(axis=1))

    def need_boundary_keys(self, sv):
        """make sure the boundary near rules exist for the orders sv (touching items of the near field's cluster exterior); True when
        the rule tables were rebuilt -- their key indices changed, a context must upload them again"""
        sv = np.unique(np.asarray(sv, dtype=np.float64))
        R = self.pw_rules()
        if np.isin(sv, R['bkeys']).all():
            return False
        have = getattr(self, '_pw_extra_bkeys', None)
        self._pw_extra_bkeys = sv if have is None else np.unique(np.concatenate([have, sv]))
        self._pw_rules = None
        return True

    def class_of_pair(self, c1, c2):
        return int(self.cls_of[self.cell_labels[c1], self.cell_labels[c2]])

    # ------------------------------------------------------------------
    def _psi_tables(self, rules):
        """merged-DoF PSI tables (FL2:662-811, FL1:255-330)."""
        dm, dim = self.dm, self.dim
        nV = dim+1
        dpe, dpv, dped = dm.dofs_per_element, dm.dofs_per_vertex, dm.dofs_per_edge
        out = {}
        for panel, qr in rules.items():
            px = dm.evalShapeFunctions(qr.nodes[:nV])
            py = dm.evalShapeFunctions(qr.nodes[nV:2*nV])
            common = -panel
            if common == nV:                                 # identical cells
                p0, p1 = px.copy(), py.copy()
            elif common == 1:
                p0 = np.zeros((2*dpe-dpv, qr.num_nodes))
                p1 = np.zeros_like(p0)
                for dof in range(dpv):
                    p0[dof], p1[dof] = px[dof], py[dof]
                for dof in range(dpv, dpe):
                    p0[dof] = px[dof]
                    p1[dpe+dof-dpv] = py[dof]
            elif common == 2:
                p0 = np.zeros((2*dpe-2*dpv-dped, qr.num_nodes))
                p1 = np.zeros_like(p0)
                for dof in range(2*dpv):
                    p0[dof], p1[dof] = px[dof], py[dof]
                for dof in range(nV*dpv, nV*dpv+dped):
                    p0[dof], p1[dof] = px[dof], py[dof]
                for dof in range(2*dpv, nV*dpv):
                    p0[dof] = px[dof]
                    p1[dpe+dof-2*dpv] = py[dof]
                for dof in range(nV*dpv+dped, dpe):
                    p0[dof] = px[dof]
                    p1[dpe+dof-2*dpv-dped] = py[dof]
            else:
                raise NotImplementedError()
            out[panel] = singularRule(qr, p0-p1, p0, p1)
        return out

    def _setup2D(self, kernel, target_order, quad_order_diagonal):
        if target_order is None:
            target_order = 0.5                               # FL2:600-604
        self.target_order = target_order
        smax = max(-0.5*(kernel.max_singularity+2), 0.)
        logh = abs(np.log(self.hmin/self.H0))
        if quad_order_diagonal is None:
            qd = max(np.ceil((target_order+1.+smax)/0.43*logh), 4)
            qdV = max(np.ceil((target_order+1.+smax)/0.7*logh), 4)
        else:
            qd = qdV = quad_order_diagonal
        self.quad_order_diagonal, self.quad_order_diagonalV = int(qd), int(qdV)
        sing = kernel.getSingularityValue()
        # singularity cancellation: 2 orders within and (for continuous elements) across elements
        across = 0. if self.dm.polynomialOrder == 0 else 2.   # FL2:594-598: discontinuous elements cancel nothing across elements
        rules = {COMMON_FACE: singularityCancelationQuadRule2D(COMMON_FACE, 2.+sing, self.quad_order_diagonal, self.quad_order_diagonalV),
                 COMMON_EDGE: singularityCancelationQuadRule2D(COMMON_EDGE, across+sing, self.quad_order_diagonal, self.quad_order_diagonalV),
                 COMMON_VERTEX: singularityCancelationQuadRule2D(COMMON_VERTEX, across+sing, self.quad_order_diagonal, self.quad_order_diagonalV)}
        self.singular = self._psi_tables(rules)
        self.sing_fac = 4.0                                  # FL2:851
        s = max(-0.5*(sing+2), 0.)
        c = (0.5*target_order+0.5)*np.log(self.num_dofs*self.H0**2)
        self.qo = orderFormula(c, s-1., 1., s, 0.4, False)

    def _setup1D(self, kernel, target_order, quad_order_diagonal):
        smin = max(-0.5*(kernel.min_singularity+1), 0.)
        smax = max(-0.5*(kernel.max_singularity+1), 0.)
        if target_order is None:
            target_order = self.dm.polynomialOrder+1-smin    # FL1:219-222
        self.target_order = target_order
        if quad_order_diagonal is None:
            quad_order_diagonal = max(np.ceil(((target_order+2.)*np.log(self.num_dofs*self.H0) +
                                               (2.*smax-1.)*abs(np.log(self.hmin/self.H0)))/0.8), 2)
        self.quad_order_diagonal = self.quad_order_diagonalV = int(quad_order_diagonal)
        sing = kernel.getSingularityValue()
        dm_order = max(self.dm.polynomialOrder, 1)
        across = 0. if self.dm.polynomialOrder == 0 else 2.   # FL1:212-216
        rules = {COMMON_EDGE: singularityCancelationQuadRule1D(COMMON_EDGE, 2.+sing, self.quad_order_diagonal, 2*dm_order),
                 COMMON_VERTEX: singularityCancelationQuadRule1D(COMMON_VERTEX, across+sing, self.quad_order_diagonal, 2*dm_order)}
        self.singular = self._psi_tables(rules)
        self.sing_fac = 1.0                                  # FL1:374
        s = max(-0.5*(sing+1), 0.)
        c = (target_order+2.)*np.log(self.num_dofs*self.H0)
        self.qo = orderFormula(c, 2.*s-1., 0., 2.*s, 0.8, False)

    def _setup2D_boundary(self, bk, target_order, quad_order_diagonal):
        smax = max(0.5*(-bk.max_singularity-1.), 0.)
        if target_order is None:
            target_order = 0.5
        if quad_order_diagonal is None:
            quad_order_diagonal = max(np.ceil((target_order+0.5+smax)/0.35*abs(np.log(self.hmin/self.H0))), 2)
        self.bquad_order_diagonal = qd = int(quad_order_diagonal)
        sing = bk.getSingularityValue()
        sg_edge = sing if sing > -2.+1e-3 else 2.+sing       # FL2:1271-1274
        rules = {COMMON_EDGE: singularityCancelationQuadRule2D_boundary(COMMON_EDGE, sg_edge, qd, qd),
                 COMMON_VERTEX: singularityCancelationQuadRule2D_boundary(COMMON_VERTEX, sing, qd, qd)}
        self.bsingular = {p: singularRule(qr, self.dm.evalShapeFunctions(qr.nodes[:3])) for p, qr in rules.items()}
        self.bsing_fac = -2.0                                # FL2:1375
        s = max(0.5*(-sing-1.), 0.)
        c = (0.5*target_order+0.25)*np.log(self.num_dofs*self.H0**2)
        self.bqo = orderFormula(c, s-1., 1., s, 0.35, True)

    def _setup1D_boundary(self, bk, target_order, quad_order_diagonal):
        smin = max(0.5*(-bk.min_singularity), 0.)
        smax = max(0.5*(-bk.max_singularity), 0.)
        if target_order is None:
            target_order = self.dm.polynomialOrder+1-smin
        if quad_order_diagonal is None:
            quad_order_diagonal = max(np.ceil(((target_order+1.)*np.log(self.num_dofs*self.H0) +
                                               (2.*smax-1.)*abs(np.log(self.hmin/self.H0)))/0.8), 2)
        self.bquad_order_diagonal = qd = int(quad_order_diagonal)
        sing = bk.getSingularityValue()
        sg = sing if sing > -1.+1e-3 else 2.+sing            # FL1:686-689
        qr = singularityCancelationQuadRule1D_boundary(COMMON_VERTEX, sg, qd, 1)
        self.bsingular = {COMMON_VERTEX: singularRule(qr, self.dm.evalShapeFunctions(qr.nodes[:2]))}
        self.bsing_fac = 1.0
        s = max(0.5*(-sing-1.), 0.)
        c = (target_order+1.)*np.log(self.num_dofs*self.H0)
        self.bqo = orderFormula(c, 2.*s-1., 0., 2.*s, 0.8, False)

    # ------------------------------------------------------------------
    def _distant_rules(self, qcap):
        dim, dpe = self.dim, self.dpe
        off = np.zeros(qcap+2, dtype=np.int32)
        bary, w, phi = [], [], []
        foff = np.zeros(qcap+2, dtype=np.int32)
        fbary, fw = [], []
        for q in range(qcap+1):
            if q >= 2:
                qr = simplexXiaoGimbutas(q, dim, dim)
                b = np.zeros((qr.num_nodes, 3))
                b[:, :dim+1] = qr.nodes.T
                bary.append(b)
                w.append(qr.weights)
                phi.append(self.dm.evalShapeFunctions(qr.nodes).T)
                off[q+1] = off[q]+qr.num_nodes
                fr = simplexDuffyTransformation(q, dim, dim-1)       # NO:999
                fb = np.zeros((fr.num_nodes, 2))
                fb[:, :dim] = fr.nodes.T
                fbary.append(fb)
                fw.append(fr.weights)
                foff[q+1] = foff[q]+fr.num_nodes
            else:
                off[q+1] = off[q]
                foff[q+1] = foff[q]
        self.dist_off = off
        self.dist_bary = np.ascontiguousarray(np.concatenate(bary))
        self.dist_w = np.ascontiguousarray(np.concatenate(w))
        self.dist_phi = np.ascontiguousarray(np.concatenate(phi))
        self.bfacet_off = foff
        self.bfacet_bary = np.ascontiguousarray(np.concatenate(fbary))
        self.bfacet_w = np.ascontiguousarray(np.concatenate(fw))

    def num_points(self, q):
        return int(self.dist_off[q+1]-self.dist_off[q])

    def _boundary_mesh(self):
        surface = self.dm.mesh.get_surface_mesh()
        self.bcells = np.ascontiguousarray(surface.cells, dtype=np.int32)

    def describe(self):
        return dict(dim=self.dim, dpe=self.dpe, num_dofs=self.num_dofs, H0=self.H0, hmin=self.hmin,
                    target_order=self.target_order, quad_order_diagonal=self.quad_order_diagonal,
                    quad_order_diagonalV=self.quad_order_diagonalV,
                    singular_points={p: r.num_nodes for p, r in self.singular.items()})
