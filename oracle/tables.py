"""The oracle's OWN host tables -- TEST INFRASTRUCTURE ONLY (imported by tests/ only).

Everything the reference's local-matrix constructors precompute before the element-pair loop runs -- kernel exponent and
scaling, the constants of the quadrature-order formula, the singularity-cancelling near rules with their merged-DoF PSI
tables (face / edge / vertex, P1 and P2, 1D and 2D), the boundary (Gauss theorem) twins, the DoF permutation table, the
class tables of a piecewise-constant variable order -- restated here with numpy / scipy straight from the reference text,
sharing NO code with pynucleus_amd/local_matrix.py, quadrature.py, kernels.py or fractionalOrders.py.  tests/test_oracle_tables.py
asserts that the product's tables equal these entry by entry, and the GPU parity tests feed nl_oracle.c from THESE tables, so
a wrong constant on the product side can no longer cancel out between the GPU path and its checker.

Reference text followed (paths under /root/reference):
  fem/PyNucleus_fem/quadrature.pyx:451-478      GaussJacobi (tensor rule, scipy js_roots = roots_sh_jacobi)
  fem/PyNucleus_fem/quadrature.pyx:481-520      simplexDuffyTransformation (facet rules of the boundary term)
  nl/PyNucleus_nl/fractionalLaplacian2D.pyx:36-399     singularityCancelationQuadRule2D
  nl/PyNucleus_nl/fractionalLaplacian2D.pyx:402-563    singularityCancelationQuadRule2D_boundary
  nl/PyNucleus_nl/fractionalLaplacian2D.pyx:587-642    setKernel (quad_order_diagonal, -V), getQuadOrder
  nl/PyNucleus_nl/fractionalLaplacian2D.pyx:644-813    getNearQuadRule: merged-DoF PSI tables
  nl/PyNucleus_nl/fractionalLaplacian2D.pyx:851, 1375  4 vol1 vol2 / -2 vol1 vol2
  nl/PyNucleus_nl/fractionalLaplacian2D.pyx:1207-1314  boundary setKernel, getQuadOrder, getNearQuadRule
  nl/PyNucleus_nl/fractionalLaplacian1D.pyx:35-141, 144-178, 203-330, 378, 626-712     the same in 1D
  nl/PyNucleus_nl/nonlocalOperator_{SCALAR}.pxi:66-109, 435     DoF permutation table, H0 = diam / sqrt(8)
  nl/PyNucleus_nl/nonlocalOperator.pyx:49-79           PermutationIndexer.rank (Lehmer code = lexicographic rank)
  nl/PyNucleus_nl/kernelsCy.pyx:168-174, 216-231, 89-100, 1594-1630, 1982-2010  kernel values, singularities, boundary kernel (phi = 1/s)
  nl/PyNucleus_nl/kernelNormalization.pyx:70-91, 225-260               scaling constants
  nl/PyNucleus_nl/fractionalOrders.pyx:203-218, 282-336, 826-896       varconst / leftRight / layers orders
  fem/PyNucleus_fem/DoFMaps.pyx:1854-2025          P1 / P2 shape functions and nodes
  fem/PyNucleus_fem/meshCy.pyx:1811-1845           boundary edges, oriented as in their cell
  fem/PyNucleus_fem/mesh.py:1658-1661              diam

The ONE input taken from outside is the table of triangle rules of the distant pairs (the reference takes them from the
un-vendored package modepy; SURVEY 8c: unpinned): passed in explicitly as (off, bary, w).
"""
from itertools import permutations, product
import numpy as np
from scipy.special import roots_sh_jacobi, gamma

FRACTIONAL, INDICATOR, PERIDYNAMIC, GAUSSIAN, EXPONENTIAL = 0, 1, 2, 3, 4      # ktype codes of nl_oracle.h
COMMON_VERTEX, COMMON_EDGE, COMMON_FACE = -1, -2, -3


# ---- quadrature primitives -----------------------------------------------------------------------------------------------
def gauss_jacobi(specs):
    """tensor Gauss-Jacobi rule on [0,1]^n; factor (order, alpha, beta) carries the weight x^alpha (1-x)^beta
    (Q:451-478: k = smallest count with 2k-1 >= order; js_roots(k, alpha+beta+1, alpha+1)); last factor fastest."""
    n1, w1 = [], []
    for order, alpha, beta in specs:
        k = (int(order)+1)//2
        if 2*k-1 != int(order):
            k += 1
        x, w = roots_sh_jacobi(k, alpha+beta+1., alpha+1.)
        n1.append(x)
        w1.append(w)
    idx = list(product(*[range(len(x)) for x in n1]))
    nodes = np.array([[n1[m][i[m]] for i in idx] for m in range(len(specs))])
    weights = np.array([np.prod([w1[m][i[m]] for m in range(len(specs))]) for i in idx])
    return nodes, weights


def duffy_simplex(order, manifold_dim):
    """Q:481-520: barycentric nodes [manifold_dim+1, n] and weights of the Duffy rule on a simplex"""
    if manifold_dim == 0:
        return np.ones((1, 1)), np.ones(1)
    e, w = gauss_jacobi([(order+manifold_dim-d-1, 0, manifold_dim-d-1) for d in range(manifold_dim)])
    nodes = np.zeros((manifold_dim+1, e.shape[1]))
    for j in range(manifold_dim-1, -1, -1):
        nodes[j+1] = e[j]
        for k in range(j):
            nodes[j+1] *= (1.-e[k])
    nodes[0] = 1.-nodes[1:].sum(axis=0)
    return nodes, w*{1: 1., 2: 2., 3: 6.}[manifold_dim]


def _tri(x1, x2):
    return np.stack([1.-x1, x1-x2, x2])


def near_rule_2d(panel, sigma, qd, qdV):
    """FL2:36-399.  Returns (nodes[6, M], weights[M]): barycentric coordinates of x (rows 0-2) and y (rows 3-5)."""
    parts = []
    if panel == COMMON_FACE:
        (e0, e1, e2, e3), w = gauss_jacobi([(1, 3+sigma, 0), (1, 2+sigma, 0), (1, 1+sigma, 0), (qd, 0, 0)])
        ww = 2.0*w*(e0*e1*e2)**(-sigma)
        parts.append((e0, e0*e1*(1-e2+e2*e3), e0*(1-e1*e2), e0*e1*(1-e2), ww))
        parts.append((e0, e0*e1, e0*(1-e1*e2*e3), e0*e1*(1-e2), ww))
        parts.append((e0, e0*e1*(1-e2), e0*(1-e1*e2*e3), e0*e1*(1-e2*e3), ww))
    elif panel == COMMON_EDGE:
        (e0, e1, e2, e3), w = gauss_jacobi([(1, 3+sigma, 0), (1, 2+sigma, 0), (qd, 0, 0), (qd, 0, 0)])
        ww = w*(e0*e1)**(-sigma)
        parts.append((e0*(1-e1*e2), e0*e1*(1-e2), e0, e0*e1*e3, ww))
        parts.append((e0, e0*e1*e3, e0*(1-e1*e2), e0*e1*(1-e2), ww))
        (e0, e1, e2, e3), w = gauss_jacobi([(1, 3+sigma, 0), (1, 2+sigma, 0), (qd, 1, 0), (qd, 0, 0)])
        ww = w*(e0*e1)**(-sigma)
        parts.append((e0*(1-e1*e2*e3), e0*e1*e2*(1-e3), e0, e0*e1, ww))
        parts.append((e0, e0*e1, e0*(1-e1*e2*e3), e0*e1*e2*(1-e3), ww))
    elif panel == COMMON_VERTEX:
        (e0, e1, e2, e3), w = gauss_jacobi([(1, 3+sigma, 0), (qdV, 0, 0), (qdV, 1, 0), (qdV, 0, 0)])
        ww = w*e0**(-sigma)
        parts.append((e0, e0*e1, e0*e2, e0*e2*e3, ww))
        parts.append((e0*e2, e0*e2*e3, e0, e0*e1, ww))
    else:
        raise ValueError(panel)
    nodes = np.concatenate([np.concatenate([_tri(x1, x2), _tri(y1, y2)]) for x1, x2, y1, y2, _ in parts], axis=1)
    return nodes, np.concatenate([p[4] for p in parts])


def near_rule_2d_boundary(panel, sigma, qd, qreg):
    """FL2:402-563: cell x facet.  nodes[5, M]: barycentric coordinates of x (rows 0-2) and of y on the facet (rows 3-4)."""
    parts = []
    if panel == COMMON_EDGE:
        (e0, e1, e2), w = gauss_jacobi([(qreg, 1.+sigma, 1.), (qd, 0., 0.), (qd, 0., 0.)])
        ww = w*e0**(-sigma)
        parts.append(([1-e0-(1-e0)*e2, e0+(1-e0)*e2-e0*e1, e0*e1], [1-e2*(1-e0), e2*(1-e0)], ww))
        parts.append(([1-e0-e2+e0*e2, e2-e0*e2, e0], [1-e2+e0*e2+e0*e1-e0, e2-e0*e2-e0*e1+e0], ww))
        parts.append(([1-e2+e0*e2-e0*e1, e2-e0*e2, e0*e1], [1-e2+e0*e2-e0, e2-e0*e2+e0], ww))
    elif panel == COMMON_VERTEX:
        (e0, e1, e2), w = gauss_jacobi([(qreg, 2.0+sigma, 0), (qd, 0, 0), (qd, 0, 0)])
        parts.append(([1-e0, e0*(1-e1), e0*e1], [1-e0*e2, e0*e2], w*e0**(-sigma)))
        (e0, e1, e2), w = gauss_jacobi([(qreg, 2.0+sigma, 0), (qd, 1, 0), (qd, 0, 0)])
        parts.append(([1-e0*e1, e0*e1*(1-e2), e0*e1*e2], [1-e0, e0], w*e0**(-sigma)))
    else:
        raise ValueError(panel)
    nodes = np.concatenate([np.stack(bx+by) for bx, by, _ in parts], axis=1)
    return nodes, np.concatenate([p[2] for p in parts])


def near_rule_1d(panel, sigma, qd, qreg):
    """FL1:35-141.  nodes[4, M]: (1-x, x, 1-y, y)."""
    if panel == COMMON_EDGE:                  # identical cells
        (e0, e1), w = gauss_jacobi([(qreg, 1+sigma, 0), (qreg, 0+sigma, 0)])
        x, y = e0*(1-e1), e0
        return np.stack([1-x, x, 1-y, y]), 2.0*w*(e0*e1)**(-sigma)
    if panel == COMMON_VERTEX:
        (e0, e1), w = gauss_jacobi([(qreg, 1+sigma, 0), (qd, 0, 0)])
        ww = w*e0**(-sigma)
        xs, ys = np.concatenate([e0*e1, e0]), np.concatenate([e0, e0*e1])
        return np.stack([1-xs, xs, 1-ys, ys]), np.concatenate([ww, ww])
    raise ValueError(panel)


def near_rule_1d_boundary(sigma, qd):
    """FL1:144-178.  nodes[3, M]: (1-eta, eta, 1)."""
    (e,), w = gauss_jacobi([(qd, sigma, 0)])
    return np.stack([1-e, e, np.ones_like(e)]), w*e**(-sigma)


# ---- finite elements -----------------------------------------------------------------------------------------------------
def element_nodes(dim, order):
    """DoFMaps.pyx: barycentric coordinates of the local DoFs (P1: vertices; P2: vertices, then edges (0,1), (1,2), (0,2))"""
    I = np.eye(dim+1)
    if order == 0:                                             # P0: the barycentre (DoFMaps.pyx:1794-1801)
        return np.full((1, dim+1), 1./(dim+1))
    if order == 1:
        return I
    if order == 3:                                             # P3 on intervals (DoFMaps.pyx:2117-2120)
        assert dim == 1
        return np.vstack([I, [[2./3., 1./3.], [1./3., 2./3.]]])
    if dim == 1:
        return np.vstack([I, [[0.5, 0.5]]])
    return np.vstack([I, [[0.5, 0.5, 0.], [0., 0.5, 0.5], [0.5, 0., 0.5]]])


def element_layout(dim, order):
    """(dofs per vertex, dofs per edge) of the element (DoFMap constructors, DoFMaps.pyx:1797, 1880-1890, 2121)"""
    return (0 if order == 0 else 1), (1 if (order == 2 and dim == 2) else 0)


def shape_functions(dim, order, lam):
    """values [dpe, n] of the local shape functions at barycentric points lam[dim+1, n] (DoFMaps.pyx:1854-1960)"""
    lam = np.asarray(lam, dtype=np.float64)
    if order == 0:
        return np.ones((1, lam.shape[1]))
    if order == 1:
        return lam[:dim+1].copy()
    if order == 3:                                             # DoFMaps.pyx:2044-2045 (vertex), :2068-2069 (edge (v1, v2))
        assert dim == 1
        l0, l1 = lam[0], lam[1]
        return np.stack([4.5*l0*(l0-1./3.)*(l0-2./3.), 4.5*l1*(l1-1./3.)*(l1-2./3.), 13.5*l0*l1*(l0-1./3.), 13.5*l1*l0*(l1-1./3.)])
    vert = [lam[k]*(2.*lam[k]-1.) for k in range(dim+1)]
    edges = [(0, 1)] if dim == 1 else [(0, 1), (1, 2), (0, 2)]
    return np.stack(vert+[4.*lam[a]*lam[b] for a, b in edges])


def dof_permutations(dim, order):
    """NO:66-109: table[rank(perm), dofPerm] = dofOrig with nodes[dofPerm, j] == nodes[dofOrig, perm[j]]; itertools yields the
    permutations in lexicographic order, which is the order of their Lehmer rank (nonlocalOperator.pyx:64-79)"""
    nodes = element_nodes(dim, order)
    rows = []
    for perm in permutations(range(dim+1)):
        row = []
        for dofPerm in range(nodes.shape[0]):
            hit = [dofOrig for dofOrig in range(nodes.shape[0]) if np.abs(nodes[dofPerm]-nodes[dofOrig][list(perm)]).max() <= 1e-10]
            row.append(hit[0])
        rows.append(row)
    return np.array(rows, dtype=np.int32)


def merged_psi(dim, order, panel, nodes):
    """FL2:662-811, FL1:255-330: rows = DoFs of the union of the two cells, shared DoFs first.  Returns (psi, phi_x part,
    phi_y part) with psi = px - py."""
    nV = dim+1
    dpv, dped = element_layout(dim, order)
    px = shape_functions(dim, order, nodes[:nV])
    py = shape_functions(dim, order, nodes[nV:2*nV])
    dpe = px.shape[0]
    common = -panel
    M = nodes.shape[1]
    if common == nV:
        p0, p1 = px, py
    elif common == 1:
        p0, p1 = np.zeros((2*dpe-dpv, M)), np.zeros((2*dpe-dpv, M))
        p0[:dpv], p1[:dpv] = px[:dpv], py[:dpv]
        p0[dpv:dpe] = px[dpv:]
        p1[dpe:] = py[dpv:]
    elif common == 2 and dim == 2:
        rows = 2*dpe-2*dpv-dped
        p0, p1 = np.zeros((rows, M)), np.zeros((rows, M))
        for dof in list(range(2*dpv))+list(range(nV*dpv, nV*dpv+dped)):      # the two shared vertices, the shared edge
            p0[dof], p1[dof] = px[dof], py[dof]
        for dof in range(2*dpv, nV*dpv):                                      # the third vertex of either cell
            p0[dof] = px[dof]
            p1[dpe+dof-2*dpv] = py[dof]
        for dof in range(nV*dpv+dped, dpe):                                   # the other edges
            p0[dof] = px[dof]
            p1[dpe+dof-2*dpv-dped] = py[dof]
    else:
        raise ValueError(panel)
    return p0-p1, p0, p1


# ---- mesh-level inputs ---------------------------------------------------------------------------------------------------
def mesh_boundary_facets(mesh):
    """The facet list of the mesh is an INPUT like its cells (facet numbers index the boundary items and labels the callers
    hand over; refinement carries the list along, meshCy.pyx:535-560).  It is checked against the definition: exactly the edges
    that belong to one cell, oriented as in that cell (boundary_facets below)."""
    given = np.ascontiguousarray(mesh.get_surface_mesh().cells, dtype=np.int32)
    own = boundary_facets(mesh.cells)
    assert sorted(map(tuple, given.tolist())) == sorted(map(tuple, own.tolist())), 'boundary facets of the mesh do not match its cells'
    return given


def boundary_facets(cells):
    """meshCy.pyx:1811-1845 (2D): edges that belong to one cell, oriented as in that cell; 1D: vertices that occur once"""
    cells = np.asarray(cells)
    if cells.shape[1] == 2:
        ids, cnt = np.unique(cells.ravel(), return_counts=True)
        return ids[cnt == 1].reshape(-1, 1).astype(np.int32)
    seen = {}
    for c in cells:
        for a, b in ((c[0], c[1]), (c[1], c[2]), (c[2], c[0])):
            key = (min(a, b), max(a, b))
            if key in seen:
                del seen[key]
            else:
                seen[key] = (a, b)
    return np.array(list(seen.values()), dtype=np.int32).reshape(-1, 2)


# ---- kernels ---------------------------------------------------------------------------------------------------------------
def fractional_scaling(dim, s, horizon, normalized):
    """kernelNormalization.pyx:70-91 (constantFractionalLaplacianScaling) / :122-131: C(d, s) * 1/2; 1/2 when not normalised"""
    if not normalized:
        return 0.5
    if np.isfinite(horizon):
        return (2.-2.*s)*horizon**(2.*s-2.)*dim*gamma(0.5*dim)/np.pi**(0.5*dim)*0.5
    return 2.**(2.*s)*s*gamma(s+0.5*dim)/np.pi**(0.5*dim)/gamma(1.-s)*0.5


def integrable_scaling(ktype, dim, horizon, normalized, variance=1., rate=1.):
    """kernelNormalization.pyx:225-290 (constantIntegrableScaling), interaction ball2 / full space"""
    from math import erf, exp, sqrt, pi
    if not normalized:
        return 0.5
    if ktype == GAUSSIAN:
        if dim == 1:
            return 4.0/sqrt(pi)/(erf(3.0)-6.0*exp(-9.0)/sqrt(pi))/(horizon/3.0)**3/2. if np.isfinite(horizon) else 1.0/sqrt(2.0*pi*variance)/2.
        return 4.0/pi/(1.0-10.0*exp(-9.0))/(horizon/3.0)**4/2. if np.isfinite(horizon) else 1.0/(2.0*pi*variance)/2.
    if ktype == EXPONENTIAL:
        assert dim == 1
        if np.isfinite(horizon):
            return rate**3/(2.0-exp(-rate*horizon)*(2.0+2.0*rate*horizon+(rate*horizon)**2))/2.
        return rate**3/2.0/2.
    if ktype == INDICATOR:
        return 3./horizon**3/2. if dim == 1 else 8./np.pi/horizon**4/2.
    if ktype == PERIDYNAMIC:
        return 2./horizon**2/2. if dim == 1 else 6./np.pi/horizon**3/2.
    raise NotImplementedError(ktype)


class KernelBlock:
    """the POD block of nl_oracle.h's nlo_kernel: gamma(x, y) = scale |x-y|^(2 exponent) inside the horizon"""

    def __init__(self, ktype, exponent, scale, horizon, interaction=0):
        self.ktype, self.exponent, self.scale, self.horizon, self.interaction = int(ktype), float(exponent), float(scale), float(horizon), int(interaction)

    def device_params(self):
        return dict(ktype=self.ktype, exponent=self.exponent, scale=self.scale, interaction=self.interaction,
                    horizon2=self.horizon**2 if np.isfinite(self.horizon) else np.inf)


class Formula:
    """order = max(ceil((c0 + a L_other + b L_max - e logdh_other) / (max(logdh_self, 0) + den0)), 2); clip_num: the logdh of the
    numerator is clipped at 0 too (the boundary formulas FL2:1226-1243, FL1:644-660 take max(log(d/h), 0) up front)"""

    def __init__(self, c0, a, b, e, den0, clip_num):
        self.c0, self.a, self.b, self.e, self.den0, self.clip_num = float(c0), float(a), float(b), float(e), float(den0), bool(clip_num)

    def astuple(self):
        return (self.c0, self.a, self.b, self.e, self.den0, self.clip_num)


class Rule:
    def __init__(self, nodes, weights, psi, phi0=None, phi1=None):
        self.nodes, self.weights, self.psi = np.ascontiguousarray(nodes), np.ascontiguousarray(weights), np.ascontiguousarray(psi)
        self.num_nodes, self.rows = int(self.weights.shape[0]), int(self.psi.shape[0])
        self.phi0, self.phi1 = phi0, phi1


# ---- piecewise-constant variable orders ---------------------------------------------------------------------------------
def order_table(spec):
    """(sVals[L, L], label function of points[n, dim]) of a piecewise-constant order.
    spec = ('varconst', s) | ('leftRight', sll, srr, slr, srl, interface) | ('layers', boundaries, orders)"""
    kind = spec[0]
    if kind == 'varconst':
        return np.array([[float(spec[1])]]), (lambda pts: np.zeros(len(pts), dtype=np.int32))
    if kind == 'leftRight':
        sll, srr, slr, srl, interface = [float(v) for v in spec[1:6]]
        if not np.isfinite(slr):
            slr = 0.5*(sll+srr)
        if not np.isfinite(srl):
            srl = 0.5*(sll+srr)
        # fractionalOrders.pyx:282-299: "left" is x[0] < interface
        return np.array([[sll, slr], [srl, srr]]), (lambda pts: (np.asarray(pts)[:, 0] >= interface).astype(np.int32))
    if kind == 'layers':
        bnd, orders = np.asarray(spec[1], dtype=np.float64), np.asarray(spec[2], dtype=np.float64)
        nl = bnd.shape[0]-1

        def labels(pts):
            # fractionalOrders.pyx:826-855: last coordinate; first layer whose closed interval holds it
            c = np.asarray(pts)[:, -1]
            out = np.zeros(c.shape[0], dtype=np.int32)
            for k, v in enumerate(c):
                if v <= bnd[0]:
                    out[k] = 0
                elif v >= bnd[nl]:
                    out[k] = nl-1
                else:
                    out[k] = next(i for i in range(nl) if bnd[i] <= v <= bnd[i+1])
            return out
        return orders, labels
    if kind == 'innerOuter':
        # fractionalOrders.pyx:664-691: r2x < r^2 is "inner"
        dim, sii, soo, r, center, sio, soi = spec[1:8]
        if not np.isfinite(sio):
            sio = 0.5*(sii+soo)
        if not np.isfinite(soi):
            soi = 0.5*(sii+soo)
        center = np.asarray(center, dtype=np.float64)[:dim]
        return (np.array([[sii, sio], [soi, soo]], dtype=np.float64),
                lambda pts: (((np.asarray(pts)[:, :dim]-center)**2).sum(axis=1) >= r*r).astype(np.int32))
    if kind == 'islands':
        # fractionalOrders.pyx:754-789: in an island iff r <= |x_i| <= r2 for every coordinate
        sii, soo, r, r2, sio, soi = spec[1:7]
        if not np.isfinite(sio):
            sio = 0.5*(sii+soo)
        if not np.isfinite(soi):
            soi = 0.5*(sii+soo)

        def labels(pts):
            out = np.zeros(len(pts), dtype=np.int32)
            for k, p in enumerate(np.asarray(pts)):
                out[k] = 0 if all(r <= abs(v) <= r2 for v in p[:2]) else 1
            return out
        return np.array([[sii, sio], [soi, soo]], dtype=np.float64), labels
    if kind == 'product':
        # fractionalOrders.pyx:733-752 (sumFractionalOrder.eval multiplies the two values)
        sv1, lab1 = order_table(spec[1])
        sv2, lab2 = order_table(spec[2])
        L2 = sv2.shape[0]
        sv = np.kron(sv1, sv2)
        return sv, (lambda pts: (lab1(pts)*L2+lab2(pts)).astype(np.int32))
    raise NotImplementedError(kind)


# ---- the tables ---------------------------------------------------------------------------------------------------------------
class OracleTables:
    """Tables of one (DoFMap, kernel spec, params) triple in the attribute layout oracle.OracleProblem reads.

    dm: mesh + DoF numbering (inputs: mesh.vertices, mesh.cells, dm.dofs, volumes / diameters are read by OracleProblem itself);
    spec: dict(kernelType=FRACTIONAL|INDICATOR|PERIDYNAMIC, s=float or an order_table spec, horizon=float, normalized=bool);
    distant = (off[qcap+2], bary[n, 3], w[n]): the triangle / interval rules of distant pairs (the shared, unpinned input)."""

    def __init__(self, dm, spec, params=None, zeroExterior=True, distant=None, _sing_range=None, _interior_defaults=False):
        params = dict(params or {})
        mesh = dm.mesh
        self.dm = dm
        self.dim = dim = int(np.asarray(mesh.vertices).shape[1])
        self.dpe = int(np.asarray(dm.dofs).shape[1])
        # the element from the number of local DoFs: P0 1, P1 dim+1, P2 3 / 6, P3 on intervals 4
        self.order = order = {1: 0, dim+1: 1, (3 if dim == 1 else 6): 2, (4 if dim == 1 else 10): 3}[self.dpe]
        self.num_dofs = int(dm.num_dofs)
        self.pointwise = False
        self.spec = spec
        ktype, horizon, normalized = spec['kernelType'], float(spec.get('horizon', np.inf)), bool(spec.get('normalized', True))
        finite = np.isfinite(horizon)
        self.zeroExterior = bool(zeroExterior) and not finite                      # NA:919-922
        self.interaction_transform = None
        if spec.get('ellipse') is not None:
            # ellipse_retriangulation / _barycenter (interactionDomains.pyx:1579-1630): T = [[cos t / a, -sin t / a], [sin t / b, cos t / b]]
            a, b, th = spec['ellipse']
            self.interaction_transform = np.array([[np.cos(th)/a, -np.sin(th)/a], [np.sin(th)/b, np.cos(th)/b]])
        verts = np.asarray(mesh.vertices, dtype=np.float64)
        cells = np.asarray(mesh.cells)
        self.H0 = float(np.linalg.norm(verts.max(axis=0)-verts.min(axis=0)))/np.sqrt(8.)   # NO:435, mesh.py:1658-1661
        edges = [(a, b) for a in range(dim+1) for b in range(a+1, dim+1)]
        # meshCy.pyx:1661-1728: hmin is the shortest EDGE of the mesh (the per-cell h is the longest edge of the cell)
        self.hmin = float(np.min([np.linalg.norm(verts[cells[:, a]]-verts[cells[:, b]], axis=1) for a, b in edges]))
        self.dof_perm_table = dof_permutations(dim, order)
        self.qcap = int(distant[0].shape[0]-2)
        self.dist_off = np.ascontiguousarray(distant[0], dtype=np.int32)
        self.dist_bary = np.ascontiguousarray(distant[1], dtype=np.float64)
        self.dist_w = np.ascontiguousarray(distant[2], dtype=np.float64)
        self.dist_phi = np.ascontiguousarray(shape_functions(dim, order, self.dist_bary[:, :dim+1].T).T)
        # facet rules of the boundary term (NO:999: simplexDuffyTransformation(order, dim, dim-1)), orders 2 .. qcap
        foff, fb, fw = np.zeros(self.qcap+2, dtype=np.int32), [], []
        for q in range(self.qcap+1):
            foff[q+1] = foff[q]
            if q >= 2:
                nd, w = duffy_simplex(q, dim-1)
                b = np.zeros((nd.shape[1], 2))
                b[:, :dim] = nd.T
                fb.append(b)
                fw.append(w)
                foff[q+1] += nd.shape[1]
        self.bfacet_off, self.bfacet_bary, self.bfacet_w = foff, np.concatenate(fb), np.concatenate(fw)

        s = spec.get('s', None)
        self.classes, self.nonsym = None, False
        if isinstance(s, tuple):
            # NO:509-513: the order of a pair is s(centre1, centre2); one class per distinct value with the tables of that
            # constant order, the near-field quadrature orders from the extreme singularities of the variable kernel
            sVals, labels = order_table(s)
            vals = np.unique(sVals)
            self.class_s = vals
            self.cls_of = np.searchsorted(vals, sVals).astype(np.int32)
            self.num_labels = int(sVals.shape[0])
            self.nonsym = bool(np.abs(sVals-sVals.T).max() >= 1e-10)
            self.cell_labels = labels(verts[cells].mean(axis=1))
            sr = (float(sVals.min()), float(sVals.max()))
            self.classes = [OracleTables(dm, dict(spec, s=float(v)), params, zeroExterior, distant, _sing_range=sr, _interior_defaults=self.nonsym)
                            for v in vals]
            c0 = self.classes[0]
            self.has_boundary_tables = c0.has_boundary_tables
            self.singular = c0.singular
            self.target_order = c0.target_order
            self.bcells = mesh_boundary_facets(mesh) if self.has_boundary_tables else None
            self.facet_labels = labels(verts[self.bcells].mean(axis=1)) if self.has_boundary_tables else np.zeros(0, dtype=np.int32)
            return

        # ---- one constant-order (or integrable) kernel --------------------------------------------------------------------
        if ktype == FRACTIONAL:
            s = float(s)
            sing = -dim-2.*s                                   # kernelsCy.pyx:1618-1621
            self.kernel = KernelBlock(FRACTIONAL, 0.5*sing, fractional_scaling(dim, s, horizon, normalized), horizon, spec.get('interaction', 0))
        elif ktype == INDICATOR:
            sing = 0.
            self.kernel = KernelBlock(INDICATOR, 0., integrable_scaling(ktype, dim, horizon, normalized), horizon, spec.get('interaction', 0))
        elif ktype == PERIDYNAMIC:
            sing = -1.
            self.kernel = KernelBlock(PERIDYNAMIC, -0.5, integrable_scaling(ktype, dim, horizon, normalized), horizon, spec.get('interaction', 0))
        elif ktype == GAUSSIAN:
            # kernelsCy.pyx:388-416, 687-692: C exp(-d2 invD), invD = 1 / (horizon/3)^2 (finite horizon) or 1 / (2 variance^dim)
            sing = 0.
            var = float(spec.get('variance', 1.))
            invD = 1.0/(horizon/3.)**2 if finite else 0.5/var**dim
            self.kernel = KernelBlock(GAUSSIAN, -invD, integrable_scaling(ktype, dim, horizon, normalized, var), horizon, spec.get('interaction', 0))
        elif ktype == EXPONENTIAL:
            # kernelsCy.pyx:448-462: C exp(-a |x-y|)
            sing = 0.
            rate = float(spec.get('exponentialRate', 1.))
            self.kernel = KernelBlock(EXPONENTIAL, -rate, integrable_scaling(ktype, dim, horizon, normalized, 1., rate), horizon, spec.get('interaction', 0))
        else:
            raise NotImplementedError(ktype)
        self.singularityValue = sing
        smin_sing, smax_sing = (sing, sing) if _sing_range is None else (-dim-2.*_sing_range[0], -dim-2.*_sing_range[1])
        target = None if _interior_defaults else params.get('target_order', None)
        qd_in = None if _interior_defaults else params.get('quad_order_diagonal', None)
        logh = abs(np.log(self.hmin/self.H0))
        N = self.num_dofs
        if dim == 2:
            # FL2:587-620
            if target is None:
                target = 0.5
            smax = max(-0.5*(smax_sing+2.), 0.)
            if qd_in is None:
                qd = max(np.ceil((target+1.+smax)/0.43*logh), 4)
                qdV = max(np.ceil((target+1.+smax)/0.7*logh), 4)
            else:
                qd = qdV = qd_in
            self.quad_order_diagonal, self.quad_order_diagonalV = int(qd), int(qdV)
            # FL2:590-598: the integrand cancels two orders of the singularity within an element, and across elements for continuous
            # elements only (P0: none)
            across = 0. if order == 0 else 2.
            rules = {p: near_rule_2d(p, (2. if p == COMMON_FACE else across)+sing, int(qd), int(qdV)) for p in (COMMON_FACE, COMMON_EDGE, COMMON_VERTEX)}
            self.sing_fac = 4.0
            sq = max(-0.5*(sing+2.), 0.)                       # FL2:622-642
            self.qo = Formula((0.5*target+0.5)*np.log(N*self.H0**2), sq-1., 1., sq, 0.4, False)
        else:
            # FL1:203-253
            smin = max(-0.5*(smin_sing+1.), 0.)
            smax = max(-0.5*(smax_sing+1.), 0.)
            if target is None:
                target = order+1-smin
            if qd_in is None:
                qd_in = max(np.ceil(((target+2.)*np.log(N*self.H0)+(2.*smax-1.)*logh)/0.8), 2)
            self.quad_order_diagonal = self.quad_order_diagonalV = int(qd_in)
            across = 0. if order == 0 else 2.                  # FL1:208-216
            rules = {p: near_rule_1d(p, (2. if p == COMMON_EDGE else across)+sing, int(qd_in), 2*max(order, 1)) for p in (COMMON_EDGE, COMMON_VERTEX)}
            self.sing_fac = 1.0
            sq = max(-0.5*(sing+1.), 0.)
            self.qo = Formula((target+2.)*np.log(N*self.H0), 2.*sq-1., 0., 2.*sq, 0.8, False)
        self.target_order = target
        self.singular = {}
        for p, (nodes, w) in rules.items():
            psi, p0, p1 = merged_psi(dim, order, p, nodes)
            self.singular[p] = Rule(nodes, w, psi, p0, p1)

        # ---- the Gauss-theorem twin for Omega x Omega^c (NA:953-955; kernelsCy.pyx:1982-2010: same s, phi = 1/s, one power less)
        # finite horizon, fractional kernel of constant order, l2 ball: the cluster method integrates the exterior of a cluster pair with the
        # twin of the SAME kernel on the full space -- getModifiedKernel(horizon = inf) keeps the scaling (NA:953-955, kernelsCy.pyx:1085-1107)
        fh_near = finite and ktype == FRACTIONAL and spec.get('interaction', 0) == 1 and spec.get('ellipse') is None
        self.has_boundary_tables = (ktype in (FRACTIONAL, GAUSSIAN, EXPONENTIAL) and not finite) or fh_near
        if self.zeroExterior and not self.has_boundary_tables:
            raise NotImplementedError('zeroExterior needs a fractional, Gaussian or exponential kernel on the full space')
        if not self.has_boundary_tables:
            return
        if ktype == FRACTIONAL:
            bsing = 1.-dim-2.*s
            # (a class of a piecewise-constant order -- _sing_range is set -- keeps the TRUNCATED twin: NA:1966-2156 integrate with
            # kernel.getBoundaryKernel(), facets beyond the horizon drop out)
            self.boundaryKernel = KernelBlock(FRACTIONAL, 0.5*bsing, self.kernel.scale/s, np.inf if (fh_near and _sing_range is None) else horizon,
                                              spec.get('interaction', 0) if (fh_near and _sing_range is not None) else 0)
        else:
            # kernelsCy.pyx:1194-1218: the same type, scaling and exponentInverse with boundary = True; kernelFun :418-477 (block ids 5 / 7
            # Gaussian in 1D / 2D, 6 exponential: nl_oracle.c kernel_eval); singularity 0 (:657-664)
            bsing = 0.
            self.boundaryKernel = KernelBlock((5 if dim == 1 else 7) if ktype == GAUSSIAN else 6, self.kernel.exponent, self.kernel.scale, horizon)
        bmin_sing, bmax_sing = (bsing, bsing) if (_sing_range is None or ktype != FRACTIONAL) else (1.-dim-2.*_sing_range[0], 1.-dim-2.*_sing_range[1])
        bt = params.get('target_order', None)
        bqd = params.get('quad_order_diagonal', None)
        if dim == 2:
            # FL2:1207-1314
            smax = max(0.5*(-bmax_sing-1.), 0.)
            if bt is None:
                bt = 0.5
            if bqd is None:
                bqd = max(np.ceil((bt+0.5+smax)/0.35*logh), 2)
            bqd = int(bqd)
            sg_edge = bsing if bsing > -2.+1e-3 else 2.+bsing
            brules = {COMMON_EDGE: near_rule_2d_boundary(COMMON_EDGE, sg_edge, bqd, bqd),
                      COMMON_VERTEX: near_rule_2d_boundary(COMMON_VERTEX, bsing, bqd, bqd)}
            self.bsing_fac = -2.0
            sq = max(0.5*(-bsing-1.), 0.)
            self.bqo = Formula((0.5*bt+0.25)*np.log(N*self.H0**2), sq-1., 1., sq, 0.35, True)
        else:
            # FL1:626-712
            smin = max(0.5*(-bmin_sing), 0.)
            smax = max(0.5*(-bmax_sing), 0.)
            if bt is None:
                bt = order+1-smin
            if bqd is None:
                bqd = max(np.ceil(((bt+1.)*np.log(N*self.H0)+(2.*smax-1.)*logh)/0.8), 2)
            bqd = int(bqd)
            sg = bsing if bsing > -1.+1e-3 else 2.+bsing
            brules = {COMMON_VERTEX: near_rule_1d_boundary(sg, bqd)}
            self.bsing_fac = 1.0
            sq = max(0.5*(-bsing-1.), 0.)
            self.bqo = Formula((bt+1.)*np.log(N*self.H0), 2.*sq-1., 0., 2.*sq, 0.8, False)
        self.bquad_order_diagonal = bqd
        self.bsingular = {p: Rule(nodes, w, shape_functions(dim, order, nodes[:dim+1])) for p, (nodes, w) in brules.items()}
        self.bcells = mesh_boundary_facets(mesh)

    def num_points(self, q):
        return int(self.dist_off[q+1]-self.dist_off[q])
