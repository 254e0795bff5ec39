"""Quadrature tables for the nonlocal element-pair integrals (host side, numpy).

Everything here is *table construction*: the tables are flattened to SoA arrays
and uploaded once to HBM; the HIP kernels only ever read them.

Reference behaviour followed (file:line under /root/reference):
  fem/PyNucleus_fem/quadrature.pyx:451-478   GaussJacobi (tensor Gauss-Jacobi on [0,1]^d,
                                             weight x^alpha (1-x)^beta, scipy js_roots)
  fem/PyNucleus_fem/quadrature.pyx:481-518   simplexDuffyTransformation
  fem/PyNucleus_fem/quadrature.pyx:521-545   simplexXiaoGimbutas (modepy tables, weights sum to 1)
  fem/PyNucleus_fem/quadrature.pyx:209-229   doubleSimplexQuadratureRule
  nl/PyNucleus_nl/fractionalLaplacian2D.pyx:36-399   singularityCancelationQuadRule2D
  nl/PyNucleus_nl/fractionalLaplacian2D.pyx:402-563  singularityCancelationQuadRule2D_boundary
  nl/PyNucleus_nl/fractionalLaplacian1D.pyx:35-141   singularityCancelationQuadRule1D
  nl/PyNucleus_nl/fractionalLaplacian1D.pyx:144-179  singularityCancelationQuadRule1D_boundary

Triangle rules: the reference takes them from the un-vendored third-party package
``modepy`` (XiaoGimbutasSimplexQuadrature, version unpinned).  If modepy is
importable we use exactly that call; otherwise the fully symmetric, positive,
interior rules shipped in ``data/triangle_rules.json`` (generated and verified by
``tools/gen_triangle_rules.py``) are used, and beyond their range a conical
Gauss-Jacobi product rule of the requested degree.
"""
import json
import os
import numpy as np
from scipy.special import roots_sh_jacobi

REAL = np.float64

COMMON_VERTEX = -1
COMMON_EDGE = -2
COMMON_FACE = -3


class quadratureRule:
    def __init__(self, nodes, weights, dim, manifold_dim=None):
        nodes = np.ascontiguousarray(nodes, dtype=REAL)
        weights = np.ascontiguousarray(weights, dtype=REAL)
        assert nodes.shape[1] == weights.shape[0]
        self.nodes = nodes
        self.weights = weights
        self.dim = dim
        self.manifold_dim = dim if manifold_dim is None else manifold_dim
        self.num_nodes = nodes.shape[1]


class simplexQuadratureRule(quadratureRule):
    pass


class GaussJacobi(quadratureRule):
    """Tensor Gauss-Jacobi rule on [0,1]^d; entry (order, alpha, beta) integrates
    p(x) x^alpha (1-x)^beta exactly for deg p <= order (rounded up to odd)."""

    def __init__(self, order_weight_exponents):
        nodes1D, weights1D, self.orders = [], [], []
        for order, alpha, beta in order_weight_exponents:
            k = (int(order)+1)//2
            if 2*k-1 != order:
                k += 1
            self.orders.append(2*k-1)
            # quadrature.pyx:464-466: js_roots(k, beta+alpha+1, alpha+1)
            n1D, w1D = roots_sh_jacobi(k, alpha+beta+1., alpha+1.)
            nodes1D.append(np.asarray(n1D, dtype=REAL))
            weights1D.append(np.asarray(w1D, dtype=REAL))
        dim = len(nodes1D)
        grids = np.meshgrid(*nodes1D, indexing='ij')
        nodes = np.stack([g.reshape(-1) for g in grids], axis=0)
        wg = np.meshgrid(*weights1D, indexing='ij')
        weights = np.ones(nodes.shape[1], dtype=REAL)
        for m in range(dim):
            weights = weights*wg[m].reshape(-1)
        super().__init__(nodes, weights, dim)


class simplexDuffyTransformation(simplexQuadratureRule):
    def __init__(self, order, dim, manifold_dim=None):
        if manifold_dim is None:
            manifold_dim = dim
        if manifold_dim == 0:
            super().__init__(np.ones((1, 1)), np.ones((1)), dim, manifold_dim)
            self.orders = [100]
            return
        exps = [(order+manifold_dim-d-1, 0, manifold_dim-d-1) for d in range(manifold_dim)]
        qr = GaussJacobi(exps)
        n = qr.num_nodes
        nodes = np.empty((manifold_dim+1, n), dtype=REAL)
        for j in range(manifold_dim-1, -1, -1):
            nodes[j+1] = qr.nodes[j]
            for k in range(j):
                nodes[j+1] *= (1.-qr.nodes[k])
        nodes[0] = 1.
        for j in range(manifold_dim):
            nodes[0] -= nodes[j+1]
        w = qr.weights.copy()
        if manifold_dim == 2:
            w *= 2.
        elif manifold_dim == 3:
            w *= 6.
        super().__init__(nodes, w, dim, manifold_dim)
        self.orders = qr.orders


_TRIANGLE_TABLES = None
_TRIANGLE_SOURCE = None


def _load_triangle_tables():
    global _TRIANGLE_TABLES
    if _TRIANGLE_TABLES is None:
        fn = os.path.join(os.path.dirname(__file__), 'data', 'triangle_rules.json')
        tables = {}
        if os.path.exists(fn):
            with open(fn) as f:
                raw = json.load(f)
            for k, v in raw['rules'].items():
                tables[int(k)] = (np.array(v['nodes'], dtype=REAL), np.array(v['weights'], dtype=REAL))
        _TRIANGLE_TABLES = tables
    return _TRIANGLE_TABLES


def triangleRuleSource():
    """Source of the triangle rules for distant pairs: 'builtin' (the repo's tables, data/triangle_rules.json) unless the
    environment asks for the reference's own source with PNL_TRIANGLE_RULES=modepy and modepy is importable.  Deterministic by
    default: the assembled operators, the golden fixtures and the point counts the tile kernels key on do not depend on what
    happens to be installed."""
    global _TRIANGLE_SOURCE
    if _TRIANGLE_SOURCE is None:
        import os
        _TRIANGLE_SOURCE = 'builtin'
        if os.environ.get('PNL_TRIANGLE_RULES', 'builtin') == 'modepy':
            try:
                import modepy  # noqa: F401
                _TRIANGLE_SOURCE = 'modepy'
            except Exception:
                raise RuntimeError('PNL_TRIANGLE_RULES=modepy but modepy cannot be imported')
    return _TRIANGLE_SOURCE


def triangleRule(order):
    """degree-`order` rule on the reference triangle: bary nodes [3, n], weights [n] summing to 1."""
    if triangleRuleSource() == 'modepy':
        from modepy import XiaoGimbutasSimplexQuadrature
        from modepy.tools import unit_to_barycentric
        qr = XiaoGimbutasSimplexQuadrature(order, 2)
        return np.ascontiguousarray(unit_to_barycentric(qr.nodes), dtype=REAL), 0.5*np.asarray(qr.weights, dtype=REAL)
    tables = _load_triangle_tables()
    if order in tables:
        nodes, weights = tables[order]
        return nodes.copy(), weights.copy()
    qr = simplexDuffyTransformation(order, 2, 2)
    return qr.nodes, qr.weights


class simplexXiaoGimbutas(simplexQuadratureRule):
    def __init__(self, order, dim, manifold_dim=None):
        if manifold_dim is None:
            manifold_dim = dim
        if manifold_dim in (0, 1):
            qr = simplexDuffyTransformation(order, dim, manifold_dim)
            super().__init__(qr.nodes, qr.weights, dim, manifold_dim)
        elif manifold_dim == 2:
            nodes, weights = triangleRule(int(order))
            super().__init__(nodes, weights, dim, manifold_dim)
        else:
            raise NotImplementedError('dim={}'.format(manifold_dim))
        self.order = order


class doubleSimplexQuadratureRule(quadratureRule):
    def __init__(self, rule1, rule2):
        self.rule1 = rule1
        self.rule2 = rule2
        w = (rule1.weights[:, None]*rule2.weights[None, :]).reshape(-1)
        super().__init__(np.zeros((0, w.shape[0])), w, rule1.dim+rule2.dim, rule1.manifold_dim+rule2.manifold_dim)


def _bary2(x1, x2):
    return np.stack([1-x1, x1-x2, x2])


class singularityCancelationQuadRule2D(quadratureRule):
    """Rules on [0,1]^4 for triangle pairs sharing a face / an edge / a vertex.
    nodes[0:3] = barycentric coords of x, nodes[3:6] of y."""

    def __init__(self, panel, singularity, quad_order_diagonal, quad_order_diagonalV, quad_order_regular=1):
        dim = 2
        sg = singularity
        if panel == COMMON_FACE:
            qr = GaussJacobi(((1, 3+sg, 0), (1, 2+sg, 0), (1, 1+sg, 0), (quad_order_diagonal, 0, 0)))
            e0, e1, e2, e3 = qr.nodes
            w = 2.0*qr.weights*(e0*e1*e2)**(-sg)
            bx = [_bary2(e0, e0*e1*(1-e2+e2*e3)), _bary2(e0, e0*e1), _bary2(e0, e0*e1*(1-e2))]
            by = [_bary2(e0*(1-e1*e2), e0*e1*(1-e2)),
                  _bary2(e0*(1-e1*e2*e3), e0*e1*(1-e2)),
                  _bary2(e0*(1-e1*e2*e3), e0*e1*(1-e2*e3))]
            bary = np.concatenate([np.concatenate(bx, axis=1), np.concatenate(by, axis=1)], axis=0)
            super().__init__(bary, np.concatenate([w, w, w]), dim+1)
        elif panel == COMMON_EDGE:
            q0 = GaussJacobi(((1, 3+sg, 0), (1, 2+sg, 0), (quad_order_diagonal, 0, 0), (quad_order_diagonal, 0, 0)))
            q1 = GaussJacobi(((1, 3+sg, 0), (1, 2+sg, 0), (quad_order_diagonal, 1, 0), (quad_order_diagonal, 0, 0)))
            e0, e1, e2, e3 = q0.nodes
            w0 = q0.weights*(e0*e1)**(-sg)
            bx = [_bary2(e0*(1-e1*e2), e0*e1*(1-e2)), _bary2(e0, e0*e1*e3)]
            by = [_bary2(e0, e0*e1*e3), _bary2(e0*(1-e1*e2), e0*e1*(1-e2))]
            e0, e1, e2, e3 = q1.nodes
            w1 = q1.weights*(e0*e1)**(-sg)
            bx += [_bary2(e0*(1-e1*e2*e3), e0*e1*e2*(1-e3)), _bary2(e0, e0*e1)]
            by += [_bary2(e0, e0*e1), _bary2(e0*(1-e1*e2*e3), e0*e1*e2*(1-e3))]
            bary = np.concatenate([np.concatenate(bx, axis=1), np.concatenate(by, axis=1)], axis=0)
            super().__init__(bary, np.concatenate([w0, w0, w1, w1]), 2*dim)
        elif panel == COMMON_VERTEX:
            qv = GaussJacobi(((1, 3+sg, 0), (quad_order_diagonalV, 0, 0), (quad_order_diagonalV, 1, 0), (quad_order_diagonalV, 0, 0)))
            e0, e1, e2, e3 = qv.nodes
            w = qv.weights*e0**(-sg)
            bx = [_bary2(e0, e0*e1), _bary2(e0*e2, e0*e2*e3)]
            by = [_bary2(e0*e2, e0*e2*e3), _bary2(e0, e0*e1)]
            bary = np.concatenate([np.concatenate(bx, axis=1), np.concatenate(by, axis=1)], axis=0)
            super().__init__(bary, np.concatenate([w, w]), 2*dim+1)
        else:
            raise NotImplementedError('Unknown panel type: {}'.format(panel))


class singularityCancelationQuadRule2D_boundary(quadratureRule):
    """Rules on [0,1]^3 for a triangle and a boundary edge sharing an edge / a vertex.
    nodes[0:3] = barycentric coords of x (triangle), nodes[3:5] of y (edge)."""

    def __init__(self, panel, singularity, quad_order_diagonal, quad_order_regular):
        dim = 2
        sg = singularity
        if panel == COMMON_EDGE:
            q = GaussJacobi(((quad_order_regular, 1.+sg, 1.), (quad_order_diagonal, 0., 0.), (quad_order_diagonal, 0., 0.)))
            e0, e1, e2 = q.nodes
            w = q.weights*e0**(-sg)
            bx = [np.stack([1-e0-(1-e0)*e2, e0+(1-e0)*e2-e0*e1, e0*e1]),
                  np.stack([1-e0-e2+e0*e2, e2-e0*e2, e0]),
                  np.stack([1-e2+e0*e2-e0*e1, e2-e0*e2, e0*e1])]
            by = [np.stack([1-e2*(1-e0), e2*(1-e0)]),
                  np.stack([1-e2+e0*e2+e0*e1-e0, e2-e0*e2-e0*e1+e0]),
                  np.stack([1-e2+e0*e2-e0, e2-e0*e2+e0])]
            bary = np.concatenate([np.concatenate(bx, axis=1), np.concatenate(by, axis=1)], axis=0)
            super().__init__(bary, np.concatenate([w, w, w]), 2*dim+1)
        elif panel == COMMON_VERTEX:
            q0 = GaussJacobi(((quad_order_regular, 2.0+sg, 0), (quad_order_diagonal, 0, 0), (quad_order_diagonal, 0, 0)))
            q1 = GaussJacobi(((quad_order_regular, 2.0+sg, 0), (quad_order_diagonal, 1, 0), (quad_order_diagonal, 0, 0)))
            e0, e1, e2 = q0.nodes
            w0 = q0.weights*e0**(-sg)
            bx = [np.stack([1-e0, e0*(1-e1), e0*e1])]
            by = [np.stack([1-e0*e2, e0*e2])]
            e0, e1, e2 = q1.nodes
            w1 = q1.weights*e0**(-sg)
            bx.append(np.stack([1-e0*e1, e0*e1*(1-e2), e0*e1*e2]))
            by.append(np.stack([1-e0, e0]))
            bary = np.concatenate([np.concatenate(bx, axis=1), np.concatenate(by, axis=1)], axis=0)
            super().__init__(bary, np.concatenate([w0, w1]), 2*dim+1)
        else:
            raise NotImplementedError('Unknown panel type: {}'.format(panel))


class singularityCancelationQuadRule1D(quadratureRule):
    """nodes[0:2] = barycentric coords of x, nodes[2:4] of y (intervals)."""

    def __init__(self, panel, singularity, quad_order_diagonal, quad_order_regular):
        dim = 1
        sg = singularity
        if panel == COMMON_EDGE:
            q = GaussJacobi(((quad_order_regular, 1+sg, 0), (quad_order_regular, 0+sg, 0)))
            e0, e1 = q.nodes
            x, y = e0*(1-e1), e0
            bary = np.stack([1-x, x, 1-y, y])
            super().__init__(bary, 2.0*q.weights*(e0*e1)**(-sg), 2*dim+2)
        elif panel == COMMON_VERTEX:
            q = GaussJacobi(((quad_order_regular, 1+sg, 0), (quad_order_diagonal, 0, 0)))
            e0, e1 = q.nodes
            w = q.weights*e0**(-sg)
            x = np.concatenate([e0*e1, e0])
            y = np.concatenate([e0, e0*e1])
            bary = np.stack([1-x, x, 1-y, y])
            super().__init__(bary, np.concatenate([w, w]), 2*dim)
        else:
            raise NotImplementedError('Unknown panel type: {}'.format(panel))


class singularityCancelationQuadRule1D_boundary(quadratureRule):
    """nodes[0:2] = barycentric coords of x (interval), nodes[2] = 1 (boundary point)."""

    def __init__(self, panel, singularity, quad_order_diagonal, quad_order_regular):
        dim = 1
        if panel == COMMON_VERTEX:
            q = GaussJacobi(((quad_order_diagonal, singularity, 0), ))
            eta = q.nodes[0]
            bary = np.stack([1-eta, eta, np.ones_like(eta)])
            super().__init__(bary, q.weights*eta**(-singularity), 2*dim+1)
        else:
            raise NotImplementedError('Unknown panel type: {}'.format(panel))
