"""Quadrature tables: exactness of the triangle rules, Gauss-Jacobi, and the geometry of the
singularity-cancelling transformations (with singularity 0 they must integrate polynomials over
K x K exactly: the sub-domains tile the product domain and carry the right Jacobians)."""
import numpy as np
import pytest
from math import factorial
from pynucleus_amd.quadrature import (GaussJacobi, simplexDuffyTransformation, simplexXiaoGimbutas, triangleRule,
                                      singularityCancelationQuadRule2D, singularityCancelationQuadRule2D_boundary,
                                      singularityCancelationQuadRule1D, singularityCancelationQuadRule1D_boundary,
                                      COMMON_FACE, COMMON_EDGE, COMMON_VERTEX)


def tri_moment(a, b, c):
    return 2.*factorial(a)*factorial(b)*factorial(c)/factorial(a+b+c+2)


@pytest.mark.parametrize('q', list(range(2, 31)))
def test_triangle_rule_exact(q):
    nodes, w = triangleRule(q)
    assert np.isclose(w.sum(), 1.) and (w > 0).all() and (nodes > 0).all()
    assert np.allclose(nodes.sum(axis=0), 1.)
    for a in range(q+1):
        for b in range(q+1-a):
            c = q-a-b
            assert abs((w*nodes[0]**a*nodes[1]**b*nodes[2]**c).sum()-tri_moment(a, b, c)) < 2e-15


def test_triangle_rule_point_counts():
    # the counts of the Xiao-Gimbutas family the reference uses for the orders that dominate
    assert [triangleRule(q)[0].shape[1] for q in range(2, 9)] == [3, 6, 6, 7, 12, 15, 16]


def test_gauss_jacobi():
    qr = GaussJacobi(((5, 1.5, 0), (3, 0, 1)))
    x, y = qr.nodes
    for i in range(6):
        for j in range(4):
            exact = 1./(i+2.5)*(1./(j+1)-1./(j+2))
            assert np.isclose((qr.weights*x**i*y**j).sum(), exact, rtol=1e-13)


@pytest.mark.parametrize('q', [2, 3, 5, 8])
def test_duffy(q):
    qr = simplexDuffyTransformation(q, 2, 2)
    assert np.isclose(qr.weights.sum(), 1.)
    for a in range(q+1):
        for b in range(q+1-a):
            assert np.isclose((qr.weights*qr.nodes[1]**a*qr.nodes[2]**b).sum(), tri_moment(0, a, b), atol=1e-14)
    g = simplexXiaoGimbutas(q, 1, 1)
    assert np.isclose((g.weights*g.nodes[1]**q).sum(), 1./(q+1))


def ref_tri_pair_moment(ax, ay):
    """int_{K x K} over two reference triangles of prod lambda^a (x) prod lambda^b (y), each normalised to area 1/2"""
    return 0.25*tri_moment(*ax)*tri_moment(*ay)


@pytest.mark.parametrize('panel', [COMMON_FACE, COMMON_EDGE, COMMON_VERTEX])
def test_singular_rule_2d_geometry(panel):
    """With a constant kernel (rule parameter = the two cancelled orders) the rule must reproduce
    int_{K1} int_{K2} psi_r psi_s exactly, psi = merged P1 basis differences (FL2:662-811): the sub-domains tile
    K1 x K2 with the right Jacobians.  Exact values are products of triangle moments."""
    qr = singularityCancelationQuadRule2D(panel, 2., 8, 8)
    bx, by = qr.nodes[:3], qr.nodes[3:]
    assert (qr.nodes > -1e-14).all() and np.allclose(bx.sum(axis=0), 1.) and np.allclose(by.sum(axis=0), 1.)
    e = np.eye(3)
    z = np.zeros(3)
    rows = {COMMON_FACE: [(e[0], e[0]), (e[1], e[1]), (e[2], e[2])],
            COMMON_EDGE: [(e[0], e[0]), (e[1], e[1]), (e[2], z), (z, e[2])],
            COMMON_VERTEX: [(e[0], e[0]), (e[1], z), (e[2], z), (z, e[1]), (z, e[2])]}[panel]
    Mxx = np.array([[0.25*tri_moment(*(e[k]+e[l]).astype(int)) for l in range(3)] for k in range(3)])
    Mxy = np.full((3, 3), 1./36)
    for cxr, cyr in rows:
        for cxs, cys in rows:
            psi_r = cxr@bx-cyr@by
            psi_s = cxs@bx-cys@by
            exact = cxr@Mxx@cxs+cyr@Mxx@cys-cxr@Mxy@cys-cyr@Mxy@cxs
            assert np.isclose((qr.weights*psi_r*psi_s).sum(), exact, rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize('panel', [COMMON_EDGE, COMMON_VERTEX])
def test_singular_rule_2d_boundary_geometry(panel):
    qr = singularityCancelationQuadRule2D_boundary(panel, 0., 8, 8)
    bx, by = qr.nodes[:3], qr.nodes[3:]
    assert (qr.nodes > -1e-14).all() and np.allclose(bx.sum(axis=0), 1.) and np.allclose(by.sum(axis=0), 1.)
    # triangle (area 1/2) x edge (length 1)
    assert np.isclose(qr.weights.sum(), 0.5)
    val = (qr.weights*bx[1]*bx[2]*by[1]**2).sum()
    assert np.isclose(val, 0.5*tri_moment(0, 1, 1)/3., rtol=1e-12)


def test_singular_rule_1d_geometry():
    qr = singularityCancelationQuadRule1D(COMMON_EDGE, 0., 6, 6)
    assert np.isclose(qr.weights.sum(), 1.)
    # identical cells: the x <-> y mirror image is folded in (factor 2, FL1:78), so symmetric integrands only
    assert np.isclose((qr.weights*qr.nodes[1]*qr.nodes[3]).sum(), 0.25)
    assert np.isclose((qr.weights*(qr.nodes[1]-qr.nodes[3])**2).sum(), 1./6)
    qr = singularityCancelationQuadRule1D(COMMON_VERTEX, 0., 6, 6)
    assert np.isclose(qr.weights.sum(), 1.)
    assert np.isclose((qr.weights*qr.nodes[1]**2*qr.nodes[3]).sum(), 1./3*0.5)
    qr = singularityCancelationQuadRule1D_boundary(COMMON_VERTEX, 0., 6, 1)
    assert np.isclose(qr.weights.sum(), 1.)


def test_singular_rule_integrates_singularity():
    """int_0^1 int_0^1 |x-y|^(-1-2s) (x-y)^2 dx dy = 2/((2-2s)(3-2s)) : identical 1D cells, fractional weight"""
    s = 0.75
    sing = -1.-2*s
    qr = singularityCancelationQuadRule1D(COMMON_EDGE, 2.+sing, 8, 4)
    x, y = qr.nodes[1], qr.nodes[3]
    val = (qr.weights*np.abs(x-y)**sing*(x-y)**2).sum()
    assert np.isclose(val, 2./((2-2*s)*(3-2*s)), rtol=1e-12)
