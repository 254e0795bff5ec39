"""Non-symmetric kernels with an order s(x) per quadrature point: the oracle's restatement of
fractionalLaplacian{1,2}D_nonsym + the non-symmetric branch of getDense, pinned by the reference's stored numbers
(tests/cache_runFractional.py--domaininterval--sconstantNonSym(0.75)..., --stwoDomainNonSym(0.25,0.75)--problemknownSolution...)
and by structural identities; the GPU comparison lives in test_gpu_parity.py."""
import numpy as np
import pytest
from math import gamma
from pynucleus_amd import disc, interval, driverMesh, P1_DoFMap, PHYSICAL, NO_BOUNDARY, getFractionalKernel, nonlocalTables
from pynucleus_amd.fractionalOrders import (constantNonSymFractionalOrder, smoothedLeftRightFractionalOrder,
                                            linearLeftRightFractionalOrder, smoothedInnerOuterFractionalOrder)
from oracle.oracle import OracleProblem


def known_solution_problem(dim, kernel, beta=0.7):
    """nonlocalProblems.py:710-726 (interval), :783-799 (disc): u = (1-|x|^2)^beta, rhs through 2F1 with s = s(x)"""
    from scipy.special import hyp2f1

    def rhs(x):
        s = kernel.s(x, x)
        if dim == 1:
            return 2**(2*s)*gamma(s+0.5)*gamma(beta+1.)/np.sqrt(np.pi)/gamma(beta+1.-s)*hyp2f1(s+0.5, -beta+s, 0.5, x[0]**2)
        return 2**(2*s)*gamma(s+1.0)*gamma(beta+1.)/gamma(beta+1.-s)*hyp2f1(s+1.0, -beta+s, 1.0, float(np.dot(x, x)))

    def sol(x):
        return max(1.-float(np.dot(x, x)), 0.)**beta

    L2ex2 = np.sqrt(np.pi)*gamma(1+2*beta)/gamma(3/2+2*beta) if dim == 1 else np.pi/(1+2*beta)
    return rhs, sol, L2ex2


class Gauss1D:
    """fem/PyNucleus_fem/quadrature.pyx:303-316, the default rule of assembleRHS for P1 in 1D (femCy.pyx:2636-2642, order 3)"""

    def __init__(self, order):
        x, w = np.polynomial.legendre.leggauss((order+1)//2)
        self.nodes = np.vstack([(x+1.)/2., 1.-(x+1.)/2.])
        self.weights = w/2.
        self.num_nodes = x.shape[0]


def l2_error(dm, u, sol, L2ex2):
    """discretizedProblems.py:77-93: sqrt(|L2_ex^2 - 2 z.u + u.M u|); the difference of O(1) numbers is dominated by the
    quadrature of z near the end points, so z uses the reference's default rule"""
    z = np.asarray(dm.assembleRHS(sol, qr=Gauss1D(3)))
    M = dm.assembleMass()
    return float(np.sqrt(abs(L2ex2-2*z@u+u@(M@u))))


def test_order_functions():
    """fractionalOrders.pyx:390-540: the device / oracle / host encodings agree (the oracle evaluates s(x) itself)"""
    mesh = disc(2)
    dm = P1_DoFMap(mesh, PHYSICAL)
    for sF in (smoothedLeftRightFractionalOrder(0.25, 0.75, r=0.3), linearLeftRightFractionalOrder(0.6, 0.3, r=0.25),
               smoothedInnerOuterFractionalOrder(0.3, 0.6, r=0.2), constantNonSymFractionalOrder(0.4)):
        k = getFractionalKernel(2, sF)
        assert not k.symmetric and not k.piecewise and k.variable
        T = nonlocalTables(dm, k, {})
        O = OracleProblem(T)
        from oracle.oracle import lib
        import ctypes as C
        sv = np.array([lib().nlo_pw_svalue(C.byref(O.P), c, (c*7+3) % mesh.num_cells) for c in range(mesh.num_cells)])
        ref = np.maximum(T.cell_smax, T.cell_smax[(np.arange(mesh.num_cells)*7+3) % mesh.num_cells])
        assert np.abs(sv-ref).max() < 1e-15
        assert sF.min-1e-15 <= T.cell_smax.min() and T.cell_smax.max() <= sF.max+1e-15


@pytest.mark.parametrize('s,stored,tol', [(0.75, 0.04184297664965481, 1e-6), (0.25, 0.09611243700814974, 1e-9)])
def test_interval_constant_nonsym_stored_error(s, stored, tol):
    """runFractional --domain interval --s constantNonSym(s) --problem constant --element P1 --matrixFormat dense:
    stored Hs errors 0.04184297664965481 (s=0.75, gmres tolerance of the stored run: 3e-7 here) and 0.09611243700814974
    (s=0.25, reproduced to 4e-12); the matrix is the constant-order one (same rules: the nonsym class drops the
    caller's target order and falls back to P+1-s, which is what the driver passes to the symmetric class)"""
    mesh = driverMesh('interval', 6)
    dm = P1_DoFMap(mesh, PHYSICAL)
    T = nonlocalTables(dm, getFractionalKernel(1, constantNonSymFractionalOrder(s)), {'target_order': 5.})
    assert T.target_order == 2.-s
    A, cnt, _ = OracleProblem(T).get_dense()
    b = np.asarray(dm.assembleRHS(1.0))
    u = np.linalg.solve(A, b)
    C = 2.**(-2.*s)*gamma(0.5)/gamma((1+2.*s)/2.)/gamma(1.+s)
    ex = C*np.sqrt(np.pi)*gamma(s+1)/gamma(s+3/2)
    hs = np.sqrt(abs(b@u-ex))
    assert abs(hs-stored) <= tol*stored, hs
    A0 = OracleProblem(nonlocalTables(dm, getFractionalKernel(1, s), {'target_order': 2.-s})).get_dense()[0]
    assert np.abs(A-A0).max() <= 1e-8*np.abs(A0).max()      # touching pairs: the two orientations differ at quadrature-error level
    assert cnt['numIntegrations'] > 0 and cnt['numAssembledCellPairs'] == 128*129//2


def test_interval_two_domain_nonsym_known_solution():
    """runFractional --domain interval --s twoDomainNonSym(0.25,0.75) --problem knownSolution --element P1 --solver lu
    --matrixFormat dense: stored L2 error 0.0020560901451394443 (compared there at rTol 3e-2, discretizedProblems.py:227)"""
    mesh = driverMesh('interval', 6)
    dm = P1_DoFMap(mesh, PHYSICAL)
    kernel = getFractionalKernel(1, smoothedLeftRightFractionalOrder(0.25, 0.75))
    A = OracleProblem(nonlocalTables(dm, kernel, {})).get_dense()[0]
    assert np.abs(A-A.T).max() > 1e-3*np.abs(A).max()
    rhs, sol, L2ex2 = known_solution_problem(1, kernel)
    from pynucleus_amd.quadrature import simplexXiaoGimbutas
    b = np.asarray(dm.assembleRHS(rhs, qr=simplexXiaoGimbutas(3, 1, 1)))     # discretizedProblems.py:561
    u = np.linalg.solve(A, b)
    err = l2_error(dm, u, sol, L2ex2)
    assert abs(err-0.0020560901451394443) <= 3e-2*0.0020560901451394443, err


@pytest.mark.parametrize('dim,noRef', [(2, 2), (1, 4)])
def test_nonsym_structure(dim, noRef):
    """[u(x) g(x,y) - u(y) g(y,x)] [v(x) - v(y)]: the test-function factor is a difference, so A 1 = 0 on an all-vertex DoF map
    without the exterior term although the operator is not symmetric; a constant order gives back the symmetric operator away
    from touching pairs (their two orientations differ at quadrature-error level)"""
    mesh = disc(noRef) if dim == 2 else interval(noRef)
    dm = P1_DoFMap(mesh, NO_BOUNDARY)
    sF = smoothedLeftRightFractionalOrder(0.3, 0.6, r=0.4)
    A = OracleProblem(nonlocalTables(dm, getFractionalKernel(dim, sF), {}, zeroExterior=False)).get_dense()[0]
    assert np.abs(A.sum(axis=1)).max() < 1e-10*np.abs(A).max()
    assert np.abs(A-A.T).max() > 1e-3*np.abs(A).max()
    dmI = P1_DoFMap(mesh, PHYSICAL)
    Tn = nonlocalTables(dmI, getFractionalKernel(dim, constantNonSymFractionalOrder(0.4)), {})
    An = OracleProblem(Tn).get_dense()[0]
    A0 = OracleProblem(nonlocalTables(dmI, getFractionalKernel(dim, 0.4), {'target_order': Tn.target_order})).get_dense()[0]
    assert np.abs(An-A0).max() < 1e-5*np.abs(A0).max()
    assert np.abs(An-An.T).max() < 1e-13*np.abs(A0).max()


def test_host_tables_reject_what_is_not_built():
    with pytest.raises(NotImplementedError):
        getFractionalKernel(2, constantNonSymFractionalOrder(0.4), horizon=0.5)


def test_pointwise_p2_oracle():
    """P2 elements for the kernels with an order per quadrature point (round 3; FL2:894-1184 / FL1:410-604 are element-agnostic):
    with a constant order the non-symmetric P2 operator is the symmetric one -- exactly in 1D (every rule is Gauss-Jacobi), up to
    the dropped target_order / quad_order_diagonal of the _nonsym constructors in 2D"""
    from pynucleus_amd import P2_DoFMap
    s = 0.4
    dm = P2_DoFMap(interval(4), PHYSICAL)
    T = nonlocalTables(dm, getFractionalKernel(1, constantNonSymFractionalOrder(s)), {})
    A = OracleProblem(T).get_dense()[0]
    A0 = OracleProblem(nonlocalTables(dm, getFractionalKernel(1, s), {'target_order': T.target_order})).get_dense()[0]
    assert np.abs(A-A0).max() <= 1e-13*np.abs(A0).max()
    dm = P2_DoFMap(disc(1), PHYSICAL)
    T = nonlocalTables(dm, getFractionalKernel(2, constantNonSymFractionalOrder(s)), {})
    A, cnt, _ = OracleProblem(T).get_dense()
    A0, c0, _ = OracleProblem(nonlocalTables(dm, getFractionalKernel(2, s), {'target_order': T.target_order})).get_dense()
    assert cnt['numAssembledCellPairs'] == c0['numAssembledCellPairs'] and np.abs(A-A0).max() <= 1e-4*np.abs(A0).max()
    assert np.abs(A-A.T).max() <= 1e-6*np.abs(A).max()


def test_piecewise_nonsymmetric_order_oracle():
    """s(l1, l2) != s(l2, l1), piecewise: both orientations with the parameters of each (NA:1411-1428).  With parameters frozen
    per orientation the local matrix is the symmetric one, so without the exterior term the operator is the average of the
    two symmetric operators built from s12 and from s21, up to the touching pairs' quadrature asymmetry"""
    from pynucleus_amd.fractionalOrders import leftRightFractionalOrder
    mesh = disc(2)
    dm = P1_DoFMap(mesh, PHYSICAL)

    def mat(*a):
        k = getFractionalKernel(2, leftRightFractionalOrder(*a))
        T = nonlocalTables(dm, k, {}, zeroExterior=False)
        return OracleProblem(T).get_dense()[0], k, T
    A, k, T = mat(0.25, 0.75, 0.3, 0.6)
    assert not k.symmetric and k.piecewise and T.nonsym
    A1 = mat(0.25, 0.75, 0.3, 0.3)[0]
    A2 = mat(0.25, 0.75, 0.6, 0.6)[0]
    assert np.abs(A-A.T).max() <= 1e-13*np.abs(A).max()
    assert np.abs(A-0.5*(A1+A2)).max() <= 1e-6*np.abs(A).max()
    assert np.abs(A1-A2).max() > 1e-2*np.abs(A).max()


def test_golden_nonsymmetric():
    """regression net for the non-symmetric oracle paths (tests/golden/make_golden.py)"""
    import os
    from pynucleus_amd.fractionalOrders import leftRightFractionalOrder
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'oracle_golden.npz'))
    mesh = disc(2)
    dm = P1_DoFMap(mesh, PHYSICAL)
    A = OracleProblem(nonlocalTables(dm, getFractionalKernel(2, smoothedLeftRightFractionalOrder(0.25, 0.75, r=0.3)), {})).get_dense()[0]
    ref = gold['dense_disc2_smoothedLeftRight']
    assert np.abs(A-ref).max() <= 1e-13*np.abs(ref).max()
    A = OracleProblem(nonlocalTables(dm, getFractionalKernel(2, leftRightFractionalOrder(0.25, 0.75, 0.3, 0.6)), {})).get_dense()[0]
    ref = gold['dense_disc2_leftRight_nonsym']
    assert np.abs(A-ref).max() <= 1e-13*np.abs(ref).max()
