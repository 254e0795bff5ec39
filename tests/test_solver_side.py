"""SURVEY 8(f) row 4, the solver side: geometric multigrid on hierarchies of assembled nonlocal operators, multigrid-
preconditioned CG and the theta time stepper of the fractional heat equation.

CPU part: the numpy restatement (oracle/solver_oracle.py) against the reference's own stored numbers
(tests/cache_runFractionalHeat.py--..., tests/cache_runFractional.py--...--solvercg-mg--..., compared there at
rTol 1e-2 ... 3e-2) and the host-side transfer operators of the product against the oracle's cell walk.
GPU part: the device cycle / CG / time step (libpnl_hip.so through the C ABI) against the oracle on the same hierarchy.
"""
import numpy as np
import pytest
from math import gamma
from scipy.special import hyp2f1
from pynucleus_amd import P1_DoFMap, P2_DoFMap, PHYSICAL, getFractionalKernel, nonlocalTables
from pynucleus_amd.multigrid import buildProlongation, buildRestriction, _seed_mesh, determineTimeSteps
from pynucleus_amd.quadrature import simplexXiaoGimbutas
from oracle.oracle import OracleProblem
from oracle import solver_oracle as SO


def oracle_hierarchy(domain, noRef, s, params, mass=False, element='P1'):
    mesh = _seed_mesh(domain)
    dim = mesh.manifold_dim
    levels = []
    from pynucleus_amd import dofmapFactory
    build_restriction = {'P0': SO.build_restriction_P0, 'P1': SO.build_restriction_P1, 'P2': SO.build_restriction_P2,
                         'P3': SO.build_restriction_P3}[element]
    for l in range(noRef+1):
        if l > 0:
            mesh = mesh.refine()
        dm = dofmapFactory(element, mesh, PHYSICAL)
        L = {'mesh': mesh, 'DoFMap': dm, 'A': OracleProblem(nonlocalTables(dm, getFractionalKernel(dim, s), dict(params))).get_dense()[0]}
        if mass:
            L['M'] = dm.assembleMass().toarray()
        if l > 0:
            L['R'] = build_restriction(levels[-1]['DoFMap'], dm)
            L['P'] = L['R'].T.copy()
        levels.append(L)
    return levels


def transient_problem(dim, s, problem):
    """nonlocalProblems.py:651-662 ('constant'), :710-727 ('knownSolution') on the interval, transient version :1641-1672:
    u(t, x) = cos(t) u_ss(x), f(t) = -sin(t) u_ss + cos(t) f_ss"""
    assert dim == 1
    beta = 0.7
    if problem == 'knownSolution':
        def uss(x):
            return max(1.-x[0]**2, 0.)**beta

        def fss(x):
            return 2**(2*s)*gamma(s+0.5)*gamma(beta+1.)/np.sqrt(np.pi)/gamma(beta+1.-s)*hyp2f1(s+0.5, -beta+s, 0.5, x[0]**2)
        L2ex2 = np.sqrt(np.pi)*gamma(1+2*beta)/gamma(1.5+2*beta)
    else:
        C = 2.**(-2.*s)*gamma(0.5)/gamma((1+2.*s)/2.)/gamma(1.+s)

        def uss(x):
            return C*max(1.-x[0]**2, 0.)**s

        def fss(x):
            return 1.
        L2ex2 = C**2*np.sqrt(np.pi)*gamma(1+2*s)/gamma(1.5+2*s)
    return uss, fss, L2ex2


# the reference's stored numbers: (final L2 error, L2(0,T;L2) error, L2(0,T;L2) norm)
HEAT_FIXTURES = {
    # cache_runFractionalHeat.py--domaininterval--sconst(0.25)--problemconstant--elementP1--solvercg-mg--matrixFormatdense
    (0.25, 'constant'): (0.01455872345929613, 0.03218338586612875, 1.7018299503210628, 2e-6),
    # ...--sconst(0.25)--problemknownSolution--elementP1--solvercg-jacobi--matrixFormatH2
    (0.25, 'knownSolution'): (0.0008912208343986159, 0.0018388585398440504, 1.3228634831094461, 2e-5),
    # ...--sconst(0.75)--problemknownSolution--elementP1--solvercg-mg--matrixFormatH2 (H2 approximation in the stored run)
    (0.75, 'knownSolution'): (0.001419170068070812, 0.002695812286552434, 1.3216806035576412, 5e-4),
}


def heat_setup(levels, s, problem):
    dm = levels[-1]['DoFMap']
    uss, fss, L2ex2 = transient_problem(1, s, problem)
    qr = simplexXiaoGimbutas(3, 1, 1)                        # buildQuadratureRule, discretizedProblems.py:771-772
    z_ss, f_ss = np.asarray(dm.assembleRHS(uss, qr)), np.asarray(dm.assembleRHS(fss, qr))

    def load(t):
        return -np.sin(t)*z_ss+np.cos(t)*f_ss
    return uss, load, z_ss, L2ex2


@pytest.mark.parametrize('s,problem', sorted(HEAT_FIXTURES))
def test_oracle_heat_reproduces_the_stored_errors(s, problem):
    """runFractionalHeat --domain interval --element P1 (noRef 6, Crank-Nicolson, dt = sqrt(h) = 1/8, finalTime 1)"""
    levels = oracle_hierarchy('interval', 6, s, {'target_order': 2.-s}, mass=True)
    L = levels[-1]
    uss, load, z_ss, L2ex2 = heat_setup(levels, s, problem)
    dt, nt = SO.heat_time_steps(L['mesh'].h)
    assert (dt, nt) == (0.125, 8)
    theta = 0.5
    trans = [dict(K, A=K['M']/dt+theta*K['A']) for K in levels]
    mg = SO.Multigrid(trans)
    times = np.linspace(0., 1., nt+1)
    u = np.asarray(L['DoFMap'].interpolate(uss), dtype=float)
    us, its = [u.copy()], []

    def solve(rhs, x0):
        x, it, _ = SO.cg(trans[-1]['A'], rhs, x0=x0, tol=1e-10, maxiter=100, B=mg.precondition)
        its.append(it)
        return x
    for k in range(nt):
        forcing = (1-theta)*load(times[k])+theta*load(times[k+1])
        u = SO.theta_step(L['A'], L['M'], dt, theta, forcing, u, solve)
        us.append(u.copy())
    e_final, e_l2, norm = SO.transient_errors(us, times, L['M'], lambda t: np.cos(t)*z_ss, lambda t: np.cos(t)**2*L2ex2)
    ref = HEAT_FIXTURES[(s, problem)]
    rtol = ref[3]
    assert abs(e_final-ref[0]) <= rtol*ref[0] and abs(e_l2-ref[1]) <= rtol*ref[1], (e_final, e_l2, ref)
    assert abs(norm-ref[2]) <= 1e-6*ref[2], (norm, ref[2])
    assert max(its) <= 8                                      # multigrid-preconditioned CG: mesh-independent iteration counts


@pytest.mark.parametrize('element,s,noRef,stored_norm,rtol,stored_errors', [
    ('P0', 0.25, 6, 1.7025600858867103, 1e-8, (0.007567757829891671, 0.0149413089985309)),
    ('P3', 0.25, 5, 1.7026331344615124, 1e-4, None), ('P3', 0.75, 5, 0.9834064913824577, 1e-5, None)])
def test_oracle_heat_P0_P3(element, s, noRef, stored_norm, rtol, stored_errors):
    """runFractionalHeat --domain interval --element P0 / P3 --solver cg-mg --matrixFormat dense: hierarchies of P0 (prolongation =
    injection, restriction_1D_P0.pxi) and P3 spaces (restriction_1D_P3.pxi) through the same time stepper.  P0 (noRef 6, dt = 1/8):
    the stored L2(0,T;L2) norm to 1e-10 and both stored error norms to 5e-6; P3 (noRef 5, dt = 1/6): the norm to 1e-5 / 3e-7, the
    error norms are not reproduced, as for P2 below (they hinge on the quadrature of the boundary-singular solution in
    z = assembleRHS(u)) -- the P3 operators are pinned by the steady fixtures (tests/test_oracle_pinning.py, 2e-12)"""
    levels = oracle_hierarchy('interval', noRef, s, {'target_order': int(element[1])+1.-s}, mass=True, element=element)
    L = levels[-1]
    uss, load, z_ss, L2ex2 = heat_setup(levels, s, 'constant')
    dt, nt = SO.heat_time_steps(L['mesh'].h)
    trans = [dict(K, A=K['M']/dt+0.5*K['A']) for K in levels]
    mg = SO.Multigrid(trans)
    times = np.linspace(0., 1., nt+1)
    u = np.asarray(L['DoFMap'].interpolate(uss), dtype=float)
    us, its = [u.copy()], []
    for k in range(nt):
        forcing = 0.5*load(times[k])+0.5*load(times[k+1])

        def solve(rhs, x0):
            x, it, _ = SO.cg(trans[-1]['A'], rhs, x0=x0, tol=1e-10, maxiter=200, B=mg.precondition)
            its.append(it)
            return x
        u = SO.theta_step(L['A'], L['M'], dt, 0.5, forcing, u, solve)
        us.append(u.copy())
    e_final, e_l2, norm = SO.transient_errors(us, times, L['M'], lambda t: np.cos(t)*z_ss, lambda t: np.cos(t)**2*L2ex2)
    assert abs(norm-stored_norm) <= rtol*stored_norm, (norm, e_final, e_l2, max(its))
    if stored_errors is not None:
        assert abs(e_final-stored_errors[0]) <= 1e-4*stored_errors[0] and abs(e_l2-stored_errors[1]) <= 1e-4*stored_errors[1], (e_final, e_l2)
    assert max(its) <= 10, max(its)


@pytest.mark.parametrize('s,stored_norm,rtol', [(0.25, 1.7019259587916384, 1e-5), (0.75, 0.9832074391209417, 1e-6)])
def test_oracle_heat_P2_norm(s, stored_norm, rtol):
    """runFractionalHeat --domain interval --element P2 --solver cg-mg --matrixFormat dense (noRef 5, dt = 1/6): the P2 hierarchy
    (transfer weights of restriction_1D_P2.pxi) through the same time stepper reproduces the stored L2(0,T;L2) NORM; the stored
    error norms (0.0124 / 0.00044) are not reproduced (0.0139 / 0.00074 here: they hinge on the quadrature of the boundary-
    singular load, which that run takes from a rule this container does not have) -- those stay unpinned for P2"""
    levels = oracle_hierarchy('interval', 5, s, {'target_order': 2.-s}, mass=True, element='P2')
    L = levels[-1]
    uss, load, z_ss, L2ex2 = heat_setup(levels, s, 'constant')
    dt, nt = SO.heat_time_steps(L['mesh'].h)
    assert nt == 6
    trans = [dict(K, A=K['M']/dt+0.5*K['A']) for K in levels]
    mg = SO.Multigrid(trans)
    times = np.linspace(0., 1., nt+1)
    u = np.asarray(L['DoFMap'].interpolate(uss), dtype=float)
    us, its = [u.copy()], []
    for k in range(nt):
        forcing = 0.5*load(times[k])+0.5*load(times[k+1])

        def solve(rhs, x0):
            x, it, _ = SO.cg(trans[-1]['A'], rhs, x0=x0, tol=1e-10, maxiter=100, B=mg.precondition)
            its.append(it)
            return x
        u = SO.theta_step(L['A'], L['M'], dt, 0.5, forcing, u, solve)
        us.append(u.copy())
    norm = SO.transient_errors(us, times, L['M'], lambda t: np.cos(t)*z_ss, lambda t: np.cos(t)**2*L2ex2)[2]
    assert abs(norm-stored_norm) <= rtol*stored_norm, norm
    assert max(its) <= 12


def test_oracle_multigrid_is_a_solver_for_the_stored_steady_run():
    """runFractional --domain interval --s const(0.25) --solver cg-mg --matrixFormat dense: the multigrid-preconditioned CG
    reaches the solution whose Hs error the reference stores (0.09611243700804001), in a handful of iterations; the
    stand-alone V cycle contracts at a mesh-independent rate"""
    s = 0.25
    levels = oracle_hierarchy('interval', 6, s, {'target_order': 2.-s})
    dm = levels[-1]['DoFMap']
    b = np.asarray(dm.assembleRHS(1.0))
    mg = SO.Multigrid(levels)
    x, its, res = SO.cg(levels[-1]['A'], b, tol=1e-10, B=mg.precondition)
    C = 2.**(-2.*s)*gamma(0.5)/gamma((1+2.*s)/2.)/gamma(1.+s)
    ex = C*np.sqrt(np.pi)*gamma(s+1)/gamma(s+3/2)
    hs = np.sqrt(abs(b@x-ex))
    assert abs(hs-0.09611243700804001) <= 1e-8*0.09611243700804001 and its <= 10
    x2, it2, r2 = mg.solve(b, tol=1e-9*np.linalg.norm(b), maxiter=50)
    assert it2 < 50 and np.abs(x2-x).max() <= 1e-7*np.abs(x).max()
    rates = [r2[k+1]/r2[k] for k in range(2, len(r2)-1)]
    assert max(rates) < 0.5


@pytest.mark.parametrize('domain,element', [('interval', 'P1'), ('disc', 'P1'), ('interval', 'P2'), ('disc', 'P2'), ('interval', 'P0'),
                                            ('disc', 'P0'), ('interval', 'P3')])
def test_transfer_operators_equal_the_cell_walk(domain, element):
    """the product's prolongation (coarse shape functions at the fine nodes) against the reference's tabulated weights
    (restriction_{1,2}D_P{0,1,2}.pxi, restriction_1D_P3.pxi) walked cell by cell"""
    from pynucleus_amd.dofmap import P0_DoFMap, P3_DoFMap
    mesh = _seed_mesh(domain)
    DM, walk = {'P0': (P0_DoFMap, SO.build_restriction_P0), 'P1': (P1_DoFMap, SO.build_restriction_P1),
                'P2': (P2_DoFMap, SO.build_restriction_P2), 'P3': (P3_DoFMap, SO.build_restriction_P3)}[element]
    for _ in range(3):
        fine = mesh.refine()
        dc, df = DM(mesh, PHYSICAL), DM(fine, PHYSICAL)
        P, R = buildProlongation(dc, df), buildRestriction(dc, df)
        Ro = walk(dc, df)
        tol = 1e-15 if element == 'P3' else 0.                # cubic shape functions at thirds and sixths: rounding of 13.5 l0 l1 (l0 - 1/3)
        assert np.abs(R.toarray()-Ro).max() <= tol and np.abs(P.toarray().T-Ro).max() <= tol
        assert P.shape == (df.num_dofs, dc.num_dofs) and abs(P.toarray().max()-1.) <= tol
        # the prolongation reproduces coarse functions: the interpolant of a coarse FE function at the fine nodes
        uc = np.random.default_rng(0).standard_normal(dc.num_dofs)
        xf = df.getDoFCoordinates()
        if element == 'P1' and domain == 'interval':
            xc = dc.getDoFCoordinates()[:, 0]
            order = np.argsort(xc)
            ref = np.interp(xf[:, 0], np.concatenate(([-1.], xc[order], [1.])), np.concatenate(([0.], uc[order], [0.])))
            assert np.abs(P@uc-ref).max() < 1e-14
        mesh = fine


def test_time_step_rule():
    assert determineTimeSteps(2./128, 1.0) == (0.125, 8)
    dt, n = determineTimeSteps(2./128, 1.0, 'Implicit Euler')
    assert n == 64 and dt == 1./64


# ---- device -----------------------------------------------------------------------------------------------------------------
def device_hierarchy(domain, noRef, s, params, mass=False, element='P1'):
    from pynucleus_amd.multigrid import fractionalHierarchy
    dim = 1 if domain == 'interval' else 2
    return fractionalHierarchy(domain, noRef, getFractionalKernel(dim, s), params, buildMass=mass, element=element)


def as_oracle_levels(H):
    out = []
    for L in H.getLevelList():
        K = {'A': L['A'].toarray().copy()}
        if 'M' in L:
            K['M'] = L['M'].toarray()
        if 'R' in L:
            K['R'], K['P'] = L['R'].toarray(), L['P'].toarray()
        out.append(K)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize('domain,noRef,s,element', [('interval', 6, 0.25, 'P1'), ('disc', 3, 0.75, 'P1'), ('interval', 4, 0.75, 'P2'),
                                                    ('disc', 2, 0.4, 'P2'), ('interval', 5, 0.25, 'P0'), ('interval', 4, 0.75, 'P3'),
                                                    ('disc', 3, 0.25, 'P0')])
def test_gpu_cycle_solve_and_cg_against_the_oracle(domain, noRef, s, element):
    from pynucleus_amd.multigrid import multigrid
    params = {'target_order': int(element[1])+1.-s} if domain == 'interval' else {}
    H = device_hierarchy(domain, noRef, s, params, element=element)
    levels_o = oracle_hierarchy(domain, noRef, s, params, element=element)
    # the device hierarchy is the oracle's: operators at the assembly tolerance, transfer operators exactly
    for L, K in zip(H.getLevelList(), levels_o):
        assert np.abs(L['A'].toarray()-K['A']).max() <= 1e-10*np.abs(K['A']).max()
    # solver parity on identical data: the oracle runs on the matrices the GPU assembled
    lv = as_oracle_levels(H)
    mo = SO.Multigrid(lv)
    mg = multigrid(H)
    dm = H.finest['DoFMap']
    b = np.asarray(dm.assembleRHS(1.0))
    n = b.shape[0]
    rng = np.random.default_rng(0)
    x0 = rng.standard_normal(n)
    # one cycle from zero and from a guess
    xo = np.zeros(n); mo.solveOnLevel(len(lv)-1, b, xo, True)
    assert np.abs(mg.cycle(b)-xo).max() <= 1e-12*np.abs(xo).max()
    xo = x0.copy(); mo.solveOnLevel(len(lv)-1, b, xo, False)
    assert np.abs(mg.cycle(b, x0)-xo).max() <= 1e-12*np.abs(xo).max()
    # stationary iteration
    tol = 1e-9*np.linalg.norm(b)
    xs, its, res = mg.solve(b, tol=tol, maxiter=60)
    xso, itso, reso = mo.solve(b, tol=tol, maxiter=60)
    assert its == itso and its < 60
    assert np.allclose(res, reso, rtol=1e-6, atol=1e-14*res[0])
    assert np.abs(xs-xso).max() <= 1e-10*np.abs(xso).max()
    # multigrid-preconditioned CG
    xc, itc, resc = mg.cg(b, tol=1e-10)
    xco, itco, resco = SO.cg(lv[-1]['A'], b, tol=1e-10, B=mo.precondition)
    assert itc == itco and itc <= 12
    assert np.allclose(resc, resco, rtol=1e-5, atol=1e-13)
    assert np.abs(xc-xco).max() <= 1e-9*np.abs(xco).max()
    assert np.abs(lv[-1]['A']@xc-b).max() <= 1e-8*np.abs(b).max()
    # the generic Krylov loop of the package with the cycle as a callable preconditioner runs the same iteration
    B = mg.asPreconditioner()
    z = B(b)
    assert np.abs(z-mo.precondition(b)).max() <= 1e-12*np.abs(z).max()


@pytest.mark.gpu
def test_gpu_steady_run_cg_mg_reproduces_the_stored_hs_errors():
    """the reference's stored Hs errors through multigrid-preconditioned CG on the device:
    interval s = 0.25 (cache_runFractional.py--domaininterval--sconst(0.25)--problemconstant--elementP1--solvercg-mg--
    matrixFormatdense: 0.09611243700804001) and disc s = 0.25, noRef 5 (...domaindisc--sconst(0.25)...: 0.1839933908571473)"""
    from pynucleus_amd.multigrid import multigrid
    for domain, noRef, s, params, stored, rtol, nd in (('interval', 6, 0.25, {'target_order': 1.75}, 0.09611243700804001, 1e-8, 127),
                                                       ('disc', 5, 0.25, {'target_order': 0.5}, 0.1839933908571473, 1e-5, 2977)):
        H = device_hierarchy(domain, noRef, s, params)
        dm = H.finest['DoFMap']
        assert dm.num_dofs == nd
        b = np.asarray(dm.assembleRHS(1.0))
        x, its, res = multigrid(H).cg(b, tol=1e-10)
        dim = 1 if domain == 'interval' else 2
        C = 2.**(-2.*s)*gamma(dim/2.)/gamma((dim+2.*s)/2.)/gamma(1.+s)
        ex = C*np.sqrt(np.pi)*gamma(s+1)/gamma(s+3/2) if dim == 1 else C*np.pi/(s+1)
        hs = np.sqrt(abs(b@x-ex))
        assert abs(hs-stored) <= rtol*stored, (domain, hs)
        assert its <= 12, its


@pytest.mark.gpu
@pytest.mark.parametrize('s,problem', sorted(HEAT_FIXTURES))
def test_gpu_fractional_heat_reproduces_the_stored_errors(s, problem):
    from pynucleus_amd.multigrid import solveFractionalHeat
    H = device_hierarchy('interval', 6, s, {'target_order': 2.-s}, mass=True)
    L = H.finest
    uss, load, z_ss, L2ex2 = heat_setup(H.getLevelList(), s, problem)
    times, us, stepper = solveFractionalHeat(H, uss, load, finalTime=1.0, tol=1e-10)
    M = L['M'].toarray()
    e_final, e_l2, norm = SO.transient_errors(us, times, M, lambda t: np.cos(t)*z_ss, lambda t: np.cos(t)**2*L2ex2)
    ref = HEAT_FIXTURES[(s, problem)]
    assert abs(e_final-ref[0]) <= ref[3]*ref[0] and abs(e_l2-ref[1]) <= ref[3]*ref[1], (e_final, e_l2, ref)
    assert abs(norm-ref[2]) <= 1e-6*ref[2]
    assert len(us) == 9 and max(stepper.iterations) <= 8
    # step-by-step parity with the oracle's time stepper on the same (device-assembled) matrices
    lv = as_oracle_levels(H)
    dt, theta = 0.125, 0.5
    trans = [dict(K, A=K['M']/dt+theta*K['A']) for K in lv]
    mo = SO.Multigrid(trans)
    u = us[0].copy()
    for k in range(8):
        forcing = (1-theta)*load(times[k])+theta*load(times[k+1])
        u = SO.theta_step(lv[-1]['A'], lv[-1]['M'], dt, theta, forcing, u,
                          lambda rhs, x0: SO.cg(trans[-1]['A'], rhs, x0=x0, tol=1e-10, maxiter=100, B=mo.precondition)[0])
        assert np.abs(u-us[k+1]).max() <= 1e-9*np.abs(u).max(), k


@pytest.mark.gpu
def test_gpu_fractional_heat_P0_reproduces_the_stored_errors():
    """runFractionalHeat --domain interval --s const(0.25) --problem constant --element P0 --solver cg-mg --matrixFormat dense through
    the product path (P0 hierarchy assembled on the device, Crank-Nicolson with cg-mg in the library): stored L2(Omega) error at
    t = 1 0.007567757829891671, L2(0,T;L2) error 0.0149413089985309, norm 1.7025600858867103"""
    from pynucleus_amd.multigrid import solveFractionalHeat
    s = 0.25
    H = device_hierarchy('interval', 6, s, {'target_order': 1.-s}, mass=True, element='P0')
    uss, load, z_ss, L2ex2 = heat_setup(H.getLevelList(), s, 'constant')
    times, us, stepper = solveFractionalHeat(H, uss, load, finalTime=1.0, tol=1e-10)
    e_final, e_l2, norm = SO.transient_errors(us, times, H.finest['M'].toarray(), lambda t: np.cos(t)*z_ss, lambda t: np.cos(t)**2*L2ex2)
    assert abs(e_final-0.007567757829891671) <= 1e-4*0.007567757829891671 and abs(e_l2-0.0149413089985309) <= 1e-4*0.0149413089985309, (e_final, e_l2)
    assert abs(norm-1.7025600858867103) <= 1e-8*1.7025600858867103, norm
    assert len(us) == 9 and max(stepper.iterations) <= 8


@pytest.mark.gpu
def test_gpu_solver_kernels():
    """pnl_gemv_axpby / pnl_csr_matvec against numpy (odd sizes, unaligned views, beta paths)"""
    import torch
    import scipy.sparse as sp
    from pynucleus_amd import _lib
    from pynucleus_amd.multigrid import _DevCSR
    ctx = _lib.Context(0)
    ctx.set_stream(torch.cuda.current_stream(0).cuda_stream)
    rng = np.random.default_rng(1)
    for (nr, nc, ld) in ((1, 1, 1), (7, 5, 8), (129, 777, 784), (300, 1025, 1025)):
        A = rng.standard_normal((nr, ld)); x = rng.standard_normal(nc+1); b = rng.standard_normal(nr)
        Ad, xd, bd = torch.from_numpy(A).cuda(), torch.from_numpy(x).cuda(), torch.from_numpy(b).cuda()
        for off in (0, 1):
            xv = xd[off:off+nc]
            y = torch.empty(nr, dtype=torch.float64, device='cuda')
            torch.cuda.synchronize()
            ctx.gemv_axpby(Ad.data_ptr(), ld, nr, nc, xv.data_ptr(), -1., 1., bd.data_ptr(), y.data_ptr())
            ctx.synchronize()
            ref = b-A[:, :nc]@x[off:off+nc]
            assert np.abs(y.cpu().numpy()-ref).max() <= 1e-12*max(1., np.abs(ref).max())
            ctx.gemv_axpby(Ad.data_ptr(), ld, nr, nc, xv.data_ptr(), 2., 0., 0, y.data_ptr())
            ctx.synchronize()
            assert np.abs(y.cpu().numpy()-2*A[:, :nc]@x[off:off+nc]).max() <= 1e-12*max(1., np.abs(ref).max())
    S = sp.random(211, 97, density=0.05, random_state=3, format='csr')
    D = _DevCSR(S, torch.device('cuda', 0))
    x = rng.standard_normal(97); y0 = rng.standard_normal(211)
    y = torch.from_numpy(y0.copy()).cuda()
    torch.cuda.synchronize()
    D.matvec(ctx, torch.from_numpy(x).cuda(), alpha=0.5, beta=-2., y=y)
    ctx.synchronize()
    assert np.abs(y.cpu().numpy()-(0.5*S@x-2*y0)).max() <= 1e-13


@pytest.mark.gpu
def test_gpu_generic_cycle_equals_the_library_cycle_and_takes_h2_levels():
    """the operator-agnostic cycle (levels of any operator type with a device matvec) runs the same iteration as the library
    cycle on dense levels; with the H2 operator on the finest level (runFractional --matrixFormat H2 --solver cg-mg) the
    multigrid-preconditioned CG reproduces the reference's stored Hs error 0.059725648882225826 (disc, s = 0.75, noRef 5;
    compared there at relTol 1e-2)"""
    from pynucleus_amd.multigrid import multigrid, fractionalHierarchy
    H = device_hierarchy('disc', 3, 0.75, {})
    b = np.asarray(H.finest['DoFMap'].assembleRHS(1.0))
    lib, gen = multigrid(H), multigrid(H, native=False)
    assert lib._native and not gen._native
    x1, x2 = lib.cycle(b), gen.cycle(b)
    assert np.abs(x1-x2).max() <= 1e-13*np.abs(x1).max()
    x0 = np.random.default_rng(2).standard_normal(b.shape[0])
    assert np.abs(lib.cycle(b, x0)-gen.cycle(b, x0)).max() <= 1e-13*np.abs(x0).max()
    (xa, ia, ra), (xb, ib, rb) = lib.cg(b, tol=1e-10), gen.cg(b, tol=1e-10)
    assert ia == ib and np.abs(xa-xb).max() <= 1e-10*np.abs(xa).max()
    (xs, is_, rs), (xt, it, rt) = lib.solve(b, tol=1e-9, maxiter=40), gen.solve(b, tol=1e-9, maxiter=40)
    assert is_ == it and np.abs(xs-xt).max() <= 1e-10*np.abs(xs).max()
    # H2 on the finest level
    s = 0.75
    H2 = fractionalHierarchy('disc', 5, getFractionalKernel(2, s), {'target_order': 0.5}, matrixFormat='H2', h2MinDoFs=2000)
    from pynucleus_amd.h2 import H2Matrix
    assert isinstance(H2.finest['A'], H2Matrix) and H2.finest['DoFMap'].num_dofs == 2977
    b = np.asarray(H2.finest['DoFMap'].assembleRHS(1.0))
    mgl, mgg = multigrid(H2), multigrid(H2, native=False)
    # the H2 operator on the finest level runs INSIDE the library cycle (pnl_mg_level_desc.kind = 1: near-field CSR product + far
    # field of the operator set up in the context) and does what the operator-agnostic cycle does
    assert mgl._native and mgl._h2_top is H2.finest['A'] and not mgg._native
    c1, c2 = mgl.cycle(b), mgg.cycle(b)
    assert np.abs(c1-c2).max() <= 1e-12*np.abs(c2).max()
    x, its, res = mgl.cg(b, tol=1e-9)
    xg, itg, resg = mgg.cg(b, tol=1e-9)
    assert its == itg and np.abs(x-xg).max() <= 1e-9*np.abs(xg).max()
    (xs, is_, rs), (xt, it, rt) = mgl.solve(b, tol=1e-9, maxiter=40), mgg.solve(b, tol=1e-9, maxiter=40)
    assert is_ == it and np.abs(xs-xt).max() <= 1e-9*np.abs(xs).max() and len(rs) == is_+1
    C = 2.**(-2.*s)*gamma(1.)/gamma((2+2.*s)/2.)/gamma(1.+s)
    hs = np.sqrt(abs(b@x-C*np.pi/(s+1)))
    assert abs(hs-0.059725648882225826) <= 1e-2*0.059725648882225826, hs
    assert its <= 15, its


@pytest.mark.gpu
def test_gpu_fractional_heat_on_the_disc_reproduces_the_stored_errors():
    """runFractionalHeat --domain disc --s const(0.25) --problem constant --element P1 --solver cg-mg --matrixFormat dense
    (noRef 5, 2977 DoFs, Crank-Nicolson): stored errors 0.03181790573759944 / 0.07058538202611951, norm 1.489665512411283
    (compared by the reference at rTol 3e-2).  The norm is reproduced to 7e-6.  The error norms are differences of O(1)
    numbers, exactL2^2 - 2 z.u + u.Mu with z = int u_ss phi_i, and u_ss = C (1 - r^2)^(1/4) is singular at the boundary: the
    value of z there depends on the triangle rule (this package's degree-3 rule, not the reference's Xiao-Gimbutas table) at
    the level of the error itself -- observed 0.032798 / 0.072812, 3.1 % above the stored numbers"""
    from pynucleus_amd.multigrid import solveFractionalHeat
    s = 0.25
    H = device_hierarchy('disc', 5, s, {'target_order': 0.5}, mass=True)
    dm = H.finest['DoFMap']
    assert dm.num_dofs == 2977
    C = 2.**(-2.*s)*gamma(1.)/gamma((2+2.*s)/2.)/gamma(1.+s)
    L2ex2 = C**2*np.pi/(1+2*s)                              # nonlocalProblems.py:747

    def uss(x):
        return C*max(1.-x[0]**2-x[1]**2, 0.)**s
    qr = simplexXiaoGimbutas(3, 2, 2)
    z_ss, f_ss = np.asarray(dm.assembleRHS(uss, qr)), np.asarray(dm.assembleRHS(1.0, qr))
    times, us, stepper = solveFractionalHeat(H, uss, lambda t: -np.sin(t)*z_ss+np.cos(t)*f_ss, finalTime=1.0, tol=1e-10)
    M = H.finest['M']
    e_final, e_l2, norm = SO.transient_errors(us, times, M, lambda t: np.cos(t)*z_ss, lambda t: np.cos(t)**2*L2ex2)
    assert abs(norm-1.489665512411283) <= 2e-5*1.489665512411283, norm
    assert abs(e_final-0.03181790573759944) <= 5e-2*0.03181790573759944, e_final
    assert abs(e_l2-0.07058538202611951) <= 5e-2*0.07058538202611951, e_l2
    assert max(stepper.iterations) <= 10


def test_oracle_gmres_jacobi_reproduces_the_stored_nonsymmetric_run():
    """runFractional --domain interval --s constantNonSym(0.25) --problem constant --element P1 --solver gmres-jacobi
    --matrixFormat dense: stored Hs error 0.09611243700814974"""
    from pynucleus_amd import driverMesh
    from pynucleus_amd.fractionalOrders import constantNonSymFractionalOrder
    s = 0.25
    dm = P1_DoFMap(driverMesh('interval', 6), PHYSICAL)
    A = OracleProblem(nonlocalTables(dm, getFractionalKernel(1, constantNonSymFractionalOrder(s)), {'target_order': 5.})).get_dense()[0]
    b = np.asarray(dm.assembleRHS(1.0))
    dinv = 1./np.diag(A)
    x, its, res = SO.gmres(A, b, tol=1e-10, maxiter=100, B=lambda r: dinv*r)
    C = 2.**(-2.*s)*gamma(0.5)/gamma((1+2.*s)/2.)/gamma(1.+s)
    hs = np.sqrt(abs(b@x-C*np.sqrt(np.pi)*gamma(s+1)/gamma(s+3/2)))
    assert abs(hs-0.09611243700814974) <= 1e-8*0.09611243700814974, hs
    assert res[-1] < 1e-10 and its < 100 and np.abs(A@x-b).max() < 1e-8
    # right preconditioning and restarts reach the same solution
    x2 = SO.gmres(A, b, tol=1e-10, maxiter=20, restarts=20, B=lambda r: dinv*r, left=False)[0]
    assert np.abs(x2-x).max() <= 1e-7*np.abs(x).max()
    # BiCGStab (bicgstab_solver) on the same system
    x3, it3, res3 = SO.bicgstab(A, b, tol=1e-10, maxiter=200, B=lambda r: dinv*r)
    assert it3 < 200 and np.abs(x3-x).max() <= 1e-7*np.abs(x).max()


@pytest.mark.gpu
@pytest.mark.parametrize('order', ['constantNonSym', 'twoDomainNonSym'])
def test_gpu_gmres_on_a_nonsymmetric_operator_against_the_oracle(order):
    """solvers.gmres (Krylov basis in HBM) on the device-assembled operators of the non-symmetric code path (order per quadrature
    point): the same iteration as the oracle's, Jacobi and multigrid preconditioners (the drivers' gmres-jacobi / gmres-mg).
    constantNonSym(0.25): the stored Hs error 0.09611243700814974; twoDomainNonSym(0.25, 0.75): a genuinely non-symmetric matrix"""
    from pynucleus_amd.fractionalOrders import constantNonSymFractionalOrder, smoothedLeftRightFractionalOrder
    from pynucleus_amd.multigrid import fractionalHierarchy, multigrid
    from pynucleus_amd.solvers import gmres
    sF = constantNonSymFractionalOrder(0.25) if order == 'constantNonSym' else smoothedLeftRightFractionalOrder(0.25, 0.75)
    H = fractionalHierarchy('interval', 6, getFractionalKernel(1, sF), {'target_order': 5.})
    Aop = H.finest['A']
    A = Aop.toarray().copy()
    if order == 'twoDomainNonSym':
        assert np.abs(A-A.T).max() > 1e-3*np.abs(A).max()
    b = np.asarray(H.finest['DoFMap'].assembleRHS(1.0))
    dinv = 1./np.diag(A)
    x, its, res = gmres(Aop, b, tol=1e-10, maxiter=120, preconditioner='jacobi')
    xo, ito, reso = SO.gmres(A, b, tol=1e-10, maxiter=120, B=lambda r: dinv*r)
    assert its == ito and np.allclose(res, reso, rtol=1e-5, atol=1e-13)
    assert np.abs(x-xo).max() <= 1e-8*np.abs(xo).max()
    assert np.abs(A@x-b).max() <= 1e-7*np.abs(b).max()
    if order == 'constantNonSym':
        s = 0.25
        C = 2.**(-2.*s)*gamma(0.5)/gamma((1+2.*s)/2.)/gamma(1.+s)
        hs = np.sqrt(abs(b@x-C*np.sqrt(np.pi)*gamma(s+1)/gamma(s+3/2)))
        assert abs(hs-0.09611243700814974) <= 1e-8*0.09611243700814974, hs
    # BiCGStab with the Jacobi preconditioner: the oracle's iteration
    from pynucleus_amd.solvers import bicgstab
    xb, itb, resb = bicgstab(Aop, b, tol=1e-10, maxiter=200, preconditioner='jacobi')
    xbo, itbo, resbo = SO.bicgstab(A, b, tol=1e-10, maxiter=200, B=lambda r: dinv*r)
    assert abs(itb-itbo) <= 1 and itb < 200 and np.abs(xb-xbo).max() <= 1e-7*np.abs(xbo).max()
    assert np.abs(A@xb-b).max() <= 1e-7*np.abs(b).max()
    # gmres-mg: the V cycle of the hierarchy as preconditioner, far fewer iterations
    mg = multigrid(H)
    xm, itm, resm = gmres(Aop, b, tol=1e-10, maxiter=120, preconditioner=mg.asPreconditioner())
    mo = SO.Multigrid(as_oracle_levels(H))
    xmo, itmo, resmo = SO.gmres(A, b, tol=1e-10, maxiter=120, B=mo.precondition)
    assert itm == itmo and itm < its and np.abs(xm-xmo).max() <= 1e-8*np.abs(xmo).max()
    assert np.abs(A@xm-b).max() <= 1e-7*np.abs(b).max()


@pytest.mark.gpu
def test_gpu_cg_mg_at_scale_is_mesh_independent():
    """disc, s = 1/2, levels 0 .. 6 (12,097 DoFs on the finest): the multigrid-preconditioned CG needs the same handful of
    iterations as on the coarse hierarchies, agrees with Jacobi-CG, and one Crank-Nicolson step of the heat equation at that
    size conserves the discrete balance M (u1 - u0)/dt + S (u1 + u0)/2 = forcing"""
    import torch
    from pynucleus_amd.multigrid import multigrid, CrankNicolson
    H = device_hierarchy('disc', 6, 0.5, {'target_order': 0.5}, mass=True)
    L = H.finest
    dm, A = L['DoFMap'], L['A']
    assert dm.num_dofs == 12097
    b = np.asarray(dm.assembleRHS(1.0))
    mg = multigrid(H)
    x, its, res = mg.cg(b, tol=1e-9)
    assert its <= 8 and res[-1] <= 1e-9
    xj, itj, _ = A.solve_cg_jacobi(b, tol=1e-10, maxiter=3000)
    assert itj > 3*its and np.abs(x-xj).max() <= 1e-7*np.abs(xj).max()
    r = b-A*x
    assert np.abs(r).max() <= 1e-7*np.abs(b).max()
    # one theta step
    dt = 0.05
    st = CrankNicolson(H, dt, theta=0.5, tol=1e-11)
    u0 = np.asarray(dm.interpolate(lambda p: max(1.-p[0]**2-p[1]**2, 0.)**0.5))
    u = torch.from_numpy(u0.copy()).cuda()
    st.step(0., u, b)
    u1 = u.cpu().numpy()
    M = L['M']
    bal = M@(u1-u0)/dt+0.5*(A*(u1+u0))-b
    assert np.abs(bal).max() <= 1e-7*np.abs(b).max()
    assert st.iterations[0] <= 10


def test_oracle_chebyshev_smoother():
    """chebyshevSmoother (smoothers.pyx:390-457) in the oracle cycle: degree-3 polynomial on [rho/30, 1.1 rho]; the cycle is a
    contraction and cg-mg with it reaches the stored Hs error of the interval run"""
    s = 0.25
    levels = oracle_hierarchy('interval', 6, s, {'target_order': 2.-s})
    b = np.asarray(levels[-1]['DoFMap'].assembleRHS(1.0))
    mg = SO.Multigrid(levels, chebyshev={'degree': 3})
    x, its, res = mg.solve(b, tol=1e-9*np.linalg.norm(b), maxiter=60)
    assert its < 60 and max(res[k+1]/res[k] for k in range(2, len(res)-1)) < 0.6
    xc, itc, _ = SO.cg(levels[-1]['A'], b, tol=1e-10, B=mg.precondition)
    C = 2.**(-2.*s)*gamma(0.5)/gamma((1+2.*s)/2.)/gamma(1.+s)
    hs = np.sqrt(abs(b@xc-C*np.sqrt(np.pi)*gamma(s+1)/gamma(s+3/2)))
    assert abs(hs-0.09611243700804001) <= 1e-8*0.09611243700804001 and itc <= 10


@pytest.mark.gpu
def test_gpu_chebyshev_smoother_against_the_oracle():
    from pynucleus_amd.multigrid import multigrid
    H = device_hierarchy('disc', 3, 0.75, {})
    lv = as_oracle_levels(H)
    b = np.asarray(H.finest['DoFMap'].assembleRHS(1.0))
    mo = SO.Multigrid(lv, chebyshev={'degree': 3})
    mg = multigrid(H, smoother=('chebyshev', {'degree': 3}))
    assert not mg._native and 'Chebyshev' in str(mg)
    for l in range(1, len(lv)):
        assert np.allclose(mg._g[l]['cheb'], mo.cheb[l], rtol=1e-10)
    xo = np.zeros(b.shape[0]); mo.solveOnLevel(len(lv)-1, b, xo, True)
    assert np.abs(mg.cycle(b)-xo).max() <= 1e-11*np.abs(xo).max()
    xs, its, res = mg.solve(b, tol=1e-9*np.linalg.norm(b), maxiter=60)
    xso, itso, reso = mo.solve(b, tol=1e-9*np.linalg.norm(b), maxiter=60)
    assert its == itso and its < 60 and np.abs(xs-xso).max() <= 1e-9*np.abs(xso).max()
    xc, itc, _ = mg.cg(b, tol=1e-10)
    assert itc <= 12 and np.abs(lv[-1]['A']@xc-b).max() <= 1e-8*np.abs(b).max()
