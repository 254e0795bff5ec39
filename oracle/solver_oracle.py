"""TEST INFRASTRUCTURE -- CPU restatement (numpy, dense) of the solver side of the reference for SURVEY 8(f) row 4:
geometric multigrid on a hierarchy of assembled nonlocal operators, multigrid-preconditioned CG and the theta time
stepper of the fractional heat equation.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product (pynucleus_amd/) never does.

Parity: pinned by the reference's own stored numbers (tests/test_solver_side.py):
  tests/cache_runFractionalHeat.py--domaininterval--sconst(0.25)--problemconstant--elementP1--solvercg-mg--matrixFormatdense
  tests/cache_runFractionalHeat.py--domaininterval--sconst(0.25)--problemknownSolution--elementP1--solvercg-jacobi--matrixFormatH2
  tests/cache_runFractionalHeat.py--domaininterval--sconst(0.75)--problemknownSolution--elementP1--solvercg-mg--matrixFormatH2
  tests/cache_runFractional.py--domaininterval--sconst(0.25)--problemconstant--elementP1--solvercg-mg--matrixFormatdense

Reference routines restated (file:line):
  restriction_1D_P1.pxi / restriction_2D_P1.pxi:8-72       buildRestriction_{1,2}D_P1 (children 2c, 2c+1 / 4c .. 4c+3)
  multigrid_{SCALAR}.pxi:237-292                            multigrid.solveOnLevel
  multigrid_{SCALAR}.pxi:296-390                            multigrid.solve
  smoothers_{SCALAR}.pxi:88-108, 118-131                    separableSmoother.eval, jacobiPreconditioner
  base/PyNucleus_base/solvers.pyx:363-444                   cg_solver.solve
  base/PyNucleus_base/timestepping.py:64-112                CrankNicolson.setRHS / step
  nl/PyNucleus_nl/discretizedProblems.py:740-785            buildTransientHierarchy, determineTimeSteps
  nl/PyNucleus_nl/discretizedProblems.py:276-333            L2 errors of a transient solution
  nl/PyNucleus_nl/nonlocalProblems.py:651-662, 710-727, 1641-1672   the 'constant' / 'knownSolution' problems, transient version
"""
import numpy as np


def build_restriction_P1(dm_coarse, dm_fine):
    """R[coarse dof, fine dof]: 1 at the coarse vertex, 1/2 at the midpoints of its edges.  Walks the coarse cells and
    their children like the reference does (the refinement numbers the children of cell c 2c, 2c+1 in 1D:
    (c0, m), (m, c1); 4c .. 4c+3 in 2D: (c0, m01, m02), (c1, m12, m01), (c2, m02, m12), (m01, m12, m02))."""
    dim = dm_coarse.mesh.manifold_dim
    R = np.zeros((dm_coarse.num_dofs, dm_fine.num_dofs))

    def enter(I, J, v):
        if I >= 0 and J >= 0:
            R[I, J] = v
    for c in range(dm_coarse.mesh.num_cells):
        if dim == 1:
            s0, s1 = 2*c, 2*c+1
            enter(dm_coarse.dofs[c, 0], dm_fine.dofs[s0, 0], 1.0)
            enter(dm_coarse.dofs[c, 0], dm_fine.dofs[s0, 1], 0.5)
            enter(dm_coarse.dofs[c, 1], dm_fine.dofs[s0, 1], 0.5)
            enter(dm_coarse.dofs[c, 1], dm_fine.dofs[s1, 1], 1.0)
        else:
            s0, s1, s2 = 4*c, 4*c+1, 4*c+2
            d0, d1, d2 = dm_coarse.dofs[c]
            enter(d0, dm_fine.dofs[s0, 0], 1.0); enter(d0, dm_fine.dofs[s0, 1], 0.5); enter(d0, dm_fine.dofs[s0, 2], 0.5)
            enter(d1, dm_fine.dofs[s0, 1], 0.5); enter(d1, dm_fine.dofs[s1, 0], 1.0); enter(d1, dm_fine.dofs[s1, 1], 0.5)
            enter(d2, dm_fine.dofs[s0, 2], 0.5); enter(d2, dm_fine.dofs[s1, 1], 0.5); enter(d2, dm_fine.dofs[s2, 0], 1.0)
    return R


# buildRestriction_1D_P2 (restriction_1D_P2.pxi:44-62) and buildRestriction_2D_P2 (restriction_2D_P2.pxi:84-134) as tables:
# coarse local DoF -> [(child, fine local DoF, weight)]
_R_P2 = {
    1: {0: [(0, 0, 1.0), (0, 2, 0.375), (1, 2, -0.125)],
        1: [(0, 2, -0.125), (1, 1, 1.0), (1, 2, 0.375)],
        2: [(0, 1, 1.0), (0, 2, 0.75), (1, 2, 0.75)]},
    2: {0: [(0, 0, 1.0), (0, 3, 0.375), (0, 5, 0.375), (1, 4, -0.125), (1, 5, -0.125), (2, 3, -0.125), (2, 4, -0.125)],
        1: [(0, 3, -0.125), (0, 4, -0.125), (1, 0, 1.0), (1, 3, 0.375), (1, 5, 0.375), (2, 4, -0.125), (2, 5, -0.125)],
        2: [(0, 4, -0.125), (0, 5, -0.125), (1, 3, -0.125), (1, 4, -0.125), (2, 0, 1.0), (2, 3, 0.375), (2, 5, 0.375)],
        3: [(0, 1, 1.0), (0, 3, 0.75), (0, 4, 0.5), (1, 4, 0.5), (1, 5, 0.75), (2, 4, 0.25)],
        4: [(0, 4, 0.25), (1, 1, 1.0), (1, 3, 0.75), (1, 4, 0.5), (2, 4, 0.5), (2, 5, 0.75)],
        5: [(0, 2, 1.0), (0, 4, 0.5), (0, 5, 0.75), (1, 4, 0.25), (2, 3, 0.75), (2, 4, 0.5)]},
}


def build_restriction_P2(dm_coarse, dm_fine):
    dim = dm_coarse.mesh.manifold_dim
    nchild = 2 if dim == 1 else 4
    R = np.zeros((dm_coarse.num_dofs, dm_fine.num_dofs))
    for c in range(dm_coarse.mesh.num_cells):
        for loc, entries in _R_P2[dim].items():
            I = dm_coarse.dofs[c, loc]
            if I < 0:
                continue
            for child, floc, w in entries:
                J = dm_fine.dofs[nchild*c+child, floc]
                if J >= 0:
                    R[I, J] = w
    return R


# buildRestriction_1D_P3 (restriction_1D_P3.pxi:50-76): coarse local DoF -> [(child, fine local DoF, weight)]
_R_P3_1D = {0: [(0, 0, 1.0), (0, 1, -0.0625), (0, 2, 0.3125), (1, 3, 0.0625)],
            1: [(0, 1, -0.0625), (0, 2, 0.0625), (1, 1, 1.0), (1, 3, 0.3125)],
            2: [(0, 1, 0.5625), (0, 2, 0.9375), (0, 3, 1.0), (1, 3, -0.3125)],
            3: [(0, 1, 0.5625), (0, 2, -0.3125), (1, 2, 1.0), (1, 3, 0.9375)]}


def build_restriction_P3(dm_coarse, dm_fine):
    assert dm_coarse.mesh.manifold_dim == 1
    R = np.zeros((dm_coarse.num_dofs, dm_fine.num_dofs))
    for c in range(dm_coarse.mesh.num_cells):
        for loc, entries in _R_P3_1D.items():
            I = dm_coarse.dofs[c, loc]
            if I < 0:
                continue
            for child, floc, w in entries:
                J = dm_fine.dofs[2*c+child, floc]
                if J >= 0:
                    R[I, J] = w
    return R


def build_restriction_P0(dm_coarse, dm_fine):
    """buildRestriction_{1,2}D_P0 (restriction_1D_P0.pxi:18-38, restriction_2D_P0.pxi): weight 1 to every child of the cell"""
    nchild = 2 if dm_coarse.mesh.manifold_dim == 1 else 4
    R = np.zeros((dm_coarse.num_dofs, dm_fine.num_dofs))
    for c in range(dm_coarse.mesh.num_cells):
        I = dm_coarse.dofs[c, 0]
        if I >= 0:
            for child in range(nchild):
                R[I, dm_fine.dofs[nchild*c+child, 0]] = 1.0
    return R


class Multigrid:
    """levels[l] = {'A': dense operator, 'R': restriction to level l-1, 'P': prolongation from it}; level 0 = coarsest."""

    def __init__(self, levels, omega=2./3., presmoothingSteps=1, postsmoothingSteps=1, chebyshev=None):
        """chebyshev: None (Jacobi smoother) or the parameters of chebyshevSmoother (smoothers.pyx:439-457): degree, lowerBound,
        upperBound, rhoA (0: power method linalg.pyx:811-829 from the vector default_rng(0).standard_normal(n), relative change 1e-3, at
        most 51 steps -- the reference's absolute tolerance 0.01 stops after two steps on these operators and underestimates rho)"""
        self.levels = levels
        self.invD = [None]+[omega/np.diag(L['A']) for L in levels[1:]]
        self.pre, self.post = presmoothingSteps, postsmoothingSteps
        self.coarse_inverse = np.linalg.inv(levels[0]['A'])
        self.cheb = None
        if chebyshev is not None:
            self.cheb = [None]+[self.chebyshev_coefficients(L['A'], **chebyshev) for L in levels[1:]]

    @staticmethod
    def chebyshev_coefficients(A, degree=3, rhoA=0., lowerBound=1.0/30.0, upperBound=1.1):
        if rhoA == 0.:
            n = A.shape[0]
            x = np.random.default_rng(0).standard_normal(n)
            x /= np.linalg.norm(x)
            lold, rhoA, k = 0., 1., 0
            while abs(rhoA-lold) > 1e-3*abs(rhoA) and k <= 50:
                x = A@x
                lold = rhoA
                rhoA = np.linalg.norm(x)
                x /= rhoA
                k += 1
        a, b = rhoA*lowerBound, rhoA*upperBound
        roots = 0.5*(b-a)*(1+np.cos(np.pi*(np.arange(degree)+0.5)/degree))+a
        poly = np.poly(roots)
        poly /= np.polyval(poly, 0.)
        return -poly[:-1]

    def smooth(self, l, b, x, steps, simple):
        A = self.levels[l]['A']
        for _ in range(steps):
            res = b.copy() if simple else b-A@x
            simple = False
            if self.cheb is not None:
                c = self.cheb[l]
                y = c[0]*res
                for ck in c[1:]:
                    y = ck*res+A@y
                x += y
            else:
                x += self.invD[l]*res

    def solveOnLevel(self, l, b, x, simple=False):
        if l == 0:
            x[:] = self.coarse_inverse@b
            return
        L = self.levels[l]
        self.smooth(l, b, x, self.pre, simple)
        res = b-L['A']@x
        defect = L['R']@res
        solcg = np.zeros(self.levels[l-1]['A'].shape[0])
        self.solveOnLevel(l-1, defect, solcg, True)
        x += L['P']@solcg
        self.smooth(l, b, x, self.post, False)

    def solve(self, b, x=None, tol=1e-8, maxiter=50):
        simple = x is None
        x = np.zeros_like(b) if x is None else x.copy()
        A = self.levels[-1]['A']
        residuals = [np.linalg.norm(b if simple else b-A@x)]
        it = 0
        while residuals[-1] > tol and it < maxiter:
            it += 1
            self.solveOnLevel(len(self.levels)-1, b, x, simple)
            simple = False
            residuals.append(np.linalg.norm(b-A@x))
        return x, it, residuals

    def precondition(self, r):
        z = np.zeros_like(r)
        self.solveOnLevel(len(self.levels)-1, r, z, True)
        return z


def cg(A, b, x0=None, tol=1e-8, maxiter=1000, B=None):
    """cg_solver.solve: convergence on sqrt(r.Br), residual recomputed every 50 iterations."""
    B = B or (lambda r: r.copy())
    x = np.zeros_like(b) if x0 is None else x0.copy()
    r = b.copy() if x0 is None else b-A@x
    p = B(r)
    betaOld = r@p
    residuals = [np.sqrt(abs(betaOld))]
    its = 0
    if residuals[-1] > tol:
        k = 0
        its = maxiter
        for i in range(maxiter):
            Ap = A@p
            alpha = betaOld/(p@Ap)
            x += alpha*p
            r -= alpha*Ap
            if k == 50:
                r = b-A@x
                k = 0
            Br = B(r)
            beta = r@Br
            residuals.append(np.sqrt(abs(beta)))
            its = i
            if residuals[-1] <= tol:
                break
            p = Br+(beta/betaOld)*p
            betaOld = beta
            k += 1
            if i == maxiter-1:
                its = maxiter
    return x, its, residuals


def theta_step(S, M, dt, theta, forcing, u, solve):
    """CrankNicolson.step: (M/dt + theta S) u_new = M u / dt - (1 - theta) S u + forcing"""
    rhs = M@u/dt-(1.-theta)*(S@u)+forcing
    return solve(rhs, u)


def heat_time_steps(h, finalTime=1.0, timeStepperType='Crank-Nicolson'):
    """determineTimeSteps (discretizedProblems.py:774-783)"""
    dt = np.sqrt(h) if timeStepperType == 'Crank-Nicolson' else h
    n = int(np.around(finalTime/dt))
    return finalTime/n, n


def transient_errors(us, times, M, z_of_t, exactL2Squared_of_t):
    """(final L2 error, L2(0,T;L2) error, L2(0,T;L2) norm) as discretizedProblems.py:276-333 computes them"""
    nt = len(times)-1

    def fac(k):
        if k == 0:
            return times[1]-times[0]
        if k == nt:
            return times[k]-times[k-1]
        return times[k+1]-times[k-1]

    def err2(k):
        return abs(exactL2Squared_of_t(times[k])-2*(z_of_t(times[k])@us[k])+us[k]@(M@us[k]))
    return (np.sqrt(err2(nt)), np.sqrt(sum(fac(k)*err2(k) for k in range(nt+1))),
            np.sqrt(sum(fac(k)*abs(us[k]@(M@us[k])) for k in range(nt+1))))


def gmres(A, b, x0=None, tol=1e-8, maxiter=50, restarts=1, B=None, left=True):
    """gmres_solver.solve (base/PyNucleus_base/solvers.pyx:504-659), dense numpy"""
    x = np.zeros_like(b) if x0 is None else x0.copy()
    L = B if left else None
    R = B if not left else None
    n = b.shape[0]
    Q = np.zeros((maxiter+1, n))
    H = np.zeros((maxiter+1, maxiter))
    c, sn, gamma, y = np.zeros(maxiter), np.zeros(maxiter), np.zeros(maxiter+1), np.zeros(maxiter+1)
    residuals, allIter, breakout = [], 0, False
    for _ in range(restarts):
        if breakout:
            break
        r = b-A@x
        if L is not None:
            r = L(r)
        gamma[0] = np.linalg.norm(r)
        if not residuals:
            residuals.append(abs(gamma[0]))
        if abs(gamma[0]) < tol:
            break
        Q[0] = r/gamma[0]
        i = -1
        for i in range(maxiter):
            w = L(A@Q[i]) if L is not None else (A@R(Q[i]) if R is not None else A@Q[i])
            w = np.array(w, copy=True)
            for j in range(i+1):
                H[j, i] = Q[j]@w
                w -= H[j, i]*Q[j]
            H[i+1, i] = np.linalg.norm(w)
            if not abs(H[i+1, i]) > 1e-15:
                breakout = True
                break
            Q[i+1] = w/H[i+1, i]
            for j in range(i):
                rho, sigma = H[j, i], H[j+1, i]
                H[j, i] = c[j]*rho+sn[j]*sigma
                H[j+1, i] = -sn[j]*rho+c[j]*sigma
            beta = np.sqrt(H[i, i]**2+H[i+1, i]**2)
            c[i], sn[i] = H[i, i]/beta, H[i+1, i]/beta
            H[i, i] = beta
            gamma[i+1] = -sn[i]*gamma[i]
            gamma[i] = c[i]*gamma[i]
            residuals.append(abs(gamma[i+1]))
            if abs(gamma[i+1]) < tol:
                breakout = True
                break
        allIter += i
        for j in range(i, -1, -1):
            t = gamma[j]
            for l in range(j+1, i+1):
                t -= H[j, l]*y[l]
            y[j] = t/H[j, j]
        upd = y[:i+1]@Q[:i+1]
        x += R(upd) if R is not None else upd
    return x, allIter, residuals


def bicgstab(A, b, x0=None, tol=1e-8, maxiter=50, B=None):
    """bicgstab_solver.solve (base/PyNucleus_base/solvers.pyx:716-787), dense numpy"""
    x = np.zeros_like(b) if x0 is None else x0.copy()
    r = b.copy() if x0 is None else b-A@x
    p = r.copy()
    r0 = B(r) if B is not None else r.copy()
    kappa = r@r0
    residuals = [np.sqrt(abs(kappa))]
    for k in range(maxiter):
        p2 = B(p) if B is not None else p
        temp = A@p2
        alpha = kappa/(temp@r0)
        s = r-alpha*temp
        s2 = B(s) if B is not None else s
        temp2 = A@s2
        omega = (temp2@s)/(temp2@temp2)
        x = x+alpha*p2+omega*s2
        r = s-omega*temp2
        residuals.append(np.linalg.norm(r))
        if residuals[-1] < tol:
            return x, k, residuals
        kappaNew = r@r0
        beta = kappaNew/kappa*alpha/omega
        kappa = kappaNew
        p = r+beta*(p-omega*temp)
    return x, maxiter, residuals
