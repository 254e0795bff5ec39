"""Krylov solvers for the operators of this package (dense, CSR / SSS, H2, distributed): vectors stay in HBM, the operator
is whatever ``A.matvec`` does (GEMV, SpMV, H2 passes, all-reduce).

``cg`` follows the reference's cg_solver.solve (base/PyNucleus_base/solvers.pyx:363-444) step by step: preconditioned
residual norm sqrt(r.Br) as convergence criterion, the residual recomputed from scratch every 50 iterations, update order
x, r, (refresh), Br, beta, p.  ``jacobi`` = the reference's jacobi_solver as a preconditioner (solvers.pyx:229-245).
The dense operator additionally has this loop as one library call (pnl_cg_jacobi, Dense_LinearOperator.solve_cg_jacobi).
``gmres`` follows gmres_solver.solve (solvers.pyx:504-659; the drivers' gmres-jacobi / gmres-mg for non-symmetric orders):
left (default) or right preconditioner, modified Gram-Schmidt, Givens rotations, the rotated right-hand side as residual
estimate, restarts; the Krylov basis stays in HBM, the small Hessenberg problem lives on the host."""
import numpy as np


def _dev_vector(v, device):
    import torch
    if isinstance(v, torch.Tensor):
        return v.to(device=device, dtype=torch.float64)
    return torch.from_numpy(np.ascontiguousarray(np.asarray(v, dtype=np.float64))).to(device)


def cg(A, b, x0=None, tol=1e-8, maxiter=1000, preconditioner='jacobi'):
    """Solve A x = b for a symmetric positive definite operator.  Returns (x, iterations, residuals); x is a torch tensor
    on the operator's device if b is one, else a numpy array."""
    import torch
    device = getattr(A, 'device', None)
    if device is None:
        device = A.A.device if hasattr(A, 'A') else torch.device('cuda', torch.cuda.current_device())
    bd = _dev_vector(b, device)
    x = torch.zeros_like(bd) if x0 is None else _dev_vector(x0, device).clone()
    dinv = None
    B = None
    if preconditioner == 'jacobi':
        d = A.diagonal
        d = d() if callable(d) else d
        dinv = 1./_dev_vector(d, device)
    elif callable(preconditioner):
        # r -> B r on device vectors, e.g. multigrid.asPreconditioner() (multigridPreconditioner, multigrid_{SCALAR}.pxi:470-497)
        B = preconditioner
    elif preconditioner is not None:
        raise NotImplementedError(preconditioner)
    r = bd-A.matvec(x) if x0 is not None else bd.clone()
    residuals = []
    if B is not None:
        p = B(r).clone()
        betaOld = float(torch.dot(r, p))
    elif dinv is None:
        p = r.clone()
        betaOld = float(torch.dot(r, p))
    else:
        p = dinv*r
        betaOld = float(torch.dot(r, p))
    conv = float(np.sqrt(abs(betaOld)))
    residuals.append(conv)
    its = 0
    if conv > tol:
        k = 0
        for i in range(maxiter):
            Ap = A.matvec(p)
            alpha = betaOld/float(torch.dot(p, Ap))
            x.add_(p, alpha=alpha)
            r.add_(Ap, alpha=-alpha)
            if k == 50:
                r = bd-A.matvec(x)                      # recalculate the residual to avoid rounding errors
                k = 0
            Br = B(r) if B is not None else (r if dinv is None else dinv*r)
            beta = float(torch.dot(r, Br))
            conv = float(np.sqrt(abs(beta)))
            residuals.append(conv)
            its = i
            if conv <= tol:
                break
            p = Br+(beta/betaOld)*p
            betaOld = beta
            k += 1
        else:
            its = maxiter
    import torch as _t
    if isinstance(b, _t.Tensor):
        return x, its, residuals
    return x.cpu().numpy(), its, residuals


def gmres(A, b, x0=None, tol=1e-8, maxiter=50, restarts=1, preconditioner=None, left=True):
    """Solve A x = b for a general operator.  ``preconditioner``: 'jacobi', None or a callable r -> B r on device vectors
    (multigrid.asPreconditioner()).  Returns (x, iterations, residuals): the residuals are the norms of the (left-
    preconditioned) residual, the first one computed, the others from the rotated right-hand side like the reference."""
    import torch
    device = getattr(A, 'device', None)
    if device is None:
        device = A.A.device if hasattr(A, 'A') else torch.device('cuda', torch.cuda.current_device())
    bd = _dev_vector(b, device)
    x = torch.zeros_like(bd) if x0 is None else _dev_vector(x0, device).clone()
    B = None
    if preconditioner == 'jacobi':
        d = A.diagonal
        d = d() if callable(d) else d
        dinv = 1./_dev_vector(d, device)
        B = lambda r: dinv*r                                   # noqa: E731
    elif callable(preconditioner):
        B = preconditioner
    elif preconditioner is not None:
        raise NotImplementedError(preconditioner)
    L = B if left else None
    R = B if not left else None
    n = bd.shape[0]
    Q = torch.empty((maxiter+1, n), dtype=torch.float64, device=device)
    H = np.zeros((maxiter+1, maxiter))
    c, sn, gamma, y = np.zeros(maxiter), np.zeros(maxiter), np.zeros(maxiter+1), np.zeros(maxiter+1)
    residuals, allIter, breakout, eps = [], 0, False, 1e-15
    for _ in range(restarts):
        if breakout:
            break
        r = bd-A.matvec(x)
        if L is not None:
            r = L(r)
        gamma[0] = float(torch.linalg.norm(r))
        if not residuals:
            residuals.append(abs(gamma[0]))
        if abs(gamma[0]) < tol:
            break
        Q[0] = r/gamma[0]
        i = -1
        for i in range(maxiter):
            # Arnoldi step
            if L is not None:
                w = L(A.matvec(Q[i].contiguous()))
            elif R is not None:
                w = A.matvec(R(Q[i].contiguous()))
            else:
                w = A.matvec(Q[i].contiguous())
            w = w.clone()
            for j in range(i+1):
                H[j, i] = float(torch.dot(Q[j], w))
                w.add_(Q[j], alpha=-H[j, i])
            H[i+1, i] = float(torch.linalg.norm(w))
            if not abs(H[i+1, i]) > eps:
                breakout = True
                break
            Q[i+1] = w/H[i+1, i]
            # previous Givens rotations on the new column, then the new rotation
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
        # back substitution and update
        for j in range(i, -1, -1):
            t = gamma[j]
            for l in range(j+1, i+1):
                t -= H[j, l]*y[l]
            y[j] = t/H[j, j]
        upd = torch.from_numpy(y[:i+1].copy()).to(device)@Q[:i+1]
        x.add_(R(upd) if R is not None else upd)
    if isinstance(b, torch.Tensor):
        return x, allIter, residuals
    return x.cpu().numpy(), allIter, residuals


def bicgstab(A, b, x0=None, tol=1e-8, maxiter=50, preconditioner=None):
    """bicgstab_solver.solve (solvers.pyx:716-787): right-preconditioned stabilised BiCG, r0 = B r, stopping on the 2-norm of
    the residual.  Returns (x, iterations, residuals)."""
    import torch
    device = getattr(A, 'device', None)
    if device is None:
        device = A.A.device if hasattr(A, 'A') else torch.device('cuda', torch.cuda.current_device())
    bd = _dev_vector(b, device)
    x = torch.zeros_like(bd) if x0 is None else _dev_vector(x0, device).clone()
    B = None
    if preconditioner == 'jacobi':
        d = A.diagonal
        d = d() if callable(d) else d
        dinv = 1./_dev_vector(d, device)
        B = lambda r: dinv*r                                   # noqa: E731
    elif callable(preconditioner):
        B = preconditioner
    elif preconditioner is not None:
        raise NotImplementedError(preconditioner)
    r = bd.clone() if x0 is None else bd-A.matvec(x)
    p = r.clone()
    r0 = B(r).clone() if B is not None else r.clone()
    kappa = float(torch.dot(r, r0))
    residuals = [float(np.sqrt(abs(kappa)))]
    its = maxiter
    for k in range(maxiter):
        p2 = B(p) if B is not None else p
        temp = A.matvec(p2.contiguous())
        alpha = kappa/float(torch.dot(temp, r0))
        s_ = r-alpha*temp
        s2 = B(s_) if B is not None else s_
        temp2 = A.matvec(s2.contiguous())
        omega = float(torch.dot(temp2, s_))/float(torch.dot(temp2, temp2))
        x = x+alpha*p2+omega*s2
        r = s_-omega*temp2
        residuals.append(float(torch.linalg.norm(r)))
        if residuals[-1] < tol:
            its = k
            break
        kappaNew = float(torch.dot(r, r0))
        beta = kappaNew/kappa*alpha/omega
        kappa = kappaNew
        p = r+beta*(p-omega*temp)
    if isinstance(b, torch.Tensor):
        return x, its, residuals
    return x.cpu().numpy(), its, residuals
