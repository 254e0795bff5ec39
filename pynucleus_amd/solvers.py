"""Krylov solvers for the operators of this package (dense, CSR / SSS, H2, distributed): vectors stay in HBM, the operator
is whatever ``A.matvec`` does (GEMV, SpMV, H2 passes, all-reduce).

``cg`` follows the reference's cg_solver.solve (base/PyNucleus_base/solvers.pyx:363-444) step by step: preconditioned
residual norm sqrt(r.Br) as convergence criterion, the residual recomputed from scratch every 50 iterations, update order
x, r, (refresh), Br, beta, p.  ``jacobi`` = the reference's jacobi_solver as a preconditioner (solvers.pyx:229-245).
The dense operator additionally has this loop as one library call (pnl_cg_jacobi, Dense_LinearOperator.solve_cg_jacobi)."""
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
