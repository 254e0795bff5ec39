#!/usr/bin/env python3
"""The reference's runFractionalHeat driver in miniature (drivers/runFractionalHeat.py --domain interval|disc --s const(s)
--problem constant --element P1 --solver cg-mg --matrixFormat dense): u_t + (-Delta)^s u = f on the GPU.

One nonlocal operator per refinement level (assembled on the device), Crank-Nicolson with dt = sqrt(h), every step solved by
multigrid-preconditioned CG on M/dt + theta S inside libpnl_hip.so (pnl_theta_step).  The problem is the reference's transient
'constant' problem (nonlocalProblems.py:1641-1672): u(t, x) = cos(t) u_ss(x) with u_ss = C (1 - |x|^2)_+^s, the solution of
(-Delta)^s u = 1, and f = -sin(t) u_ss + cos(t); errors as discretizedProblems.py:276-333 reports them.

    python examples/fractional_heat.py [interval|disc] [noRef] [s]
"""
import sys
import time
from math import gamma, pi
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from pynucleus_amd import getFractionalKernel  # noqa: E402
from pynucleus_amd.multigrid import fractionalHierarchy, solveFractionalHeat  # noqa: E402
from pynucleus_amd.quadrature import simplexXiaoGimbutas  # noqa: E402

domain = sys.argv[1] if len(sys.argv) > 1 else 'interval'
dim = 1 if domain == 'interval' else 2
noRef = int(sys.argv[2]) if len(sys.argv) > 2 else (6 if dim == 1 else 5)
s = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25
params = {'target_order': 2.-s} if dim == 1 else {'target_order': 0.5}

t0 = time.time()
H = fractionalHierarchy(domain, noRef, getFractionalKernel(dim, s), params, buildMass=True)
t1 = time.time()
dm = H.finest['DoFMap']
C = 2.**(-2.*s)*gamma(dim/2.)/gamma((dim+2.*s)/2.)/gamma(1.+s)
# |u_ss|^2_{L2}: interval nonlocalProblems.py:661, disc :747
L2ex2 = C**2*np.sqrt(pi)*gamma(1+2*s)/gamma(1.5+2*s) if dim == 1 else C**2*pi/(2*s+1)


def uss(x):
    return C*max(1.-float(np.dot(x, x)), 0.)**s


qr = simplexXiaoGimbutas(3, dim, dim)
z_ss, f_ss = np.asarray(dm.assembleRHS(uss, qr)), np.asarray(dm.assembleRHS(1.0, qr))
times, us, stepper = solveFractionalHeat(H, uss, lambda t: -np.sin(t)*z_ss+np.cos(t)*f_ss, finalTime=1.0, tol=1e-10)
t2 = time.time()
M = H.finest['M']
nt = len(times)-1


def err2(k):
    return abs(np.cos(times[k])**2*L2ex2-2*np.cos(times[k])*(z_ss@us[k])+us[k]@(M@us[k]))


def fac(k):
    return times[1]-times[0] if k == 0 else (times[k]-times[k-1] if k == nt else times[k+1]-times[k-1])


print('{}: levels {} DoFs, hierarchy assembled in {:.2f} s; {} Crank-Nicolson steps (dt = {:.4f}) in {:.2f} s, cg-mg iterations per step {}'.format(
    domain, [L['DoFMap'].num_dofs for L in H.getLevelList()], t1-t0, nt, times[1]-times[0], t2-t1, stepper.iterations))
print('L^2(Omega) error at t=finalTime: {:.6e}'.format(np.sqrt(err2(nt))))
print('L^2(0,T; L^2(Omega)) error:     {:.6e}'.format(np.sqrt(sum(fac(k)*err2(k) for k in range(nt+1)))))
print('L^2(0,T; L^2(Omega)) norm:      {:.10f}'.format(np.sqrt(sum(fac(k)*abs(us[k]@(M@us[k])) for k in range(nt+1)))))
if domain == 'interval' and noRef == 6 and s == 0.25:
    print('reference (tests/cache_runFractionalHeat.py--domaininterval--sconst(0.25)--problemconstant--elementP1--solvercg-mg--'
          'matrixFormatdense): 0.01455872345929613, 0.03218338586612875, 1.7018299503210628')
