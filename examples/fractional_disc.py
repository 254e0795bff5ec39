#!/usr/bin/env python3
"""The reference's runFractional driver in miniature (drivers/runFractional.py --domain disc --s const(0.75) --problem
constant): assemble the fractional Laplacian on the disc as a dense operator and as an H2 operator on the GPU, solve
(-Delta)^s u = 1 with Jacobi-CG and compare with the analytic energy (nonlocalProblems.py:742-748).

    python examples/fractional_disc.py [noRef] [s]
"""
import sys
import time
from math import gamma, pi
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from pynucleus_amd import driverMesh, P1_DoFMap, PHYSICAL, getFractionalKernel, nonlocalBuilder  # noqa: E402
from pynucleus_amd.solvers import cg  # noqa: E402

noRef = int(sys.argv[1]) if len(sys.argv) > 1 else 5
s = float(sys.argv[2]) if len(sys.argv) > 2 else 0.75
mesh = driverMesh('disc', noRef)
dm = P1_DoFMap(mesh, PHYSICAL)
kernel = getFractionalKernel(2, s)
builder = nonlocalBuilder(dm, kernel, {'target_order': 0.5, 'eta': 3.}, zeroExterior=True)
b = np.asarray(dm.assembleRHS(1.0))
C = 2.**(-2.*s)*gamma(1.)/gamma((2+2.*s)/2.)/gamma(1.+s)
exactHsSquared = C*pi/(s+1)

t0 = time.time()
A = builder.getDense()
u, its, res = A.solve_cg_jacobi(b, tol=1e-8, maxiter=5000)
t1 = time.time()
print('dense: N = {}, {} element pairs, assembly + CG ({} iterations) {:.2f} s, Hs error {:.4e}'.format(
    dm.num_dofs, A.info['counters']['numAssembledCellPairs'], its, t1-t0, np.sqrt(abs(b@u-exactHsSquared))))

t0 = time.time()
H = builder.getH2()
uh, its, res = cg(H, b, tol=1e-8, maxiter=5000)
t1 = time.time()
print('{}: tree + near field + far field + CG ({} iterations) {:.2f} s, Hs error {:.4e}, |u_h2 - u_dense| / |u_dense| = {:.2e}'.format(
    H, its, t1-t0, np.sqrt(abs(b@uh-exactHsSquared)), np.linalg.norm(uh-u)/np.linalg.norm(u)))
