#!/usr/bin/env python3
"""Generate fully symmetric, positive, interior quadrature rules on the triangle.

The reference takes its triangle rules from modepy's Xiao-Gimbutas tables (an
un-vendored dependency that is not installed here).  This script computes rules
with the same defining properties (fully symmetric, all weights positive, all
nodes strictly inside, exact to a given total degree) by solving the moment
equations in an orthogonal (Dubiner) basis with Levenberg-Marquardt from random
starts, and writes them to pynucleus_amd/data/triangle_rules.json.  Each rule is
verified against all moments of its degree to 2e-15 before it is stored.

Usage: python tools/gen_triangle_rules.py [max_degree] [restarts]
"""
import json
import os
import sys
import time
import numpy as np
from scipy.optimize import least_squares
from scipy.special import eval_jacobi
from multiprocessing import Pool

# orbit structures [n0, n1, n2] (centroid, (a,a,1-2a), (a,b,1-a-b)); point count n0+3n1+6n2
# candidates per degree, tried in order (smallest first)
STRUCTS = {
    2: [(0, 1, 0)],
    3: [(0, 2, 0)],
    4: [(0, 2, 0)],
    5: [(1, 2, 0)],
    6: [(0, 2, 1)],
    7: [(0, 3, 1), (1, 3, 1)],
    8: [(1, 3, 1)],
    9: [(1, 4, 1)],
    10: [(1, 2, 3), (1, 4, 2)],
    11: [(1, 5, 2), (1, 3, 3)],
    12: [(0, 5, 3)],
    13: [(1, 6, 3)],
    14: [(0, 6, 4)],
    15: [(1, 6, 5), (0, 7, 5)],
    16: [(1, 7, 5), (1, 6, 6)],
    17: [(0, 8, 6), (1, 8, 6)],
    18: [(1, 6, 8), (1, 9, 7)],
    19: [(1, 8, 8)],
    20: [(1, 10, 8), (1, 8, 9)],
}


def dubiner(deg, l):
    """orthogonal basis on the triangle evaluated at barycentric points l[3, n]; rows = basis fns.
    Reference triangle (0,0),(1,0),(0,1), x = l[1], y = l[2]."""
    x, y = l[1], l[2]
    out = []
    with np.errstate(divide='ignore', invalid='ignore'):
        xi = np.where(np.abs(1-y) > 1e-300, 2*x/(1-y)-1, 0.)
    eta = 2*y-1
    for m in range(deg+1):
        pm = eval_jacobi(m, 0, 0, xi)*(1-y)**m
        for n in range(deg+1-m):
            pn = eval_jacobi(n, 2*m+1, 0, eta)
            out.append(pm*pn*np.sqrt((2*m+1)*(2*m+2*n+2)))
    return np.array(out)


def expand(struct, p):
    n0, n1, n2 = struct
    pts, wts = [], []
    k = 0
    if n0:
        pts.append(np.array([[1/3], [1/3], [1/3]]))
        wts.append(np.array([p[k]]))
        k += 1
    for _ in range(n1):
        a, w = p[k], p[k+1]
        k += 2
        b = 1-2*a
        pts.append(np.array([[a, a, b], [a, b, a], [b, a, a]]))
        wts.append(np.array([w, w, w]))
    for _ in range(n2):
        a, b, w = p[k], p[k+1], p[k+2]
        k += 3
        c = 1-a-b
        pts.append(np.array([[a, a, b, b, c, c], [b, c, a, c, a, b], [c, b, c, a, b, a]]))
        wts.append(np.full(6, w))
    return np.concatenate(pts, axis=1), np.concatenate(wts)


def residual(p, struct, deg, target):
    l, w = expand(struct, p)
    return dubiner(deg, l) @ w-target


def target_moments(deg):
    t = np.zeros((deg+1)*(deg+2)//2)
    # int over reference triangle (area 1/2) normalised to weights summing to 1:
    # only the constant basis function (value sqrt(2)) has a non-zero mean
    t[0] = np.sqrt(2.)
    return t


def attempt(args):
    deg, struct, seed = args
    rng = np.random.default_rng(seed)
    n0, n1, n2 = struct
    N = n0+3*n1+6*n2
    p = []
    if n0:
        p.append(1./N)
    for _ in range(n1):
        p += [rng.uniform(0.02, 0.49), 1./N]
    for _ in range(n2):
        a = rng.uniform(0.02, 0.45)
        b = rng.uniform(0.02, 0.9*(1-a))
        p += [a, min(b, 1-a-0.02), 1./N]
    p = np.array(p)
    target = target_moments(deg)
    try:
        sol = least_squares(residual, p, args=(struct, deg, target), method='lm', xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=4000)
    except Exception:
        return None
    if np.abs(sol.fun).max() > 1e-13:
        return None
    # polish
    sol = least_squares(residual, sol.x, args=(struct, deg, target), method='lm', xtol=3e-16, ftol=3e-16, gtol=3e-16, max_nfev=200)
    l, w = expand(struct, sol.x)
    if np.abs(sol.fun).max() > 2e-15 or w.min() <= 1e-6/len(w) or l.min() <= 1e-4:
        return None
    # reject (near-)duplicate nodes
    d = np.abs(l[:, :, None]-l[:, None, :]).max(axis=0)+np.eye(l.shape[1])
    if d.min() < 1e-4:
        return None
    return sol.x.tolist(), float(np.abs(sol.fun).max()), float(w.min()), float(l.min())


def verify(l, w, deg):
    """independent check against the closed form  int l0^a l1^b l2^c / area = 2 a! b! c! / (a+b+c+2)!"""
    from math import factorial
    err = 0.
    for a in range(deg+1):
        for b in range(deg+1-a):
            for c in range(deg+1-a-b):
                exact = 2.*factorial(a)*factorial(b)*factorial(c)/factorial(a+b+c+2)
                approx = (w*l[0]**a*l[1]**b*l[2]**c).sum()
                err = max(err, abs(exact-approx))
    return err


def main():
    max_deg = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    restarts = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    out_fn = os.path.join(os.path.dirname(__file__), '..', 'pynucleus_amd', 'data', 'triangle_rules.json')
    rules = {}
    if os.path.exists(out_fn):
        with open(out_fn) as f:
            rules = json.load(f)['rules']
    pool = Pool(8)
    for deg in range(2, max_deg+1):
        if str(deg) in rules:
            print('degree', deg, 'already present with', len(rules[str(deg)]['weights']), 'points')
            continue
        t0 = time.time()
        found = None
        for struct in STRUCTS.get(deg, []):
            jobs = [(deg, struct, 1000*deg+s) for s in range(restarts)]
            best = None
            for res in pool.imap_unordered(attempt, jobs, chunksize=4):
                if res is not None:
                    # prefer the most interior / most uniform-weight solution
                    if best is None or res[2]*res[3] > best[2]*best[3]:
                        best = res
            if best is not None:
                found = (struct, best)
                break
        if found is None:
            print('degree', deg, ': no rule found ({:.0f}s) -> conical product fallback at run time'.format(time.time()-t0))
            continue
        struct, (p, res, wmin, lmin) = found
        l, w = expand(struct, np.array(p))
        order = np.lexsort((l[2], l[1], l[0]))
        l, w = l[:, order], w[order]
        err = verify(l, w, deg)
        assert err < 5e-15, (deg, err)
        # the rule must NOT be exact one degree higher only by accident of tolerance; informational
        print('degree {:2d}: {:3d} points, struct {}, moment err {:.1e}, monomial err {:.1e}, wmin {:.2e}, lmin {:.2e} ({:.0f}s)'.format(
            deg, len(w), struct, res, err, wmin, lmin, time.time()-t0))
        rules[str(deg)] = {'struct': list(struct), 'nodes': l.tolist(), 'weights': w.tolist()}
        with open(out_fn, 'w') as f:
            json.dump({'comment': 'fully symmetric positive interior triangle rules; barycentric nodes [3][n], weights sum to 1; generated by tools/gen_triangle_rules.py',
                       'rules': rules}, f)
    pool.close()


if __name__ == '__main__':
    main()
