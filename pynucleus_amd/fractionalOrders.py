"""Variable fractional orders s(x, y) (host side).

Mirrors /root/reference/nl/PyNucleus_nl/fractionalOrders.pyx for the orders that are piecewise constant on a finite
partition of the domain: variableConstFractionalOrder :203-217, piecewiseConstantFractionalOrder :219-283,
leftRightFractionalOrder :285-336, layersFractionalOrder :826-882.  With ``piecewise=True`` the reference evaluates the
order once per element pair at the two cell centres (Kernel.evalParams, kernelsCy.pyx:1852-1867); an order of this family
is therefore a label per cell plus a small table sVals[label_x, label_y], which is what the GPU path consumes.
Orders of one variable s(x) (constantNonSym, smoothedLeftRight, linearLeftRight, smoothedInnerOuter) are non-symmetric and
evaluated per quadrature point (second half of this file); feFractionalOrder and lambda orders are not built.
"""
import numpy as np


class fractionalOrderBase:
    numParameters = 1


class variableFractionalOrder(fractionalOrderBase):
    """s(x, y) = sVals[label(x), label(y)]"""

    def __init__(self, sVals):
        self.sVals = np.ascontiguousarray(sVals, dtype=np.float64)
        assert self.sVals.ndim == 2 and self.sVals.shape[0] == self.sVals.shape[1]
        self.min, self.max = float(self.sVals.min()), float(self.sVals.max())
        self.symmetric = bool(np.abs(self.sVals-self.sVals.T).max() < 1e-10)

    @property
    def numLabels(self):
        return self.sVals.shape[0]

    def labels(self, points):
        """label of every point, points[n, dim] -> int array"""
        raise NotImplementedError()

    def __call__(self, x, y):
        x = np.atleast_2d(np.asarray(x, dtype=float))
        y = np.atleast_2d(np.asarray(y, dtype=float))
        return float(self.sVals[self.labels(x)[0], self.labels(y)[0]])

    def __repr__(self):
        return '{}(sym={})'.format(type(self).__name__, self.symmetric)


class variableConstFractionalOrder(variableFractionalOrder):
    """constant order that takes the variable-order code path (used by the reference's tests to compare the two)"""

    def __init__(self, s):
        super().__init__([[float(s)]])
        self.value = float(s)

    def labels(self, points):
        return np.zeros(np.atleast_2d(points).shape[0], dtype=np.int32)

    def spec(self):
        return ('varconst', self.value)


class leftRightFractionalOrder(variableFractionalOrder):
    """fractionalOrders.pyx:285-336: x[0] < interface is 'left'"""

    def __init__(self, sll, srr, slr=np.nan, srl=np.nan, interface=0.):
        if not np.isfinite(slr):
            slr = 0.5*(sll+srr)
        if not np.isfinite(srl):
            srl = 0.5*(sll+srr)
        super().__init__([[sll, slr], [srl, srr]])
        self.symmetric = bool(slr == srl)
        self.interface = float(interface)

    def labels(self, points):
        return (np.atleast_2d(points)[:, 0] >= self.interface).astype(np.int32)

    def spec(self):
        return ('leftRight', self.sVals[0, 0], self.sVals[1, 1], self.sVals[0, 1], self.sVals[1, 0], self.interface)


class layersFractionalOrder(variableFractionalOrder):
    """fractionalOrders.pyx:826-882: layers along the last coordinate"""

    def __init__(self, dim, layerBoundaries, layerOrders):
        self.dim = int(dim)
        self.layerBoundaries = np.ascontiguousarray(layerBoundaries, dtype=np.float64)
        super().__init__(layerOrders)
        assert self.sVals.shape[0] == self.layerBoundaries.shape[0]-1
        self.symmetric = bool((self.sVals == self.sVals.T).all())

    def labels(self, points):
        c = np.atleast_2d(points)[:, self.dim-1]
        b = self.layerBoundaries
        n = b.shape[0]-1
        # first layer i with b[i] <= c <= b[i+1] (fractionalOrders.pyx:832-843)
        lab = np.clip(np.searchsorted(b, c, side='left')-1, 0, n-1)
        lab = np.where(c <= b[0], 0, np.where(c >= b[n], n-1, lab))
        return lab.astype(np.int32)

    def spec(self):
        return ('layers', self.layerBoundaries.copy(), self.sVals.copy())


class innerOuterFractionalOrder(variableFractionalOrder):
    """fractionalOrders.pyx:664-731: inside / outside the ball of radius r around center (|x - center|^2 < r^2 is 'inner')"""

    def __init__(self, dim, sii, soo, r, center, sio=np.nan, soi=np.nan):
        if not np.isfinite(sio):
            sio = 0.5*(sii+soo)
        if not np.isfinite(soi):
            soi = 0.5*(sii+soo)
        super().__init__([[sii, sio], [soi, soo]])
        self.symmetric = bool(sio == soi)
        self.dim, self.r = int(dim), float(r)
        self.center = np.ascontiguousarray(center, dtype=np.float64)[:self.dim]

    def labels(self, points):
        p = np.atleast_2d(points)[:, :self.dim]
        return (((p-self.center)**2).sum(axis=1) >= self.r*self.r).astype(np.int32)

    def spec(self):
        return ('innerOuter', self.dim, self.sVals[0, 0], self.sVals[1, 1], self.r, self.center.copy(), self.sVals[0, 1], self.sVals[1, 0])


class islandsFractionalOrder(variableFractionalOrder):
    """fractionalOrders.pyx:754-822 (2D): the 'islands' are the points whose coordinates all satisfy r <= |x_i| <= r2"""

    def __init__(self, sii, soo, r, r2, sio=np.nan, soi=np.nan):
        if not np.isfinite(sio):
            sio = 0.5*(sii+soo)
        if not np.isfinite(soi):
            soi = 0.5*(sii+soo)
        super().__init__([[sii, sio], [soi, soo]])
        self.symmetric = bool(sio == soi)
        self.r, self.r2 = float(r), float(r2)

    def labels(self, points):
        p = np.abs(np.atleast_2d(points)[:, :2])
        return (~((p >= self.r) & (p <= self.r2)).all(axis=1)).astype(np.int32)

    def spec(self):
        return ('islands', self.sVals[0, 0], self.sVals[1, 1], self.r, self.r2, self.sVals[0, 1], self.sVals[1, 0])


class sumFractionalOrder(variableFractionalOrder):
    """fractionalOrders.pyx:733-752: s(x, y) = s1(x, y) * s2(x, y) (the reference's eval multiplies the two values; fac1 / fac2 are
    stored and not used there).  Piecewise-constant factors: the labels are the pairs of labels."""

    def __init__(self, s1, fac1, s2, fac2):
        assert isinstance(s1, variableFractionalOrder) and isinstance(s2, variableFractionalOrder)
        self.s1, self.s2, self.fac1, self.fac2 = s1, s2, float(fac1), float(fac2)
        L1, L2 = s1.numLabels, s2.numLabels
        sv = np.zeros((L1*L2, L1*L2))
        for a in range(L1):
            for b in range(L2):
                for c in range(L1):
                    for d in range(L2):
                        sv[a*L2+b, c*L2+d] = s1.sVals[a, c]*s2.sVals[b, d]
        super().__init__(sv)
        self.symmetric = bool(s1.symmetric and s2.symmetric)

    def labels(self, points):
        return (self.s1.labels(points)*self.s2.numLabels+self.s2.labels(points)).astype(np.int32)

    def spec(self):
        return ('product', self.s1.spec(), self.s2.spec())


class piecewiseConstantFractionalOrder(variableFractionalOrder):
    """fractionalOrders.pyx:219-283: blockIndicator(x) -> block number"""

    def __init__(self, dim, blockIndicator, sVals):
        super().__init__(sVals)
        self.dim = int(dim)
        self.blockIndicator = blockIndicator

    def labels(self, points):
        return np.array([int(self.blockIndicator(p)) for p in np.atleast_2d(points)], dtype=np.int32)


# ---- orders s(x) of one variable: non-symmetric, evaluated per quadrature point ------------------------------------------
# fractionalOrders.pyx:338-540 (extendedFunction family), :153-183 singleVariableUnsymmetricFractionalOrder, :631-657.
# The reference switches such kernels to piecewise=False (kernels.py:147-149): gamma(x, y) = C(s(x)) |x-y|^(-d-2 s(x)).
# Every function knows its device encoding (type id + up to 6 parameters, include/pnl_hip.h pnl_order_function).

class lambdaFractionalOrder(variableFractionalOrder):
    """fractionalOrders.pyx:176-201: s(x, y) given by a Python callable.  The reference calls it per element pair at the two cell
    centres (evalParams, NO:509-513); a callable cannot run on the device, so it is TABULATED on the host: the points the assembly
    asks about (cell centres, facet centres) are grouped into labels -- two points share a label when the callable cannot tell them
    apart against the representatives of all labels, in either argument -- and the order becomes a table sVals[label(x), label(y)]
    like every other piecewise-constant order.  A callable that is not piecewise constant yields one label per point and is refused
    beyond maxLabels."""

    def __init__(self, dim, smin, smax, symmetric, fun, maxLabels=64):
        self.dim, self.fun, self.maxLabels = int(dim), fun, int(maxLabels)
        self.min, self.max, self.symmetric = float(smin), float(smax), bool(symmetric)
        self._reps = []                                  # one representative point per label
        self.sVals = np.zeros((0, 0))

    def _sig(self, p):
        return tuple(float(self.fun(p, r)) for r in self._reps)+tuple(float(self.fun(r, p)) for r in self._reps)

    def _rebuild(self):
        L = len(self._reps)
        self.sVals = np.array([[float(self.fun(self._reps[a], self._reps[b])) for b in range(L)] for a in range(L)]).reshape(L, L)
        if L:
            assert self.sVals.min() >= self.min-1e-12 and self.sVals.max() <= self.max+1e-12, 'the callable leaves [smin, smax]'

    def labels(self, points):
        pts = np.atleast_2d(np.asarray(points, dtype=np.float64))
        while True:
            sigs = {self._sig(r)+(float(self.fun(r, r)),): l for l, r in enumerate(self._reps)}
            lab = np.full(pts.shape[0], -1, dtype=np.int32)
            grew = False
            for i, p in enumerate(pts):
                l = sigs.get(self._sig(p)+(float(self.fun(p, p)),))
                if l is None:
                    # the representatives so far cannot place p: a new label, and every signature has one more column
                    if len(self._reps) >= self.maxLabels:
                        raise NotImplementedError('lambdaFractionalOrder: more than {} distinct labels -- the callable is not piecewise '
                                                  'constant on the mesh (orders evaluated per quadrature point: singleVariableUnsymmetric'
                                                  'FractionalOrder)'.format(self.maxLabels))
                    self._reps.append(np.array(p, copy=True))
                    grew = True
                    break
                lab[i] = l
            if not grew:
                break
        self._rebuild()
        return lab

    @property
    def numLabels(self):
        return len(self._reps)

    def __call__(self, x, y):
        return float(self.fun(np.asarray(x, dtype=float), np.asarray(y, dtype=float)))

    def spec(self):
        return ('lambda', self.sVals.copy())


class extendedFunction:
    device_type = 0

    def device_params(self):
        raise NotImplementedError()

    def __call__(self, x):
        """vectorised: x[..., dim] -> s[...]"""
        raise NotImplementedError()


class constantExtended(extendedFunction):
    device_type = 1

    def __init__(self, value):
        self.value = float(value)

    def __call__(self, x):
        return np.full(np.asarray(x).shape[:-1], self.value)

    def device_params(self):
        return [self.value, 0., 0., 0., 0., 0.]

    def __repr__(self):
        return '{}'.format(self.value)


class smoothStep(extendedFunction):
    """:390-445, cubic blend of sl and sr across |x0 - interface| <= r"""
    device_type = 2

    def __init__(self, sl, sr, r, interface=0.):
        self.sl, self.sr, self.r, self.interface = float(sl), float(sr), float(r), float(interface)
        self.slope = 0.5/self.r

    def __call__(self, x):
        x0 = np.asarray(x, dtype=float)[..., 0]
        t = (x0-self.interface)*self.slope+0.5
        mid = self.sl+(self.sr-self.sl)*(3.0*t**2-2.0*t**3)
        return np.where(x0 < self.interface-self.r, self.sl, np.where(x0 > self.interface+self.r, self.sr, mid))

    def device_params(self):
        return [self.sl, self.sr, self.r, self.interface, self.slope, 0.]

    def __repr__(self):
        return 'smoothStep(sl={},sr={},r={},interface={})'.format(self.sl, self.sr, self.r, self.interface)


class linearStep(extendedFunction):
    """:447-498"""
    device_type = 3

    def __init__(self, sl, sr, r, interface=0.):
        self.sl, self.sr, self.r, self.interface = float(sl), float(sr), float(r), float(interface)
        self.slope = 0.5*(self.sr-self.sl)/self.r

    def __call__(self, x):
        x0 = np.asarray(x, dtype=float)[..., 0]
        mid = self.sl+self.slope*(x0-self.interface+self.r)
        return np.where(x0 < self.interface-self.r, self.sl, np.where(x0 > self.interface+self.r, self.sr, mid))

    def device_params(self):
        return [self.sl, self.sr, self.r, self.interface, self.slope, 0.]

    def __repr__(self):
        return 'linearStep(sl={},sr={},r={})'.format(self.sl, self.sr, self.r)


class smoothStepRadial(extendedFunction):
    """:500-539, the same blend in |x| across radius +- r"""
    device_type = 4

    def __init__(self, sl, sr, r, radius=0.5):
        self.sl, self.sr, self.r, self.radius = float(sl), float(sr), float(r), float(radius)
        self.slope = 0.5/self.r

    def __call__(self, x):
        x = np.asarray(x, dtype=float)
        rr = np.sqrt((x**2).sum(axis=-1))
        t = (rr-self.radius)*self.slope+0.5
        mid = self.sl+(self.sr-self.sl)*(3.0*t**2-2.0*t**3)
        return np.where(rr < self.radius-self.r, self.sl, np.where(rr > self.radius+self.r, self.sr, mid))

    def device_params(self):
        return [self.sl, self.sr, self.r, self.radius, self.slope, 0.]

    def __repr__(self):
        return 'smoothStepRadial(sl={},sr={},r={},radius={})'.format(self.sl, self.sr, self.r, self.radius)


class lookupExtended(extendedFunction):
    """fractionalOrders.pyx:541-625: a finite-element function looked up in the cell that holds x.  Here: continuous P1
    functions -- the values at the mesh vertices (u[dof] of the vertex's DoF, 0 for a boundary DoF that the DoFMap dropped)."""
    device_type = 5

    def __init__(self, mesh, dm, u):
        assert dm.dofs_per_element == mesh.dim+1, 'feFractionalOrder: P1 functions'
        self.mesh, self.dm, self.u = mesh, dm, np.ascontiguousarray(u, dtype=np.float64)
        assert self.u.shape[0] == dm.num_dofs
        vals = np.zeros(mesh.num_vertices)
        dofs, cells = np.asarray(dm.dofs), np.asarray(mesh.cells)
        ok = dofs >= 0
        vals[cells[ok]] = self.u[dofs[ok]]
        self.vertex_values = vals

    def device_params(self):
        return [0.]*6

    def cell_values(self, cells):
        """values at the vertices of the given simplices (cells[n, k] vertex numbers)"""
        return self.vertex_values[np.asarray(cells)]

    def __call__(self, x):
        """host evaluation by point location (barycentric coordinates over all cells; small meshes / tests)"""
        x = np.atleast_2d(np.asarray(x, dtype=np.float64))
        V = self.mesh.vertices[self.mesh.cells]                     # [nc, nV, dim]
        out = np.zeros(x.shape[0])
        for i, pt in enumerate(x):
            if self.mesh.dim == 1:
                l1 = (pt[0]-V[:, 0, 0])/(V[:, 1, 0]-V[:, 0, 0])
                lam = np.stack([1.-l1, l1], axis=1)
            else:
                d0, d1, dp = V[:, 1]-V[:, 0], V[:, 2]-V[:, 0], pt[None, :2]-V[:, 0]
                det = d0[:, 0]*d1[:, 1]-d1[:, 0]*d0[:, 1]
                l1 = (dp[:, 0]*d1[:, 1]-d1[:, 0]*dp[:, 1])/det
                l2 = (d0[:, 0]*dp[:, 1]-dp[:, 0]*d0[:, 1])/det
                lam = np.stack([1.-l1-l2, l1, l2], axis=1)
            c = int(np.argmax(lam.min(axis=1)))
            out[i] = (lam[c]*self.vertex_values[self.mesh.cells[c]]).sum()
        return out if out.shape[0] > 1 else float(out[0])

    def __repr__(self):
        return 'lookupExtended({} vertex values)'.format(self.vertex_values.shape[0])


class singleVariableUnsymmetricFractionalOrder(fractionalOrderBase):
    """s(x, y) = sFun(x) (:153-183)"""
    symmetric = False

    def __init__(self, sFun, smin, smax, numParameters=0):
        self.sFun = sFun
        self.min, self.max = float(smin), float(smax)
        self.numParameters = numParameters

    def __call__(self, x, y=None):
        return float(self.sFun(np.atleast_1d(np.asarray(x, dtype=float))))

    def evalPoints(self, x):
        return self.sFun(x)

    def __repr__(self):
        return '{}({})'.format(type(self).__name__, self.sFun)


class constantNonSymFractionalOrder(singleVariableUnsymmetricFractionalOrder):
    def __init__(self, s):
        super().__init__(constantExtended(s), s, s, 1)
        self.value = float(s)


class smoothedLeftRightFractionalOrder(singleVariableUnsymmetricFractionalOrder):
    def __init__(self, sl, sr, r=0.1, slope=200., interface=0.):
        super().__init__(smoothStep(sl, sr, r, interface), min(sl, sr), max(sl, sr), 2)


class linearLeftRightFractionalOrder(singleVariableUnsymmetricFractionalOrder):
    def __init__(self, sl, sr, r=0.1, interface=0.):
        super().__init__(linearStep(sl, sr, r, interface), min(sl, sr), max(sl, sr), 2)


class feFractionalOrder(singleVariableUnsymmetricFractionalOrder):
    """fractionalOrders.pyx:660-668: the order is a finite-element function, s(x) = sum_i vec_i phi_i(x).  ``vec``: the coefficient
    vector (an object with ``.dm`` like the reference's fe_vector, or an array together with ``dm``); P1 spaces on the mesh of the
    assembly."""

    def __init__(self, vec, smin, smax, dm=None):
        dm = dm if dm is not None else vec.dm
        u = np.asarray(getattr(vec, 'toarray', lambda: vec)())
        self.vec = vec
        super().__init__(lookupExtended(dm.mesh, dm, u), smin, smax, numParameters=dm.num_dofs)

    def evalPoints(self, x):
        x = np.asarray(x, dtype=np.float64)
        flat = x.reshape(-1, x.shape[-1])
        return np.atleast_1d(self.sFun(flat)).reshape(x.shape[:-1])


class smoothedInnerOuterFractionalOrder(singleVariableUnsymmetricFractionalOrder):
    def __init__(self, sl, sr, r=0.1, slope=200., radius=0.5):
        super().__init__(smoothStepRadial(sl, sr, r, radius), min(sl, sr), max(sl, sr))
