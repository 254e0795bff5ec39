"""Variable fractional orders s(x, y) (host side).

Mirrors /root/reference/nl/PyNucleus_nl/fractionalOrders.pyx for the orders that are piecewise constant on a finite
partition of the domain: variableConstFractionalOrder :203-217, piecewiseConstantFractionalOrder :219-283,
leftRightFractionalOrder :285-336, layersFractionalOrder :826-882.  With ``piecewise=True`` the reference evaluates the
order once per element pair at the two cell centres (Kernel.evalParams, kernelsCy.pyx:1852-1867); an order of this family
is therefore a label per cell plus a small table sVals[label_x, label_y], which is what the GPU path consumes.
Smoothly varying orders (smoothedLeftRight, feFractionalOrder, ...) are not built.
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


class piecewiseConstantFractionalOrder(variableFractionalOrder):
    """fractionalOrders.pyx:219-283: blockIndicator(x) -> block number"""

    def __init__(self, dim, blockIndicator, sVals):
        super().__init__(sVals)
        self.dim = int(dim)
        self.blockIndicator = blockIndicator

    def labels(self, points):
        return np.array([int(self.blockIndicator(p)) for p in np.atleast_2d(points)], dtype=np.int32)
