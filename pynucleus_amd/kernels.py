"""Kernel objects gamma(x,y) and their factories (host side).

Keeps the reference's factory signatures and the attributes the builder reads
(/root/reference/nl/PyNucleus_nl/kernels.py:109-231, kernelsCy.pxd:20-50,88-97).
A kernel is reduced to a small POD parameter block for the HIP library:

    gamma(x,y) = scale * (|x-y|^2)^exponent   [ * 1{|x-y|^2 <= horizon^2} ]

Reference formulas followed (file:line under /root/reference/nl/PyNucleus_nl):
  kernelsCy.pyx:159-183  fracKernelInfinite{1,2,3}D     C*pow(d2, -d/2-s)
  kernelsCy.pyx:75-114   fracKernelFinite{1,2,3}D       same inside the interaction set
  kernelsCy.pyx:216-240  boundary variants              C*pow(d2, -(d-1)/2-s)
  kernelsCy.pyx:273-294  indicator kernel               C inside the interaction set
  kernelsCy.pyx:321-360  peridynamic kernel             C/sqrt(d2) inside the interaction set
  kernelsCy.pyx:1982-2027 getBoundaryKernel             phi = 1/s
  kernelNormalization.pyx:70-91   constantFractionalLaplacianScaling
  kernelNormalization.pyx:224-256 constantIntegrableScaling (indicator / peridynamic)
"""
import numpy as np
from math import gamma as Gamma, pi

FRACTIONAL = 0
INDICATOR = 1
PERIDYNAMIC = 2
GAUSSIAN = 3
EXPONENTIAL = 4
# device ids of the Gauss-theorem twins of the integrable kernels on the full space (kernelsCy.pyx:418-477)
GAUSSIAN_BOUNDARY = 5
EXPONENTIAL_BOUNDARY = 6

_KERNEL_NAMES = {'FRACTIONAL': FRACTIONAL, 'INDICATOR': INDICATOR, 'CONSTANT': INDICATOR,
                 'PERIDYNAMIC': PERIDYNAMIC, 'INVERSEDISTANCE': PERIDYNAMIC, 'INVERSEOFDISTANCE': PERIDYNAMIC,
                 'GAUSSIAN': GAUSSIAN, 'EXPONENTIAL': EXPONENTIAL}


def getKernelEnum(kernelTypeString):
    if isinstance(kernelTypeString, (int, np.integer)):
        return int(kernelTypeString)
    try:
        return _KERNEL_NAMES[kernelTypeString.upper()]
    except KeyError:
        raise NotImplementedError(kernelTypeString)


class constant:
    def __init__(self, value):
        self.value = float(value)

    def __call__(self, x):
        return self.value

    def __repr__(self):
        return '{}'.format(self.value)


class constFractionalOrder:
    """s(x,y) = const (fractionalOrders.pyx:72-93)"""
    symmetric = True
    numParameters = 1

    def __init__(self, s):
        self.value = float(s)
        self.min = self.max = self.value

    def __call__(self, x, y):
        return self.value

    def __repr__(self):
        return 's={}'.format(self.value)


class constantTwoPoint:
    symmetric = True

    def __init__(self, value):
        self.value = float(value)

    def __call__(self, x, y):
        return self.value

    def __repr__(self):
        return '{}'.format(self.value)


class constantFractionalLaplacianScaling(constantTwoPoint):
    def __init__(self, dim, s, horizon, tempered=0.):
        self.dim = dim
        if 1. < s < 2.:
            s = s-1.
        self.s = s
        self.horizon = horizon
        self.tempered = tempered
        if horizon <= 0. or s <= 0. or s >= 1.:
            value = np.nan
        elif horizon < np.inf:
            value = (2.-2*s)*pow(horizon, 2*s-2.)*dim*Gamma(0.5*dim)/pow(pi, 0.5*dim)*0.5
        elif tempered == 0. or s == 0.5:
            value = 2.0**(2.0*s)*s*Gamma(s+0.5*dim)/pow(pi, 0.5*dim)/Gamma(1.0-s)*0.5
        else:
            raise NotImplementedError('tempered kernels')
        super().__init__(value)

    def __repr__(self):
        return 'constantFractionalLaplacianScaling({},{} -> {})'.format(self.s, self.horizon, self.value)


class constantIntegrableScaling(constantTwoPoint):
    """kernelNormalization.pyx:225-290"""

    def __init__(self, kType, interaction, dim, horizon, gaussian_variance=1.0, exponentialRate=1.0):
        from math import erf, exp, sqrt
        self.kType, self.dim, self.horizon = kType, dim, horizon
        if horizon <= 0.:
            value = np.nan
        elif kType == INDICATOR:
            if dim == 1:
                value = 3./horizon**3/2.
            elif dim == 2:
                value = 8./pi/horizon**4/2.
            else:
                raise NotImplementedError()
        elif kType == PERIDYNAMIC:
            if dim == 1:
                value = 2./horizon**2/2.
            elif dim == 2:
                value = 6./pi/horizon**3/2.
            else:
                raise NotImplementedError()
        elif kType == GAUSSIAN:
            if dim == 1:
                value = (4.0/sqrt(pi)/(erf(3.0)-6.0*exp(-9.0)/sqrt(pi))/(horizon/3.0)**3/2. if horizon < np.inf
                         else 1.0/sqrt(2.0*pi*gaussian_variance)/2.)
            elif dim == 2:
                value = (4.0/pi/(1.0-10.0*exp(-9.0))/(horizon/3.0)**4/2. if horizon < np.inf else 1.0/(2.0*pi*gaussian_variance)/2.)
            else:
                raise NotImplementedError()
        elif kType == EXPONENTIAL:
            a = exponentialRate
            if dim == 1:
                value = (a**3/(2.0-exp(-a*horizon)*(2.0+2.0*a*horizon+(a*horizon)**2))/2. if horizon < np.inf else a**3/2.0/2.)
            else:
                raise NotImplementedError()
        else:
            raise NotImplementedError()
        super().__init__(value)


class interactionDomain:
    symmetric = True

    def __init__(self, horizon):
        self.horizon = horizon


class fullSpace(interactionDomain):
    device_id = 0

    def __init__(self):
        super().__init__(np.inf)

    def __repr__(self):
        return 'R^d'


class ball2_retriangulation(interactionDomain):
    """l2 ball |x-y| <= horizon; elements cut by the horizon are re-triangulated
    (interactionDomains.pyx:395-822, 866-980)."""
    device_id = 1

    def __repr__(self):
        return 'ball2({})'.format(self.horizon)


class ball2_barycenter(interactionDomain):
    """l2 ball; a cut element interacts entirely or not at all, decided by its barycentre
    (interactionDomains.pyx:340-392, 982-1067)."""
    device_id = 2

    def __repr__(self):
        return 'ball2_barycenter({})'.format(self.horizon)


class Kernel:
    """gamma(x,y) = scale*(|x-y|^2)^exponent on the interaction set."""

    def __init__(self, dim, kType, horizon, interaction=None, scaling=None, phi=None, piecewise=True, boundary=False,
                 valueSize=1, max_horizon=np.nan):
        self.dim = int(dim)
        self.kernelType = kType
        self.horizon = horizon if isinstance(horizon, constant) else constant(np.inf if horizon is None else horizon)
        self.horizonValue = self.horizon.value
        self.finiteHorizon = self.horizonValue != np.inf
        if interaction is None:
            interaction = fullSpace() if not self.finiteHorizon else ball2_retriangulation(self.horizonValue)
        self.interaction = interaction
        self.complement = False
        self.scalingPrePhi = scaling
        self.phi = phi
        self.scaling = scaling
        self.scalingValue = scaling.value*(phi.value if phi is not None else 1.)
        self.piecewise = piecewise
        self.boundary = boundary
        self.valueSize = valueSize
        self.max_horizon = self.horizonValue if np.isnan(max_horizon) else max_horizon
        self.variable = self.variableOrder = self.variableHorizon = self.variableScaling = self.variableSingularity = False
        self.symmetric = True
        self.exponentInverse = None
        if kType in (INDICATOR, GAUSSIAN, EXPONENTIAL):
            self.singularityValue = 0.                   # kernelsCy.pyx:649-664
        elif kType == PERIDYNAMIC:
            self.singularityValue = -1. if not boundary else 0.
        elif kType != FRACTIONAL:
            raise NotImplementedError(kType)
        if kType != FRACTIONAL:
            self.min_singularity = self.max_singularity = self.singularityValue

    # reference: Kernel.getSingularityValue / getHorizonValue / getScalingValue
    def getSingularityValue(self):
        return self.singularityValue

    def getHorizonValue(self):
        return self.horizonValue

    def getHorizonValue2(self):
        return self.horizonValue**2

    def getScalingValue(self):
        return self.scalingValue

    @property
    def exponent(self):
        """power of |x-y|^2; Gaussian / exponential kernels: the factor in exp(exponent |x-y|^2) / exp(exponent |x-y|)"""
        if self.kernelType in (GAUSSIAN, EXPONENTIAL):
            return -self.exponentInverse
        return 0.5*self.singularityValue

    def device_params(self):
        """POD block handed to pnl_set_kernel: (type, dim, exponent, scale, horizon^2)."""
        h2 = self.horizonValue**2 if self.finiteHorizon else np.inf
        ktype = int(self.kernelType)
        if self.boundary and self.kernelType in (GAUSSIAN, EXPONENTIAL):
            ktype = GAUSSIAN_BOUNDARY if self.kernelType == GAUSSIAN else EXPONENTIAL_BOUNDARY
        return dict(ktype=ktype, dim=self.dim, exponent=float(self.exponent),
                    scale=float(self.scalingValue), horizon2=float(h2), boundary=bool(self.boundary),
                    interaction=int(getattr(self.interaction, 'device_id', 0)) if self.finiteHorizon else 0)

    def __call__(self, x, y):
        x = np.atleast_1d(np.asarray(x, dtype=float))
        y = np.atleast_1d(np.asarray(y, dtype=float))
        if getattr(self, 'pointwise', False):
            return float(self.evalPoints(x, y))
        d2 = float(((x-y)**2).sum())
        if self.finiteHorizon and not d2 <= self.horizonValue**2:
            return 0.
        if self.kernelType == GAUSSIAN:
            if self.boundary:
                # kernelsCy.pyx:418-445 with gammainc(a, x) = Gamma(a) * gammaincc(a, x) (:39-40): Gamma(1/2, z) = sqrt(pi) erfc(sqrt z),
                # Gamma(1, z) = exp(-z)
                from scipy.special import erfc
                z = d2*self.exponentInverse
                return self.scalingValue*(np.sqrt(np.pi/self.exponentInverse)*erfc(np.sqrt(z)) if self.dim == 1
                                          else np.exp(-z)/(self.exponentInverse*np.sqrt(d2)))
            return self.scalingValue*np.exp(self.exponent*d2)
        if self.kernelType == EXPONENTIAL:
            if self.boundary:
                return 2.0*self.scalingValue*np.exp(self.exponent*np.sqrt(d2))/self.exponentInverse      # kernelsCy.pyx:463-477
            return self.scalingValue*np.exp(self.exponent*np.sqrt(d2))
        return self.scalingValue*d2**self.exponent

    def _integrable_boundary_scaling(self):
        raise NotImplementedError('Gauss-theorem boundary kernels exist for fractional kernels only')

    def getModifiedKernel(self, horizon=None, scaling=None):
        raise NotImplementedError()

    def getBoundaryKernel(self):
        """kernelsCy.pyx:1194-1218: the same type with boundary = True, the same scaling and exponentInverse"""
        if self.kernelType not in (GAUSSIAN, EXPONENTIAL) or self.finiteHorizon:
            raise NotImplementedError('Gauss-theorem twin of kernel type {} / a finite horizon'.format(self.kernelType))
        k = getIntegrableKernel(self.dim, self.kernelType, self.horizon, scaling=self.scaling, phi=self.phi, piecewise=self.piecewise,
                                boundary=True, variance=getattr(self, 'variance', 1.0), exponentialRate=getattr(self, 'exponentialRate', 1.0))
        k.exponentInverse = self.exponentInverse
        k.normalized = getattr(self, 'normalized', True)
        return k

    def __repr__(self):
        name = {FRACTIONAL: 'fractional', INDICATOR: 'indicator', PERIDYNAMIC: 'peridynamic', GAUSSIAN: 'gaussian',
                EXPONENTIAL: 'exponential'}[self.kernelType]
        return 'kernel({}{}, {}, {})'.format(name, '-boundary' if self.boundary else '', self.interaction, self.scalingValue)


class FractionalKernel(Kernel):
    def __init__(self, dim, s, horizon, interaction, scaling, phi=None, piecewise=True, boundary=False,
                 derivative=0, tempered=0., max_horizon=np.nan, manifold=False, normalized=True):
        if derivative != 0 or tempered != 0. or manifold:
            raise NotImplementedError('derivative / tempered / manifold fractional kernels')
        from .fractionalOrders import variableFractionalOrder, singleVariableUnsymmetricFractionalOrder
        self.s = s
        self.derivative = derivative
        self.temperedValue = tempered
        self.manifold = manifold
        self.normalized = normalized
        # orders of one variable s(x): evaluated per quadrature point, non-symmetric (kernels.py:147-149 forces piecewise=False)
        self.pointwise = isinstance(s, singleVariableUnsymmetricFractionalOrder)
        if self.pointwise:
            piecewise = False
            if horizon is not None and getattr(horizon, 'value', np.inf) != np.inf:
                raise NotImplementedError('pointwise variable order with a finite horizon')
        variableOrder = isinstance(s, variableFractionalOrder) or self.pointwise
        if variableOrder:
            # kernelsCy.pyx:1603-1621: parameters are set per element pair by evalParams
            if not piecewise and not self.pointwise:
                raise NotImplementedError('two-variable orders that are not piecewise constant per element pair')
            self.sValue = np.nan
            self.singularityValue = np.nan
            shift = 0. if not boundary else 1.
            self.min_singularity = shift-dim-2*s.min
            self.max_singularity = shift-dim-2*s.max
            self._scalingFun = scaling
            scaling = constantTwoPoint(np.nan)
        else:
            if not isinstance(s, constFractionalOrder):
                raise NotImplementedError('fractional order {}'.format(s))
            self.sValue = s.value
            # kernelsCy.pyx:1622-1634
            if not boundary:
                self.singularityValue = -dim-2*self.sValue
            else:
                self.singularityValue = 1.-dim-2*self.sValue
            self.min_singularity = self.max_singularity = self.singularityValue
        super().__init__(dim, FRACTIONAL, horizon, interaction, scaling, phi, piecewise, boundary, 1, max_horizon)
        if variableOrder:
            self.variable = self.variableOrder = self.variableSingularity = self.variableScaling = True
            self.symmetric = bool(s.symmetric)

    def scalingOfOrder(self, sv):
        """variableFractionalLaplacianScaling (kernelNormalization.pyx:329-364) times phi = 1/s of the boundary twin; vectorised"""
        from scipy.special import gamma as G
        sv = np.asarray(sv, dtype=float)
        C = 2.0**(2.0*sv)*sv*G(sv+0.5*self.dim)*pi**(-0.5*self.dim)/G(1.0-sv)*0.5 if self.normalized else np.full(sv.shape, 0.5)
        return C/sv if self.boundary else C

    def evalPoints(self, x, y):
        """pointwise kernels: gamma(x, y) = C(s(x)) |x-y|^(shift-d-2 s(x)) for arrays of points (updateAndEvalFractional,
        kernelsCy.pyx:596-622)"""
        x = np.asarray(x, dtype=float)
        y = np.asarray(y, dtype=float)
        sv = self.s.evalPoints(x)
        d2 = ((x-y)**2).sum(axis=-1)
        return self.scalingOfOrder(sv)*d2**(0.5*((1. if self.boundary else 0.)-self.dim-2*sv))

    def evalParams(self, x, y):
        """kernelsCy.pyx:1852-1867 (piecewise): order, singularity and scaling of the element pair with centres x, y"""
        if not self.variable:
            return
        if self.pointwise:
            return                                          # evalParams does nothing for piecewise == False
        sv = self.s(x, y)
        self.sValue = sv
        self.singularityValue = (1. if self.boundary else 0.)-self.dim-2*sv
        C = constantFractionalLaplacianScaling(self.dim, sv, self.horizonValue).value if self.normalized else 0.5
        self.scalingValue = C/sv if self.boundary else C

    def constantOrderKernel(self, sv):
        """the constant-order kernel an element pair with s(x, y) = sv sees after evalParams (same family, same boundary flag)"""
        k = getFractionalKernel(self.dim, float(sv), self.horizon, self.interaction, None, self.normalized, self.piecewise, None,
                                self.boundary)
        # the near-field quadrature orders are chosen from the extreme singularities of the VARIABLE kernel
        # (fractionalLaplacian2D.pyx:606, 1210; fractionalLaplacian1D.pyx:218-219, 629-630)
        k.min_singularity, k.max_singularity = self.min_singularity, self.max_singularity
        return k

    def getModifiedKernel(self, s=None, horizon=None, scaling=None):
        s = self.s if s is None else s
        if horizon is None:
            horizon = self.horizon
            interaction = self.interaction
            if scaling is None:
                scaling = self.scalingPrePhi
        else:
            interaction = None
            if scaling is None and isinstance(self.scalingPrePhi, constantFractionalLaplacianScaling):
                scaling = None if self.normalized else self.scalingPrePhi
        return getFractionalKernel(self.dim, s, horizon, interaction, scaling, self.normalized, self.piecewise,
                                   None, self.boundary)

    def getFullSpaceKernel(self):
        """getModifiedKernel(horizon = inf) (kernelsCy.pyx:1085-1107) for a constant order: the same kernel on the full space with the
        SAME scaling (the normalisation of the finite horizon is kept); the cluster method integrates cluster exteriors with its
        boundary twin (NA:953-955)"""
        if self.variable:
            raise NotImplementedError('full-space twin of a variable-order kernel')
        return FractionalKernel(self.dim, self.s, None, None, self.scalingPrePhi, phi=self.phi, piecewise=self.piecewise, boundary=self.boundary,
                                normalized=self.normalized)

    def getBoundaryKernel(self):
        """Kernel obtained by eliminating the exterior via Gauss' theorem:
        Gamma_b = (C/s) |x-y|^{-(d-1)-2s}."""
        if self.variable:
            return FractionalKernel(self.dim, self.s, None if self.pointwise else self.horizon, None, None, phi=None,
                                    piecewise=self.piecewise, boundary=True, normalized=self.normalized)
        return FractionalKernel(self.dim, self.s, self.horizon, self.interaction if self.finiteHorizon else None, self.scalingPrePhi,
                                phi=constantTwoPoint(1./self.sValue), piecewise=self.piecewise, boundary=True,
                                normalized=self.normalized)

    def __repr__(self):
        return 'kernel(fractional{}, {}, {}, {})'.format('-boundary' if self.boundary else '', self.s, self.interaction, self.scalingValue)


def _getFractionalOrder(s):
    from .fractionalOrders import variableFractionalOrder, singleVariableUnsymmetricFractionalOrder
    if isinstance(s, (constFractionalOrder, variableFractionalOrder, singleVariableUnsymmetricFractionalOrder)):
        return s
    if isinstance(s, (float, int, np.floating)):
        return constFractionalOrder(float(s))
    raise NotImplementedError(s)


def _getHorizon(horizon):
    if horizon is None:
        return constant(np.inf)
    if isinstance(horizon, constant):
        return horizon
    return constant(float(horizon))


class ellipse_retriangulation(ball2_retriangulation):
    """interactionDomains.pyx:1579-1604 (constant a, b, theta): {y: |T (y - x)| <= horizon}, T = [[cos t / a, -sin t / a],
    [sin t / b, cos t / b]] -- the ellipse with semi-axes a horizon, b horizon turned by theta; one of a, b must be 1.  The kernel
    is evaluated at the transformed distance |T (x - y)| (linearTransformInteraction.evalPtr :1417-1423)."""

    def __init__(self, horizon, a, b, theta=0.):
        a, b, theta = float(getattr(a, 'value', a)), float(getattr(b, 'value', b)), float(getattr(theta, 'value', theta))
        assert a == 1. or b == 1., 'One of the two axes must be equal to 1.'
        super().__init__(horizon)
        self.a, self.b, self.theta = a, b, theta
        self.transform = np.array([[np.cos(theta)/a, -np.sin(theta)/a], [np.sin(theta)/b, np.cos(theta)/b]])

    def __repr__(self):
        return 'ellipse({}, {}, {}; {})'.format(self.a, self.b, self.theta, self.horizon)


class ellipse_barycenter(ball2_barycenter):
    """interactionDomains.pyx:1606-1630: the same set, cut elements decided by their barycentre"""

    def __init__(self, horizon, a, b, theta=0.):
        a, b, theta = float(getattr(a, 'value', a)), float(getattr(b, 'value', b)), float(getattr(theta, 'value', theta))
        assert a == 1. or b == 1., 'One of the two axes must be equal to 1.'
        super().__init__(horizon)
        self.a, self.b, self.theta = a, b, theta
        self.transform = np.array([[np.cos(theta)/a, -np.sin(theta)/a], [np.sin(theta)/b, np.cos(theta)/b]])

    def __repr__(self):
        return 'ellipse_barycenter({}, {}, {}; {})'.format(self.a, self.b, self.theta, self.horizon)


def _getInteraction(interaction, horizon):
    if isinstance(interaction, interactionDomain):
        return interaction
    if horizon.value == np.inf:
        return fullSpace()
    if interaction is None or interaction == 'ball2':
        return ball2_retriangulation(horizon.value)
    if interaction == 'ball2_barycenter':
        return ball2_barycenter(horizon.value)
    if isinstance(interaction, str) and interaction.startswith('ellipse'):
        # the drivers' "ellipse(a,b,theta)" / "ellipse_barycenter(a,b,theta)" (nonlocalProblems.py interaction argument)
        name, args = interaction.split('(', 1)
        a, b, theta = [float(v) for v in args.rstrip(')').split(',')]
        return (ellipse_barycenter if 'barycenter' in name else ellipse_retriangulation)(horizon.value, a, b, theta)
    raise NotImplementedError('Interaction: {}'.format(interaction))


def getFractionalKernel(dim, s, horizon=None, interaction=None, scaling=None, normalized=True, piecewise=True, phi=None,
                        boundary=False, derivative=0, tempered=0., max_horizon=np.nan, manifold=False):
    sFun = _getFractionalOrder(s)
    horizonFun = _getHorizon(horizon)
    interaction = _getInteraction(interaction, horizonFun)
    from .fractionalOrders import variableFractionalOrder, singleVariableUnsymmetricFractionalOrder
    if isinstance(sFun, (variableFractionalOrder, singleVariableUnsymmetricFractionalOrder)):
        # variableFractionalLaplacianScaling (kernelNormalization.pyx:329-364) is evaluated per element pair in evalParams
        return FractionalKernel(dim, sFun, horizonFun, interaction, None, None, piecewise, boundary, derivative, tempered,
                                max_horizon, manifold, normalized)
    if scaling is None:
        if normalized:
            scaling = constantFractionalLaplacianScaling(dim, sFun.value, horizonFun.value, tempered)
        else:
            scaling = constantTwoPoint(0.5)
        if boundary:
            fac = constantTwoPoint(1./sFun.value)
            phi = fac if phi is None else constantTwoPoint(fac.value*phi.value)
    return FractionalKernel(dim, sFun, horizonFun, interaction, scaling, phi, piecewise, boundary, derivative, tempered,
                            max_horizon, manifold, normalized)


def getIntegrableKernel(dim, kernel, horizon, scaling=None, interaction=None, normalized=True, piecewise=True, phi=None,
                        boundary=False, max_horizon=np.nan, **kwargs):
    kType = getKernelEnum(kernel)
    horizonFun = _getHorizon(horizon)
    interaction = _getInteraction(interaction, horizonFun)
    variance, rate = float(kwargs.get('variance', 1.0)), float(kwargs.get('exponentialRate', 1.0))
    if scaling is None:
        if normalized:
            scaling = constantIntegrableScaling(kType, interaction, dim, horizonFun.value, variance, rate)
        else:
            scaling = constantTwoPoint(0.5)
    k = Kernel(dim, kType, horizonFun, interaction, scaling, phi, piecewise, boundary, 1, max_horizon)
    k.normalized = bool(normalized)
    if kType == GAUSSIAN:
        # kernelsCy.pyx:687-692: 1 / (horizon / 3)^2 for a finite horizon, 1 / (2 variance^dim) on the full space
        k.exponentInverse = 1.0/(k.horizonValue/3.)**2 if k.finiteHorizon else 0.5/variance**dim
        k.variance = variance
    elif kType == EXPONENTIAL:
        k.exponentInverse = rate
        k.exponentialRate = rate
    return k


def getKernel(dim, s=None, horizon=None, scaling=None, interaction=None, normalized=True, piecewise=True, phi=None,
              kernel=FRACTIONAL, boundary=False, max_horizon=np.nan, variance=1., exponentialRate=1.0):
    kType = getKernelEnum(kernel)
    if kType == FRACTIONAL:
        return getFractionalKernel(dim, s, horizon, interaction, scaling, normalized, piecewise, phi, boundary,
                                   max_horizon=max_horizon)
    return getIntegrableKernel(dim, kernel=kType, horizon=horizon, scaling=scaling, interaction=interaction,
                               normalized=normalized, piecewise=piecewise, phi=phi, max_horizon=max_horizon, variance=variance,
                               exponentialRate=exponentialRate)


class _kernelFactory:
    """nonlocalProblems.py:123-131"""

    def __call__(self, name, **kwargs):
        return self.build(name, **kwargs)

    def build(self, name, **kwargs):
        n = name.lower()
        if n == 'fractional':
            return getFractionalKernel(**kwargs)
        if n in ('indicator', 'constant'):
            return getIntegrableKernel(kernel=INDICATOR, **kwargs)
        if n in ('inversedistance', 'inverseofdistance', 'peridynamic'):
            return getIntegrableKernel(kernel=PERIDYNAMIC, **kwargs)
        if n == 'gaussian':
            return getIntegrableKernel(kernel=GAUSSIAN, **kwargs)
        if n == 'exponential':
            return getIntegrableKernel(kernel=EXPONENTIAL, **kwargs)
        raise NotImplementedError(name)


kernelFactory = _kernelFactory()
