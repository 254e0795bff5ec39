"""MI355X-native nonlocal-operator assembly behind the PyNucleus_nl builder API.

The hot path (element-pair quadrature filling dense / near-field blocks) runs in
hand-written HIP kernels for gfx950 behind a C-ABI shared library
(include/pnl_hip.h, pynucleus_amd/csrc/).  This package is the host-side mirror
of the reference interface: meshes, DoF maps, kernels and nonlocalBuilder.
"""
from .mesh import (mesh1d, mesh2d, simpleInterval, uniformSquare, uniform_disc, disc, interval, driverMesh, intervalWithInteraction,  # noqa: F401
                   PHYSICAL, NO_BOUNDARY, INTERIOR, INTERIOR_NONOVERLAPPING)
from .dofmap import P0_DoFMap, P1_DoFMap, P2_DoFMap, P3_DoFMap, dofmapFactory, fe_vector  # noqa: F401
from .kernels import (getKernel, getFractionalKernel, getIntegrableKernel, kernelFactory,  # noqa: F401
                      FRACTIONAL, INDICATOR, PERIDYNAMIC, GAUSSIAN, EXPONENTIAL, constFractionalOrder, constant, ball2_retriangulation, ball2_barycenter, ellipse_retriangulation, ellipse_barycenter)
from .local_matrix import nonlocalTables  # noqa: F401
from .fractionalOrders import (variableConstFractionalOrder, leftRightFractionalOrder, layersFractionalOrder,  # noqa: F401
                               piecewiseConstantFractionalOrder, lambdaFractionalOrder, constantNonSymFractionalOrder,
                               smoothedLeftRightFractionalOrder, linearLeftRightFractionalOrder,
                               smoothedInnerOuterFractionalOrder, feFractionalOrder, innerOuterFractionalOrder,
                               islandsFractionalOrder, sumFractionalOrder)
from .builder import nonlocalBuilder, assembleNonlocalOperator  # noqa: F401
