"""mimi_amd -- MI355X-native element integration / assembly for mimi's NURBS nonlinear
solids.  The compute path is libmimi_hip.so (hand-written HIP for gfx950, C ABI in
include/mimi_hip.h); this package is the host-side mirror of the reference's operator
surface.  There is no CPU fallback."""
from .materials import (CompressibleOgdenNeoHookean, J2, StVenantKirchhoff, J2Linear, J2Simo, J2Log, Material, HardeningBase, PowerLawHardening,
                        VoceHardening, JohnsonCookHardening, JohnsonCookRateDependentHardening,
                        JohnsonCookTemperatureAndRateDependentHardening,
                        JohnsonCookConstantTemperatureHardening)
from .splines import BSplinePatch
from . import integrators
from .integrators import RigidSphere, RigidPlane, RigidSpline, NearestDistanceToSplines
from .solid import NonlinearSolid, Solid, BoundaryConditions, RuntimeCommunication

__all__ = ["CompressibleOgdenNeoHookean", "J2", "Material", "BSplinePatch", "integrators", "NonlinearSolid",
           "Solid", "BoundaryConditions", "RuntimeCommunication", "RigidSphere", "RigidPlane"]
