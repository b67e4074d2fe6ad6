"""centroidalcontrolcollection_amd -- MI355X-native batched centroidal-MPC planOnce() path.

Drop-in for ONE hot path of isri-aist/CentroidalControlCollection: many independent planOnce() problem
instances solved at once by hand-written HIP kernels (gfx950) behind the C-ABI of include/ccc_amd.h.
See DESIGN.md (scope, kernels, rooflines) and INTEGRATION.md (how the reference binds to it).
"""
from .ddp import DdpCentroidal, DdpSingleRigidBody  # noqa: F401
from .linear_mpc_xy import LinearMpcXY  # noqa: F401
from .intrinsically_stable_mpc import IntrinsicallyStableMpc  # noqa: F401
from .linear_mpc_z import LinearMpcZ  # noqa: F401
from .ddp_zmp import DdpZmp  # noqa: F401
from .linear_mpc_zmp import InitialParam, LinearMpcZmp, RefData  # noqa: F401

__all__ = ["LinearMpcZmp", "RefData", "InitialParam", "DdpCentroidal", "DdpSingleRigidBody", "LinearMpcXY",
           "IntrinsicallyStableMpc", "LinearMpcZ", "DdpZmp"]
