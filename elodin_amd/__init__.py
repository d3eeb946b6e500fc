"""elodin_amd — MI355X-native backend for ONE path of elodin-sys/elodin: the `six_dof` integrator.

Everything computes through the C ABI in include/sixdof_hip.h (hand-written gfx950 kernels);
the Python here is the host-side mirror of the reference's interface for that path.
"""
from ._lib import (BackendError, RK4, SEMI_IMPLICIT, EFF_CONST_WRENCH, EFF_UNIFORM_GRAVITY, EFF_BODY_TORQUE,
                   EFF_BODY_FORCE, EFF_BALL_DRAG, EFF_EDGE_GRAVITY_NEWTON, EFF_EDGE_GRAVITY_SOFTENED,
                   EFF_ALLPAIRS_GRAVITY_SOFTENED, component_id)
from .exec import Effector, HipExec, TickTimings
from .api import (Body, C, Edge, EntityId, Exec, GravityEdge, skew, Integrator, Quaternion, SpatialForce, SpatialInertia,
                  SpatialMotion, SpatialTransform, System, World, ball_drag, body_force, body_torque, constant_wrench,
                  gravity_newton, gravity_softened, six_dof, uniform_gravity)

__all__ = ["BackendError", "HipExec", "Effector", "TickTimings", "component_id", "RK4", "SEMI_IMPLICIT"]
