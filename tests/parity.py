"""Shared parity helpers: same inputs -> oracle (CPU, reference op order) vs HipExec (GPU)."""
import numpy as np

from oracle import oracle as orc

FIELDS = ("world_pos", "world_vel", "world_accel", "force")
# north_star tolerance: 1e-9 relative on f64 state.  "Relative" = per entity and field, scaled by
# that field vector's largest component (a quaternion or velocity component crossing zero has
# no meaningful element-wise relative error).
F64_RTOL = 1e-9


def field_rel_err(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    scale = np.maximum(np.max(np.abs(ref), axis=1, keepdims=True), 1e-300)
    return float(np.max(np.abs(got - ref) / scale)) if got.size else 0.0


def pos_rel_err(got, ref):
    """world_pos mixes a unit quaternion and a position: scale the halves separately."""
    return max(field_rel_err(got[:, :4], ref[:, :4]), field_rel_err(got[:, 4:], ref[:, 4:]))


def state_errors(hip, oracle_world):
    errs = {"world_pos": pos_rel_err(hip.world_pos, oracle_world.world_pos)}
    for f in FIELDS[1:]:
        g, r = getattr(hip, f), getattr(oracle_world, f)
        errs[f] = max(field_rel_err(g[:, :3], r[:, :3]), field_rel_err(g[:, 3:], r[:, 3:]))
    return errs


def elementwise_rel_err(got, ref):
    """SURVEY §8(d)'s own definition: |s_i - s_i^ref| / max(|s_i^ref|, floor) element by element, floor = 1e-12 x the largest
    component of that entity's field vector (so an exact zero does not divide by zero).  It bounds every component on its own
    scale — a component that is small NEXT TO its vector (one nearly-zero quaternion element) is held to a relative error of its
    own size, which the vector-scaled figure above does not ask for; both are reported."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    scale = np.maximum(np.max(np.abs(ref), axis=1, keepdims=True), 1e-300)
    return float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-12 * scale))) if got.size else 0.0


def state_errors_elementwise(hip, oracle_world):
    errs = {}
    for f in FIELDS:
        g, r = getattr(hip, f), getattr(oracle_world, f)
        cut = 4 if f == "world_pos" else 3
        errs[f] = max(elementwise_rel_err(g[:, :cut], r[:, :cut]), elementwise_rel_err(g[:, cut:], r[:, cut:]))
    return errs


def to_oracle_ops(effectors):
    return [(e.kind, tuple(e.p), e.aux) for e in effectors]
